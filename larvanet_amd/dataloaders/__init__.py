"""Data-loader plugins, discovered like the reference does it (train_larva.py:58-59):
importlib.import_module('<package>.dataloaders.' + args.dataloader).create_loader()."""
