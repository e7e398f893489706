"""Model plugins, discovered exactly like the reference does it (train_larva.py:69-70):
importlib.import_module('<package>.models.' + args.model).create_model()."""
