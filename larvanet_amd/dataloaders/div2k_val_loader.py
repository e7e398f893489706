"""DIV2K validation pairs (dataloaders/div2k_val_loader.py of the reference): whole images,
always cached, uint8 (no float cast).  The reference hard-codes c:/aim2020/... paths (:28,108,125);
here they are flags."""
import os

from .div2k_train_loader import DIV2KLoader as _TrainLoader


def create_loader():
    return DIV2KValLoader()


class DIV2KValLoader(_TrainLoader):
    float_images = False

    def _add_args(self, parser):
        parser.add_argument("--val_input_path", type=str, default="data/DIV2K_valid_LR_bicubic")
        parser.add_argument("--val_truth_path", type=str, default="data/DIV2K_valid_HR")

    def parse_args(self, args):
        parsed, remaining = super().parse_args(args)
        self._alias()
        return parsed, remaining

    def _alias(self):
        self.args.data_input_path = self.args.val_input_path
        self.args.data_truth_path = self.args.val_truth_path
        self.args.data_cached = True
        self.args.data_seed = 0

    def prepare(self, scales):
        if not hasattr(self, "args"):  # the reference never calls parse_args on the val loader (train_larva.py:63-65)
            self.parse_args([])
        super().prepare(scales)
