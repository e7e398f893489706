"""The data-loader plugin contract of the reference (dataloaders/base.py:9-111), restated.
Images travel as numpy CHW arrays on the 0-255 scale (float32 for training patches, uint8 for
validation images); nothing is normalised anywhere."""


def create_loader():
    return BaseLoader()


class BaseLoader:
    def __init__(self):
        self.is_threaded = False

    def parse_args(self, args):
        """Consume known flags from `args`; returns (Namespace copy, leftover list)."""
        raise NotImplementedError

    def prepare(self, scales):
        raise NotImplementedError

    def get_num_images(self):
        raise NotImplementedError

    def get_patch_batch(self, batch_size, scale, input_patch_size):
        """-> (list of LR patches, list of HR patches)"""
        raise NotImplementedError

    def get_random_image_patch_pair(self, scale, input_patch_size):
        raise NotImplementedError

    def get_image_patch_pair(self, image_index, scale, input_patch_size):
        raise NotImplementedError

    def get_image_pair(self, image_index, scale):
        """-> (LR image, HR image, image name)"""
        raise NotImplementedError

    # threaded loaders only
    def start_training_queue_runner(self, batch_size, input_patch_size):
        raise NotImplementedError

    def stop_queue_runners(self):
        raise NotImplementedError

    def get_queue_data(self, scale):
        raise NotImplementedError
