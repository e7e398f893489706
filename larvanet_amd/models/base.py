"""The model-plugin contract of the reference (models/base.py:1-85), restated.

A driver loads a plugin with importlib and only ever talks to it through these methods, so any
class that honours them can replace the reference's model file of the same name.
"""


def create_model():
    return BaseModel()


class BaseModel:
    """Interface only; every method must be provided by the concrete model."""

    def __init__(self):
        self.global_step = 0
        self.loss_dict = {}

    def parse_args(self, args):
        """Consume the flags this model knows from `args` (list of str).
        Returns (parsed Namespace copy, list of leftover args)."""
        raise NotImplementedError

    def prepare(self, is_training, scales, global_step=0):
        """Build network (and, when is_training, loss/optimizer/scheduler). Must precede any other call."""
        raise NotImplementedError

    def save(self, base_path):
        """Write a checkpoint of the current weights under directory `base_path`."""
        raise NotImplementedError

    def restore(self, ckpt_path, target=None):
        """Load weights from file `ckpt_path` (`target` optionally names a sub-part)."""
        raise NotImplementedError

    def get_model(self):
        """The underlying torch.nn.Module (may be None)."""
        raise NotImplementedError

    def get_next_train_scale(self):
        """Scale factor to train on next."""
        raise NotImplementedError

    def train_step(self, input_list, scale, truth_list, summary=None):
        """One optimisation step on lists of numpy patches; returns a representative loss."""
        raise NotImplementedError

    def upscale(self, input_list, scale):
        """Super-resolve a list of CHW numpy images without training."""
        raise NotImplementedError
