"""Training driver: counterpart of the reference's train_larvaV2.py -- train_larva.py's loop with `--steps_per_epoch`
(train_larvaV2.py:29, 73-81, 112, 142) in place of volume_per_step: an "epoch" is 300 MiB of input values unless given,
model.steps_per_epoch is set, the timing lines are printed for the first two epochs, and -- as in the reference, which
never sets volume_per_step there -- the volume-triggered validation / checkpoint of train_step_larva does not fire
after step 1.

    python -m larvanet_amd.train_larvaV2 --model=LarvaNetV2 --num_modules=4 --num_blocks=4,4,4,4 ...
"""
from .train_larva import main as _main


def main(argv=None):
    return _main(argv, v2=True)


if __name__ == "__main__":
    main()
