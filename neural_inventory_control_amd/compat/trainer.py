"""Flat alias of `neural_inventory_control_amd.trainer` with the reference's import chain (trainer.py:1-3): a script that
starts with `from trainer import *` (main_run.py:3) gets every class of the hot path plus the shared third-party names."""
from shared_imports import *  # noqa: F401,F403
from environment import *  # noqa: F401,F403
from loss_functions import *  # noqa: F401,F403
from neural_inventory_control_amd.trainer import *  # noqa: F401,F403
from neural_inventory_control_amd.trainer import Trainer  # noqa: F401
