"""Flat alias of `neural_inventory_control_amd.loss_functions` (reference module: loss_functions.py)."""
from shared_imports import *  # noqa: F401,F403
from neural_inventory_control_amd.loss_functions import *  # noqa: F401,F403
from neural_inventory_control_amd.loss_functions import PolicyLoss  # noqa: F401
