"""Flat alias of `neural_inventory_control_amd.data_handling` (reference module: data_handling.py)."""
from shared_imports import *  # noqa: F401,F403
from neural_inventory_control_amd.data_handling import *  # noqa: F401,F403
from neural_inventory_control_amd.data_handling import DatasetCreator, DeviceBatches, MyDataset, Scenario, Scenarios  # noqa: F401
