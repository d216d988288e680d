"""Flat alias of `neural_inventory_control_amd.quantile_forecaster` (reference module: quantile_forecaster.py)."""
from shared_imports import *  # noqa: F401,F403
from neural_inventory_control_amd.quantile_forecaster import FullyConnectedForecaster  # noqa: F401
