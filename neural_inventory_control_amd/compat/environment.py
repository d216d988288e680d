"""Flat alias of `neural_inventory_control_amd.environment` with the reference's import chain (environment.py:1-3):
`from environment import *` also brings the data-handling and policy names."""
from shared_imports import *  # noqa: F401,F403
from data_handling import *  # noqa: F401,F403
from neural_networks import *  # noqa: F401,F403
from neural_inventory_control_amd.environment import *  # noqa: F401,F403
from neural_inventory_control_amd.environment import Simulator  # noqa: F401
