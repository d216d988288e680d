"""Flat alias of `neural_inventory_control_amd.neural_networks` (reference module: neural_networks.py)."""
from shared_imports import *  # noqa: F401,F403
from neural_inventory_control_amd.neural_networks import *  # noqa: F401,F403
from neural_inventory_control_amd.neural_networks import MyNeuralNetwork, NeuralNetworkCreator  # noqa: F401
