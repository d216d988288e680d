"""Flat-module surface of the reference (module names are part of its API — user scripts and its own
`main_run.py:1-3` start with `from trainer import *`, which chains through `environment`, `data_handling`,
`neural_networks`, `loss_functions` and `shared_imports`, trainer.py:1-3 / environment.py:1-3).

Importing this package puts its directory on `sys.path`, after which a reference-style script runs unchanged on the HIP
engine:

    import neural_inventory_control_amd.compat   # the one added line (or: PYTHONPATH=<this directory>)
    from trainer import *

Each flat module star-exports the package's implementation of the same-named reference module, with the same chain of
star-imports, so every name a reference script picks up that way (`Trainer`, `Simulator`, `Scenario`, `DatasetCreator`,
`MyDataset`, `PolicyLoss`, `NeuralNetworkCreator`, `MyNeuralNetwork`, `DataLoader`, `torch`, `nn`, `np`, `DefaultDict`, ...)
resolves.  Nothing here computes anything.
"""
import os
import sys

PATH = os.path.dirname(os.path.abspath(__file__))
if PATH not in sys.path:
    sys.path.insert(0, PATH)
