"""Third-party names the reference's modules share through `from shared_imports import *` (shared_imports.py:1-12)."""
import copy  # noqa: F401
import datetime  # noqa: F401
import os  # noqa: F401
from collections import defaultdict as DefaultDict  # noqa: F401

import numpy as np  # noqa: F401
import pandas as pd  # noqa: F401
import torch  # noqa: F401
from torch import nn  # noqa: F401
from torch.nn.modules.loss import _Loss  # noqa: F401
from torch.utils.data import DataLoader, Dataset  # noqa: F401

try:  # plotting is optional on a headless GPU box; Trainer.plot_losses imports it on demand
    import matplotlib.pyplot as plt  # noqa: F401
except Exception:  # pragma: no cover
    plt = None
