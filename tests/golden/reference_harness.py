"""Import harness for the upstream reference (THIS CONTAINER ONLY).

The reference lives read-only at /root/reference and is never copied into this
repo nor shipped to the GPU box.  This module makes `import trainer` etc. work
so that golden vectors can be generated (make_golden.py) and the oracle can be
pinned bit-for-bit against it (tests/test_oracle_vs_reference.py, skipped when
/root/reference is absent).

gymnasium is not installed in the image; the reference only uses
`gym.Env`, `spaces.Box`, `spaces.Dict` (environment.py:7,21-22,352-358,380-389)
so a tiny stub is injected.
"""
import contextlib
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("NIC_REFERENCE_ROOT", "/root/reference")


def reference_available():
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "environment.py"))


def _install_gym_stub():
    if "gymnasium" in sys.modules:
        return
    gym = types.ModuleType("gymnasium")
    spaces = types.ModuleType("gymnasium.spaces")

    class Env:  # noqa: D401 - stub
        pass

    class Box:
        def __init__(self, low=None, high=None, shape=None, dtype=None):
            self.low, self.high, self.shape, self.dtype = low, high, shape, dtype

    class Dict(dict):
        def __init__(self, d=None):
            super().__init__(d or {})

    gym.Env = Env
    spaces.Box = Box
    spaces.Dict = Dict
    gym.spaces = spaces
    sys.modules["gymnasium"] = gym
    sys.modules["gymnasium.spaces"] = spaces


def _install_cpu_load_shim():
    """The reference's quantile-forecaster checkpoint was saved from a CUDA tensor and is loaded without a map_location
    (neural_networks.py:536); on this CPU-only container torch.load needs one.  Harness-level shim like the gymnasium stub."""
    import torch
    if getattr(torch.load, "_nic_cpu_shim", False):
        return
    orig = torch.load

    def load(f, *a, **k):
        k.setdefault("map_location", "cpu")
        return orig(f, *a, **k)
    load._nic_cpu_shim = True
    torch.load = load


_REF = None


def load_reference():
    """Returns a namespace with the reference's public classes."""
    global _REF
    if _REF is not None:
        return _REF
    if not reference_available():
        raise RuntimeError("reference tree not present at %s" % REFERENCE_ROOT)
    _install_gym_stub()
    _install_cpu_load_shim()
    sys.dont_write_bytecode = True
    import matplotlib
    matplotlib.use("Agg")
    sys.path.insert(0, REFERENCE_ROOT)
    try:
        import trainer as ref_trainer  # star-imports everything else
        import data_handling as ref_data
        import environment as ref_env
        import neural_networks as ref_nn
        import loss_functions as ref_loss
    finally:
        sys.path.remove(REFERENCE_ROOT)
    ns = types.SimpleNamespace(
        Scenario=ref_data.Scenario,
        MyDataset=ref_data.MyDataset,
        DatasetCreator=ref_data.DatasetCreator,
        Simulator=ref_env.Simulator,
        PolicyLoss=ref_loss.PolicyLoss,
        Trainer=ref_trainer.Trainer,
        NeuralNetworkCreator=ref_nn.NeuralNetworkCreator,
        nn=ref_nn,
    )
    _REF = ns
    return ns


@contextlib.contextmanager
def in_reference_dir():
    """Some reference paths are relative (config_files/, data_files/)."""
    cwd = os.getcwd()
    os.chdir(REFERENCE_ROOT)
    try:
        yield
    finally:
        os.chdir(cwd)


def load_reference_configs(setting, policy):
    import yaml
    with open(os.path.join(REFERENCE_ROOT, "config_files/settings/%s.yml" % setting)) as f:
        cs = yaml.safe_load(f)
    with open(os.path.join(REFERENCE_ROOT, "config_files/policies_and_hyperparams/%s.yml" % policy)) as f:
        ch = yaml.safe_load(f)
    return cs, ch
