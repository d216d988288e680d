import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    # GPU tests are skipped (not failed) when no device is visible, e.g. plain `pytest tests/` on the CPU container.
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


# ---- gradient-parity record: every golden-gradient comparison of the GPU suite notes its worst relative error per test
# (tests/grad_parity_log.py); the table is written to gpurun_out/grad_parity.json at the end of a GPU session (the round's copy is
# committed under profiles/)
def pytest_sessionfinish(session, exitstatus):
    import grad_parity_log
    grad_parity_log.dump(os.path.join(ROOT, "gpurun_out", "grad_parity.json"))
