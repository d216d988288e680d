"""Worst relative error of every golden-gradient comparison, per test (filled by the GPU suite, dumped by conftest)."""
import json
import os

RECORDS = {}


def note(worst, tol):
    test_id = os.environ.get("PYTEST_CURRENT_TEST", "?").split(" ")[0]
    rec = RECORDS.setdefault(test_id, {"worst_rel_l2": 0.0, "tol": tol})
    rec["worst_rel_l2"] = max(rec["worst_rel_l2"], float(worst))


def dump(path):
    if not RECORDS:
        return
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        json.dump({"note": "worst relative L2 error of d(mean_loss)/d(theta) per parameter tensor against the reference's golden "
                           "gradients (or the oracle's), per GPU test; bar = north_star's 1e-5",
                   "worst_overall": max(v["worst_rel_l2"] for v in RECORDS.values()),
                   "tests": dict(sorted(RECORDS.items()))}, open(path, "w"), indent=1)
    except OSError:
        pass
