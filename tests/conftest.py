import ast
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long CPU test, enabled with EGOM2P_SLOW=1")


def load_golden(case):
    path = os.path.join(GOLDEN_DIR, f"{case}.npz")
    if not os.path.exists(path):
        pytest.skip(f"golden fixture {case}.npz not present")
    g = np.load(path, allow_pickle=False)
    meta = ast.literal_eval(str(g["meta"])) if "meta" in g.files else {}
    if "budgets" in meta:
        meta["budgets"] = ast.literal_eval(meta["budgets"])
    return g, meta


@pytest.fixture(scope="session")
def golden():
    return load_golden


def rel_l2(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


# ------------------------------------------------------------------------------------------------------------------------------
# Recorded parity baselines.  A tolerance in a test states what the arithmetic allows; it does not notice a
# kernel drifting from 3e-3 to 8e-3 under a 1e-2 bar.  tests/golden/parity_bars.json holds the value every such check MEASURED on
# MI355X on the tree that recorded it (seeded inputs, deterministic kernels: the values reproduce); `bar()` holds a measurement
# to BOTH the stated tolerance and its recorded baseline plus a margin (30 % for errors, +2 for counts of near-tie flips).
#   EGOM2P_RECORD_BARS=<file>: append {"name", "value"} lines instead of checking the baseline (tools/merge_bars.py folds them in).
# ------------------------------------------------------------------------------------------------------------------------------
BARS_PATH = os.path.join(GOLDEN_DIR, "parity_bars.json")
_BARS = None


def _bars():
    global _BARS
    if _BARS is None:
        import json
        _BARS = json.load(open(BARS_PATH)) if os.path.exists(BARS_PATH) else {}
    return _BARS


def bar(name, value, hard, rel_margin=0.30, abs_margin=0.0):
    """assert value <= hard (the stated tolerance) and value <= recorded[name] * (1 + rel_margin) + abs_margin"""
    value = float(value)
    rec_path = os.environ.get("EGOM2P_RECORD_BARS")
    if rec_path:
        import json
        with open(rec_path, "a") as f:
            f.write(json.dumps({"name": name, "value": value}) + "\n")
    assert value <= hard, (name, value, "stated tolerance", hard)
    base = _bars().get(name)
    if base is not None and not rec_path:
        lim = float(base) * (1.0 + rel_margin) + abs_margin
        assert value <= lim, (name, value, "recorded baseline", base, "limit", lim)
    return value
