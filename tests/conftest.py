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
