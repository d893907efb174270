"""ORACLE tooling - outputs of the reference's OWN schedule functions (egom2p/utils/scheduler.py, imported by path: pure
numpy) for a handful of argument sets -> tests/golden/schedules.npz.  Runs only in the build container.

    python oracle/make_goldens_schedules.py
"""
import importlib.util
import io
import os
from contextlib import redirect_stdout

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("EGOM2P_REFERENCE", "/root/reference")

CASES = {
    "cos_a": ("cosine_scheduler", dict(base_value=1e-3, final_value=1e-6, epochs=5, niter_per_ep=37, warmup_epochs=1)),
    "cos_b": ("cosine_scheduler", dict(base_value=4e-4, final_value=0.0, epochs=3, niter_per_ep=50, warmup_steps=17)),
    "cos_wd": ("cosine_scheduler", dict(base_value=0.05, final_value=0.01, epochs=4, niter_per_ep=25)),
    "const": ("constant_scheduler", dict(base_value=2e-4, epochs=2, niter_per_ep=31)),
    "isq_a": ("inverse_sqrt_scheduler", dict(base_value=1e-3, final_value=1e-5, epochs=6, niter_per_ep=40, warmup_steps=20, cooldown_steps=30, timescale=100)),
    "isq_b": ("inverse_sqrt_scheduler", dict(base_value=5e-4, final_value=0.0, epochs=4, niter_per_ep=25, warmup_epochs=1, cooldown_epochs=1, timescale=10000)),
    "isq_wd": ("inverse_sqrt_scheduler", dict(base_value=0.05, final_value=0.05, epochs=3, niter_per_ep=20, cooldown_steps=10, timescale=10000)),
}


def main():
    spec = importlib.util.spec_from_file_location("ref_scheduler", os.path.join(REF, "egom2p/utils/scheduler.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    gold = {}
    with redirect_stdout(io.StringIO()):
        for name, (fn, kw) in CASES.items():
            gold[name] = np.asarray(getattr(mod, fn)(**kw), dtype=np.float64)
            gold[name + ".args"] = np.array(repr((fn, kw)))
    path = os.path.join(ROOT, "tests", "golden", "schedules.npz")
    np.savez_compressed(path, **gold)
    print(f"[goldens] schedules -> {path} ({os.path.getsize(path) / 1e3:.1f} kB)")


if __name__ == "__main__":
    main()
