"""Host-side schedule arrays (egom2p_amd/scheduler.py) against the reference's own functions' outputs
(tests/golden/schedules.npz, made by oracle/make_goldens_schedules.py from egom2p/utils/scheduler.py)."""
import ast
import os
from types import SimpleNamespace

import numpy as np
import pytest

from egom2p_amd import scheduler as S

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "schedules.npz"), allow_pickle=False)


@pytest.mark.parametrize("name", [k for k in GOLD.files if not k.endswith(".args")])
def test_schedule_equals_the_reference(name):
    fn, kw = ast.literal_eval(str(GOLD[name + ".args"]))
    got = getattr(S, fn)(**kw)
    assert got.shape == GOLD[name].shape
    assert np.array_equal(got, GOLD[name])                     # same numpy expressions: bit for bit


def test_build_schedules_frozen_phase_then_main():
    a = SimpleNamespace(lr=1e-3, min_lr=1e-6, frozen_model_lr=2e-4, weight_decay=0.05, weight_decay_end=0.01, scheduler="cosine",
                        warmup_epochs=1, warmup_steps=-1, cooldown_epochs=0, cooldown_steps=-1, frozen_model_epochs=2, epochs=6)
    lr, wd = S.build_schedules(a, 10)
    assert len(lr) == len(wd) == 60
    assert np.all(lr[:20] == 2e-4) and np.all(wd[:20] == 0.05)                       # constant frozen phase (:524-531)
    assert np.array_equal(lr[20:], S.cosine_scheduler(1e-3, 1e-6, 4, 10, warmup_epochs=1))
    assert np.array_equal(wd[20:], S.cosine_scheduler(0.05, 0.01, 4, 10))
    a.scheduler, a.cooldown_steps, a.frozen_model_epochs = "inverse_sqrt-500", 5, 0
    lr, wd = S.build_schedules(a, 10)
    assert np.array_equal(lr, S.inverse_sqrt_scheduler(1e-3, 1e-6, 6, 10, warmup_epochs=1, cooldown_steps=5, timescale=500))
    assert np.array_equal(wd, S.inverse_sqrt_scheduler(0.05, 0.01, 6, 10, cooldown_steps=5, timescale=500))
    a.scheduler = "linear"
    with pytest.raises(NotImplementedError):
        S.build_schedules(a, 10)
