"""SURVEY section 8 row f4, pinned: the host restatement of the reference's token-budget sampler
(`synth.masking_budgets_host`, the checker of the device kernel) follows the law of the REAL reference's
`UnifiedMasking.input_token_budget` / `target_token_budget` draws (tests/golden/budget_stats.npz)."""
import numpy as np
import pytest

from conftest import load_golden
from _budget_check import check_against_reference
from egom2p_amd import synth
from egom2p_amd.config import MODEL_CFGS


@pytest.mark.parametrize("case", ["fixed", "ranged"])
def test_host_budget_sampler_follows_the_reference_draws(case):
    g, _ = load_golden("budget_stats")
    cfg = MODEL_CFGS["egom2p_base_12e_12d_swiglu_nobias"]
    assert [m.name for m in cfg.mods] == [str(x) for x in g["mods"]]
    (lo_i, hi_i), (lo_t, hi_t) = g[f"{case}.range"]
    k_in, k_tg = synth.masking_budgets_host(cfg, 24000, (int(lo_i), int(hi_i)), (int(lo_t), int(hi_t)), seed=11)
    rep = check_against_reference(g, case, k_in, k_tg)
    print(case, {k: np.round(v["ks"], 4) for k, v in rep.items()})
