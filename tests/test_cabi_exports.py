"""CPU-side checks of the drop-in boundary: the C-ABI library builds, loads, and exports exactly the
entry points include/egom2p_hip.h declares (no compute is launched without a GPU)."""
import os
import re
import subprocess

from egom2p_amd import _lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "egom2p_hip.h")).read()
    return set(re.findall(r"^\s*(?:int|long)\s+(ego_\w+)\s*\(", text, flags=re.M))


def test_header_and_binding_agree():
    assert _declared() == set(L.EXPORTS)


def test_library_exports_every_declared_symbol():
    if not os.path.exists(L.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    lib = L.load()
    for name in _declared():
        assert hasattr(lib, name), name
    out = subprocess.run(["nm", "-D", "--defined-only", L.LIB_PATH], capture_output=True, text=True).stdout
    exported = set(re.findall(r" T (ego_\w+)", out))
    assert _declared() <= exported
    hdr = open(os.path.join(ROOT, "include", "egom2p_hip.h")).read()
    assert lib.ego_abi_version() == L.ABI_VERSION == int(re.search(r"#define EGO_ABI_VERSION (\d+)", hdr).group(1))


def test_no_oracle_import_in_product_path():
    """The product package must never import the oracle (a fallback would void parity claims)."""
    pkg = os.path.join(ROOT, "egom2p_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, f


def test_profiler_cost_functions_accept_the_ops_signatures():
    """bench.py wraps every `ops` entry with a cost function of the same signature: a keyword added to an op but not to
    its cost function only fails on the GPU box, inside the bench - check the binding here."""
    import inspect
    from unittest import mock
    import torch
    from egom2p_amd import ops
    from egom2p_amd.profiler import KernelTimer
    kt = KernelTimer()
    with mock.patch.object(torch.cuda, "Event", lambda **k: None):
        with kt.capture(12):
            table = dict(kt.cost_table)
    for name, cost in table.items():
        op = inspect.signature(getattr(ops, name))
        cs = inspect.signature(cost)
        var_pos = any(p.kind == p.VAR_POSITIONAL for p in cs.parameters.values())
        var_kw = any(p.kind == p.VAR_KEYWORD for p in cs.parameters.values())
        n_pos_op = sum(1 for p in op.parameters.values() if p.default is p.empty)
        n_pos_cost = sum(1 for p in cs.parameters.values() if p.kind in (p.POSITIONAL_ONLY, p.POSITIONAL_OR_KEYWORD))
        assert var_pos or n_pos_cost >= n_pos_op, f"cost function of ops.{name} takes fewer positional arguments than the op"
        for n, p in op.parameters.items():                   # parameters with defaults are the ones callers pass by keyword
            if p.default is not p.empty:
                assert n in cs.parameters or var_kw, f"profiler cost function of ops.{name} does not accept keyword '{n}'"


def test_bench_counts_the_flops_the_survey_states():
    """bench.py's `algorithmic_tflops_per_gpu` / `mfma_frac_end_to_end` come from `flops_per_clip`: for the canonical ego-b
    split it must be SURVEY.md section 8(d)'s block-sparse count (1.397 TF forward, 4.19 TF forward + backward per clip =
    406.9 MFLOP per clip position) - a reporting bug here would move every efficiency figure."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    from egom2p_amd import synth
    from egom2p_amd.config import MODEL_CFGS
    f = bench.flops_per_clip(MODEL_CFGS["egom2p_base_12e_12d_swiglu_nobias"], synth.CANONICAL_BUDGETS, 2048, 2048)
    assert abs(f / 1.397e12 - 1.0) < 1e-3
    assert abs(3.0 * f / 10300 / 406.9e6 - 1.0) < 1e-3


def test_head_storage_layouts():
    """Host logic of the padded storage (engine.head_layout): heads of 64 unchanged (even counts only); other head dimensions
    padded to 96 / 128 with as many all-zero phantom heads as make the row width a multiple of 128 - the narrowest rows win."""
    import pytest
    from egom2p_amd.engine import head_layout
    assert head_layout(12, 64) == (64, 12) and head_layout(18, 64) == (64, 18)          # ego-b, ego-L (D = 1152)
    assert head_layout(15, 68) == (96, 16)              # registered ego-L (egom2p_model.py:1080-1092): 1536 wide, one phantom head
    assert head_layout(15, 68, min_pad=128) == (128, 15)                                # EGOM2P_HEAD_PAD=128: the round-3 layout
    assert head_layout(31, 66) == (96, 32)              # registered ego-XL (:1100-1118)
    assert head_layout(16, 120) == (128, 16) and head_layout(4, 96) == (96, 4)
    for H, HD in ((15, 64), (4, 200)):
        with pytest.raises(L.EgoHipError):
            head_layout(H, HD)
    for H in range(1, 40):                              # whatever the head count: rows are multiples of 128, at most 8 phantom heads
        for HD in (66, 68, 80, 96, 100, 128):
            hdp, hs = head_layout(H, HD)
            assert hdp >= HD and hs >= H and hs - H <= 8 and (hdp * hs) % 128 == 0


def test_top_k_count_follows_the_reference_argument_forms():
    """`top_k_top_p_filtering` (egom2p/models/generate.py:335-342): an int top_k is a count, a float a share of the vocabulary
    (`int(top_k * V)`), both capped at V; 0 / 0.0 switch the filter off; a share that keeps no token raises there (torch.topk(., 0)[..., -1])."""
    import pytest
    from egom2p_amd import ops
    assert ops.top_k_count(0, 64000) == 0 and ops.top_k_count(0.0, 64000) == 0 and ops.top_k_count(None, 64000) == 0
    assert ops.top_k_count(50, 64000) == 50 and ops.top_k_count(70000, 64000) == 64000
    assert ops.top_k_count(0.001, 64000) == 64 and ops.top_k_count(0.5, 256) == 128 and ops.top_k_count(1.0, 256) == 256
    with pytest.raises(ValueError):
        ops.top_k_count(1e-9, 64000)
    with pytest.raises(ValueError):
        ops.top_k_count("3", 64000)
