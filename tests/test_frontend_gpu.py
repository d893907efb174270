"""Front-end parity on the GPU: compaction / embedding gather / loss permutation / embedding backward
through the C-ABI against the CPU oracle (integer outputs bit-exact, fp32 rows exact)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from egom2p_amd import ops, synth  # noqa: E402
from egom2p_amd.config import MODEL_CFGS  # noqa: E402
from egom2p_amd.posemb import build_pos_emb  # noqa: E402
from oracle import egom2p_oracle as O  # noqa: E402

DEV = "cuda"


def _alloc(B, n_keep, n_mods):
    i32 = dict(device=DEV, dtype=torch.int32)
    return dict(ids_keep=torch.full((B, n_keep), -7, device=DEV, dtype=torch.int64),
                pad=torch.full((B, n_keep), 9, device=DEV, dtype=torch.uint8),
                mod_mask=torch.full((B, n_keep), 99, device=DEV, dtype=torch.int16),
                slot=torch.full((B, n_keep), 99, **i32), local=torch.full((B, n_keep), -9, **i32),
                tok=torch.full((B, n_keep), -9, **i32), ks=torch.full((B, n_keep), -9, **i32),
                ke=torch.full((B, n_keep), -9, **i32), n_valid=torch.zeros(B, **i32),
                seg=torch.zeros(B, n_mods, 2, **i32), err=torch.zeros(1, **i32))


def _run_compact(cfg, md, mods, n_keep, is_decoder):
    B = md[mods[0].name]["input_mask"].shape[0]
    key = "target_mask" if is_decoder else "input_mask"
    masks = [md[m.name][key].to(DEV).contiguous() for m in mods]
    ids = [md[m.name]["tensor"].reshape(B, -1).to(DEV).contiguous() for m in mods]
    dams = [md[m.name]["decoder_attention_mask"].to(DEV).contiguous() for m in mods] if is_decoder else None
    out = _alloc(B, n_keep, len(mods))
    ops.compact(masks, ids, dams, [m.max_tokens for m in mods], [m.id for m in mods], n_keep, is_decoder, out, B)
    torch.cuda.synchronize()
    return out


@pytest.mark.parametrize("cfg_name,B,n,budgets", [
    ("ego_tiny_2e_2d", 4, 32, {"tok_cam": [(10, 12), (3, 0), (15, 15), (0, 7)], "tok_gaze": [(9, 5), (20, 10), (1, 29), (16, 0)]}),
    ("ego_tiny8_2e_2d", 3, 64, {n: ([(4, 0), (0, 6), (3, 3)] if n == "tok_aux3" else [(5 + (b + j) % 7, 3 + (2 * b + j) % 5) for b in range(3)])
                                for j, n in enumerate(("tok_cam", "tok_gaze") + tuple(f"tok_aux{i}" for i in range(6)))}),   # EGO_MAX_MODS modalities
    ("ego_b_2e_2d", 3, 2048, None),
    ("ego_b_2e_2d", 4, 2048, "dirichlet"),
    ("ego_b_2e_2d", 2, 700, None),          # truncation: more valid tokens than kept
])
def test_compact_bit_exact(cfg_name, B, n, budgets):
    cfg = MODEL_CFGS[cfg_name]
    if budgets == "dirichlet":
        budgets = synth.dirichlet_budgets(cfg, B, n, n, seed=7)
    md = synth.make_clip_batch(cfg, B, budgets, seed=7)
    mods = cfg.mods
    # encoder side
    out = _run_compact(cfg, md, mods, n, False)
    ce = O.compact_encoder(md, mods, n)
    assert np.array_equal(out["ids_keep"].cpu().numpy(), ce["ids_keep"])
    assert np.array_equal(out["pad"].cpu().numpy().astype(bool), ce["pad"])
    assert np.array_equal(out["mod_mask"].cpu().numpy(), ce["mod_mask"])
    valid = ~ce["pad"]
    assert np.array_equal(out["tok"].cpu().numpy()[valid], ce["tok"][valid])
    assert np.array_equal(out["local"].cpu().numpy()[valid], ce["local"][valid])
    assert np.array_equal(out["slot"].cpu().numpy()[valid], ce["slot"][valid])
    assert np.array_equal(out["n_valid"].cpu().numpy(), valid.sum(1))
    assert (out["ks"].cpu().numpy() == 0).all()
    assert np.array_equal(out["ke"].cpu().numpy(), np.repeat(valid.sum(1)[:, None], n, 1))
    # decoder side, shuffled modality order
    order = list(reversed(mods)) if len(mods) == 2 else [mods[2], mods[0], mods[3], mods[1]]
    out = _run_compact(cfg, md, order, n, True)
    cd = O.compact_decoder(md, order, n)
    assert np.array_equal(out["ids_keep"].cpu().numpy(), cd["ids_keep"])
    assert np.array_equal(out["pad"].cpu().numpy().astype(bool), cd["pad"])
    assert np.array_equal(out["mod_mask"].cpu().numpy(), cd["mod_mask"])
    assert np.array_equal(out["tok"].cpu().numpy(), cd["target_ids"])
    ks, ke, ok = O.attention_ranges(cd["dam"], cd["mod_mask_pre"], cd["pad"])
    assert ok and out["err"].item() == 0
    gks, gke = out["ks"].cpu().numpy(), out["ke"].cpu().numpy()
    nonempty = ke > ks
    assert np.array_equal(gks[nonempty], ks[nonempty]) and np.array_equal(gke[nonempty], ke[nonempty])
    assert (gke[~nonempty] <= gks[~nonempty]).all()


def test_embed_and_backward_and_perm():
    cfg = MODEL_CFGS["ego_b_2e_2d"]
    B, n = 3, 512
    budgets = synth.dirichlet_budgets(cfg, B, n, n, seed=3)
    md = synth.make_clip_batch(cfg, B, budgets, seed=3)
    mods = cfg.mods
    D = cfg.dim
    tables = [synth.normal(f"t{m.name}", (m.vocab_size, D), 0.02).to(DEV) for m in mods]
    pos = [build_pos_emb(m, D)[0].to(DEV).contiguous() for m in mods]
    modv = [synth.normal(f"m{m.name}", (D,), 0.02).to(DEV) for m in mods]
    out = _run_compact(cfg, md, mods, n, False)
    x = torch.empty(B * n, D, device=DEV)
    emb = torch.empty(B * n, D, device=DEV)
    ops.embed_fwd(tables, pos, modv, None, out["slot"], out["local"], out["tok"], x, emb, B * n, D)
    slot = out["slot"].view(-1).long()
    ref_e = torch.zeros(B * n, D, device=DEV)
    ref_x = torch.zeros(B * n, D, device=DEV)
    for i in range(len(mods)):
        sel = slot == i
        e = pos[i][out["local"].view(-1)[sel].long()] + modv[i]
        ref_e[sel] = e
        ref_x[sel] = tables[i][out["tok"].view(-1)[sel].long()] + e
    assert torch.equal(emb, ref_e) and torch.equal(x, ref_x)

    # backward: scatter-add into tables, modality sums, base vector
    dx = torch.randn(B * n, D, device=DEV)
    d2 = torch.randn(B * n, D, device=DEV)
    dtab = [torch.zeros_like(t) for t in tables]
    dmod = [torch.zeros(D, device=DEV) for _ in mods]
    dbase = torch.zeros(D, device=DEV)
    ops.embed_bwd(dtab, dmod, dbase, dx, d2, out["slot"], out["tok"], B * n, D)
    for i in range(len(mods)):
        sel = slot == i
        ref_t = torch.zeros_like(tables[i]).index_add_(0, out["tok"].view(-1)[sel].long(), dx[sel])
        assert (dtab[i] - ref_t).abs().max().item() < 1e-4
        assert (dmod[i] - (dx[sel] + d2[sel]).sum(0)).abs().max().item() < 2e-3
    assert (dbase - dx[slot >= 0].sum(0)).abs().max().item() < 2e-3

    # loss permutation on the decoder side
    order = [mods[2], mods[0], mods[3], mods[1]]
    outd = _run_compact(cfg, md, order, n, True)
    canon = torch.tensor([mods.index(m) for m in order], device=DEV, dtype=torch.int32)
    perm = torch.empty(B * n, device=DEV, dtype=torch.int32)
    tperm = torch.full((B * n,), -1, device=DEV, dtype=torch.int32)
    ranges = torch.zeros(len(mods), 2, device=DEV, dtype=torch.int32)
    base = torch.zeros(B, len(mods), device=DEV, dtype=torch.int32)
    ops.loss_perm(outd["seg"], canon, outd["slot"], outd["tok"], B, n, len(mods), perm, tperm, ranges, base)
    mm = outd["mod_mask"].view(-1).cpu().numpy()
    tg = outd["tok"].view(-1).cpu().numpy()
    p = perm.cpu().numpy()
    off = 0
    for c, m in enumerate(mods):
        rows = np.flatnonzero(mm == m.id)                    # row order of y[decoder_mod_mask == id]
        assert ranges[c, 0].item() == off and ranges[c, 1].item() == rows.size
        assert np.array_equal(p[rows], off + np.arange(rows.size))
        assert np.array_equal(tperm.cpu().numpy()[off:off + rows.size], tg[rows])
        off += rows.size
    assert (p[mm == -1] == -1).all()


def test_clip_synth_on_device_is_bit_identical_to_host():
    """the device generator of the input contract (masking.py:236-266 equivalent) reproduces synth.make_clip_batch"""
    from egom2p_amd import synth
    from egom2p_amd.config import MODEL_CFGS
    cfg = MODEL_CFGS["egom2p_base_12e_12d_swiglu_nobias"]
    for budgets, seed, off in ((None, 3, 0), (synth.dirichlet_budgets(cfg, 3, 2048, 2048, 9), 9, 5)):
        B = 3
        host = synth.make_clip_batch(cfg, B, budgets, seed=seed, sample_offset=off)
        dev = synth.make_clip_batch_device(cfg, B, budgets, seed=seed, sample_offset=off)
        for m in cfg.mods:
            for k in ("tensor", "input_mask", "target_mask", "decoder_attention_mask"):
                assert torch.equal(dev[m.name][k].cpu(), host[m.name][k]), (m.name, k)


@pytest.mark.parametrize("case", ["fixed", "ranged"])
def test_device_budget_sampler_follows_the_reference_draws(case):
    """SURVEY section 8 row f4, pinned to the reference: `ego_budget_dirichlet` against 24,000 draws of the REAL
    `UnifiedMasking.input_token_budget` / `target_token_budget` (tests/golden/budget_stats.npz, made by
    oracle/make_goldens_masking.py): per-draw invariants, means / deviations / P(0) / P(cap) / one-hot share and the
    Kolmogorov-Smirnov distance per modality, for fixed (2048 / 2048) and ranged token counts (bars: tests/_budget_check.py)."""
    from conftest import load_golden
    from _budget_check import check_against_reference
    g, _ = load_golden("budget_stats")
    cfg = MODEL_CFGS["egom2p_base_12e_12d_swiglu_nobias"]
    (lo_i, hi_i), (lo_t, hi_t) = g[f"{case}.range"]
    k_in, k_tg = synth.masking_budgets_device(cfg, 24000, (int(lo_i), int(hi_i)), (int(lo_t), int(hi_t)), seed=13)
    torch.cuda.synchronize()
    rep = check_against_reference(g, case, k_in.cpu().numpy(), k_tg.cpu().numpy())
    print(case, {k: np.round(v["ks"], 4) for k, v in rep.items()})


def test_device_budget_sampler_follows_the_reference_mixture():
    """the Dirichlet-mixture token budgets of `UnifiedMasking` (masking.py:181-234, :530-541)
    sampled on the device.  Checked against the host restatement of the same algorithm (independent random stream):
    per-sample invariants exactly, distribution by moments and by the share of near-one-hot draws (alpha 0.01 / 0.1
    components put almost every token on one modality until it hits its cap)."""
    cfg = MODEL_CFGS["ego_b_2e_2d"]
    B, N = 20000, 2048
    k_in, k_tg = synth.masking_budgets_device(cfg, B, (N, N), (N, N), seed=5)
    torch.cuda.synchronize()
    di, dt = k_in.cpu().numpy().astype(np.int64), k_tg.cpu().numpy().astype(np.int64)
    hi, ht = synth.masking_budgets_host(cfg, B, (N, N), (N, N), seed=5)
    cap = np.array([m.max_tokens for m in cfg.mods])[:, None]
    for a, t in ((di, dt), (hi, ht)):
        assert (a >= 0).all() and (t >= 0).all() and (a <= cap).all() and (a + t <= cap).all()
        assert (a.sum(0) <= N).all() and (t.sum(0) <= N).all()
        # the sum is exact unless a clamp took tokens away (then some modality sits at its cap)
        short = a.sum(0) < N
        assert ((a == cap).any(0) | ~short).all()
    # moments per modality (rgb / depth share the law, so do cam / gaze): 20,000 samples each side
    for arr_d, arr_h in ((di, hi), (dt, ht)):
        md, mh = arr_d.mean(1), arr_h.mean(1)
        sd_, sh = arr_d.std(1), arr_h.std(1)
        assert np.all(np.abs(md - mh) < 0.03 * mh + 0.5), (md, mh)
        assert np.all(np.abs(sd_ - sh) < 0.05 * sh + 0.5), (sd_, sh)
        # share of clips whose budget is (almost) all on one modality
        one_d = ((arr_d.max(0) >= 0.95 * arr_d.sum(0).clip(min=1))).mean()
        one_h = ((arr_h.max(0) >= 0.95 * arr_h.sum(0).clip(min=1))).mean()
        assert abs(one_d - one_h) < 0.02, (one_d, one_h)
    # budgets -> masks -> compaction on the device: a clip batch with nothing but stream keys from the host
    md = synth.make_clip_batch_device_masked(cfg, 6, N, N, seed=9)
    for j, m in enumerate(cfg.mods):
        kin = (~md[m.name]["input_mask"]).sum(1).cpu().numpy()
        ktg = (~md[m.name]["target_mask"]).sum(1).cpu().numpy()
        overlap = (~md[m.name]["input_mask"] & ~md[m.name]["target_mask"]).sum().item()
        assert overlap == 0 and (kin + ktg <= m.max_tokens).all()
        assert (md[m.name]["decoder_attention_mask"].sum(1).cpu().numpy() == ktg).all()
    eng_out = _run_compact(cfg, {k: {kk: vv.cpu() for kk, vv in v.items()} for k, v in md.items()}, cfg.mods, N, True)
    assert eng_out["err"].item() == 0


def test_row_list_kernels_and_sparse_exchange_single_rank():
    """The ego_rows_* kernels under dp.SparseTableExchange on the GPU: a one-rank exchange must
    leave the table as it was, the row list is the ascending set of flagged rows, and the embedding backward sets the
    flags of exactly the token rows it adds to."""
    from egom2p_amd.dp import SparseTableExchange
    V, D, cap = 64000, 768, 5000
    g = torch.zeros(V, D, device=DEV)
    touched = torch.zeros(V, dtype=torch.uint8, device=DEV)
    rows = torch.randperm(V, device=DEV)[:3000]
    g[rows] = torch.randn(3000, D, device=DEV)
    touched[rows] = 1
    ref = g.clone()
    ex = SparseTableExchange([(g, touched)], cap_rows=cap)
    ex.exchange()
    torch.cuda.synchronize()
    t = ex.tables[0]
    n = int(t["count"].item())
    assert n == 3000 and torch.equal(t["rows"][:n].long(), torch.sort(rows).values) and (t["rows"][n:] == -1).all()
    assert torch.equal(g, ref) and int(touched.sum()) == 0
    assert torch.equal(t["send"][:n], ref[t["rows"][:n].long()]) and float(t["send"][n:].abs().max()) == 0.0
    # overflow is reported
    touched[:] = 1
    small = SparseTableExchange([(g, touched)], cap_rows=100)
    small.exchange()
    assert small.overflowed()
    # flags from the embedding backward
    cfg = MODEL_CFGS["ego_b_2e_2d"]
    from egom2p_amd.engine import Engine
    eng = Engine(cfg, "cuda:0", max_batch=2, n_enc=256, n_dec=256)
    eng.init_random(1)
    tabs = eng.track_touched_table_rows(True)
    md = synth.make_clip_batch(cfg, 2, None, seed=4)
    mdg = {k: {kk: vv.cuda() for kk, vv in v.items()} for k, v in md.items()}
    eng.forward(mdg)
    eng.zero_grad()
    eng.backward(1.0)
    torch.cuda.synchronize()
    for (gt, fl), m in zip(tabs, cfg.mods):
        nz = (gt.abs().sum(1) > 0)
        assert torch.equal(fl.bool() | ~nz, torch.ones_like(nz))              # every row with a gradient is flagged
        slot_rows = eng.ce["tok"][:2][eng.ce["slot"][:2] == cfg.mods.index(m)].long().unique()
        assert torch.equal(torch.nonzero(fl).flatten(), slot_rows)            # and exactly the kept token ids are


def test_unified_masking_on_real_token_tensors():
    """egom2p_amd.masking.UnifiedMasking: the reference's `UnifiedMasking.__call__` (data/masking.py:519-560, image_mask
    :236-266) for a batch of EXISTING token tensors on the device - the contract of every returned entry, per clip and
    modality, and that the engine trains on it."""
    from egom2p_amd.engine import Engine
    from egom2p_amd.masking import UnifiedMasking
    from egom2p_amd.model import MODALITY_INFO
    cfg = MODEL_CFGS["ego_b_2e_2d"]
    B = 12
    info = {m.name: MODALITY_INFO[m.name] for m in cfg.mods}
    toks = {m.name: synth.randint(f"um.{m.name}", (B,) + (m.grid if m.kind == "video" else (m.max_tokens,)), m.vocab_size, seed=5).cuda()
            for m in cfg.mods}
    masker = UnifiedMasking(info, None, input_tokens_range=(1024, 2048), target_tokens_range=2048, seed=7)
    md = masker(toks)
    n_in_tot, n_tg_tot = np.zeros(B, np.int64), np.zeros(B, np.int64)
    for m in cfg.mods:
        e = md[m.name]
        assert e["tensor"] is toks[m.name]                                    # tokens pass through untouched
        im, tm, dam = e["input_mask"].cpu().numpy(), e["target_mask"].cpu().numpy(), e["decoder_attention_mask"].cpu().numpy()
        assert im.shape == tm.shape == dam.shape == (B, m.max_tokens) and im.dtype == bool
        assert not ((~im) & (~tm)).any()                                      # inputs and targets are disjoint
        for b in range(B):
            k_tg = int((~tm[b]).sum())
            n_in_tot[b] += int((~im[b]).sum()); n_tg_tot[b] += k_tg
            nz = np.nonzero(dam[b])[0]
            if k_tg == 0:
                assert nz.size == 0
            else:                                                             # target count at the first target position (:262-264)
                assert nz.tolist() == [int(np.argmax(~tm[b]))] and dam[b, nz[0]] == k_tg
    assert (n_in_tot >= 1) .all() and (n_in_tot <= 2048).all() and (n_tg_tot <= 2048).all()
    assert n_in_tot.min() < n_in_tot.max()                                    # the token count itself is drawn from the range
    # a second call masks differently, a second instance with the same seed reproduces the first call
    md2 = masker(toks)
    assert not torch.equal(md2["tok_rgb"]["input_mask"], md["tok_rgb"]["input_mask"])
    again = UnifiedMasking(info, None, input_tokens_range=(1024, 2048), target_tokens_range=2048, seed=7)(toks)
    assert all(torch.equal(again[k]["input_mask"], md[k]["input_mask"]) and torch.equal(again[k]["target_mask"], md[k]["target_mask"]) for k in md)
    eng = Engine(cfg, "cuda:0", max_batch=B, n_enc=2048, n_dec=2048)
    eng.init_random(3)
    loss, mod_loss = eng.forward(md, dec_order=[m.name for m in cfg.mods])
    eng.backward(1.0)
    assert np.isfinite(float(loss)) and 5.0 < float(loss) < 12.0 and torch.isfinite(eng.G).all()
    with pytest.raises(NotImplementedError):
        UnifiedMasking({"caption": {"type": "seq", "max_tokens": 256}}, None, 128, 128)


def test_embed_bwd_skewed_tokens_and_bitwise_reproducible():
    """The atomic-free embedding backward (csrc/embed.hip: embed_sums_kernel + embed_tables_kernel) on a skewed batch:
    one token carries 6000 rows (whole-workgroup path, several flushes of its owner), a few carry 64-300, the rest are
    random; padded rows (slot -1) in between.  Against index_add_ within fp32 summation-order noise, and twice bit for bit."""
    torch.manual_seed(3)
    D, rows, n_mods, V = 768, 40000, 4, [64000, 64000, 256, 256]
    slot = torch.randint(0, n_mods, (rows,), device=DEV, dtype=torch.int32)
    slot[torch.rand(rows, device=DEV) < 0.05] = -1
    tok = torch.zeros(rows, device=DEV, dtype=torch.int32)
    for m in range(n_mods):
        sel = slot == m
        tok[sel] = torch.randint(0, V[m], (int(sel.sum()),), device=DEV, dtype=torch.int32)
    idx0 = (slot == 0).nonzero()[:, 0]
    tok[idx0[:6000]] = 513                         # heavy key: 6000 rows of table 0
    tok[idx0[6000:6300]] = 1025                    # same owner workgroup (token id mod 512 == 1)
    tok[idx0[6300:6364]] = 7
    dx = torch.randn(rows, D, device=DEV)
    d2 = torch.randn(rows, D, device=DEV)
    outs = []
    for rep in range(2):
        dtab = [torch.zeros(v, D, device=DEV) for v in V]
        dtab[3] = None                             # a modality without a table gradient
        dmod = [torch.zeros(D, device=DEV) for _ in range(n_mods)]
        dbase = torch.zeros(D, device=DEV)
        touched = [torch.zeros(v, device=DEV, dtype=torch.uint8) if t is not None else None for v, t in zip(V, dtab)]
        ops.embed_bwd(dtab, dmod, dbase, dx, d2, slot, tok, rows, D, touched=touched)
        ops.embed_bwd(dtab, dmod, dbase, dx, d2, slot, tok, rows, D, touched=touched)      # gradients ACCUMULATE
        torch.cuda.synchronize()
        outs.append((dtab, dmod, dbase, touched))
    dtab, dmod, dbase, touched = outs[0]
    for m in range(n_mods):
        sel = slot == m
        if dtab[m] is not None:
            ref = torch.zeros(V[m], D, device=DEV, dtype=torch.float64).index_add_(0, tok[sel].long(), dx[sel].double()) * 2
            err = (dtab[m].double() - ref).abs().max().item()
            assert err < 2e-4 * max(1.0, ref.abs().max().item() / 50), (m, err)
            assert torch.equal(touched[m].bool(), ref.abs().sum(1) > 0)
        refm = (dx[sel].double() + d2[sel].double()).sum(0) * 2
        assert (dmod[m].double() - refm).abs().max().item() < 1e-3 * max(1.0, refm.abs().max().item() / 50), m
    refb = dx[slot >= 0].double().sum(0) * 2
    assert (dbase.double() - refb).abs().max().item() < 1e-3 * max(1.0, refb.abs().max().item() / 50)
    for a, b in zip(outs[0][:3], outs[1][:3]):
        for x, y in zip(a if isinstance(a, list) else [a], b if isinstance(b, list) else [b]):
            if x is not None:
                assert torch.equal(x, y)
