"""End-to-end parity of the HIP engine against outputs of the REAL reference model (tests/golden/*.npz,
made by oracle/make_goldens.py from the reference's own EgoM2P in fp32).

Bars (north_star: integer/index outputs bit-exact; floats within 1e-3 rel of the reference loss; the
engine computes GEMMs/attention in bf16 like autocast(bf16), so activation/gradient taps are compared
by relative L2 with a bf16-sized tolerance, stated per tap below).
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from conftest import bar, load_golden, rel_l2  # noqa: E402
from egom2p_amd import _lib as L  # noqa: E402
from egom2p_amd import ops, synth  # noqa: E402
from egom2p_amd.config import MODEL_CFGS  # noqa: E402
from egom2p_amd.engine import Engine  # noqa: E402

LOSS_RTOL = 1e-3        # north_star: outputs within 1e-3 rel of reference
ACT_TOL = 2e-2          # bf16 GEMM/attention vs fp32 reference, relative L2 over a tap
GRAD_TOL = 4e-2         # same for gradients (two bf16 passes)


def _setup(case):
    case, _, head_pad = case.partition("@")              # "L1020@128": the same fixture on the heads-of-128 storage layout
    g, meta = load_golden(case)
    cfg = MODEL_CFGS[meta["cfg"]]
    sd = synth.build_state_dict(cfg, meta["seed"])
    md = synth.make_clip_batch(cfg, meta["batch"], meta["budgets"], meta["seed"])
    import os
    old = os.environ.get("EGOM2P_HEAD_PAD")
    if head_pad:
        os.environ["EGOM2P_HEAD_PAD"] = head_pad
    try:
        eng = Engine(cfg, "cuda:0", max_batch=meta["batch"], n_enc=meta["n_enc"], n_dec=meta["n_dec"])
    finally:
        if head_pad:
            os.environ.pop("EGOM2P_HEAD_PAD") if old is None else os.environ.__setitem__("EGOM2P_HEAD_PAD", old)
    if head_pad:
        assert eng.HDP == int(head_pad) and eng.Hs == eng.H
    eng.load_state_dict(sd)
    mdg = {k: {kk: vv.cuda() for kk, vv in v.items()} for k, v in md.items()}
    return g, meta, cfg, sd, mdg, eng


def _tap(g, key, t):
    """compare tensor t (B,n,D) against the golden tap (full tensor, or head/tail slices + row norms)."""
    t = t.float().cpu()
    if f"tap.{key}" in g.files:
        return rel_l2(t.numpy(), g[f"tap.{key}"])
    e1 = rel_l2(t[:, :6, :24].numpy(), g[f"tap_head.{key}"])
    e2 = rel_l2(t[:, -4:, -16:].numpy(), g[f"tap_tail.{key}"])
    e3 = rel_l2(t.double().norm(dim=-1).numpy(), g[f"tap_rownorm.{key}"])
    return max(e1, e2, e3)


# L1020 = the REGISTERED ego-L geometry (dim 1020, 15 heads of 68, F = 2720: egom2p_model.py:1080-1092) at 2 + 2 layers: stored in rows
# of 1024 with 16 heads of 96 (15 padded heads + one all-zero phantom head: engine.py; "@128" = the round-3 layout, 15 heads of 128) -
# the pad columns must stay exact zeros in activations and gradients
# XL2046 = the registered ego-XL geometry (dim 2046, 31 heads of 66, F = 5456) at 1 + 1 layers, same storage scheme (32 heads of 96)
# b2_reg4 = ego-b width with FOUR REGISTER TOKENS (egom2p_model.py:170-171, 381-387): R rows in front of every sample's encoder tokens
@pytest.mark.parametrize("case", ["tiny", "tiny_pad", "tiny8", "b2", "b2_ragged", "b2_untied", "b2_reg4", "b12", "L2", "L24", "L1020", "L1020@128", "XL2046"])
def test_engine_matches_reference(case):
    g, meta, cfg, sd, md, eng = _setup(case)
    B, N, M, D = meta["batch"], meta["n_enc"], meta["n_dec"], cfg.dim
    R = eng.R
    assert R == cfg.num_register_tokens and eng.Ne == N + R
    Ne = N + R                                           # encoder rows per sample: the reference's encoder_tokens.shape[1]
    Dp = eng.D                                           # row pitch of the engine's activations (== D unless stored padded)

    def act(t, rows, n):
        v = t[:rows].view(B, n, Dp)
        assert Dp == D or not bool(v[..., D:].any()), "pad columns must be zero"
        return v[..., :D]
    order = [str(x) for x in g["dec_order"]]
    loss, mod_loss = eng.forward(md, dec_order=order)
    torch.cuda.synchronize()

    # ---- integer / index outputs: bit-exact with the reference
    assert np.array_equal(eng.ce["ids_keep"][:B, R:].cpu().numpy(), g["enc_ids_keep"])    # (the reference's ids_keep has no register entries)
    assert np.array_equal(eng.cd["ids_keep"][:B].cpu().numpy(), g["dec_ids_keep"])
    assert np.array_equal(eng.ce["pad"][:B].cpu().numpy().astype(bool), g["enc_pad"])
    assert np.array_equal(eng.cd["pad"][:B].cpu().numpy().astype(bool), g["dec_pad"])
    assert np.array_equal(eng.ce["mod_mask"][:B].cpu().numpy(), g["enc_mod_mask"])
    assert np.array_equal(eng.cd["mod_mask"][:B].cpu().numpy(), g["dec_mod_mask"])
    assert np.array_equal(eng.cd["tok"][:B].cpu().numpy(), g["target_ids"])
    assert eng.cd["err"].item() == 0
    # interval mask == the reference's dense boolean decoder mask
    blocked = np.unpackbits(g["dec_attn_mask_packed"], axis=-1)[:, :, :M].astype(bool)
    ks, ke = eng.cd["ks"][:B].cpu().numpy(), eng.cd["ke"][:B].cpu().numpy()
    j = np.arange(M)[None, None, :]
    allowed = (j >= ks[:, :, None]) & (j < ke[:, :, None])
    empty = ke <= ks
    assert np.array_equal(allowed[~empty], ~blocked[~empty]) and blocked[empty].all()

    # ---- forward taps
    RN, RM = B * Ne, B * M
    enc0 = act(eng.enc[0]["x"], RN, Ne)
    assert _tap(g, "enc_x0", enc0) < 1e-6                                     # exact fp32 gather + adds
    if R:
        assert torch.equal(enc0[:, :R].cpu(), sd["register_tokens"].expand(B, R, D))      # the register rows, bit for bit
    assert _tap(g, "dec_y0", act(eng.dec[0]["x"], RM, M)) < 1e-6
    blk0 = eng.enc[1]["x"] if cfg.encoder_depth > 1 else eng.x_enc_out
    assert _tap(g, "enc_block0", act(blk0, RN, Ne)) < ACT_TOL
    assert _tap(g, "enc_out", act(eng.xe, RN, Ne)) < ACT_TOL
    assert _tap(g, "context", act(eng.ctx, RN, Ne)) < ACT_TOL
    dblk0 = eng.dec[1]["x"] if cfg.decoder_depth > 1 else eng.y_out
    # pad rows of the decoder stream are never consumed by the reference (mod_mask -1): compare valid rows
    valid = torch.from_numpy(~g["dec_pad"]).cuda()
    if f"tap.dec_block0" in g.files:
        ref = torch.from_numpy(g["tap.dec_block0"]).cuda()
        got = act(dblk0, RM, M)
        assert rel_l2(got[valid].cpu().numpy(), ref[valid].cpu().numpy()) < ACT_TOL
        refo = torch.from_numpy(g["tap.dec_out"]).cuda()[valid]
        perm = eng.perm[:RM].view(B, M)[valid].long()
        assert rel_l2(eng.yn[perm][:, :D].float().cpu().numpy(), refo.cpu().numpy()) < ACT_TOL
    else:
        assert _tap(g, "dec_block0", act(dblk0, RM, M)) < ACT_TOL            # canonical / ragged ego-b: heads are valid rows

    # ---- loss
    ref_loss = float(g["loss"])
    got = loss.item()
    assert abs(got - ref_loss) < LOSS_RTOL * abs(ref_loss), (got, ref_loss)
    for m in cfg.mods:
        r = float(g[f"mod_loss.{m.name}"])
        assert abs(mod_loss[m.name].item() - r) < LOSS_RTOL * max(abs(r), 1.0), (m.name, mod_loss[m.name].item(), r)

    # ---- training-path logits (forward_mod_loss's `forward_logits(y[decoder_mod_mask == id])`, egom2p_model.py:633-636): the
    # fixture holds the first 4 rows x 16 columns and the arg-max of the first 64 rows of every modality, in (clip, decoder
    # position) order - the modality-grouped order of the engine's decoder_norm output (ranges[c] = first row, row count)
    rng_h = eng.ranges.cpu().numpy()
    dmm = torch.from_numpy(g["dec_mod_mask"].astype(np.int64)).cuda().view(-1)
    for c, m in enumerate(cfg.mods):
        ref_head, ref_am = g[f"logits_head.{m.name}"], g[f"logits_argmax.{m.name}"]
        sel = (dmm == m.id).nonzero()[:, 0]                     # (clip, decoder position) order = y[decoder_mod_mask == id]
        cnt = int(sel.numel())
        assert cnt == int(rng_h[c, 1]) and ref_am.shape[0] == min(64, cnt)
        if cnt == 0:
            continue
        rows = min(64, cnt)
        src = eng.perm[:RM].long()[sel[:rows]]                  # where the final LayerNorm wrote those rows (modality-grouped)
        assert int(src.min()) >= int(rng_h[c, 0]) and int(src.max()) < int(rng_h[c, 0]) + cnt
        l = eng.lin[eng.logit_key[m.name]]
        lg = torch.empty(rows, m.vocab_size, device="cuda", dtype=torch.bfloat16)
        ops.gemm_nt(eng.yn[src].contiguous(), l.wb, lg, rows, m.vocab_size, Dp, 0, lda=Dp, ldb=Dp, ldc=m.vocab_size)
        lg = lg.float()
        assert rel_l2(lg[:ref_head.shape[0], :16].cpu().numpy(), ref_head) < 3e-2, m.name
        ref_am_t = torch.from_numpy(ref_am.astype(np.int64)).cuda()
        mism = lg.argmax(-1) != ref_am_t
        # a differing arg-max must be a proven near-tie: in OUR logits the reference's token is within the accepted logit
        # error (3e-2 of the row's largest |logit|) of the row maximum (random-init heads are nearly flat over 64,000 tokens)
        if mism.any():
            gap = (lg.max(-1).values - lg.gather(1, ref_am_t[:, None])[:, 0])[mism]
            assert bool((gap <= 3e-2 * lg.abs().max(-1).values[mism]).all()), (m.name, gap.max().item())
        # (random-init heads are nearly flat: measured 2 of 60 flips on the 256-token heads; every flip is a proven near-tie
        # above).  Floor 0.95 (at most 3 of 64 rows may flip - a regression that flips a fifth of the rows inside the near-tie
        # band does not pass).
        # (modalities with few rows - 13 .. 17 in `tiny8`, dim-128 random-init heads over 256 tokens - may have two such flips.)
        # The 64,000-way heads of the full-size fixtures are flatter still: measured on MI355X 4 / 5 / 6 flips of 64 rows on
        # b12 / L2 / L24 (12+12 .. 24+24 layers of bf16 arithmetic against the fp32 reference), every one inside the near-tie
        # band above - their floor is 0.875 (8 of 64): a regression that flips a fifth of the rows inside the band fails.
        allowed = max(2, int((0.125 if m.vocab_size >= 4096 else 0.05) * rows))
        # ... and against the RECORDED count of this case and modality (tests/golden/parity_bars.json; b12 / L2 / L24 measured
        # 4 / 5 / 6): at most two more flips than recorded - the engine drifting from 5 to 8 flips turns this red
        bar(f"argmax_flips.{case}.{m.name}", int(mism.sum()), hard=allowed, rel_margin=0.0, abs_margin=2.0)

    # ---- backward + clip + AdamW
    eng.zero_grad()
    eng.backward(1.0)
    torch.cuda.synchronize()
    names = [str(n) for n in g["grad_names"]]
    sq = g["grad_sqnorm_all"]
    worst = ("", 0.0)
    for n, ref_sq in zip(names, sq):
        gr = eng.grad_of(n)
        got_sq = gr.double().pow(2).sum().item()
        if ref_sq < 0:          # the reference left .grad None (untied decoder token table: its rows are never read, :328)
            assert got_sq == 0.0, n
            continue
        err = abs(got_sq ** 0.5 - ref_sq ** 0.5) / max(ref_sq ** 0.5, 1e-12)
        if err > worst[1]:
            worst = (n, err)
    assert worst[1] < GRAD_TOL, worst
    # (tied tensors appear under both of their names in the reference's named_parameters() only once: names are unique)
    total = sum(eng.grad_of(n).double().pow(2).sum().item() for n in names) ** 0.5
    assert abs(total - float(g["grad_total_norm"])) < 1e-2 * float(g["grad_total_norm"])
    coef = min(1.0, 1.0 / (float(g["clip_total_norm"]) + 1e-6))
    for key in g.files:
        if key.startswith("grad."):
            n = key[5:]
            e = rel_l2((eng.grad_of(n) * coef).float().cpu().numpy(), g[key])
            assert e < GRAD_TOL, (n, e)
        elif key.startswith("grad_head."):
            n = key[10:]
            gr = eng.grad_of(n) * coef
            e = rel_l2(gr.reshape(-1, gr.shape[-1])[:4, :32].float().cpu().numpy(), g[key])
            assert e < 2 * GRAD_TOL, (n, e)

    if eng.padded:             # every gradient element outside the reference's tensors (pad columns / pad head rows) is an exact zero
        rest = eng.G.clone()
        for n, (o, cnt, shape) in eng.offsets.items():
            eng._logical(n, rest[o:o + cnt].view(shape)).zero_()
        assert not bool(rest.any())
        del rest

    # optimiser: flat-buffer clip + AdamW (reference: clip_grad_norm_(1.0) then AdamW lr 1e-3, wd 0.05 / 0)
    sqn = torch.zeros(1, device="cuda", dtype=torch.float64)
    ops.grad_sqnorm(eng.G, sqn)
    assert abs(sqn.item() ** 0.5 - float(g["clip_total_norm"])) < 1e-2 * float(g["clip_total_norm"])
    m_buf = torch.zeros_like(eng.P)
    v_buf = torch.zeros_like(eng.P)
    for lo, hi, nd in eng.opt_runs:
        ops.adamw_step(eng.P[lo:hi], eng.G[lo:hi], m_buf[lo:hi], v_buf[lo:hi], meta["lr"], 0.0 if nd else meta["wd"], 1,
                       gscale=1.0, max_norm=1.0, sqnorm=sqn)
    torch.cuda.synchronize()
    if eng.padded:             # ... and AdamW leaves the pads of the parameters at zero
        rest = eng.P.clone()
        for n, (o, cnt, shape) in eng.offsets.items():
            eng._logical(n, rest[o:o + cnt].view(shape)).zero_()
        assert not bool(rest.any())
        del rest
    new = eng.state_dict()
    for key in g.files:
        if key.startswith("adamw.") or key.startswith("adamw_head."):
            n = key.split(".", 1)[1]
            p1 = new[n].float().cpu()
            ref = g[key]
            got_t = p1.numpy() if key.startswith("adamw.") else p1.reshape(-1, p1.shape[-1])[:4, :32].numpy()
            p0 = sd[n]
            p0 = p0.numpy() if key.startswith("adamw.") else p0.reshape(-1, p0.shape[-1])[:4, :32].numpy()
            # Adam's first step moves every element by ~lr * sign(g): an element whose (tiny) gradient changes
            # sign under bf16 lands 2*lr away, so bound the element-wise distance and the bulk agreement
            # (the AdamW kernel itself is checked to 1e-6 against torch.optim.AdamW in test_kernels_gpu.py).
            upd_ref, upd_got = ref - p0.reshape(ref.shape), got_t.reshape(ref.shape) - p0.reshape(ref.shape)
            assert np.abs(upd_got - upd_ref).max() <= 2.2 * meta["lr"], n
            assert rel_l2(upd_got, upd_ref) < 0.6, (n, rel_l2(upd_got, upd_ref))
            assert rel_l2(got_t.reshape(ref.shape), ref) < 3e-2, n


def test_full_size_step_agrees_across_kernel_families():
    """BASELINE's full shapes (ego-b, canonical 10,300-position clips, micro-batch 16: M = 32768 rows) run on the
    persistent 256x256 GEMM kernels and the fused fc2-dgrad + SwiGLU-backward launch; the oracle cannot follow at this
    size, so the property checked is that the step is the same function whichever tile family computes it: loss and
    every gradient agree with the 128x128-kernel / unfused path (validated against the reference fixtures above) up
    to fp32 summation order (which can move a bf16-rounded intermediate by one ulp)."""
    import os
    cfg = MODEL_CFGS["egom2p_base_12e_12d_swiglu_nobias"]
    B = 16
    eng = Engine(cfg, "cuda:0", max_batch=B, n_enc=2048, n_dec=2048)
    eng.init_random(11)
    md = synth.make_clip_batch(cfg, B, None, seed=3)
    md = {k: {kk: vv.cuda() for kk, vv in v.items()} for k, v in md.items()}
    order = [m.name for m in cfg.mods]
    res = []
    try:
        for mode, fuse in ((2, "1"), (0, "0")):       # 256x256 kernels wherever legal + fused launch  vs  128x128 kernels, unfused
            ops.gemm_kernel_mode(mode, mode)
            os.environ["EGOM2P_FUSE_SWIGLU_BWD"] = fuse
            eng.zero_grad()
            loss, mod_loss = eng.forward(md, dec_order=order)
            eng.backward(1.0)
            torch.cuda.synchronize()
            res.append((float(loss), {k: float(v) for k, v in mod_loss.items()}, eng.G.clone()))
    finally:
        ops.gemm_kernel_mode(1, 1)
        os.environ.pop("EGOM2P_FUSE_SWIGLU_BWD", None)
    (l1, m1, g1), (l0, m0, g0) = res
    assert abs(l1 - l0) <= 2e-5 * abs(l0), (l1, l0)
    for k in m0:
        assert abs(m1[k] - m0[k]) <= 5e-5 * abs(m0[k]), (k, m1[k], m0[k])
    assert torch.isfinite(g1).all()
    rel = float((g1 - g0).double().norm() / g0.double().norm())
    assert rel < 3e-3, rel
    # 16 slices of the flat buffer: no region's gradient is off (a misplaced tile would show here, not in the global norm)
    n = g0.numel()
    for lo in range(0, n, n // 16):
        hi = min(n, lo + n // 16)
        d = float((g1[lo:hi] - g0[lo:hi]).double().norm() / max(float(g0[lo:hi].double().norm()), 1e-30))
        assert d < 1e-2, (lo, hi, d)


def test_forward_with_loss_grad_equals_the_two_pass_cross_entropy():
    """`forward(loss_grad=g)` (the training step: cross-entropy forward + backward in one pass over the logits) followed by
    `backward(g)` leaves bit for bit the losses and gradients of `forward()` + `backward(g)`; a different `g` in the
    backward is refused (the logits already hold d logits for the promised one)."""
    cfg = MODEL_CFGS["ego_b_2e_2d"]
    eng = Engine(cfg, "cuda:0", max_batch=3, n_enc=2048, n_dec=2048)
    eng.init_random(7)
    md = synth.make_clip_batch_device(cfg, 3, synth.dirichlet_budgets(cfg, 3, 2048, 2048, 4), seed=4, sample_offset=0, device="cuda:0")
    order = [m.name for m in reversed(cfg.mods)]
    eng.zero_grad()
    loss, mod_loss = eng.forward(md, dec_order=order)
    eng.backward(0.25)
    l2, m2, g2 = loss.clone(), {k: v.clone() for k, v in mod_loss.items()}, eng.G.clone()
    eng.zero_grad()
    loss, mod_loss = eng.forward(md, dec_order=order, loss_grad=0.25)
    with pytest.raises(L.EgoHipError):
        eng.backward(0.5)
    eng.backward(0.25)
    assert torch.equal(loss, l2) and all(torch.equal(mod_loss[k], m2[k]) for k in m2)
    assert torch.equal(eng.G, g2)
    # and the next plain forward / backward is unaffected by the earlier promise
    eng.zero_grad()
    eng.forward(md, dec_order=order)
    eng.backward(0.25)
    assert torch.equal(eng.G, g2)


def test_micro_batch_64_equals_two_accumulated_halves():
    """bench.py's default micro-batch (64 clips = 131072 rows per side, logits of 64576 x 64000 = 4.1e9 elements per video
    modality: beyond 2^31) cannot be followed by the oracle; the size-independent property is linearity of the step in the
    batch: one forward / backward over 64 canonical clips equals the accumulation of its two 32-clip halves (the regime the
    reference fixtures and the test above validate) - same loss, same per-modality losses, same gradients up to fp32
    summation order."""
    cfg = MODEL_CFGS["egom2p_base_12e_12d_swiglu_nobias"]
    B = 64
    eng = Engine(cfg, "cuda:0", max_batch=B, n_enc=2048, n_dec=2048)
    eng.init_random(5)
    md = synth.make_clip_batch_device(cfg, B, synth.CANONICAL_BUDGETS, seed=9, sample_offset=0, device="cuda:0")
    order = [m.name for m in cfg.mods]
    eng.zero_grad()
    loss, mod_loss = eng.forward(md, dec_order=order)
    eng.backward(1.0)
    torch.cuda.synchronize()
    l64, m64, g64 = float(loss), {k: float(v) for k, v in mod_loss.items()}, eng.G.clone()
    eng.zero_grad()
    lh, mh = 0.0, {k: 0.0 for k in m64}
    for half in range(2):
        part = {k: {kk: vv[half * 32:(half + 1) * 32].contiguous() for kk, vv in v.items()} for k, v in md.items()}
        loss, mod_loss = eng.forward(part, dec_order=order)
        eng.backward(0.5)
        lh += 0.5 * float(loss)
        for k, v in mod_loss.items():
            mh[k] += 0.5 * float(v)
    torch.cuda.synchronize()
    assert np.isfinite(l64) and abs(l64 - lh) <= 2e-5 * abs(lh), (l64, lh)
    for k in m64:
        assert abs(m64[k] - mh[k]) <= 5e-5 * abs(mh[k]), (k, m64[k], mh[k])
    g32 = eng.G
    assert torch.isfinite(g64).all()
    rel = float((g64 - g32).double().norm() / g32.double().norm())
    assert rel < 3e-3, rel
    n = g32.numel()
    for lo in range(0, n, n // 16):
        hi = min(n, lo + n // 16)
        d = float((g64[lo:hi] - g32[lo:hi]).double().norm() / max(float(g32[lo:hi].double().norm()), 1e-30))
        assert d < 1e-2, (lo, hi, d)
    del eng, g64
    torch.cuda.empty_cache()


@pytest.mark.parametrize("bwd", [False, True], ids=["fwd", "fwd+dgrad"])
@pytest.mark.parametrize("case", ["b2", "L2"])
def test_fp8_forward_stays_within_stated_tolerance(case, bwd):
    """BASELINE config 5 ("bf16 + fp8 MFMA GEMMs"): the forward linears on e4m3 operands (per-row activation scales,
    per-output-channel weight scales, fp32 accumulation), backward in bf16.  The reference has no fp8 path: the bar is
    a stated tolerance against its fp32 outputs (tests/golden/{b2,L2}.npz) - loss within 1e-2 relative, activation
    taps within 1e-1 (e4m3 keeps 3 mantissa bits: 4-9e-2 measured), every per-tensor gradient norm within 1e-1, total
    gradient norm within 3e-2 - next to the bf16 engine's 1e-3 / 2e-2 / 4e-2 / 1e-2 on the same fixtures.
    "fwd+dgrad" (round 5): the dgrad GEMMs dX = dY W on e4m3 too (`fp8_backward`: dY quantised per row, W^T per input channel;
    weight gradients stay bf16) - the same bars on the loss and the taps (the forward is the same), per-tensor gradient norms
    within 1.5e-1 and the total within 5e-2 (every gradient now passes through up to 2 x depth e4m3 roundings)."""
    g, meta = load_golden(case)
    cfg = MODEL_CFGS[meta["cfg"]]
    sd = synth.build_state_dict(cfg, meta["seed"])
    md = synth.make_clip_batch(cfg, meta["batch"], meta["budgets"], meta["seed"])
    eng = Engine(cfg, "cuda:0", max_batch=meta["batch"], n_enc=meta["n_enc"], n_dec=meta["n_dec"], fp8_forward=True, fp8_backward=bwd)
    assert all(l.w8 is not None for n, l in eng.lin.items() if "embeddings" not in n)
    assert all((l.wt8 is not None) == bwd for n, l in eng.lin.items() if "embeddings" not in n)
    eng.load_state_dict(sd)
    mdg = {k: {kk: vv.cuda() for kk, vv in v.items()} for k, v in md.items()}
    B, N, M, D = meta["batch"], meta["n_enc"], meta["n_dec"], cfg.dim
    order = [str(x) for x in g["dec_order"]]
    loss, mod_loss = eng.forward(mdg, dec_order=order)
    eng.zero_grad()
    eng.backward(1.0)
    torch.cuda.synchronize()
    ref_loss = float(g["loss"])
    assert abs(loss.item() - ref_loss) < 1e-2 * abs(ref_loss), (loss.item(), ref_loss)
    RN, RM = B * N, B * M
    errs = {"enc_block0": _tap(g, "enc_block0", eng.enc[1]["x"][:RN].view(B, N, D)),
            "enc_out": _tap(g, "enc_out", eng.xe[:RN].view(B, N, D)),
            "context": _tap(g, "context", eng.ctx[:RN].view(B, N, D)),
            "dec_block0": _tap(g, "dec_block0", eng.dec[1]["x"][:RM].view(B, M, D))}
    print(case, "fp8 taps", {k: f"{v:.2e}" for k, v in errs.items()}, "loss rel", abs(loss.item() - ref_loss) / ref_loss)
    assert max(errs.values()) < 1e-1, errs
    names = [str(n) for n in g["grad_names"]]
    worst = ("", 0.0)
    for n, ref_sq in zip(names, g["grad_sqnorm_all"]):
        if ref_sq <= 0:
            continue
        e = abs(eng.grad_of(n).double().pow(2).sum().item() ** 0.5 - ref_sq ** 0.5) / ref_sq ** 0.5
        if e > worst[1]:
            worst = (n, e)
    total = sum(eng.grad_of(n).double().pow(2).sum().item() for n in names) ** 0.5
    print(case, "fp8 worst grad-norm error", worst, "total", abs(total - float(g["grad_total_norm"])) / float(g["grad_total_norm"]))
    tag = f"fp8{'_dgrad' if bwd else ''}.{case}"
    bar(f"{tag}.worst_grad_norm", worst[1], hard=1.5e-1 if bwd else 1e-1)
    bar(f"{tag}.total_grad_norm", abs(total - float(g["grad_total_norm"])) / float(g["grad_total_norm"]), hard=5e-2 if bwd else 3e-2)
    bar(f"{tag}.worst_tap", max(errs.values()), hard=1e-1)


@pytest.mark.parametrize("case", ["tiny_pad", "tiny8", "b2_ragged"])
@pytest.mark.parametrize("loss_type", ["weighted_mod", "token"])
def test_loss_types_match_reference(case, loss_type):
    """`loss_type='weighted_mod'` / `'token'` (egom2p_model.py:583-612, 646-681) on the engine: the device-side loss weights
    (ego_loss_weights) in the loss AND in the CE backward, against the REAL reference's loss / per-modality losses / gradient
    norms (tests/golden/loss_types.npz), with the two-call CE and with the one-pass CE of the training step."""
    import os
    g, meta = load_golden(case)
    cfg = MODEL_CFGS[meta["cfg"]]
    lt = np.load(os.path.join(os.path.dirname(__file__), "golden", "loss_types.npz"), allow_pickle=False)
    pre = f"{case}.{loss_type}"
    sd = synth.build_state_dict(cfg, meta["seed"])
    md = synth.make_clip_batch(cfg, meta["batch"], meta["budgets"], meta["seed"])
    mdg = {k: {kk: vv.cuda() for kk, vv in v.items()} for k, v in md.items()}
    order = [str(x) for x in g["dec_order"]]
    eng = Engine(cfg, "cuda:0", max_batch=meta["batch"], n_enc=meta["n_enc"], n_dec=meta["n_dec"])
    eng.load_state_dict(sd)
    grads = []
    for fused in (False, True):
        loss, mod_loss = eng.forward(mdg, dec_order=order, loss_type=loss_type, loss_grad=1.0 if fused else None)
        eng.zero_grad()
        eng.backward(1.0)
        torch.cuda.synchronize()
        ref_loss = float(lt[f"{pre}.loss"])
        assert abs(loss.item() - ref_loss) < LOSS_RTOL * abs(ref_loss), (loss.item(), ref_loss)
        for n, r in zip(lt[f"{pre}.mod_names"], lt[f"{pre}.mod_loss"]):
            assert abs(mod_loss[str(n)].item() - float(r)) < LOSS_RTOL * max(abs(float(r)), 1.0), n
        worst = ("", 0.0)
        for n, r in zip(lt[f"{pre}.grad_names"], lt[f"{pre}.grad_sqnorm_all"]):
            got = eng.grad_of(str(n)).double().pow(2).sum().item() ** 0.5
            if r < 0:
                assert got == 0.0, n
                continue
            err = abs(got - r ** 0.5) / max(r ** 0.5, 1e-12)
            if err > worst[1]:
                worst = (str(n), err)
        assert worst[1] < GRAD_TOL, worst
        grads.append(eng.G.clone())
    assert torch.equal(grads[0], grads[1])                      # the one-pass CE is bitwise the two-call form here too


def test_invalid_loss_type_is_the_reference_error():
    cfg = MODEL_CFGS["ego_tiny_2e_2d"]
    eng = Engine(cfg, "cuda:0", max_batch=1, n_enc=30, n_dec=30)
    with pytest.raises(ValueError, match="Invalid loss type"):
        eng.forward({}, loss_type="tokens")
