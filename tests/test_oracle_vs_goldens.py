"""Pins the CPU oracle (oracle/egom2p_oracle.py) to outputs of the REAL reference model.

The fixtures under tests/golden/ were produced by oracle/make_goldens.py, which runs the
reference's own `EgoM2P` (imported by path from /root/reference) on generator-made weights and
clips.  Integer outputs must match bit-exactly; fp32 outputs to fp32 round-off (1e-5 rel).
"""
import os

import numpy as np
import pytest
import torch

from conftest import load_golden, rel_l2
from egom2p_amd import synth
from egom2p_amd.config import MODEL_CFGS
from egom2p_amd.posemb import build_pos_emb
from oracle import egom2p_oracle as O

FP32_TOL = 2e-5

CASES = ["tiny", "tiny_pad", "tiny8", "b2", "b2_ragged", "b2_reg4", "L1020"]
if os.environ.get("EGOM2P_SLOW") == "1":
    CASES += ["b12", "L2", "L24", "XL2046"]


def _setup(case):
    g, meta = load_golden(case)
    cfg = MODEL_CFGS[meta["cfg"]]
    sd = synth.build_state_dict(cfg, meta["seed"])
    md = synth.make_clip_batch(cfg, meta["batch"], meta["budgets"], meta["seed"])
    return g, meta, cfg, sd, md


@pytest.mark.parametrize("case", ["tiny", "b2"])
def test_posemb_bit_exact(case):
    g, meta, cfg, sd, md = _setup(case)
    for m in cfg.mods:
        pe = build_pos_emb(m, cfg.dim)
        assert np.array_equal(pe[0, :7].numpy(), g[f"posemb_head.{m.name}"])
        assert np.array_equal(pe[0, -3:].numpy(), g[f"posemb_tail.{m.name}"])
        s = np.array([pe.double().sum().item(), pe.double().abs().sum().item()])
        assert np.array_equal(s, g[f"posemb_sum.{m.name}"])


@pytest.mark.parametrize("case", CASES)
def test_forward_backward_matches_reference(case):
    g, meta, cfg, sd, md = _setup(case)
    torch.set_num_threads(8)
    leaf = O.make_leaf_state(sd)
    taps = {}
    loss, mod_loss = O.forward(leaf, cfg, md, meta["n_enc"], meta["n_dec"],
                               dec_order=[str(x) for x in g["dec_order"]], mode="fp32", taps=taps)

    # ---- integer / index outputs: bit-exact
    assert np.array_equal(taps["enc_ids_keep"], g["enc_ids_keep"])
    assert np.array_equal(taps["dec_ids_keep"], g["dec_ids_keep"])
    assert np.array_equal(taps["enc_pad"], g["enc_pad"])
    assert np.array_equal(taps["dec_pad"], g["dec_pad"])
    assert np.array_equal(taps["enc_mod_mask"], g["enc_mod_mask"])
    assert np.array_equal(taps["dec_mod_mask"], g["dec_mod_mask"])
    assert np.array_equal(taps["target_ids"], g["target_ids"])
    blocked = O.decoder_attention_mask(taps["dec_dam"], taps["dec_mod_mask_pre"])
    assert np.array_equal(np.packbits(blocked, axis=-1), g["dec_attn_mask_packed"])
    # interval form used by the HIP attention kernel is equivalent to the dense mask
    ks, ke, ok = O.attention_ranges(taps["dec_dam"], taps["dec_mod_mask_pre"], taps["dec_pad"])
    assert ok
    M = blocked.shape[1]
    j = np.arange(M)[None, None, :]
    allowed = (j >= ks[:, :, None]) & (j < ke[:, :, None])
    empty = (ke <= ks)
    assert np.array_equal(allowed[~empty], ~blocked[~empty])
    assert blocked[empty].all()

    # ---- fp32 taps
    for k in ("enc_x0", "enc_block0", "enc_out", "context", "dec_y0", "dec_block0", "dec_out"):
        t = taps[k].detach()
        if f"tap.{k}" in g.files:
            assert rel_l2(t.numpy(), g[f"tap.{k}"]) < FP32_TOL, k
        else:
            assert rel_l2(t[:, :6, :24].numpy(), g[f"tap_head.{k}"]) < FP32_TOL, k
            assert rel_l2(t[:, -4:, -16:].numpy(), g[f"tap_tail.{k}"]) < FP32_TOL, k
            assert rel_l2(t.double().norm(dim=-1).numpy(), g[f"tap_rownorm.{k}"]) < FP32_TOL, k
    for m in cfg.mods:
        lg = taps[f"logits.{m.name}"].detach()
        assert rel_l2(lg[:4, :16].numpy(), g[f"logits_head.{m.name}"]) < 1e-4
    assert abs(loss.item() - float(g["loss"])) < 1e-5 * abs(float(g["loss"]))
    for m in cfg.mods:
        assert abs(mod_loss[m.name].item() - float(g[f"mod_loss.{m.name}"])) < 2e-5 * max(1.0, abs(float(g[f"mod_loss.{m.name}"])))

    # ---- backward, clip, AdamW
    loss.backward()
    names = [str(n) for n in g["grad_names"]]
    sq = g["grad_sqnorm_all"]
    total = 0.0
    for n, ref_sq in zip(names, sq):
        gr = leaf[n].grad
        assert gr is not None, n
        total += gr.double().pow(2).sum().item()
    # tied tensors (shared mod_emb) appear once in named_parameters(); the sum over names is the reference's
    assert abs(total ** 0.5 - float(g["grad_total_norm"])) < 1e-4 * float(g["grad_total_norm"])
    coef = min(1.0, 1.0 / (float(g["clip_total_norm"]) + 1e-6))
    for key in g.files:
        if key.startswith("grad."):
            n = key[5:]
            assert rel_l2((leaf[n].grad * coef).numpy(), g[key]) < 1e-4, n
        elif key.startswith("grad_head."):
            n = key[10:]
            gr = (leaf[n].grad * coef)
            assert rel_l2(gr.reshape(-1, gr.shape[-1])[:4, :32].numpy(), g[key]) < 1e-4, n
            ref = g[f"grad_norm.{n}"]
            assert abs(gr.double().norm().item() - ref[0]) < 1e-4 * ref[0], n
    for key in g.files:
        if key.startswith("adamw.") or key.startswith("adamw_head."):
            n = key.split(".", 1)[1]
            p0 = sd[n]
            gr = leaf[n].grad * coef
            wd = 0.0 if O.no_decay(n) else meta["wd"]
            p1, _, _ = O.adamw_step(p0, gr, torch.zeros_like(p0), torch.zeros_like(p0), 1, meta["lr"], wd)
            if key.startswith("adamw."):
                assert rel_l2(p1.numpy(), g[key]) < 1e-5, n
            else:
                assert rel_l2(p1.reshape(-1, p1.shape[-1])[:4, :32].numpy(), g[key]) < 1e-5, n


def test_bf16_mode_close_to_fp32():
    """The autocast-emulating mode stays within bf16 round-off of the fp32 truth."""
    g, meta, cfg, sd, md = _setup("tiny")
    order = [str(x) for x in g["dec_order"]]
    with torch.no_grad():
        l32, _ = O.forward(sd, cfg, md, meta["n_enc"], meta["n_dec"], dec_order=order, mode="fp32")
        l16, _ = O.forward(sd, cfg, md, meta["n_enc"], meta["n_dec"], dec_order=order, mode="bf16")
    assert abs(l32.item() - l16.item()) < 2e-3 * abs(l32.item())


@pytest.mark.parametrize("case", ["tiny_pad", "tiny8", "b2_ragged"])
@pytest.mark.parametrize("loss_type", ["weighted_mod", "token"])
def test_loss_types_match_reference(case, loss_type):
    """`loss_type='weighted_mod'` / `'token'` (egom2p_model.py:583-612, 646-681): loss, per-modality losses and every
    parameter's gradient norm of the REAL reference (tests/golden/loss_types.npz, oracle/make_goldens_loss_types.py)."""
    g, meta, cfg, sd, md = _setup(case)
    lt = np.load(os.path.join(os.path.dirname(__file__), "golden", "loss_types.npz"), allow_pickle=False)
    pre = f"{case}.{loss_type}"
    torch.set_num_threads(8)
    leaf = O.make_leaf_state(sd)
    loss, mod_loss = O.forward(leaf, cfg, md, meta["n_enc"], meta["n_dec"], dec_order=[str(x) for x in g["dec_order"]],
                               mode="fp32", loss_type=loss_type)
    loss.backward()
    assert abs(loss.item() - float(lt[f"{pre}.loss"])) < FP32_TOL * abs(float(lt[f"{pre}.loss"]))
    for n, r in zip(lt[f"{pre}.mod_names"], lt[f"{pre}.mod_loss"]):
        assert abs(mod_loss[str(n)].item() - float(r)) < FP32_TOL * max(abs(float(r)), 1.0), n
    for n, r in zip(lt[f"{pre}.grad_names"], lt[f"{pre}.grad_sqnorm_all"]):
        t = leaf[str(n)]
        if r < 0:
            continue
        got = t.grad.double().pow(2).sum().item() ** 0.5
        assert abs(got - r ** 0.5) < 1e-4 * max(r ** 0.5, 1e-12), (n, got, r ** 0.5)
    # and the default type of the same fixture file is the case's own golden
    assert float(lt[f"{case}.mod.loss"]) == float(g["loss"])
