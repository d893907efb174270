"""ORACLE tooling - generates `tests/golden/*.npz` by running the REAL reference model.

Runs only in the build container (needs /root/reference).  The reference package cannot be
imported whole (`egom2p/utils/__init__.py` needs torchvision, an ordinary ImportError), so the
four hot-path files are loaded by file path under empty namespace stubs, as SURVEY.md section 8c
describes.  Nothing from the reference is copied: this script *calls* its code and stores
input-independent data (outputs on generator-made weights and clips) as fixtures.

    python oracle/make_goldens.py [--only tiny|tiny_pad|b2|b2_reg4|b12|L24|L1020|XL2046]

Weights/clips come from `egom2p_amd.synth` (counter-based generator) so the GPU box can
regenerate them bit-identically; fixtures hold integer outputs in full and float outputs as
full tensors (tiny) or slices + norms (ego-b).
"""
from __future__ import annotations

import argparse
import importlib.util
import os
import random
import sys
import types
from functools import partial

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = os.environ.get("EGOM2P_REFERENCE", "/root/reference")

from egom2p_amd import synth                      # noqa: E402
from egom2p_amd.config import MODEL_CFGS          # noqa: E402


def load_reference():
    """Path-import registry + egom2p_utils + {encoder,decoder}_embeddings + egom2p_model."""
    for name in ("egom2p", "egom2p.utils", "egom2p.utils.timm", "egom2p.data", "egom2p.models"):
        if name not in sys.modules:
            mod = types.ModuleType(name)
            mod.__path__ = []            # namespace stub
            sys.modules[name] = mod
    mi = types.ModuleType("egom2p.data.modality_info")
    mi.MODALITY_INFO = {}
    sys.modules["egom2p.data.modality_info"] = mi

    def _load(modname, relpath):
        spec = importlib.util.spec_from_file_location(modname, os.path.join(REF, relpath))
        mod = importlib.util.module_from_spec(spec)
        sys.modules[modname] = mod
        spec.loader.exec_module(mod)
        return mod

    _load("egom2p.utils.timm.registry", "egom2p/utils/timm/registry.py")
    _load("egom2p.models.egom2p_utils", "egom2p/models/egom2p_utils.py")
    enc = _load("egom2p.models.encoder_embeddings", "egom2p/models/encoder_embeddings.py")
    dec = _load("egom2p.models.decoder_embeddings", "egom2p/models/decoder_embeddings.py")
    model = _load("egom2p.models.egom2p_model", "egom2p/models/egom2p_model.py")
    return enc, dec, model


def build_reference_model(cfg, enc, dec, model):
    """Hand-rebuilt modality_info for the mod4 entries (`modality_info.py:59-69,75-85,116-141`)."""
    info, e_emb, d_emb = {}, {}, {}
    for m in cfg.mods:
        info[m.name] = {"vocab_size": m.vocab_size, "max_tokens": m.max_tokens, "type": m.type, "id": m.id}
        if m.kind == "video":
            e_emb[m.name] = enc.VideoTokenEncoderEmbedding(vocab_size=m.vocab_size, patch_size=(4, 8, 8), image_size=256)
            d_emb[m.name] = dec.VideoTokenDecoderEmbedding(vocab_size=m.vocab_size, patch_size=(4, 8, 8), image_size=256,
                                                           share_embedding=cfg.share_embedding)
        else:
            e_emb[m.name] = enc.GazeCamTokenEncoderEmbedding(vocab_size=m.vocab_size)
            d_emb[m.name] = dec.GazeCamTokenDecoderEmbedding(vocab_size=m.vocab_size, share_embedding=cfg.share_embedding)
    net = model.EgoM2P(
        encoder_embeddings=e_emb, decoder_embeddings=d_emb, modality_info=info,
        dim=cfg.dim, encoder_depth=cfg.encoder_depth, decoder_depth=cfg.decoder_depth,
        num_heads=cfg.num_heads, mlp_ratio=cfg.mlp_ratio, qkv_bias=False, proj_bias=False, mlp_bias=False,
        norm_layer=partial(model.LayerNorm, eps=1e-6, bias=False), act_layer=torch.nn.SiLU, gated_mlp=True,
        num_register_tokens=getattr(cfg, "num_register_tokens", 0))
    return net


GRAD_TAPS_B = [
    "encoder.0.attn.qkv.weight", "encoder.0.norm1.weight", "encoder.1.mlp.fc2.weight",
    "decoder.0.cross_attn.kv.weight", "decoder.1.mlp.fc2.weight", "decoder.1.norm2.weight",
    "encoder_embeddings.tok_rgb.token_emb.weight", "decoder_embeddings.tok_rgb.token_emb.weight",
    "encoder_embeddings.tok_rgb.mod_emb", "encoder_embeddings.tok_cam.mod_emb",
    "mask_token", "decoder_proj_context.bias", "decoder_proj_context.weight", "encoder_norm.weight",
    "decoder_norm.weight",
    "decoder_embeddings.tok_rgb.to_logits.weight", "decoder_embeddings.tok_gaze.to_logits.weight",   # untied head only
    "register_tokens",                                                                                # num_register_tokens > 0 only
]


GRAD_TAPS_TINY = [
    "encoder.0.attn.qkv.weight", "encoder.0.attn.proj.weight", "encoder.0.norm1.weight", "encoder.0.norm2.weight",
    "encoder.1.mlp.fc1.weight", "encoder.1.mlp.fc2.weight", "encoder.1.mlp.fc3.weight",
    "decoder.0.self_attn.qkv.weight", "decoder.0.self_attn.proj.weight", "decoder.0.cross_attn.q.weight",
    "decoder.0.cross_attn.kv.weight", "decoder.0.cross_attn.proj.weight", "decoder.0.query_norm.weight",
    "decoder.0.context_norm.weight", "decoder.1.mlp.fc2.weight", "decoder.1.norm2.weight", "decoder.1.norm1.weight",
    "encoder_embeddings.tok_cam.token_emb.weight", "decoder_embeddings.tok_cam.token_emb.weight",
    "encoder_embeddings.tok_gaze.token_emb.weight", "decoder_embeddings.tok_gaze.token_emb.weight",
    "encoder_embeddings.tok_cam.mod_emb", "encoder_embeddings.tok_gaze.mod_emb",
    "mask_token", "decoder_proj_context.bias", "decoder_proj_context.weight", "encoder_norm.weight",
    "decoder_norm.weight",
]


def run_case(case, cfg_name, batch, n_enc, n_dec, budgets, seed, full_float, py_seed, out_dir, enc, dec, model):
    cfg = MODEL_CFGS[cfg_name]
    torch.manual_seed(0)
    net = build_reference_model(cfg, enc, dec, model)
    sd = synth.build_state_dict(cfg, seed)
    missing, unexpected = net.load_state_dict(sd, strict=True), None
    net.train()
    mod_dict = synth.make_clip_batch(cfg, batch, budgets, seed)

    gold = {}
    # ---- positional tables (bit-exact check of egom2p_amd.posemb)
    for m in cfg.mods:
        pe = net.encoder_embeddings[m.name].pos_emb
        gold[f"posemb_sum.{m.name}"] = np.array([pe.double().sum().item(), pe.double().abs().sum().item()])
        gold[f"posemb_head.{m.name}"] = pe[0, :7, :].numpy().copy()
        gold[f"posemb_tail.{m.name}"] = pe[0, -3:, :].numpy().copy()

    # ---- staged forward: the body of EgoM2P.forward (egom2p_model.py:706-734), stage by stage
    def fresh():
        return {k: {kk: vv.clone() for kk, vv in v.items()} for k, v in mod_dict.items()}

    random.seed(py_seed)
    md = fresh()
    enc_md = {mod: net.encoder_embeddings[mod](d) for mod, d in md.items()}
    captured = []
    real_argsort = torch.argsort

    def spy(*a, **k):
        r = real_argsort(*a, **k)
        captured.append(r)
        return r

    torch.argsort = spy
    try:
        enc_tok, enc_emb, enc_mask, enc_mod = net.forward_mask_encoder(enc_md, n_enc)
        dec_md = {mod: net.decoder_embeddings[mod].forward_embed(d) for mod, d in md.items()}
        # record the shuffled decoder order the reference draws from python's global `random` (:312)
        st = random.getstate()
        order = [mod for mod, _ in random.sample(list(dec_md.items()), len(dec_md))]
        random.setstate(st)
        dec_tok, dec_emb, dec_mask, tgt_ids, dec_attn, dec_mod = net.forward_mask_decoder(dec_md, n_dec)
    finally:
        torch.argsort = real_argsort
    gold["dec_order"] = np.array(order)
    gold["enc_ids_keep"] = captured[0][:, :n_enc].numpy()
    gold["dec_ids_keep"] = captured[1][:, :n_dec].numpy()
    gold["enc_pad"] = enc_mask[:, 0].numpy()
    gold["enc_mod_mask"] = enc_mod.numpy()
    gold["dec_pad"] = dec_mask[:, 0].numpy()
    gold["dec_mod_mask"] = dec_mod.numpy()
    gold["target_ids"] = tgt_ids.numpy()
    gold["dec_attn_mask_packed"] = np.packbits(dec_attn.numpy(), axis=-1)

    x0 = enc_tok + enc_emb
    taps = {"enc_x0": x0}
    x = x0
    for i, blk in enumerate(net.encoder):
        x = blk(x, mask=enc_mask)
        if i == 0:
            taps["enc_block0"] = x
    x = net.encoder_norm(x)
    taps["enc_out"] = x
    ctx = net.decoder_proj_context(x) + enc_emb
    taps["context"] = ctx
    y = dec_tok + dec_emb
    taps["dec_y0"] = y
    for i, blk in enumerate(net.decoder):
        y = blk(y, ctx, sa_mask=dec_attn, xa_mask=enc_mask)
        if i == 0:
            taps["dec_block0"] = y
    y = net.decoder_norm(y)
    taps["dec_out"] = y
    loss_staged, mod_loss_staged = net.forward_loss(y, tgt_ids, dec_md, dec_mod, "mod")
    for mod in mod_dict:
        lg = net.decoder_embeddings[mod].forward_logits(y[dec_mod == net.modality_info[mod]["id"]])
        gold[f"logits_head.{mod}"] = lg[:4, :16].detach().numpy().copy()
        gold[f"logits_argmax.{mod}"] = lg[:64].argmax(-1).numpy() if lg.numel() else np.zeros(0, np.int64)

    # ---- end-to-end call with the same python-random state must agree with the staged run
    random.seed(py_seed)
    loss, mod_loss = net(fresh(), n_enc, n_dec, "mod")
    assert torch.allclose(loss, loss_staged, rtol=0, atol=0), (loss, loss_staged)
    gold["loss"] = np.array(loss.item(), dtype=np.float64)
    for mod, v in mod_loss.items():
        gold[f"mod_loss.{mod}"] = np.array(v.item(), dtype=np.float64)

    for k, v in taps.items():
        v = v.detach()
        if full_float:
            gold[f"tap.{k}"] = v.numpy().copy()
        else:
            gold[f"tap_head.{k}"] = v[:, :6, :24].numpy().copy()
            gold[f"tap_tail.{k}"] = v[:, -4:, -16:].numpy().copy()
            gold[f"tap_norm.{k}"] = np.array([v.double().norm().item(), v.double().sum().item()])
            gold[f"tap_rownorm.{k}"] = v.double().norm(dim=-1).numpy().astype(np.float32)

    # ---- backward + one AdamW step (train loop math: native_scaler.py:28-43, optim_factory.py:97-154,226)
    net.zero_grad()
    loss.backward()
    named = dict(net.named_parameters())
    total_sq = sum(p.grad.double().pow(2).sum().item() for p in named.values() if p.grad is not None)
    gold["grad_total_norm"] = np.array(total_sq ** 0.5)
    names = [n for n in (GRAD_TAPS_TINY if full_float else GRAD_TAPS_B) if n in named and named[n].grad is not None]
    # one scalar per trainable tensor: catches an error in any gradient without storing them all
    gold["grad_names"] = np.array(list(named.keys()))
    gold["grad_sqnorm_all"] = np.array([named[n].grad.double().pow(2).sum().item() if named[n].grad is not None
                                        else -1.0 for n in named])
    lr, wd = 1e-3, 0.05
    decay = [p for n, p in named.items() if not ("norm." in n or ".norm" in n or n.endswith(".bias"))]
    nodecay = [p for n, p in named.items() if ("norm." in n or ".norm" in n or n.endswith(".bias"))]
    opt = torch.optim.AdamW([{"params": decay, "weight_decay": wd}, {"params": nodecay, "weight_decay": 0.0}],
                            lr=lr, betas=(0.9, 0.95), eps=1e-8)
    gnorm = torch.nn.utils.clip_grad_norm_(net.parameters(), 1.0)
    gold["clip_total_norm"] = np.array(gnorm.item())
    grads = {n: named[n].grad.detach().clone() for n in names}      # post-clip grads
    opt.step()
    for n in names:
        g, p = grads[n], named[n].detach()
        if full_float:
            gold[f"grad.{n}"] = g.numpy().copy()
            gold[f"adamw.{n}"] = p.numpy().copy()
        else:
            gf, pf = g.reshape(-1, g.shape[-1]), p.reshape(-1, p.shape[-1])
            gold[f"grad_norm.{n}"] = np.array([g.double().norm().item(), g.double().sum().item()])
            gold[f"grad_head.{n}"] = gf[:4, :32].numpy().copy()
            gold[f"adamw_head.{n}"] = pf[:4, :32].numpy().copy()
            if gf.shape[0] > 4096:   # big tables: row norms of the rows the clip touches are too many; keep a checksum
                gold[f"grad_rowsq.{n}"] = gf.double().pow(2).sum(-1)[:512].numpy().astype(np.float32)

    meta = dict(cfg=cfg_name, batch=batch, n_enc=n_enc, n_dec=n_dec, seed=seed, py_seed=py_seed,
                lr=lr, wd=wd, budgets=repr(budgets))
    gold["meta"] = np.array(repr(meta))
    os.makedirs(out_dir, exist_ok=True)
    path = os.path.join(out_dir, f"{case}.npz")
    np.savez_compressed(path, **gold)
    print(f"[goldens] {case}: loss={loss.item():.6f} order={order} -> {path} "
          f"({os.path.getsize(path) / 1e6:.2f} MB)")


CASES = {
    # cfg1 ego-tiny: 2 modalities, 30+30 tokens, no padding
    "tiny": dict(cfg_name="ego_tiny_2e_2d", batch=4, n_enc=30, n_dec=30,
                 budgets={"tok_cam": (15, 15), "tok_gaze": (15, 15)}, seed=1, full_float=True, py_seed=11),
    # ego-tiny with budgets below the token counts -> padding rows, one empty modality target
    "tiny_pad": dict(cfg_name="ego_tiny_2e_2d", batch=4, n_enc=32, n_dec=32,
                     budgets={"tok_cam": [(10, 12), (3, 0), (15, 15), (0, 7)],
                              "tok_gaze": [(9, 5), (20, 10), (1, 29), (16, 0)]}, seed=2, full_float=True, py_seed=12),
    # eight modalities (EGO_MAX_MODS), ragged budgets incl. clips without inputs / without targets in a modality
    "tiny8": dict(cfg_name="ego_tiny8_2e_2d", batch=3, n_enc=64, n_dec=48,
                  budgets={n: ([(4, 0), (0, 6), (3, 3)] if n == "tok_aux3" else [(5 + (b + j) % 7, 3 + (2 * b + j) % 5) for b in range(3)])
                           for j, n in enumerate(("tok_cam", "tok_gaze") + tuple(f"tok_aux{i}" for i in range(6)))},
                  seed=8, full_float=True, py_seed=18),
    # ego-b width, 2+2 layers, canonical split, N=M=2048
    "b2": dict(cfg_name="ego_b_2e_2d", batch=2, n_enc=2048, n_dec=2048, budgets=None, seed=3,
               full_float=False, py_seed=13),
    # ego-b width, 2+2 layers, FOUR REGISTER TOKENS (egom2p_model.py:170-171, 381-387; `--num_register_tokens 4`): one clip at the
    # canonical split, one with padding rows behind its inputs and targets (and no gaze input at all)
    "b2_reg4": dict(cfg_name="ego_b_2e_2d_reg4", batch=2, n_enc=2048, n_dec=2048,
                    budgets={"tok_rgb": [(1009, 1009), (700, 900)], "tok_depth": [(1009, 1009), (600, 800)],
                             "tok_cam": [(15, 15), (10, 12)], "tok_gaze": [(15, 15), (0, 7)]}, seed=12, full_float=False, py_seed=22),
    # ego-b width, 2+2 layers, ragged budgets with padding
    "b2_ragged": dict(cfg_name="ego_b_2e_2d", batch=3, n_enc=2048, n_dec=2048, budgets="dirichlet", seed=4,
                      full_float=False, py_seed=14),
    # ego-L width (D = 1152, 18 heads of 64, F = 3072: BASELINE config 5), 2+2 layers, canonical split, B=1
    "L2": dict(cfg_name="ego_L_1152_2e_2d", batch=1, n_enc=2048, n_dec=2048, budgets=None, seed=6,
               full_float=False, py_seed=16),
    # ego-b width, 2+2 layers, UNTIED to_logits (share_embedding=False: decoder_embeddings.py:447-449 not taken), B=1
    "b2_untied": dict(cfg_name="ego_b_2e_2d_untied", batch=1, n_enc=2048, n_dec=2048, budgets=None, seed=7,
                      full_float=False, py_seed=17),
    # full-depth ego-b (400M), canonical split, B=1
    "b12": dict(cfg_name="egom2p_base_12e_12d_swiglu_nobias", batch=1, n_enc=2048, n_dec=2048, budgets=None,
                seed=5, full_float=False, py_seed=15),
    # the registered ego-L geometry (dim 1020, 15 heads of 68, F = 2720: egom2p_model.py:1080-1092), 2+2 layers, canonical split
    "L1020": dict(cfg_name="ego_L_1020_2e_2d", batch=1, n_enc=2048, n_dec=2048, budgets=None, seed=10, full_float=False, py_seed=20),
    # the registered ego-XL geometry (dim 2046, 31 heads of 66, F = 5456: egom2p_model.py:1100-1118), 1+1 layers, N = M = 1024
    "XL2046": dict(cfg_name="ego_XL_2046_1e_1d", batch=1, n_enc=1024, n_dec=1024,
                   budgets={"tok_rgb": (497, 497), "tok_depth": (497, 497), "tok_cam": (15, 15), "tok_gaze": (15, 15)},
                   seed=11, full_float=False, py_seed=21),
    # full-depth ego-L at the throughput shape (BASELINE config 5: D = 1152, 18 heads of 64, F = 3072, 24 + 24 layers,
    # 1.19 B parameters), canonical split, B=1
    # (N = M = 1024: the fp32 autograd graph of 48 layers at 2048 x 2048 scores does not fit this container's 64 GB)
    "L24": dict(cfg_name="ego_L_1152", batch=1, n_enc=1024, n_dec=1024,
                budgets={"tok_rgb": (497, 497), "tok_depth": (497, 497), "tok_cam": (15, 15), "tok_gaze": (15, 15)},
                seed=9, full_float=False, py_seed=19),
}


def init_stats(out_dir, enc, dec, model):
    """Per-parameter statistics of the reference's OWN constructor init (`init_weights`, egom2p_model.py:185-222, after
    the embedding modules' `init`): mean, std, min, max of every parameter of the tied and the untied 2+2-layer ego-b.
    They pin the distributions `Engine.init_random` / `EgoM2P.init_weights` must draw from."""
    gold = {}
    for tag, cfg_name in (("tied", "ego_b_2e_2d"), ("untied", "ego_b_2e_2d_untied")):
        torch.manual_seed(1234)
        net = build_reference_model(MODEL_CFGS[cfg_name], enc, dec, model)
        names, rows = [], []
        for n, p in net.state_dict().items():
            if n.endswith("pos_emb") or (n.endswith(".bias") and "norm" in n):
                continue
            v = p.detach().double()
            names.append(n)
            rows.append([v.mean().item(), v.std().item() if v.numel() > 1 else 0.0, v.min().item(), v.max().item(), float(v.numel())])
        gold[f"{tag}.names"] = np.array(names)
        gold[f"{tag}.stats"] = np.array(rows, dtype=np.float64)
    gold["meta"] = np.array(repr(dict(what="reference init_weights statistics", seed=1234)))
    path = os.path.join(out_dir, "init_stats.npz")
    np.savez_compressed(path, **gold)
    print(f"[goldens] init_stats -> {path} ({os.path.getsize(path) / 1e3:.1f} kB)")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None)
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden"))
    args = ap.parse_args()
    enc, dec, model = load_reference()
    torch.set_num_threads(8)
    if not args.only or "init_stats" in args.only.split(","):
        init_stats(args.out, enc, dec, model)
    for case, kw in CASES.items():
        if args.only and case not in args.only.split(","):
            continue
        kw = dict(kw)
        if kw["budgets"] == "dirichlet":
            kw["budgets"] = synth.dirichlet_budgets(MODEL_CFGS[kw["cfg_name"]], kw["batch"], kw["n_enc"], kw["n_dec"], kw["seed"])
        run_case(case, out_dir=args.out, enc=enc, dec=dec, model=model, **kw)


if __name__ == "__main__":
    main()
