"""ORACLE - test infrastructure only.  Never imported by the product path.

CPU restatement (functional torch, fp32) of the EgoM2P hot path:
embed -> mask/compact -> encoder -> context projection -> decoder -> per-modality logits + CE.
Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import it.

Each function cites the reference file:line it restates.  PINNED: `oracle/make_goldens.py`
runs the real reference model (imported by file path from /root/reference, in the build
container only) on generator-made weights/clips and stores its outputs under `tests/golden/`;
`tests/test_oracle_vs_goldens.py` checks this restatement against those fixtures.

mode='fp32'  : plain fp32 (the truth the goldens were produced in).
mode='bf16'  : inserts bf16 roundings where CUDA autocast(bf16) would (GEMM-class inputs and
               outputs bf16; softmax / layer_norm / cross_entropy fp32; residual stream fp32)
               - SURVEY.md section 2.2.  Used as the tight comparison for the HIP kernels.
mode='bf16_bwd': the same, and gradients are rounded to bf16 at the same points (what autocast's
               backward does: the gradient of a bf16 tensor is bf16).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F


class _RoundBoth(torch.autograd.Function):
    """bf16 round trip in the forward AND on the gradient: a tensor that lives in bf16 under autocast receives a bf16
    gradient (the backward of a bf16 op produces bf16)."""

    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).to(torch.float32)

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).to(torch.float32)


def _r(x: torch.Tensor, mode: str) -> torch.Tensor:
    """Round to bf16 and come back.  mode 'bf16': straight-through gradient (forward roundings only);
    mode 'bf16_bwd': the gradient is rounded to bf16 too, as the backward of autocast(bf16) does."""
    if mode == "bf16_bwd":
        return _RoundBoth.apply(x)
    if mode != "bf16":
        return x
    return x + (x.to(torch.bfloat16).to(torch.float32) - x).detach()


def linear(x, w, b=None, mode="fp32"):
    """nn.Linear under autocast: bf16 x bf16 -> bf16 (fp32 accumulate)."""
    y = F.linear(_r(x, mode), _r(w, mode), None if b is None else _r(b, mode))
    return _r(y, mode)


def layer_norm(x, w, eps=1e-6):
    """Bias-free LayerNorm, fp32 (reference `egom2p_utils.py:118-133`; bias is a zero buffer)."""
    return F.layer_norm(x, (x.shape[-1],), w, None, eps)


# ----------------------------------------------------------------------------------------
# compaction (integer, bit-exact)                     reference `egom2p_model.py:344-444`
# ----------------------------------------------------------------------------------------

def stable_partition_keep(mask: np.ndarray, n_keep: int) -> np.ndarray:
    """ids_keep = argsort(mask + arange*1e-6)[:, :n_keep]  (`egom2p_model.py:370-373, 422-425`).

    For T <= ~10^5 the fp32 keys are distinct and ordered, so this is exactly a stable
    partition: unmasked positions in ascending order, then masked ones in ascending order."""
    B, T = mask.shape
    out = np.empty((B, n_keep), dtype=np.int64)
    for b in range(B):
        keep = np.flatnonzero(~mask[b])
        drop = np.flatnonzero(mask[b])
        out[b] = np.concatenate([keep, drop])[:n_keep]
    return out


def compact_encoder(mod_dict, mods, n_enc):
    """cat_encoder_tensors + forward_mask_encoder, integer part (`egom2p_model.py:251-283, 344-396`).
    Returns ids_keep (B,N) int64, pad mask (B,N) bool, mod ids (B,N) int16 (-1 on pads),
    token ids at kept positions (B,N) int64, local position within modality (B,N) int64,
    modality slot (B,N) int64."""
    masks = np.concatenate([mod_dict[m.name]["input_mask"].numpy() for m in mods], axis=1)
    ids_all = np.concatenate([mod_dict[m.name]["tensor"].reshape(masks.shape[0], -1).numpy() for m in mods], axis=1)
    modid_all = np.concatenate([np.full(m.max_tokens, m.id, dtype=np.int16) for m in mods])
    slot_all = np.concatenate([np.full(m.max_tokens, i, dtype=np.int64) for i, m in enumerate(mods)])
    local_all = np.concatenate([np.arange(m.max_tokens, dtype=np.int64) for m in mods])
    keep = stable_partition_keep(masks, n_enc)
    pad = np.take_along_axis(masks, keep, 1)
    modid = modid_all[keep].copy()
    modid[pad] = -1
    return dict(ids_keep=keep, pad=pad, mod_mask=modid, tok=np.take_along_axis(ids_all, keep, 1),
                local=local_all[keep], slot=slot_all[keep])


def compact_decoder(mod_dict, mods_in_order, n_dec):
    """cat_decoder_tensors + forward_mask_decoder, integer part (`egom2p_model.py:285-342, 398-444`).
    `mods_in_order` is the (shuffled, `:312`) modality order, made an explicit input."""
    B = mod_dict[mods_in_order[0].name]["target_mask"].shape[0]
    masks = np.concatenate([mod_dict[m.name]["target_mask"].numpy() for m in mods_in_order], axis=1)
    ids_all = np.concatenate([mod_dict[m.name]["tensor"].reshape(B, -1).numpy() for m in mods_in_order], axis=1)
    dam_all = np.concatenate([mod_dict[m.name]["decoder_attention_mask"].numpy() for m in mods_in_order], axis=1)
    modid_all = np.concatenate([np.full(m.max_tokens, m.id, dtype=np.int16) for m in mods_in_order])
    local_all = np.concatenate([np.arange(m.max_tokens, dtype=np.int64) for m in mods_in_order])
    keep = stable_partition_keep(masks, n_dec)
    pad = np.take_along_axis(masks, keep, 1)
    tgt = np.take_along_axis(ids_all, keep, 1).copy()
    tgt[pad] = 0
    dam = np.take_along_axis(dam_all, keep, 1)
    modid_pre = modid_all[keep]            # modality ids *before* pads become -1 (used by the sep mask)
    modid = modid_pre.copy()
    modid[pad] = -1
    return dict(ids_keep=keep, pad=pad, mod_mask=modid, mod_mask_pre=modid_pre, target_ids=tgt,
                dam=dam, local=local_all[keep])


def decoder_attention_mask(dam: np.ndarray, modid_pre: np.ndarray) -> np.ndarray:
    """adapt_decoder_attention_mask (`egom2p_model.py:446-481`), non-causal + sep mask.
    True = blocked.  (B, M, M)."""
    M = dam.shape[1]
    cs = np.cumsum(dam.astype(np.int64), axis=-1)[:, :, None]
    blocked = np.arange(M)[None, None, :] >= cs
    sep = modid_pre[:, None, :] != modid_pre[:, :, None]
    return blocked | sep


def attention_ranges(dam: np.ndarray, modid_pre: np.ndarray, pad: np.ndarray):
    """Per-row allowed key interval [ks, ke) equivalent to `decoder_attention_mask` when each
    modality's kept targets are contiguous and pad keys are never allowed (always true for the
    reference data contract, `masking.py:236-266`).  Returns (ks, ke, ok) - `ok` False if the mask
    is not expressible as one interval per row."""
    full = ~decoder_attention_mask(dam, modid_pre)          # True = allowed
    B, M, _ = full.shape
    ks = np.zeros((B, M), dtype=np.int32)
    ke = np.zeros((B, M), dtype=np.int32)
    ok = True
    for b in range(B):
        for i in range(M):
            idx = np.flatnonzero(full[b, i])
            if idx.size:
                ks[b, i], ke[b, i] = idx[0], idx[-1] + 1
                ok &= (idx.size == ke[b, i] - ks[b, i])
    return ks, ke, ok


# ----------------------------------------------------------------------------------------
# transformer blocks
# ----------------------------------------------------------------------------------------

_NEG = {"fp32": -torch.finfo(torch.float32).max, "bf16": -float(torch.finfo(torch.bfloat16).max),
        "bf16_bwd": -float(torch.finfo(torch.bfloat16).max)}


def _softmax_attn(q, k, v, blocked, scale, mode):
    """(q k^T) * scale -> masked_fill(-finfo.max) -> softmax -> @ v  (`egom2p_utils.py:190-202`)."""
    attn = _r(_r(q @ k.transpose(-2, -1), mode) * scale, mode)
    if blocked is not None:
        attn = attn.masked_fill(blocked, _NEG[mode])
    attn = attn.softmax(dim=-1)
    return _r(_r(attn, mode) @ v, mode)


def self_attention(x, w_qkv, w_proj, heads, blocked, mode):
    """Attention.forward (`egom2p_utils.py:185-205`).  blocked: (B,1|N,N) bool or None."""
    B, N, C = x.shape
    qkv = linear(x, w_qkv, mode=mode).reshape(B, N, 3, heads, C // heads).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    m = None if blocked is None else blocked[:, None]
    o = _softmax_attn(q, k, v, m, (C // heads) ** -0.5, mode)
    return linear(o.transpose(1, 2).reshape(B, N, C), w_proj, mode=mode)


def cross_attention(x, ctx, w_q, w_kv, w_proj, heads, blocked, mode):
    """CrossAttention.forward (`egom2p_utils.py:222-244`).  blocked: (B,1,Nctx) bool or None."""
    B, M, C = x.shape
    N = ctx.shape[1]
    q = linear(x, w_q, mode=mode).reshape(B, M, heads, C // heads).permute(0, 2, 1, 3)
    kv = linear(ctx, w_kv, mode=mode).reshape(B, N, 2, heads, C // heads).permute(2, 0, 3, 1, 4)
    m = None if blocked is None else blocked[:, None]
    o = _softmax_attn(q, kv[0], kv[1], m, (C // heads) ** -0.5, mode)
    return linear(o.transpose(1, 2).reshape(B, M, C), w_proj, mode=mode)


def swiglu(x, w1, w2, w3, mode):
    """GatedMlp.forward: fc2(silu(fc1 x) * fc3 x)  (`egom2p_utils.py:167-169`)."""
    a = linear(x, w1, mode=mode)
    b = linear(x, w3, mode=mode)
    return linear(_r(_r(F.silu(a), mode) * b, mode), w2, mode=mode)


def encoder_block(x, sd, p, heads, blocked, mode, eps):
    """Block.forward (`egom2p_utils.py:356-359`)."""
    x = x + self_attention(layer_norm(x, sd[f"{p}.norm1.weight"], eps), sd[f"{p}.attn.qkv.weight"],
                           sd[f"{p}.attn.proj.weight"], heads, blocked, mode)
    x = x + swiglu(layer_norm(x, sd[f"{p}.norm2.weight"], eps), sd[f"{p}.mlp.fc1.weight"],
                   sd[f"{p}.mlp.fc2.weight"], sd[f"{p}.mlp.fc3.weight"], mode)
    return x


def decoder_block(y, ctx, sd, p, heads, sa_blocked, xa_blocked, mode, eps):
    """DecoderBlock.forward (`egom2p_utils.py:387-391`)."""
    y = y + self_attention(layer_norm(y, sd[f"{p}.norm1.weight"], eps), sd[f"{p}.self_attn.qkv.weight"],
                           sd[f"{p}.self_attn.proj.weight"], heads, sa_blocked, mode)
    y = y + cross_attention(layer_norm(y, sd[f"{p}.query_norm.weight"], eps),
                            layer_norm(ctx, sd[f"{p}.context_norm.weight"], eps),
                            sd[f"{p}.cross_attn.q.weight"], sd[f"{p}.cross_attn.kv.weight"],
                            sd[f"{p}.cross_attn.proj.weight"], heads, xa_blocked, mode)
    y = y + swiglu(layer_norm(y, sd[f"{p}.norm2.weight"], eps), sd[f"{p}.mlp.fc1.weight"],
                   sd[f"{p}.mlp.fc2.weight"], sd[f"{p}.mlp.fc3.weight"], mode)
    return y


# ----------------------------------------------------------------------------------------
# whole forward                                          reference `egom2p_model.py:683-734`
# ----------------------------------------------------------------------------------------

def forward(sd: Dict[str, torch.Tensor], cfg, mod_dict, n_enc: int, n_dec: int,
            dec_order: Optional[Sequence[str]] = None, mode: str = "fp32",
            taps: Optional[dict] = None, return_logits: bool = False, loss_type: str = "mod"):
    """Returns (loss, {mod: loss}).  `dec_order`: decoder modality order (names); default = dict order.
    `taps` (optional dict) receives intermediate tensors for parity tests.  `loss_type`: 'mod' (forward_mod_loss,
    egom2p_model.py:614-644), 'weighted_mod' (forward_weighted_mod_loss, :583-612) or 'token' (forward_token_loss, :646-681)."""
    if loss_type not in ("mod", "modality", "weighted_mod", "token"):
        raise ValueError("Invalid loss type")
    mods = [m for m in cfg.mods if m.name in mod_dict]
    byname = {m.name: m for m in mods}
    dmods = [byname[n] for n in (dec_order or [m.name for m in mods])]
    D, H, eps = cfg.dim, cfg.num_heads, cfg.eps
    taps = taps if taps is not None else {}

    # --- encoder side: embed (encoder_embeddings.py:181-210, 272-301) + compact (egom2p_model.py:344-396)
    ce = compact_encoder(mod_dict, mods, n_enc)
    B, N = ce["ids_keep"].shape
    pad_e = torch.from_numpy(ce["pad"])
    x_tok = torch.zeros(B, N, D)
    x_emb = torch.zeros(B, N, D)
    for i, m in enumerate(mods):
        sel = torch.from_numpy((ce["slot"] == i) & ~ce["pad"])
        if not sel.any():
            continue
        tok = torch.from_numpy(ce["tok"])[sel]
        loc = torch.from_numpy(ce["local"])[sel]
        e = f"encoder_embeddings.{m.name}"
        x_tok = x_tok.index_put((sel,), sd[f"{e}.token_emb.weight"][tok])
        x_emb = x_emb.index_put((sel,), sd[f"{e}.pos_emb"][0][loc] + sd[f"{e}.mod_emb"][0, 0])
    R = int(getattr(cfg, "num_register_tokens", 0))
    if R:                                                               # egom2p_model.py:381-387: register tokens in front, zero emb,
        reg = sd["register_tokens"].expand(B, R, D)                     # never padding, modality id -1
        x_tok = torch.cat([reg, x_tok], 1)
        x_emb = torch.cat([torch.zeros(B, R, D), x_emb], 1)
        pad_e = torch.cat([torch.zeros(B, R, dtype=torch.bool), pad_e], 1)
        ce = dict(ce, pad=pad_e.numpy(), mod_mask=np.concatenate([np.full((B, R), -1, ce["mod_mask"].dtype), ce["mod_mask"]], 1))
    x = x_tok + x_emb                                                   # egom2p_model.py:718
    taps.update(enc_ids_keep=ce["ids_keep"], enc_pad=ce["pad"], enc_mod_mask=ce["mod_mask"], enc_x0=x)

    xa_blocked = pad_e[:, None, :]                                      # (B,1,N)
    for i in range(cfg.encoder_depth):                                  # egom2p_model.py:496-497
        x = encoder_block(x, sd, f"encoder.{i}", H, xa_blocked, mode, eps)
        if i == 0:
            taps["enc_block0"] = x
    x = layer_norm(x, sd["encoder_norm.weight"], eps)                   # :499
    taps["enc_out"] = x
    ctx = linear(x, sd["decoder_proj_context.weight"], sd["decoder_proj_context.bias"], mode) + x_emb  # :722
    taps["context"] = ctx

    # --- decoder side: embed + compact (egom2p_model.py:398-444); token rows are the mask token (:328)
    cd = compact_decoder(mod_dict, dmods, n_dec)
    M = cd["ids_keep"].shape[1]
    pad_d = torch.from_numpy(cd["pad"])
    y_emb = torch.zeros(B, M, D)
    off = 0
    for m in dmods:
        keep = cd["ids_keep"]
        sel = torch.from_numpy((keep >= off) & (keep < off + m.max_tokens) & ~cd["pad"])
        off += m.max_tokens
        if not sel.any():
            continue
        loc = torch.from_numpy(cd["local"])[sel]
        d = f"decoder_embeddings.{m.name}"
        y_emb = y_emb.index_put((sel,), sd[f"{d}.pos_emb"][0][loc] + sd[f"{d}.mod_emb"][0, 0])
    y_tok = (~pad_d)[..., None].float() * sd["mask_token"][0, 0]
    y = y_tok + y_emb                                                   # :723
    sa_blocked = torch.from_numpy(decoder_attention_mask(cd["dam"], cd["mod_mask_pre"]))
    taps.update(dec_ids_keep=cd["ids_keep"], dec_pad=cd["pad"], dec_mod_mask=cd["mod_mask"],
                dec_mod_mask_pre=cd["mod_mask_pre"], dec_dam=cd["dam"],
                target_ids=cd["target_ids"], dec_y0=y)

    for i in range(cfg.decoder_depth):                                  # :520-521
        y = decoder_block(y, ctx, sd, f"decoder.{i}", H, sa_blocked, xa_blocked, mode, eps)
        if i == 0:
            taps["dec_block0"] = y
    y = layer_norm(y, sd["decoder_norm.weight"], eps)                   # :523
    taps["dec_out"] = y

    if return_logits:                                                   # :727-729, 546-547
        return {m.name: linear(y, sd[f"decoder_embeddings.{m.name}.to_logits.weight"], mode=mode) for m in mods}

    # --- per-modality logits + CE, averaged over *all* modalities (egom2p_model.py:614-644)
    mod_loss, mod_count = {}, {}
    mm = torch.from_numpy(cd["mod_mask"].astype(np.int64))
    tgt = torch.from_numpy(cd["target_ids"])
    for m in mods:        # dict order of decoder_mod_dict == mod_dict order (:712-714)
        sel = mm == m.id
        logits = linear(y[sel], sd[f"decoder_embeddings.{m.name}.to_logits.weight"], mode=mode)
        taps[f"logits.{m.name}"] = logits
        mod_count[m.name] = logits.numel()                               # rows x vocab (:676)
        if logits.numel() == 0:
            mod_loss[m.name] = logits.sum()
        else:
            mod_loss[m.name] = F.cross_entropy(logits.float(), tgt[sel], reduction="mean")
            if loss_type == "weighted_mod":                              # :607: rescaled to a 256-entry codebook
                mod_loss[m.name] = mod_loss[m.name] / math.log(m.vocab_size) * 5.545177444479562
    if loss_type == "token":                                             # :679
        loss = sum(mod_loss[k] * mod_count[k] for k in mod_loss) / sum(mod_count.values())
    else:
        loss = sum(mod_loss.values()) / len(mod_loss)
    return loss, mod_loss


def make_leaf_state(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """Clone the trainable entries as autograd leaves, preserving the ties of the reference
    (shared mod_emb, to_logits tied to decoder token_emb).  Buffers stay plain tensors."""
    out: Dict[str, torch.Tensor] = {}
    seen: Dict[int, torch.Tensor] = {}
    for k, v in sd.items():
        if k.endswith("pos_emb") or (k.endswith(".bias") and "norm" in k):
            out[k] = v
            continue
        key = v.data_ptr()
        if key not in seen:
            seen[key] = v.detach().clone().requires_grad_(True)
        out[k] = seen[key]
    return out


def adamw_step(p, g, m, v, step, lr, wd, beta1=0.9, beta2=0.95, eps=1e-8):
    """torch.optim.AdamW single-tensor math (the optimiser created at `optim_factory.py:226`)."""
    p = p * (1.0 - lr * wd)
    m = beta1 * m + (1 - beta1) * g
    v = beta2 * v + (1 - beta2) * g * g
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    denom = v.sqrt() / math.sqrt(bc2) + eps
    return p - (lr / bc1) * (m / denom), m, v


def no_decay(name: str) -> bool:
    """get_parameter_groups rule (`optim_factory.py:113`)."""
    return "norm." in name or ".norm" in name or name.endswith(".bias")
