"""Build-owned counter-based generator for weights and synthetic clips.

value = f(tensor-name, flat index, seed) using only integer arithmetic plus one exact
int->float scaling, so this container (where the reference model is run to make goldens)
and the GPU box regenerate bit-identical tensors without shipping 1.6 GB of weights.
(SURVEY.md section 8c "build-owned counter-based generator".)

Nothing here is taken from the reference; the *distributions* follow its init scheme
(`egom2p/models/egom2p_model.py:185-222`): xavier-uniform linears with qkv / kv treated as
3 / 2 separate matrices, N(0, 0.02) embeddings, LayerNorm weight 1 (we perturb LN weights and
the one Linear bias slightly so their gradients are exercised by parity tests).
"""
from __future__ import annotations

import hashlib
import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from .config import ModelCfg

_GOLD = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def _key(name: str, seed: int) -> np.uint64:
    h = hashlib.sha256(f"{seed}:{name}".encode()).digest()
    return np.uint64(int.from_bytes(h[:8], "little"))


def _hash_u64(name: str, n: int, seed: int, start: int = 0) -> np.ndarray:
    """splitmix64 finaliser over counter = key + (start+i) * golden."""
    with np.errstate(over="ignore"):
        x = (np.arange(start, start + n, dtype=np.uint64) * _GOLD) + _key(name, seed)
        x = (x ^ (x >> np.uint64(30))) * _M1
        x = (x ^ (x >> np.uint64(27))) * _M2
        x = x ^ (x >> np.uint64(31))
    return x


def _chunks(n: int, step: int = 1 << 24):
    for s in range(0, n, step):
        yield s, min(step, n - s)


def uniform(name: str, shape: Sequence[int], lo: float, hi: float, seed: int = 0) -> torch.Tensor:
    n = int(np.prod(shape))
    out = np.empty(n, dtype=np.float32)
    for s, c in _chunks(n):
        u = (_hash_u64(name, c, seed, s) >> np.uint64(40)).astype(np.float64) * (1.0 / (1 << 24))
        out[s:s + c] = (lo + (hi - lo) * u).astype(np.float32)
    return torch.from_numpy(out.reshape(tuple(shape)))


_IH_STD = math.sqrt(4.0 * ((65536.0 ** 2 - 1.0) / 12.0))  # std of the sum of four uniform u16


def normal(name: str, shape: Sequence[int], std: float, seed: int = 0, mean: float = 0.0) -> torch.Tensor:
    """Irwin-Hall(4) approximation of a Gaussian from the four 16-bit limbs of one hash."""
    n = int(np.prod(shape))
    out = np.empty(n, dtype=np.float32)
    for s, c in _chunks(n):
        x = _hash_u64(name, c, seed, s)
        m = np.uint64(0xFFFF)
        t = ((x & m) + ((x >> np.uint64(16)) & m) + ((x >> np.uint64(32)) & m) + (x >> np.uint64(48))).astype(np.float64)
        out[s:s + c] = (mean + (t - 2.0 * 65535.0) * (std / _IH_STD)).astype(np.float32)
    return torch.from_numpy(out.reshape(tuple(shape)))


def randint(name: str, shape: Sequence[int], high: int, seed: int = 0) -> torch.Tensor:
    n = int(np.prod(shape))
    x = (_hash_u64(name, n, seed) >> np.uint64(11)) % np.uint64(high)
    return torch.from_numpy(x.astype(np.int64).reshape(tuple(shape)))


def permutation(name: str, n: int, seed: int = 0) -> np.ndarray:
    return np.argsort(_hash_u64(name, n, seed), kind="stable")


# ----------------------------------------------------------------------------------------
# weights
# ----------------------------------------------------------------------------------------

def _xavier(name, out_f, in_f, seed, fan_out=None):
    fan_out = out_f if fan_out is None else fan_out
    a = math.sqrt(6.0 / float(fan_out + in_f))
    return uniform(name, (out_f, in_f), -a, a, seed)


def _ln(name, dim, seed, sd):
    sd[name + ".weight"] = normal(name + ".weight", (dim,), 0.1, seed, mean=1.0)
    sd[name + ".bias"] = torch.zeros(dim)  # zero *buffer* in the reference (LayerNorm(bias=False))


def build_state_dict(cfg: ModelCfg, seed: int = 0, posemb: bool = True) -> Dict[str, torch.Tensor]:
    """State dict with the reference's key layout."""
    from .posemb import build_pos_emb

    D, F = cfg.dim, cfg.mlp_hidden
    sd: Dict[str, torch.Tensor] = {}
    for m in cfg.mods:
        e = f"encoder_embeddings.{m.name}"
        d = f"decoder_embeddings.{m.name}"
        sd[f"{e}.mod_emb"] = normal(f"{e}.mod_emb", (1, 1, D), 0.02, seed)
        sd[f"{e}.token_emb.weight"] = normal(f"{e}.token_emb.weight", (m.vocab_size, D), 0.02, seed)
        sd[f"{d}.mod_emb"] = sd[f"{e}.mod_emb"]            # shared Parameter (egom2p_model.py:179-183)
        sd[f"{d}.token_emb.weight"] = normal(f"{d}.token_emb.weight", (m.vocab_size, D), 0.02, seed)
        if cfg.share_embedding:
            sd[f"{d}.to_logits.weight"] = sd[f"{d}.token_emb.weight"]  # tied (decoder_embeddings.py:447-449)
        else:
            sd[f"{d}.to_logits.weight"] = _xavier(f"{d}.to_logits.weight", m.vocab_size, D, seed)
        if posemb:
            pe = build_pos_emb(m, D)
            sd[f"{e}.pos_emb"] = pe
            sd[f"{d}.pos_emb"] = pe
    for i in range(cfg.encoder_depth):
        p = f"encoder.{i}"
        _ln(f"{p}.norm1", D, seed, sd)
        _ln(f"{p}.norm2", D, seed, sd)
        sd[f"{p}.attn.qkv.weight"] = _xavier(f"{p}.attn.qkv.weight", 3 * D, D, seed, fan_out=D)
        sd[f"{p}.attn.proj.weight"] = _xavier(f"{p}.attn.proj.weight", D, D, seed)
        sd[f"{p}.mlp.fc1.weight"] = _xavier(f"{p}.mlp.fc1.weight", F, D, seed)
        sd[f"{p}.mlp.fc2.weight"] = _xavier(f"{p}.mlp.fc2.weight", D, F, seed)
        sd[f"{p}.mlp.fc3.weight"] = _xavier(f"{p}.mlp.fc3.weight", F, D, seed)
    _ln("encoder_norm", D, seed, sd)
    sd["decoder_proj_context.weight"] = _xavier("decoder_proj_context.weight", D, D, seed)
    sd["decoder_proj_context.bias"] = normal("decoder_proj_context.bias", (D,), 0.02, seed)
    for i in range(cfg.decoder_depth):
        p = f"decoder.{i}"
        for n in ("norm1", "query_norm", "context_norm", "norm2"):
            _ln(f"{p}.{n}", D, seed, sd)
        sd[f"{p}.self_attn.qkv.weight"] = _xavier(f"{p}.self_attn.qkv.weight", 3 * D, D, seed, fan_out=D)
        sd[f"{p}.self_attn.proj.weight"] = _xavier(f"{p}.self_attn.proj.weight", D, D, seed)
        sd[f"{p}.cross_attn.q.weight"] = _xavier(f"{p}.cross_attn.q.weight", D, D, seed)
        sd[f"{p}.cross_attn.kv.weight"] = _xavier(f"{p}.cross_attn.kv.weight", 2 * D, D, seed, fan_out=D)
        sd[f"{p}.cross_attn.proj.weight"] = _xavier(f"{p}.cross_attn.proj.weight", D, D, seed)
        sd[f"{p}.mlp.fc1.weight"] = _xavier(f"{p}.mlp.fc1.weight", F, D, seed)
        sd[f"{p}.mlp.fc2.weight"] = _xavier(f"{p}.mlp.fc2.weight", D, F, seed)
        sd[f"{p}.mlp.fc3.weight"] = _xavier(f"{p}.mlp.fc3.weight", F, D, seed)
    _ln("decoder_norm", D, seed, sd)
    sd["mask_token"] = normal("mask_token", (1, 1, D), 0.02, seed)
    if getattr(cfg, "num_register_tokens", 0):
        sd["register_tokens"] = normal("register_tokens", (1, cfg.num_register_tokens, D), 0.02, seed)    # egom2p_model.py:170-172
    return sd


def peak_logit_table(sd: Dict[str, torch.Tensor], mod: str, seed: int = 0, cap: float = 1.0e3) -> Dict[str, torch.Tensor]:
    """Make the output head of `mod` PEAKED: row v of its (tied) logit table is multiplied by a Pareto(1) factor
    1 / (1 - u_v) (capped).  Random-init tables give nearly flat logits over 64,000 tokens, where bf16 noise flips the
    arg-max of ~10 % of the rows; with heavy-tailed row norms the top logit leads by far more than that noise, so
    generation tests can demand token equality.  Deterministic (counter-based stream), applied in place."""
    key = f"decoder_embeddings.{mod}.token_emb.weight"
    w = sd[key]
    u = uniform(f"peak.{mod}", (w.shape[0],), 0.0, 1.0, seed).double()
    scale = torch.clamp(1.0 / (1.0 - u), max=cap).float()
    w.mul_(scale[:, None])                    # to_logits.weight is the same tensor when tied
    if sd.get(f"decoder_embeddings.{mod}.to_logits.weight") is not None and \
            sd[f"decoder_embeddings.{mod}.to_logits.weight"].data_ptr() != w.data_ptr():
        sd[f"decoder_embeddings.{mod}.to_logits.weight"].mul_(scale[:, None])
    return sd


# ----------------------------------------------------------------------------------------
# synthetic clips (the `mod_dict` input contract)
# ----------------------------------------------------------------------------------------

CANONICAL_BUDGETS = {  # (k_in, k_tgt) per modality: N = M = 2048 exactly
    "tok_rgb": (1009, 1009), "tok_depth": (1009, 1009), "tok_cam": (15, 15), "tok_gaze": (15, 15),
}


def make_clip_batch(cfg: ModelCfg, batch: int, budgets: Optional[Dict[str, Sequence[Tuple[int, int]]]] = None,
                    seed: int = 0, sample_offset: int = 0) -> Dict[str, Dict[str, torch.Tensor]]:
    """Synthetic `mod_dict` for `batch` clips.

    Per modality and sample: a hash-derived permutation of its positions; the first k_in are
    encoder inputs, the next k_tgt decoder targets (disjoint, like `masking.py:248-260`);
    masks are True = ignore; `decoder_attention_mask` carries k_tgt at the first target
    position (`masking.py:262-264`).  `budgets[mod]` is one (k_in, k_tgt) pair, or a list of
    pairs, one per sample (ragged case).
    """
    out: Dict[str, Dict[str, torch.Tensor]] = {}
    for m in cfg.mods:
        n = m.max_tokens
        bud = (budgets or CANONICAL_BUDGETS)[m.name]
        if isinstance(bud[0], int):
            bud = [tuple(bud)] * batch
        shape = (batch,) + (m.grid if m.kind == "video" else (n,))
        ids = torch.empty((batch, n), dtype=torch.int64)
        in_mask = torch.ones((batch, n), dtype=torch.bool)
        tg_mask = torch.ones((batch, n), dtype=torch.bool)
        dam = torch.zeros((batch, n), dtype=torch.int32)
        for b in range(batch):
            s = sample_offset + b
            ids[b] = randint(f"clip{s}.{m.name}.ids", (n,), m.vocab_size, seed)
            perm = permutation(f"clip{s}.{m.name}.perm", n, seed)
            k_in, k_tg = bud[b]
            assert k_in + k_tg <= n
            in_mask[b, torch.from_numpy(perm[:k_in].copy())] = False
            tg_mask[b, torch.from_numpy(perm[k_in:k_in + k_tg].copy())] = False
            # first target position (argmin of mask + arange*1e-6 == first False, else 0)
            first = int(np.min(perm[k_in:k_in + k_tg])) if k_tg > 0 else 0
            dam[b, first] = k_tg
        out[m.name] = {
            "tensor": ids.reshape(shape),
            "input_mask": in_mask,
            "target_mask": tg_mask,
            "decoder_attention_mask": dam,
        }
    return out


def make_clip_batch_device(cfg: ModelCfg, batch: int, budgets: Optional[Dict[str, Sequence[Tuple[int, int]]]] = None,
                           seed: int = 0, sample_offset: int = 0, device: str = "cuda") -> Dict[str, Dict[str, torch.Tensor]]:
    """`make_clip_batch` generated on the GPU (ego_clip_synth): same bits, no host tensors, no H2D copy.  Only the per-clip
    stream keys (two sha256 per clip and modality) and the budgets are prepared on the host."""
    from . import ops
    out: Dict[str, Dict[str, torch.Tensor]] = {}
    for m in cfg.mods:
        n = m.max_tokens
        bud = (budgets or CANONICAL_BUDGETS)[m.name]
        if isinstance(bud[0], int):
            bud = [tuple(bud)] * batch
        assert all(k_in + k_tg <= n for k_in, k_tg in bud)
        keys = np.array([[_key(f"clip{sample_offset + b}.{m.name}.ids", seed), _key(f"clip{sample_offset + b}.{m.name}.perm", seed)]
                         for b in range(batch)], dtype=np.uint64)
        kd = torch.from_numpy(keys.view(np.int64)).to(device)
        kb = torch.tensor(bud, dtype=torch.int32, device=device)
        ids = torch.empty((batch, n), dtype=torch.int64, device=device)
        in_mask = torch.empty((batch, n), dtype=torch.bool, device=device)
        tg_mask = torch.empty((batch, n), dtype=torch.bool, device=device)
        dam = torch.empty((batch, n), dtype=torch.int32, device=device)
        ops.clip_synth(kd[:, 0].contiguous(), kd[:, 1].contiguous(), kb[:, 0].contiguous(), kb[:, 1].contiguous(), n, m.vocab_size,
                       ids, in_mask, tg_mask, dam)
        shape = (batch,) + (m.grid if m.kind == "video" else (n,))
        out[m.name] = {"tensor": ids.reshape(shape), "input_mask": in_mask, "target_mask": tg_mask, "decoder_attention_mask": dam}
    return out


def dirichlet_budgets(cfg: ModelCfg, batch: int, n_in: int, n_tgt: int, seed: int = 0,
                      alphas: Sequence[float] = (0.01, 0.1, 1.0, 10.0)) -> Dict[str, List[Tuple[int, int]]]:
    """Ragged budgets in the spirit of the reference's Dirichlet mixture
    (`cfgs/default/egom2p/alphas_mixture/main/mix_mod4_all2all_uni.yaml`, `masking.py:181-234`):
    per sample draw input/target shares from Dirichlet(alpha) (alpha picked from `alphas`),
    clamp to what each modality has left.  Deterministic (numpy Generator seeded per sample)."""
    mods = cfg.mods
    res: Dict[str, List[Tuple[int, int]]] = {m.name: [] for m in mods}
    for b in range(batch):
        rng = np.random.default_rng([seed, b, 0xE60])
        a_in = alphas[int(rng.integers(len(alphas)))]
        a_tg = alphas[int(rng.integers(len(alphas)))]
        p_in = rng.dirichlet([a_in] * len(mods))
        p_tg = rng.dirichlet([a_tg] * len(mods))
        for j, m in enumerate(mods):
            k_in = min(int(round(p_in[j] * n_in)), m.max_tokens)
            k_tg = min(int(round(p_tg[j] * n_tgt)), m.max_tokens - k_in)
            res[m.name].append((k_in, k_tg))
    return res


# The released mod4 mixture (cfgs/default/egom2p/alphas_mixture/main/mix_mod4_all2all_uni.yaml): four components with
# alpha = 0.01 / 0.1 / 1 / 10 for every modality, uniform sampling weights.
MOD4_MIXTURE_ALPHAS = (0.01, 0.1, 1.0, 10.0)


def masking_budgets_host(cfg: ModelCfg, batch: int, n_in_range=(2048, 2048), n_tgt_range=(2048, 2048), seed: int = 0,
                         alphas: Sequence[float] = MOD4_MIXTURE_ALPHAS, mix_weights: Optional[Sequence[float]] = None,
                         min_tokens: int = 0, max_tries: int = 100):
    """The reference's budget sampler restated on the host (`UnifiedMasking.input_token_budget` / `target_token_budget`,
    egom2p/data/masking.py:181-234; mixture / token-count draws :530-541) with numpy's generator: floor of Dirichlet * N,
    leftover tokens to the arg-max of further draws, clamp, redraw below min_tokens.  Returns int arrays [n_mods, batch]
    (k_in, k_tgt) - the checker for the device sampler `ego_budget_dirichlet` (same law, independent random stream)."""
    mods = cfg.mods
    n = len(mods)
    rng = np.random.default_rng([seed, 0xB0D6])
    w = np.ones(len(alphas)) if mix_weights is None else np.asarray(mix_weights, dtype=np.float64)
    w = w / w.sum()
    max_tok = np.array([m.max_tokens for m in mods])
    k_in = np.zeros((n, batch), dtype=np.int64)
    k_tg = np.zeros((n, batch), dtype=np.int64)

    f32_lo, f32_hi = np.float32(np.finfo(np.float32).tiny), np.nextafter(np.float32(1.0), np.float32(0.0))

    def f32_sample(a):
        # torch's Dirichlet.sample() on float32 concentrations (what masking.py:192 calls): gamma variates in double,
        # normalised, cast to float32 and CLAMPED to [FLT_MIN, 1 - 2^-24] (ATen _s_dirichlet).  The clamp is part of the
        # reference's law: a one-hot draw is never exactly 1, so floor(p * N) leaves one token over, which the arg-max of
        # the next draw hands to (3 times out of 4) ANOTHER modality - pinned by tests/golden/budget_stats.npz
        return np.clip(rng.dirichlet(a).astype(np.float32), f32_lo, f32_hi)

    def draw(alpha, total, cap):
        a = np.full(n, max(alpha, 1e-9))
        bud = np.zeros(n, dtype=np.int64)
        for _ in range(max_tries):
            bud = np.floor(f32_sample(a) * np.float32(total)).astype(np.int64)
            for _ in range(int(total - bud.sum())):
                bud[int(np.argmax(f32_sample(a)))] += 1
            bud = np.minimum(bud, cap)
            if (bud >= min_tokens).all():
                break
        return bud

    for b in range(batch):
        alpha = alphas[int(rng.choice(len(alphas), p=w))]
        n_in = int(rng.integers(n_in_range[0], n_in_range[1] + 1))
        n_tg = int(rng.integers(n_tgt_range[0], n_tgt_range[1] + 1))
        ki = draw(alpha, n_in, max_tok)
        kt = draw(alpha, n_tg, np.maximum(min_tokens, max_tok - ki))      # img / cam / gaze types: targets take what is left
        k_in[:, b], k_tg[:, b] = ki, kt
    return k_in, k_tg


def masking_budgets_device(cfg: ModelCfg, batch: int, n_in_range=(2048, 2048), n_tgt_range=(2048, 2048), seed: int = 0,
                           sample_offset: int = 0, alphas: Sequence[float] = MOD4_MIXTURE_ALPHAS,
                           mix_weights: Optional[Sequence[float]] = None, min_tokens: int = 0, device: str = "cuda"):
    """Token budgets of `batch` clips sampled ON THE DEVICE (ego_budget_dirichlet): int32 tensors [n_mods, batch]."""
    from . import ops
    mods = cfg.mods
    keys = np.array([_key(f"clip{sample_offset + b}.budget", seed) for b in range(batch)], dtype=np.uint64)
    kd = torch.from_numpy(keys.view(np.int64)).to(device)
    k_in = torch.empty(len(mods), batch, dtype=torch.int32, device=device)
    k_tg = torch.empty(len(mods), batch, dtype=torch.int32, device=device)
    al = [[a] * len(mods) for a in alphas]
    ops.budget_dirichlet(kd, al, al, [1.0] * len(alphas) if mix_weights is None else mix_weights, [m.max_tokens for m in mods],
                         [min_tokens] * len(mods), [True] * len(mods), n_in_range, n_tgt_range, k_in, k_tg)
    return k_in, k_tg


def make_clip_batch_device_masked(cfg: ModelCfg, batch: int, n_in: int = 2048, n_tgt: int = 2048, seed: int = 0, sample_offset: int = 0,
                                  device: str = "cuda") -> Dict[str, Dict[str, torch.Tensor]]:
    """Synthetic clips whose budgets come from the reference's Dirichlet mixture, budgets AND masks made on the device:
    nothing but the per-clip stream keys crosses the host-device boundary."""
    from . import ops
    k_in, k_tg = masking_budgets_device(cfg, batch, (n_in, n_in), (n_tgt, n_tgt), seed, sample_offset, device=device)
    out: Dict[str, Dict[str, torch.Tensor]] = {}
    for j, m in enumerate(cfg.mods):
        n = m.max_tokens
        keys = np.array([[_key(f"clip{sample_offset + b}.{m.name}.ids", seed), _key(f"clip{sample_offset + b}.{m.name}.perm", seed)]
                         for b in range(batch)], dtype=np.uint64)
        kd = torch.from_numpy(keys.view(np.int64)).to(device)
        ids = torch.empty((batch, n), dtype=torch.int64, device=device)
        in_mask = torch.empty((batch, n), dtype=torch.bool, device=device)
        tg_mask = torch.empty((batch, n), dtype=torch.bool, device=device)
        dam = torch.empty((batch, n), dtype=torch.int32, device=device)
        ops.clip_synth(kd[:, 0].contiguous(), kd[:, 1].contiguous(), k_in[j], k_tg[j], n, m.vocab_size, ids, in_mask, tg_mask, dam)
        shape = (batch,) + (m.grid if m.kind == "video" else (n,))
        out[m.name] = {"tensor": ids.reshape(shape), "input_mask": in_mask, "target_mask": tg_mask, "decoder_attention_mask": dam}
    return out
