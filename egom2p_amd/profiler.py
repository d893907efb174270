"""Live per-kernel-class timing with HIP events on the launch stream (the engine launches on torch's
current stream, so `torch.cuda.Event` brackets exactly the kernels of one C-ABI call) plus the
algorithmic FLOP / byte counts of each call.  Used by bench.py for the `roofline` object; the same
command under `rocprofv3 --kernel-trace --stats` gives the cross-check committed under profiles/.
"""
from __future__ import annotations

import contextlib
from collections import defaultdict
from typing import Dict

import torch

from . import ops

PEAK_BF16_TFLOPS = 2500.0     # MI355X dense bf16 MFMA (MI355X_MICROARCH.md: ~2.5 PF dense)
PEAK_HBM_GBS = 8000.0         # HBM3E spec (6.3 TB/s achievable)


def kernel_source_sha() -> str:
    """sha256 over the kernel sources the in-tree library is built from: ties a committed PMC summary
    (profiles/pmc_latest.json) to the kernels it was measured on."""
    import hashlib
    import os
    root = os.path.dirname(os.path.abspath(__file__))
    files = sorted(f for f in os.listdir(os.path.join(root, "csrc")) if f.endswith((".hip", ".h")))
    h = hashlib.sha256()
    for f in files:
        h.update(f.encode())
        h.update(open(os.path.join(root, "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def detail_names() -> bool:
    import os
    return os.environ.get("EGOM2P_PROFILE_DETAIL") == "1"


class KernelTimer:
    def __init__(self):
        self.records = []      # (class, flops, bytes, start_event, end_event)

    def _wrap(self, name, fn, cost):
        import os
        detail = os.environ.get("EGOM2P_PROFILE_DETAIL") == "1"

        def wrapped(*a, **k):
            flops, nbytes = cost(*a, **k)
            nm = name
            if detail and name == "gemm_nt":
                nm = f"gemm_nt[{a[3]}x{a[4]}x{a[5]} epi{a[6] if len(a) > 6 else k.get('epi', 0)}{' rng' if k.get('m_range') is not None else ''}]"
            elif detail and name == "gemm_tn":
                nm = f"gemm_tn[{a[5]}x{a[3]}x{a[4]} s{k.get('splits', 1)}]"
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            r = fn(*a, **k)
            e.record()
            self.records.append((nm, flops, nbytes, s, e))
            return r
        return wrapped

    @contextlib.contextmanager
    def capture(self, H: int):
        def rows_of(m_range, M):
            return int(m_range[1].item()) if m_range is not None else M

        def c_gemm_nt(A, B, C, M, N, K, epi=0, R=None, bias=None, m_range=None, **kw):
            m = min(M, rows_of(m_range, M))
            return 2.0 * m * N * K, 2.0 * (m * K + N * K) + (2.0 if epi == 0 else 8.0) * m * N

        def c_gemm_nt_swiglu_bwd(dY, W2t, ab, dab, M, F, K, **kw):
            return 2.0 * M * F * K, 2.0 * (M * K + F * K) + 8.0 * M * F       # ab read + dab write, 2 x 2 B x 2F each

        def c_gemm_nt_swiglu_fwd(X, W13, ab, h, M, F, K, **kw):
            return 4.0 * M * F * K, 2.0 * (M * K + 2 * F * K) + 6.0 * M * F       # ab write (2 x 2 B) + h write (2 B) per column

        def c_gemm_tn(P, Q, C0, Ni, Nj, M, C1=None, split_row=0, rows0=None, rows1=0, m_range=None, **kw):
            m = min(M, rows_of(m_range, M))
            return 2.0 * m * Ni * Nj, 2.0 * m * (Ni + Nj) + 8.0 * Ni * Nj

        def _pairs(ks, ke, r_bs, r_rs, B, Nq, Nk):
            if r_rs == 0:
                n = (ke[:B].clamp(max=Nk) - ks[:B]).clamp(min=0).double().sum().item() * Nq
            else:
                n = (ke[:B].clamp(max=Nk) - ks[:B]).clamp(min=0).double().sum().item()
            return n

        def c_attn_fwd(q, q_bs, q_rs, k, k_bs, k_rs, v, v_bs, v_rs, o, o_bs, o_rs, lse, ks, ke, r_bs, r_rs, B, Hh, Nq, Nk, scale, o_lo=None,
                       hd_pad=64, seg=None, seg_bad=None, hd=None):
            p = _pairs(ks, ke, r_bs, r_rs, B, Nq, Nk)
            return 4.0 * hd_pad * Hh * p, 2.0 * B * Hh * hd_pad * ((3 if o_lo is not None else 2) * Nq + 2 * Nk)

        def c_attn_bwd(*a, o_lo=None, hd_pad=64, seg=None, seg_bad=None, hd=None):
            ks, ke, r_bs, r_rs, B, Hh, Nq, Nk = a[-9], a[-8], a[-7], a[-6], a[-5], a[-4], a[-3], a[-2]
            p = _pairs(ks, ke, r_bs, r_rs, B, Nq, Nk)
            return 10.0 * hd_pad * Hh * p, 2.0 * B * Hh * hd_pad * ((5 if o_lo is not None else 4) * Nq + 4 * Nk)

        def rowcost(bytes_per_row_elem):
            def c(*a, **k):
                return 0.0, 0.0
            return c

        def c_ln_fwd(x, w, y, mean, rstd, out_row=None, eps=1e-6, q8=None, qscale=None, width=None):
            return 0.0, x.shape[0] * x.shape[1] * (7.0 if q8 is not None else 6.0)

        def c_ln_bwd(dy, x, mean, rstd, w, dx_out, dw, dx_in=None, dx_bf16=None, dy_row=None, width=None):
            n = x.shape[0] * x.shape[1]
            return 0.0, n * (2.0 + 4.0 + 4.0 + (4.0 if dx_in is not None else 0.0) + (2.0 if dx_bf16 is not None else 0.0))

        def c_ln_fwd_multi(x, ws, ys, mean, rstd, eps=1e-6, width=None):
            return 0.0, x.shape[0] * x.shape[1] * (4.0 + 2.0 * len(ws))

        def c_ln_bwd_multi(dys, x, mean, rstd, ws, dx_out, dws, dx_in=None, dx_bf16=None, width=None):
            n = x.shape[0] * x.shape[1]
            return 0.0, n * (4.0 + 2.0 * len(ws) + 4.0 + (4.0 if dx_in is not None else 0.0) + (2.0 if dx_bf16 is not None else 0.0))

        def c_swiglu_fwd(ab, h, rows, F):
            return 0.0, rows * F * 6.0

        def c_swiglu_bwd(ab, dh, dab, rows, F):
            return 0.0, rows * F * 10.0

        def c_ce(logits, ld, V, targets, rng, max_rows, *a):
            n = int(rng[1].item())
            return 0.0, n * V * 2.0

        def c_ce_bwd(logits, ld, V, targets, rng, max_rows, *a, loss_w=None):
            n = int(rng[1].item())
            return 0.0, n * V * 4.0

        def c_gemm_nt_fp8(A8, sa, B8, sb, C, M, N, K, epi=0, R=None, bias=None):
            return 2.0 * M * N * K, 1.0 * (M * K + N * K) + (2.0 if epi == 0 else 8.0) * M * N

        def c_gemm_nt_swiglu_fwd_fp8(X8, sx, W8, sw, ab, h, M, F, K):
            return 4.0 * M * F * K, 1.0 * (M * K + 2 * F * K) + 6.0 * M * F

        def c_quant(X, Q, scale, rows=None, K=None):
            r = X.shape[0] if rows is None else rows
            k = X.shape[-1] if K is None else K
            return 0.0, 3.0 * r * k

        table = {
            "gemm_nt_fp8": c_gemm_nt_fp8, "gemm_nt_swiglu_fwd_fp8": c_gemm_nt_swiglu_fwd_fp8, "quant_fp8_rows": c_quant,
            "gemm_nt": c_gemm_nt, "gemm_nt_swiglu_bwd": c_gemm_nt_swiglu_bwd, "gemm_nt_swiglu_fwd": c_gemm_nt_swiglu_fwd, "gemm_tn": c_gemm_tn, "attn_fwd": c_attn_fwd, "attn_bwd": c_attn_bwd,
            "layernorm_fwd": c_ln_fwd, "layernorm_bwd": c_ln_bwd, "layernorm_fwd_multi": c_ln_fwd_multi, "layernorm_bwd_multi": c_ln_bwd_multi, "swiglu_fwd": c_swiglu_fwd, "swiglu_bwd": c_swiglu_bwd,
            "ce_fwd": c_ce, "ce_bwd": c_ce_bwd, "ce_fwd_bwd": c_ce_bwd,          # fused: one read + one write of the logits
        }
        # HBM-bound front-end / bookkeeping kernels: algorithmic bytes (the index streams once, only
        # the kept rows of the tables / positional tables, every output once)
        def c_compact(masks, ids, dams, n_pos, mod_ids, n_keep, is_decoder, out, B, n_reg=0):
            T = sum(n_pos)
            return 0.0, float(B) * (T * (1 + (4 if is_decoder else 0)) + (n_keep + n_reg) * (8 + 31))

        def c_embed_fwd(tables, pos, mod, base_vec, slot, local, tok, x, emb, rows, D, reg=None):
            n_rw = (1 if tables is not None else 0) + 1 + 1 + (1 if emb is not None else 0)   # token row, pos row | x, emb
            return 0.0, float(rows) * D * 4 * n_rw + rows * 12.0

        def c_embed_bwd(dtables, dmods, dbase, dx, d2, slot, tok, rows, D, touched=None):
            n_rw = 1 + (1 if d2 is not None else 0) + (1 if dtables is not None else 0)       # dx, d2 | table rows (atomic adds)
            return 0.0, float(rows) * D * 4 * n_rw + rows * 8.0

        def c_loss_perm(seg, canon, slot, tok, B, M, n_mods, *a):
            return 0.0, float(B) * M * 16.0

        def c_loss_finalize(nll, ranges, n_mods, out, err=None, loss_w=None, mod_scale=None):
            return 0.0, float(ranges[:n_mods, 1].sum().item()) * 4.0

        def c_cast_weight(W, Wb=None, Wt=None, **kw):
            return 0.0, float(W.numel()) * (4 + (2 if Wb is not None else 0) + (2 if Wt is not None else 0))

        def c_cast(src, dst):
            return 0.0, float(src.numel()) * 6.0

        def c_bias_grad(g, rows, D, db):
            return 0.0, float(rows) * D * 2.0

        def c_sqnorm(g, out):
            return 0.0, float(g.numel()) * 4.0

        def c_adamw(p, g, m, v, *a, zero_grad=False, **kw):
            return 0.0, float(p.numel()) * (16 + 12 + (4 if zero_grad else 0))

        other = {"compact": c_compact, "embed_fwd": c_embed_fwd, "embed_bwd": c_embed_bwd, "loss_perm": c_loss_perm,
                 "loss_finalize": c_loss_finalize, "cast_weight": c_cast_weight, "cast_f32_bf16": c_cast, "bias_grad": c_bias_grad,
                 "grad_sqnorm": c_sqnorm, "adamw_step": c_adamw}
        self.cost_table = dict(table, **other)
        saved = {}
        # the fused launches are the same device kernel (gemm_nt256_kernel<EK>) behind other entry points: one class
        family = {"gemm_nt_swiglu_fwd": "gemm_nt", "gemm_nt_swiglu_bwd": "gemm_nt"}
        for name, cost in table.items():
            saved[name] = getattr(ops, name)
            setattr(ops, name, self._wrap(family.get(name, name) if not detail_names() else name, saved[name], cost))
        for name, cost in other.items():
            saved[name] = getattr(ops, name)
            setattr(ops, name, self._wrap("other:" + name, saved[name], cost))
        try:
            yield self
        finally:
            for name, fn in saved.items():
                setattr(ops, name, fn)

    def summary(self) -> Dict[str, Dict[str, float]]:
        torch.cuda.synchronize()
        agg = defaultdict(lambda: {"ms": 0.0, "flops": 0.0, "bytes": 0.0, "calls": 0})
        for name, fl, nb, s, e in self.records:
            a = agg[name]
            a["ms"] += s.elapsed_time(e)
            a["flops"] += fl
            a["bytes"] += nb
            a["calls"] += 1
        out = {}
        for name, a in agg.items():
            sec = a["ms"] * 1e-3
            out[name] = dict(a, tflops=(a["flops"] / sec / 1e12 if sec > 0 else 0.0), gbs=(a["bytes"] / sec / 1e9 if sec > 0 else 0.0))
        return out
