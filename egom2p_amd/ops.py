"""Thin per-kernel wrappers: torch tensors are only containers (device memory + stream); every
call goes straight to the C-ABI of libegom2p_hip.so.  Used by the engine and by the parity tests."""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence

import torch

from . import _lib as L
from ._lib import check

BF16 = torch.bfloat16


def _p(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise L.EgoHipError("egom2p_amd ops need tensors on the GPU; there is no CPU fallback")


# ------------------------------------------------------------------------------------------
def layernorm_fwd(x, w, y, mean, rstd, out_row=None, eps=1e-6, q8=None, qscale=None, width=None):
    """width: the normalised width when the rows are stored padded (x.shape[1] is the pitch; pad columns of x are zero)"""
    _need_cuda(x)
    rows, ld = x.shape
    D = ld if width is None else width
    check(L.load().ego_layernorm_fwd(_p(x), _p(w), _p(y), _p(mean), _p(rstd), _p(out_row), rows, D, ld, eps, _p(q8),
                                     0 if q8 is None else q8.stride(-2), _p(qscale), _stream()), "ego_layernorm_fwd")


_WORK = {}


def _work(device, n_floats):
    """Scratch for the atomic-free reductions (partial rows per workgroup + ordered column sums), one buffer per device,
    reused by every call in stream order; grown on demand (training steps are not graph-captured)."""
    buf = _WORK.get(device)
    if buf is None or buf.numel() < n_floats:
        buf = _WORK[device] = torch.empty(int(n_floats * 1.25) + 1024, device=device, dtype=torch.float32)
    return buf


def layernorm_bwd(dy, x, mean, rstd, w, dx_out, dw, dx_in=None, dx_bf16=None, dy_row=None, width=None):
    _need_cuda(x)
    rows, ld = x.shape
    D = ld if width is None else width
    lib = L.load()
    wk = _work(x.device, lib.ego_layernorm_bwd_work_floats(rows, D))
    check(lib.ego_layernorm_bwd(_p(dy), _p(dy_row), _p(x), _p(mean), _p(rstd), _p(w), _p(dx_in), _p(dx_out),
                                _p(dx_bf16), _p(dw), _p(wk), wk.numel(), rows, D, ld, _stream()), "ego_layernorm_bwd")


def _ptr_array(ts):
    return (C.c_void_p * len(ts))(*[t.data_ptr() for t in ts])


def layernorm_fwd_multi(x, ws, ys, mean, rstd, eps=1e-6, width=None):
    """ys[l] = bf16(LN(x) * ws[l]) for every l in one pass over x (the decoder's per-layer context_norm of the same context)"""
    _need_cuda(x)
    rows, ld = x.shape
    D = ld if width is None else width
    check(L.load().ego_layernorm_fwd_multi(_p(x), len(ws), _ptr_array(ws), _ptr_array(ys), _p(mean), _p(rstd), rows, D, ld, eps, _stream()),
          "ego_layernorm_fwd_multi")


def layernorm_bwd_multi(dys, x, mean, rstd, ws, dx_out, dws, dx_in=None, dx_bf16=None, width=None):
    """dx_out = (dx_in) + sum_l LN-backward(dys[l]; ws[l]), dws[l] += the layers' weight gradients: x is read once"""
    _need_cuda(x)
    rows, ld = x.shape
    D = ld if width is None else width
    lib = L.load()
    wk = _work(x.device, lib.ego_layernorm_bwd_multi_work_floats(rows, D, len(ws)))
    check(lib.ego_layernorm_bwd_multi(len(ws), _ptr_array(dys), _ptr_array(ws), _ptr_array(dws), _p(x), _p(mean), _p(rstd), _p(dx_in),
                                      _p(dx_out), _p(dx_bf16), _p(wk), wk.numel(), rows, D, ld, _stream()), "ego_layernorm_bwd_multi")


def gemm_nt(A, B, C_out, M, N, K, epi=L.EPI_BF16, R=None, bias=None, m_range=None, lda=None, ldb=None, ldc=None, ldr=None):
    """C[M,N] = A[M,K] @ B[N,K]^T (+ epilogue)."""
    _need_cuda(A)
    lda = A.stride(-2) if lda is None else lda
    ldb = B.stride(-2) if ldb is None else ldb
    ldc = C_out.stride(-2) if ldc is None else ldc
    ldr = 0 if R is None else (R.stride(-2) if ldr is None else ldr)
    check(L.load().ego_gemm_nt_bf16(_p(A), lda, _p(B), ldb, _p(C_out), ldc, _p(R), ldr, _p(bias), _p(m_range), M, N, K, epi,
                                    _stream()), "ego_gemm_nt_bf16")


def quant_fp8_rows(X, Q, scale, rows=None, K=None):
    """Q (uint8 e4m3) = X (bf16) / scale[row], scale[row] = amax(row) / 448."""
    _need_cuda(X)
    rows = X.shape[0] if rows is None else rows
    K = X.shape[-1] if K is None else K
    check(L.load().ego_quant_fp8_rows(_p(X), X.stride(-2), rows, K, _p(Q), Q.stride(-2), _p(scale), _stream()), "ego_quant_fp8_rows")


def gemm_nt_fp8(A8, sa, B8, sb, C_out, M, N, K, epi=L.EPI_BF16, R=None, bias=None):
    """C[M,N] = sa[:,None] * sb[None,:] * (A8[M,K] @ B8[N,K]^T) over e4m3 operands (+ epilogue)."""
    _need_cuda(A8)
    check(L.load().ego_gemm_nt_fp8(_p(A8), A8.stride(-2), _p(sa), _p(B8), B8.stride(-2), _p(sb), _p(C_out), C_out.stride(-2), _p(R),
                                   0 if R is None else R.stride(-2), _p(bias), M, N, K, epi, _stream()), "ego_gemm_nt_fp8")


def gemm_nt_swiglu_fwd_fp8(X8, sx, W8, sw, ab, h, M, F, K):
    check(L.load().ego_gemm_nt_swiglu_fwd_fp8(_p(X8), X8.stride(-2), _p(sx), _p(W8), W8.stride(-2), _p(sw), _p(ab), ab.stride(-2), _p(h),
                                              h.stride(-2), M, F, K, _stream()), "ego_gemm_nt_swiglu_fwd_fp8")


def gemm_tn(P, Q, C0, Ni, Nj, M, C1=None, split_row=0, rows0=None, rows1=0, m_range=None, splits=1, slab=None,
            ldp=None, ldq=None, ldc=None):
    """C[Ni,Nj] += P[M,Ni]^T @ Q[M,Nj]."""
    _need_cuda(P)
    ldp = P.stride(-2) if ldp is None else ldp
    ldq = Q.stride(-2) if ldq is None else ldq
    ldc = C0.stride(-2) if ldc is None else ldc
    if C1 is None:
        split_row = Ni
    rows0 = (split_row if rows0 is None else rows0)
    check(L.load().ego_gemm_tn_bf16(_p(P), ldp, _p(Q), ldq, _p(C0), _p(C1), ldc, split_row, rows0, rows1, _p(m_range), Ni, Nj,
                                    M, splits, _p(slab), _stream()), "ego_gemm_tn_bf16")


def clip_synth(key_ids, key_perm, k_in, k_tgt, n, vocab, ids, input_mask, target_mask, dam):
    """one modality of the synthetic input contract for B = len(key_ids) clips, generated on the device (ids=None: masks and
    decoder-attention marker only)"""
    _need_cuda(input_mask)
    check(L.load().ego_clip_synth(_p(key_ids), _p(key_perm), _p(k_in), _p(k_tgt), key_ids.numel(), n, vocab, _p(ids),
                                  _p(input_mask), _p(target_mask), _p(dam), _stream()), "ego_clip_synth")


def budget_dirichlet(keys, in_alphas, tgt_alphas, mix_weights, max_tokens, min_tokens, not_seq, n_in_range, n_tgt_range,
                     k_in, k_tgt, max_tries=100):
    """Dirichlet-mixture token budgets on the device.  in_alphas / tgt_alphas: [n_mix][n_mods]; k_in / k_tgt: int32 [n_mods, B]."""
    _need_cuda(k_in)
    d = L.BudgetDesc()
    d.n_mix, d.n_mods = len(in_alphas), len(in_alphas[0])
    for j in range(d.n_mix):
        d.mix_weight[j] = float(mix_weights[j])
        for i in range(d.n_mods):
            d.in_alpha[j][i] = max(float(in_alphas[j][i]), 1e-9)
            d.tgt_alpha[j][i] = max(float(tgt_alphas[j][i]), 1e-9)
    for i in range(d.n_mods):
        d.max_tokens[i], d.min_tokens[i], d.not_seq[i] = int(max_tokens[i]), int(min_tokens[i]), int(bool(not_seq[i]))
    d.n_in_lo, d.n_in_hi = int(n_in_range[0]), int(n_in_range[1])
    d.n_tgt_lo, d.n_tgt_hi = int(n_tgt_range[0]), int(n_tgt_range[1])
    d.max_tries = int(max_tries)
    check(L.load().ego_budget_dirichlet(C.byref(d), _p(keys), keys.numel(), _p(k_in), _p(k_tgt), _stream()), "ego_budget_dirichlet")


def gemm_kernel_mode(nt256=1, tn256=1):
    """1 = tile family by shape (default), 0 = 128x128 kernels only, 2 = 256x256 wherever legal"""
    check(L.load().ego_gemm_kernel_mode(nt256, tn256), "ego_gemm_kernel_mode")


def gemm_small_tiles(max_tiles128):
    """NT launches of at most this many 128x128 tiles use the 64x64-tile small-grid kernel (0 = never, < 0 = query); returns the
    previous threshold"""
    return L.load().ego_gemm_small_tiles(int(max_tiles128))


def gemm_tune(key, value):
    """ego_gemm_tune: probe / tuning hook (1: start delay of every other persistent NT workgroup, 2: force the low-latency tile
    family - 1 = 128 x 64, 2 = 128 x 128 -, 3: largest 128 x 128 tile count sent to the 128 x 64 kernel); returns the old value"""
    return L.load().ego_gemm_tune(int(key), int(value))


def tn_splits(Ni, Nj, rows, slab_numel, ranged=False, ldp=None, ldq=None):
    """Split-K factor for `gemm_tn`, asked from the launcher itself (ego_gemm_tn_plan)."""
    return L.load().ego_gemm_tn_plan(Ni, Nj, rows, Ni if ldp is None else ldp, Nj if ldq is None else ldq, slab_numel, int(ranged))


def _hd_arg(hd_pad, hd):
    """`hd_pad` argument of the ego_attn_*_hd entries: pitch of a stored head, plus (optional) the head's real dimension in bits 16.."""
    return int(hd_pad) | ((int(hd) << 16) if hd else 0)


def attn_fwd(q, q_bs, q_rs, k, k_bs, k_rs, v, v_bs, v_rs, o, o_bs, o_rs, lse, ks, ke, r_bs, r_rs, B, H, Nq, Nk, scale, o_lo=None,
             hd_pad=64, seg=None, seg_bad=None, hd=None):
    """o_lo (optional device pointer, laid out like o): receives the bf16 rounding residual of the output.
    hd_pad: elements per stored head (64: the throughput kernels; 96 / 128: zero-padded heads of another dimension; hd: that
    dimension, optional - contraction steps over all-padding columns are then skipped)
    seg (int32 [B, n_seg, 2], optional; head_dim 64 only): row groups of a block-diagonal self-attention mask, seg_bad (int32 [B],
    optional): samples that must take the per-row intervals instead (ego_attn_fwd_d64_seg)"""
    if hd_pad == 64 and seg is not None:
        check(L.load().ego_attn_fwd_d64_seg(q, q_bs, q_rs, k, k_bs, k_rs, v, v_bs, v_rs, o, o_bs, o_rs, o_lo, _p(lse), _p(ks), _p(ke),
                                            r_bs, r_rs, _p(seg), seg.shape[-2], _p(seg_bad), B, H, Nq, Nk, scale, _stream()),
              "ego_attn_fwd_d64_seg")
        return
    if hd_pad != 64:
        check(L.load().ego_attn_fwd_hd(q, q_bs, q_rs, k, k_bs, k_rs, v, v_bs, v_rs, o, o_bs, o_rs, o_lo, _p(lse), _p(ks), _p(ke),
                                       r_bs, r_rs, B, H, Nq, Nk, _hd_arg(hd_pad, hd), scale, _stream()), "ego_attn_fwd_hd")
        return
    check(L.load().ego_attn_fwd_d64(q, q_bs, q_rs, k, k_bs, k_rs, v, v_bs, v_rs, o, o_bs, o_rs, o_lo, _p(lse), _p(ks), _p(ke),
                                    r_bs, r_rs, B, H, Nq, Nk, scale, _stream()), "ego_attn_fwd_d64")


def attn_fwd_split(q, q_bs, q_rs, k, k_bs, k_rs, v, v_bs, v_rs, o, o_bs, o_rs, lse, ks, ke, r_bs, r_rs, B, H, Nq, Nk, scale, kv_splits, ws):
    """forward attention (head dim 64) with the keys of every query tile cut into kv_splits runs (under-filled grids); ws: fp32
    scratch of at least attn_fwd_split_floats(B, H, Nq, kv_splits) floats"""
    check(L.load().ego_attn_fwd_d64_split(q, q_bs, q_rs, k, k_bs, k_rs, v, v_bs, v_rs, o, o_bs, o_rs, _p(lse), _p(ks), _p(ke), r_bs, r_rs,
                                          B, H, Nq, Nk, scale, kv_splits, _p(ws), 0 if ws is None else ws.numel(), _stream()),
          "ego_attn_fwd_d64_split")


def attn_fwd_split_floats(B, H, Nq, kv_splits):
    return L.load().ego_attn_fwd_split_floats(B, H, Nq, kv_splits)


def attn_bwd(q, q_bs, q_rs, k, k_bs, k_rs, v, v_bs, v_rs, o, o_bs, o_rs, do, do_bs, do_rs, lse, delta,
             dq, dq_bs, dq_rs, dk, dk_bs, dk_rs, dv, dv_bs, dv_rs, ks, ke, r_bs, r_rs, B, H, Nq, Nk, scale, o_lo=None, hd_pad=64,
             seg=None, seg_bad=None, hd=None):
    if hd_pad == 64 and seg is not None:
        check(L.load().ego_attn_bwd_d64_seg(q, q_bs, q_rs, k, k_bs, k_rs, v, v_bs, v_rs, o, o_bs, o_rs, o_lo, do, do_bs, do_rs,
                                            _p(lse), _p(delta), dq, dq_bs, dq_rs, dk, dk_bs, dk_rs, dv, dv_bs, dv_rs,
                                            _p(ks), _p(ke), r_bs, r_rs, _p(seg), seg.shape[-2], _p(seg_bad), B, H, Nq, Nk, scale,
                                            _stream()), "ego_attn_bwd_d64_seg")
        return
    if hd_pad != 64:
        check(L.load().ego_attn_bwd_hd(q, q_bs, q_rs, k, k_bs, k_rs, v, v_bs, v_rs, o, o_bs, o_rs, o_lo, do, do_bs, do_rs,
                                       _p(lse), _p(delta), dq, dq_bs, dq_rs, dk, dk_bs, dk_rs, dv, dv_bs, dv_rs,
                                       _p(ks), _p(ke), r_bs, r_rs, B, H, Nq, Nk, _hd_arg(hd_pad, hd), scale, _stream()), "ego_attn_bwd_hd")
        return
    check(L.load().ego_attn_bwd_d64(q, q_bs, q_rs, k, k_bs, k_rs, v, v_bs, v_rs, o, o_bs, o_rs, o_lo, do, do_bs, do_rs,
                                    _p(lse), _p(delta), dq, dq_bs, dq_rs, dk, dk_bs, dk_rs, dv, dv_bs, dv_rs,
                                    _p(ks), _p(ke), r_bs, r_rs, B, H, Nq, Nk, scale, _stream()), "ego_attn_bwd_d64")


def swiglu_fwd(ab, h, rows, F):
    check(L.load().ego_swiglu_fwd(_p(ab), _p(h), rows, F, _stream()), "ego_swiglu_fwd")


def swiglu_fwd_fusable(F, K):
    """shapes the fused fc1||fc3 + gate launch accepts (else: gemm_nt + swiglu_fwd)"""
    return F % 128 == 0 and K % 64 == 0 and K >= 128 and os.environ.get("EGOM2P_FUSE_SWIGLU_FWD", "1") != "0"


def gemm_nt_swiglu_fwd(X, W13, ab, h, M, F, K, ldx=None, ldw=None):
    """ab[M,2F] = X[M,K] @ W13[2F,K]^T, h[M,F] = swiglu(ab) in one launch."""
    _need_cuda(X)
    ldx = X.stride(-2) if ldx is None else ldx
    ldw = W13.stride(-2) if ldw is None else ldw
    check(L.load().ego_gemm_nt_swiglu_fwd(_p(X), ldx, _p(W13), ldw, _p(ab), ab.stride(-2), _p(h), h.stride(-2), M, F, K, _stream()),
          "ego_gemm_nt_swiglu_fwd")


def swiglu_bwd_fusable(F, K):
    """shapes the fused fc2-dgrad + gate-backward launch accepts (else: gemm_nt + swiglu_bwd)"""
    return F % 256 == 0 and K % 64 == 0 and K >= 128 and os.environ.get("EGOM2P_FUSE_SWIGLU_BWD", "1") != "0"


def gemm_nt_swiglu_bwd(dY, W2t, ab, dab, M, F, K, ldy=None, ldw=None):
    """dab[M,2F] = swiglu_bwd(ab, dY[M,K] @ W2t[F,K]^T) without materialising dh."""
    _need_cuda(dY)
    ldy = dY.stride(-2) if ldy is None else ldy
    ldw = W2t.stride(-2) if ldw is None else ldw
    check(L.load().ego_gemm_nt_swiglu_bwd(_p(dY), ldy, _p(W2t), ldw, _p(ab), _p(dab), ab.stride(-2), M, F, K, _stream()),
          "ego_gemm_nt_swiglu_bwd")


def swiglu_bwd(ab, dh, dab, rows, F):
    check(L.load().ego_swiglu_bwd(_p(ab), _p(dh), _p(dab), rows, F, _stream()), "ego_swiglu_bwd")


def ce_fwd(logits, ld, V, targets, rng, max_rows, lse, nll):
    check(L.load().ego_ce_fwd(_p(logits), ld, V, _p(targets), _p(rng), max_rows, _p(lse), _p(nll), _stream()), "ego_ce_fwd")


def ce_bwd(logits, ld, V, targets, rng, max_rows, lse, gscale, n_mods, loss_w=None):
    """loss_w: 1-element view of this modality's entry of `loss_weights` (None: loss_type 'mod')"""
    check(L.load().ego_ce_bwd(_p(logits), ld, V, _p(targets), _p(rng), max_rows, _p(lse), _p(gscale), n_mods, _p(loss_w), _stream()),
          "ego_ce_bwd")


def ce_fusable(V):
    """ego_ce_fwd_bwd holds a logits row in one workgroup's registers"""
    return V % 8 == 0 and V <= 65536


def ce_fwd_bwd(logits, ld, V, targets, rng, max_rows, lse, nll, gscale, n_mods, loss_w=None):
    check(L.load().ego_ce_fwd_bwd(_p(logits), ld, V, _p(targets), _p(rng), max_rows, _p(lse), _p(nll), _p(gscale), n_mods, _p(loss_w),
                                  _stream()), "ego_ce_fwd_bwd")


LOSS_MODES = {"mod": 0, "modality": 0, "weighted_mod": 1, "token": 2}


def loss_weights(ranges, vocab, mode, loss_w, mod_scale):
    """per-modality loss weights of loss_type 'weighted_mod' (mode 1) / 'token' (mode 2) from the device-side row counts"""
    arr = (C.c_int * len(vocab))(*[int(v) for v in vocab])
    check(L.load().ego_loss_weights(_p(ranges), arr, len(vocab), mode, _p(loss_w), _p(mod_scale), _stream()), "ego_loss_weights")


def loss_finalize(nll, ranges, n_mods, out, err=None, loss_w=None, mod_scale=None):
    check(L.load().ego_loss_finalize(_p(nll), _p(ranges), n_mods, _p(out), _p(err), _p(loss_w), _p(mod_scale), _stream()),
          "ego_loss_finalize")


def cast_weight(W, Wb=None, Wt=None, rows_dst=None, ld_w=None, ld_t=None):
    rows, cols = W.shape
    rows_dst = rows if rows_dst is None else rows_dst
    ld_w = cols if ld_w is None else ld_w
    ld_t = rows_dst if ld_t is None else ld_t
    check(L.load().ego_cast_weight(_p(W), rows, cols, W.stride(0), _p(Wb), ld_w, _p(Wt), ld_t, rows_dst, _stream()),
          "ego_cast_weight")


def cast_f32_bf16(src, dst):
    check(L.load().ego_cast_f32_bf16(_p(src), _p(dst), src.numel(), _stream()), "ego_cast_f32_bf16")


def bias_grad(g, rows, D, db):
    lib = L.load()
    wk = _work(g.device, lib.ego_bias_grad_work_floats(rows, D))
    check(lib.ego_bias_grad(_p(g), rows, D, _p(db), _p(wk), wk.numel(), _stream()), "ego_bias_grad")


_SQ_WORK = {}


def grad_sqnorm(g, out):
    wk = _SQ_WORK.get(g.device)
    if wk is None:
        wk = _SQ_WORK[g.device] = torch.empty(2048, device=g.device, dtype=torch.float64)      # EGO_SQNORM_WORK
    check(L.load().ego_grad_sqnorm(_p(g), g.numel(), _p(out), _p(wk), _stream()), "ego_grad_sqnorm")


def adamw_step(p, g, m, v, lr, wd, step, beta1=0.9, beta2=0.95, eps=1e-8, gscale=1.0, max_norm=0.0, sqnorm=None,
               zero_grad=False):
    check(L.load().ego_adamw_step(_p(p), _p(g), _p(m), _p(v), p.numel(), lr, wd, beta1, beta2, eps, step, gscale, max_norm,
                                  _p(sqnorm), int(zero_grad), _stream()), "ego_adamw_step")


def compact(masks: Sequence[torch.Tensor], ids: Sequence[torch.Tensor], dams, n_pos, mod_ids, n_keep, is_decoder, out, B, n_reg=0):
    """`out`: dict of preallocated tensors (ids_keep,pad,mod_mask,slot,local,tok,ks,ke,n_valid,seg,err), n_reg + n_keep entries per
    sample (n_reg: register tokens in front of the kept rows, encoder only)."""
    d = L.CompactDesc()
    d.n_mods, d.n_keep, d.is_decoder, d.n_reg = len(masks), n_keep, int(is_decoder), int(n_reg)
    for i in range(len(masks)):
        d.mask[i] = masks[i].data_ptr()
        d.ids[i] = ids[i].data_ptr()
        d.dam[i] = dams[i].data_ptr() if (dams is not None and dams[i] is not None) else None
        d.n_pos[i] = n_pos[i]
        d.mod_id[i] = mod_ids[i]
    for k in ("ids_keep", "pad", "mod_mask", "slot", "local", "tok", "ks", "ke", "n_valid", "seg", "err"):
        setattr(d, k, out[k].data_ptr())
    d.seg_bad = out["seg_bad"].data_ptr() if out.get("seg_bad") is not None else None
    check(L.load().ego_compact(C.byref(d), B, _stream()), "ego_compact")


def embed_fwd(tables, pos, mod, base_vec, slot, local, tok, x, emb, rows, D, reg=None):
    d = L.EmbedDesc()
    d.reg = _p(reg)
    for i in range(len(pos)):
        d.table[i] = None if tables is None or tables[i] is None else tables[i].data_ptr()
        d.pos[i] = pos[i].data_ptr()
        d.mod[i] = mod[i].data_ptr()
    d.base_vec = _p(base_vec)
    d.slot, d.local, d.tok = slot.data_ptr(), local.data_ptr(), tok.data_ptr()
    d.x, d.emb = x.data_ptr(), _p(emb)
    d.rows, d.D = rows, D
    check(L.load().ego_embed_fwd(C.byref(d), _stream()), "ego_embed_fwd")


def embed_bwd(dtables, dmods, dbase, dx, d2, slot, tok, rows, D, touched=None):
    d = L.EmbedBwdDesc()
    for i in range(len(dmods)):
        d.dtable[i] = None if dtables is None or dtables[i] is None else dtables[i].data_ptr()
        d.vocab[i] = 0 if dtables is None or dtables[i] is None else int(dtables[i].shape[0])
        d.dmod[i] = dmods[i].data_ptr()
        d.touched[i] = None if touched is None or touched[i] is None else touched[i].data_ptr()
    d.dbase = _p(dbase)
    d.dx, d.d2 = dx.data_ptr(), _p(d2)
    d.slot, d.tok = slot.data_ptr(), tok.data_ptr()
    d.rows, d.D, d.n_mods = rows, D, len(dmods)
    lib = L.load()
    wk = _work(dx.device, lib.ego_embed_bwd_work_floats(rows, D, len(dmods)))
    d.work, d.work_floats = wk.data_ptr(), wk.numel()
    check(lib.ego_embed_bwd(C.byref(d), _stream()), "ego_embed_bwd")


def reg_grad(dx, B, rows_per_sample, n_reg, D, dreg):
    check(L.load().ego_reg_grad(_p(dx), B, rows_per_sample, n_reg, D, _p(dreg), _stream()), "ego_reg_grad")


def rows_compact(touched, cap, rows, count):
    check(L.load().ego_rows_compact(_p(touched), touched.numel(), cap, _p(rows), _p(count), _stream()), "ego_rows_compact")


def rows_gather(table, rows, count, cap, out):
    check(L.load().ego_rows_gather(_p(table), _p(rows), _p(count), cap, table.shape[-1], _p(out), _stream()), "ego_rows_gather")


def rows_scatter(table, rows, count, cap, src, add):
    check(L.load().ego_rows_scatter(_p(table), _p(rows), _p(count), cap, table.shape[-1], _p(src), int(add), _stream()), "ego_rows_scatter")


def loss_perm(seg, canon, slot, tok, B, M, n_mods, perm, tgt_perm, ranges, base):
    check(L.load().ego_loss_perm(_p(seg), _p(canon), _p(slot), _p(tok), B, M, n_mods, _p(perm), _p(tgt_perm), _p(ranges),
                                 _p(base), _stream()), "ego_loss_perm")


def sample_cfg_topp(cond, uncond, V, cfg_scale, top_p, temperature, uniforms, out_tokens, out_prob=None, ld=None, top_k=0):
    """top_k: number of tokens the top-k filter keeps (0 = off); `top_k_count` turns the reference's int / float argument into it"""
    rows = out_tokens.numel()
    ld = cond.stride(-2) if ld is None else ld
    check(L.load().ego_sample_cfg_topp(_p(cond), _p(uncond), ld, V, cfg_scale, top_p, int(top_k), temperature, _p(uniforms), _p(out_tokens),
                                       _p(out_prob), rows, _stream()), "ego_sample_cfg_topp")


def top_k_count(top_k, V: int) -> int:
    """The k of `top_k_top_p_filtering` (egom2p/models/generate.py:335-342): an int is a count, a float a share of the vocabulary;
    0 / 0.0 = no top-k filter.  k = 0 from a tiny positive share makes torch.topk(..., 0)[0][..., -1] raise in the reference."""
    if top_k is None:
        return 0
    if isinstance(top_k, bool) or not isinstance(top_k, (int, float)):
        raise ValueError(f"Invalid value for top_k: {top_k}")          # the reference's message (:342)
    if top_k <= 0:
        return 0
    k = min(top_k, V) if isinstance(top_k, int) else min(int(top_k * V), V)
    if k <= 0:
        raise ValueError(f"top_k = {top_k} keeps no token of a {V}-token vocabulary")
    return int(k)
