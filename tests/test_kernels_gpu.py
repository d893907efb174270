"""Per-kernel parity tests on a real MI355X: every C-ABI entry point against a plain PyTorch fp32
reference of the same op (integer outputs bit-exact; float tolerances stated per test)."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from egom2p_amd import _lib as L  # noqa: E402
from egom2p_amd import ops  # noqa: E402

from conftest import bar  # noqa: E402

DEV = "cuda"
ATT_FWD_TOL, ATT_BWD_TOL = 4e-3, 8e-3        # attention kernels against fp32 torch (relative L2), stated where they are used


def _rel(a, b):
    a, b = a.double(), b.double()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def _bf(x):
    return x.to(torch.bfloat16)


@pytest.fixture(autouse=True)
def _seed():
    torch.manual_seed(1234)


def test_library_loads():
    assert L.load().ego_abi_version() == L.ABI_VERSION


# ------------------------------------------------------------------------------------------ GEMM NT
@pytest.mark.parametrize("M,N,K", [(300, 256, 128), (4096, 768, 768), (2048, 2304, 768), (1000, 768, 2048), (120, 128, 384)])
def test_gemm_nt_epilogues(M, N, K):
    A = _bf(torch.randn(M, K, device=DEV))
    B = _bf(torch.randn(N, K, device=DEV) * 0.1)
    ref = A.float() @ B.float().t()
    C = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
    ops.gemm_nt(A, B, C, M, N, K, L.EPI_BF16)
    assert _rel(C.float(), ref) < 4e-3
    C32 = torch.empty(M, N, device=DEV)
    ops.gemm_nt(A, B, C32, M, N, K, L.EPI_F32)
    assert _rel(C32, ref) < 1e-5
    R = torch.randn(M, N, device=DEV)
    out = torch.empty(M, N, device=DEV)
    ops.gemm_nt(A, B, out, M, N, K, L.EPI_RESID, R=R)
    assert _rel(out, R + _bf(ref).float()) < 2e-3
    bias = torch.randn(N, device=DEV)
    ops.gemm_nt(A, B, out, M, N, K, L.EPI_BIAS_RESID, R=R, bias=bias)
    assert _rel(out, R + _bf(ref + _bf(bias).float()).float()) < 2e-3


def test_gemm_nt_row_range():
    M, N, K = 1024, 256, 128
    A = _bf(torch.randn(M, K, device=DEV))
    B = _bf(torch.randn(N, K, device=DEV) * 0.1)
    C = torch.full((M, N), 7.0, device=DEV)
    rng = torch.tensor([130, 333], device=DEV, dtype=torch.int32)
    ops.gemm_nt(A, B, C, M, N, K, L.EPI_F32, m_range=rng)
    ref = A.float() @ B.float().t()
    assert _rel(C[130:463], ref[130:463]) < 1e-5
    assert (C[:130] == 7).all() and (C[463:] == 7).all()


@pytest.mark.parametrize("M,N,K", [(1707, 768, 768), (1706, 2304, 768), (1707, 768, 2048), (1707, 4096, 768), (3414, 1536, 768),
                                   (65, 72, 64), (5120, 768, 128)])
def test_gemm_nt_small_grid_kernel_equals_the_128_tile_kernel(M, N, K):
    """Under-filled NT launches (the 1707-row linears of the generation path) run on 64x64 tiles with a 4-deep LDS-DMA ring
    (gemm_nt64_kernel): same K order, same MFMA chain per output element -> bit for bit the 128x128 kernel's result, for every
    epilogue, ragged edges and device-side row ranges included."""
    A = _bf(torch.randn(M, K, device=DEV))
    B = _bf(torch.randn(N, K, device=DEV) * 0.1)
    R = torch.randn(M, N, device=DEV)
    bias = torch.randn(N, device=DEV)
    rng = torch.tensor([M // 3, M // 2], device=DEV, dtype=torch.int32)
    outs = {}
    old = ops.gemm_small_tiles(-1)
    try:
        # t128x64 / t128x128 (round 5): the same kernel template on 128 x 64 and 128 x 128 tiles with a 3-deep ring (the generation
        # path's encoder linears: one to three rounds of 128 x 128 tiles)
        for mode, thr, force in (("t128", 0, 0), ("t64", 1 << 30, 0), ("t128x64", 0, 1), ("t128x128", 0, 2)):
            ops.gemm_small_tiles(thr)
            ops.gemm_tune(2, force)
            ops.gemm_kernel_mode(0, 1)                       # keep the 256x256 family out of the comparison
            C = torch.full((M + 2, N), 5.0, device=DEV, dtype=torch.bfloat16)
            ops.gemm_nt(A, B, C, M, N, K, L.EPI_BF16)
            C32 = torch.full((M + 2, N), 3.0, device=DEV)
            ops.gemm_nt(A, B, C32, M, N, K, L.EPI_F32)
            o1 = torch.empty(M, N, device=DEV)
            ops.gemm_nt(A, B, o1, M, N, K, L.EPI_RESID, R=R)
            o2 = torch.empty(M, N, device=DEV)
            ops.gemm_nt(A, B, o2, M, N, K, L.EPI_BIAS_RESID, R=R, bias=bias)
            Cr = torch.full((M, N), 7.0, device=DEV)
            ops.gemm_nt(A, B, Cr, M, N, K, L.EPI_F32, m_range=rng)
            outs[mode] = (C, C32, o1, o2, Cr)
    finally:
        ops.gemm_small_tiles(old)
        ops.gemm_tune(2, 0)
        ops.gemm_kernel_mode(1, 1)
    for other in ("t64", "t128x64", "t128x128"):
        for a, b in zip(outs["t128"], outs[other]):
            assert torch.equal(a, b), other
    C, C32, o1, o2, Cr = outs["t64"]
    ref = A.float() @ B.float().t()
    assert _rel(C32[:M], ref) < 1e-5 and (C32[M:] == 3).all() and (C[M:] == 5).all()
    assert _rel(C[:M].float(), ref) < 4e-3
    lo, n = M // 3, M // 2
    assert _rel(Cr[lo:lo + n], ref[lo:lo + n]) < 1e-5 and (Cr[:lo] == 7).all() and (Cr[lo + n:] == 7).all()


@pytest.mark.parametrize("M,N,K", [(8292, 5120, 192), (8192, 5120, 64), (16384 + 7, 2560, 128), (65536, 768, 768),
                                   (40000, 1152, 1152)])      # ego-L width: the last column tile is half empty
def test_gemm_nt256_persistent(M, N, K):
    """>= 640 tiles of 256 x 256 with a bf16 output run on the persistent staggered kernel: several tiles per
    workgroup, odd / single K-step counts (stage parity carries over the tile seam), ragged last row tile."""
    A = _bf(torch.randn(M, K, device=DEV))
    B = _bf(torch.randn(N, K, device=DEV) * 0.1)
    C = torch.full((M + 3, N), 5.0, device=DEV, dtype=torch.bfloat16)
    ops.gemm_nt(A, B, C, M, N, K, L.EPI_BF16)
    ref = A.float() @ B.float().t()
    assert _rel(C[:M].float(), ref) < 4e-3
    assert (C[M:] == 5).all()
    # exact per-element check on a strided sample (catches a misplaced tile, which a norm could hide)
    idx = torch.arange(0, M, 97, device=DEV)
    assert (C[idx].float() - ref[idx]).abs().max().item() <= ref[idx].abs().max().item() * 2 ** -7     # one bf16 ulp


def test_gemm_nt256_fp32_epilogues():
    M, N, K = 8292, 5120, 192
    A = _bf(torch.randn(M, K, device=DEV))
    B = _bf(torch.randn(N, K, device=DEV) * 0.1)
    ref = A.float() @ B.float().t()
    C32 = torch.full((M + 2, N), 3.0, device=DEV)
    ops.gemm_nt(A, B, C32, M, N, K, L.EPI_F32)
    assert _rel(C32[:M], ref) < 1e-5 and (C32[M:] == 3).all()
    R = torch.randn(M, N, device=DEV)
    out = torch.empty(M, N, device=DEV)
    ops.gemm_nt(A, B, out, M, N, K, L.EPI_RESID, R=R)
    assert _rel(out, R + _bf(ref).float()) < 2e-3
    assert (out - R - ref).abs().max().item() <= ref.abs().max().item() * 2 ** -7
    ops.gemm_nt(A, B, R, M, N, K, L.EPI_RESID, R=R)                # in place on the residual stream, like the engine
    assert torch.equal(R, out)
    bias = torch.randn(N, device=DEV)
    R2 = torch.randn(M, N, device=DEV)
    ops.gemm_nt(A, B, out, M, N, K, L.EPI_BIAS_RESID, R=R2, bias=bias)
    assert _rel(out, R2 + _bf(ref + _bf(bias).float()).float()) < 2e-3


def test_gemm_nt_swiglu_fwd_fused_equals_two_calls():
    """fc1||fc3 + gate in one launch writes bit for bit what gemm_nt(EPI_BF16) + swiglu_fwd write (same kernel family)"""
    try:
        ops.gemm_kernel_mode(2, 1)                      # reference GEMM on the 256x256 kernel too: same summation order
        for M, F, K in [(4200, 2048, 768), (8192 + 9, 384, 128)]:
            X = _bf(torch.randn(M, K, device=DEV))
            W13 = _bf(torch.randn(2 * F, K, device=DEV) * 0.05)
            ab_ref = torch.empty(M, 2 * F, device=DEV, dtype=torch.bfloat16)
            h_ref = torch.empty(M, F, device=DEV, dtype=torch.bfloat16)
            ops.gemm_nt(X, W13, ab_ref, M, 2 * F, K, L.EPI_BF16)
            ops.swiglu_fwd(ab_ref, h_ref, M, F)
            ab = torch.full((M + 1, 2 * F), 9.0, device=DEV, dtype=torch.bfloat16)
            h = torch.full((M + 1, F), 9.0, device=DEV, dtype=torch.bfloat16)
            ops.gemm_nt_swiglu_fwd(X, W13, ab, h, M, F, K)
            if (2 * F) % 256 == 0:
                assert torch.equal(ab[:M], ab_ref)
                assert torch.equal(h[:M], h_ref)
            else:                                        # reference ran on the 128x128 kernel: rounding-level agreement
                assert _rel(ab[:M].float(), ab_ref.float()) < 1e-3 and _rel(h[:M].float(), h_ref.float()) < 2e-3
            assert (ab[M:] == 9).all() and (h[M:] == 9).all()
            # against plain torch
            ref = X.float() @ W13.float().t()
            assert _rel(ab[:M].float(), ref) < 4e-3
    finally:
        ops.gemm_kernel_mode(1, 1)


def test_gemm_nt_swiglu_bwd_fused_equals_two_calls():
    """fc2 dgrad + gate backward in one launch writes bit for bit what gemm_nt(EPI_BF16) + swiglu_bwd write"""
    for M, F, K in [(4200, 2048, 768), (8192 + 9, 512, 128)]:
        dY = _bf(torch.randn(M, K, device=DEV))
        W2t = _bf(torch.randn(F, K, device=DEV) * 0.05)          # fc2 weight transposed: [F, K]
        ab = _bf(torch.randn(M, 2 * F, device=DEV) * 2)
        dh = torch.empty(M, F, device=DEV, dtype=torch.bfloat16)
        ref = torch.empty(M, 2 * F, device=DEV, dtype=torch.bfloat16)
        ops.gemm_nt(dY, W2t, dh, M, F, K, L.EPI_BF16)
        ops.swiglu_bwd(ab, dh, ref, M, F)
        out = torch.full((M + 1, 2 * F), 9.0, device=DEV, dtype=torch.bfloat16)
        ops.gemm_nt_swiglu_bwd(dY, W2t, ab, out, M, F, K)
        assert torch.equal(out[:M], ref)
        assert (out[M:] == 9).all()


def test_gemm_nt256_row_range():
    M, N, K = 9000, 5120, 128                  # 36 x 20 = 720 tiles by the host bound, 33 x 20 by the device range
    A = _bf(torch.randn(M, K, device=DEV))
    B = _bf(torch.randn(N, K, device=DEV) * 0.1)
    C = torch.full((M, N), 7.0, device=DEV, dtype=torch.bfloat16)
    rng = torch.tensor([77, 8300], device=DEV, dtype=torch.int32)
    ops.gemm_nt(A, B, C, M, N, K, L.EPI_BF16, m_range=rng)
    ref = A[77:8377].float() @ B.float().t()
    assert _rel(C[77:8377].float(), ref) < 4e-3
    assert (C[:77] == 7).all() and (C[8377:] == 7).all()


# ------------------------------------------------------------------------------------------ GEMM TN
@pytest.mark.parametrize("M,Ni,Nj,splits", [(1000, 256, 128, 1), (4096, 768, 768, 4), (777, 2304, 768, 3), (64, 128, 128, 1),
                                             # whole-step shapes with >= 128 (tile, split) pairs run on the 256x256 staggered kernel
                                             (8192, 2304, 768, 9), (2368, 2304, 768, 9), (16384, 1024, 1024, 8), (64 * 5, 2048, 1024, 5),
                                             # ego-L width: 13.5 x 4.5 tiles, the last row / column tiles are half empty
                                             (4096 + 33, 3456, 1152, 4)])
def test_gemm_tn(M, Ni, Nj, splits):
    P = _bf(torch.randn(M, Ni, device=DEV))
    Q = _bf(torch.randn(M, Nj, device=DEV))
    C0 = torch.randn(Ni, Nj, device=DEV)
    ref = C0 + P.float().t() @ Q.float()
    slab = torch.empty(splits, Ni, Nj, device=DEV) if splits > 1 else None
    ops.gemm_tn(P, Q, C0, Ni, Nj, M, splits=splits, slab=slab)
    assert _rel(C0, ref) < 1e-5


def test_gemm_tn_split_rows_and_range():
    M, Ni, Nj = 900, 256, 128           # fused/padded output: rows [0,128) -> C0 (100 valid), [128,256) -> C1 (90 valid)
    P = _bf(torch.randn(M, Ni, device=DEV))
    Q = _bf(torch.randn(M, Nj, device=DEV))
    C0 = torch.zeros(100, Nj, device=DEV)
    C1 = torch.zeros(90, Nj, device=DEV)
    rng = torch.tensor([100, 700], device=DEV, dtype=torch.int32)
    for splits in (1, 2):
        C0.zero_(); C1.zero_()
        slab = torch.empty(splits, Ni, Nj, device=DEV) if splits > 1 else None
        ops.gemm_tn(P, Q, C0, Ni, Nj, M, C1=C1, split_row=128, rows0=100, rows1=90, m_range=rng, splits=splits, slab=slab)
        ref = P[100:800].float().t() @ Q[100:800].float()
        assert _rel(C0, ref[:100]) < 1e-5
        assert _rel(C1, ref[128:218]) < 1e-5


def test_gemm_tn256_range_ragged_and_split_rows():
    """the 256x256 kernel takes device row ranges and partial last steps through its buffer bounds (zero fill)"""
    M, Ni, Nj = 900, 8192, 1024          # 32 x 4 tiles of 256 x 256 -> the launcher picks the 256 kernel
    P = _bf(torch.randn(M, Ni, device=DEV))
    Q = _bf(torch.randn(M, Nj, device=DEV))
    C0 = torch.zeros(4000, Nj, device=DEV)
    C1 = torch.zeros(4090, Nj, device=DEV)
    rng = torch.tensor([100, 701], device=DEV, dtype=torch.int32)      # 701 rows: 10 full steps + 61 rows
    ops.gemm_tn(P, Q, C0, Ni, Nj, M, C1=C1, split_row=4096, rows0=4000, rows1=4090, m_range=rng)
    ref = P[100:801].float().t() @ Q[100:801].float()
    assert _rel(C0, ref[:4000]) < 1e-5
    assert _rel(C1, ref[4096:4096 + 4090]) < 1e-5
    # empty range: nothing is added
    C0.zero_()
    ops.gemm_tn(P, Q, C0, Ni, Nj, M, C1=C1, split_row=4096, rows0=4000, rows1=4090,
                m_range=torch.tensor([5, 0], device=DEV, dtype=torch.int32))
    assert float(C0.abs().max()) == 0.0
    # ragged M without a range, split-K
    M2, Ni2, Nj2 = 64 * 7 + 13, 2304, 768
    P2 = _bf(torch.randn(M2, Ni2, device=DEV)); Q2 = _bf(torch.randn(M2, Nj2, device=DEV))
    G = torch.randn(Ni2, Nj2, device=DEV)
    ref2 = G + P2.float().t() @ Q2.float()
    ops.gemm_tn(P2, Q2, G, Ni2, Nj2, M2, splits=8, slab=torch.empty(8, Ni2, Nj2, device=DEV))
    assert _rel(G, ref2) < 1e-5


def test_ring_kernels_are_race_free_and_deterministic():
    """The persistent NT and the TN 256-kernels order their LDS rings only by counted vmcnt + barriers (no fences): an
    early read would still pass a tolerance check whenever the DMA happens to land first.  Screen for it: the kernels
    are deterministic, so repeated launches (with other traffic in between) must reproduce the first result bit for bit."""
    noise = torch.empty(64 * 1024 * 1024, device=DEV)
    for M, N, K in [(16384 + 5, 2560, 128), (32768, 768, 768), (12288, 1152, 1152)]:
        A = _bf(torch.randn(M, K, device=DEV)); B = _bf(torch.randn(N, K, device=DEV) * 0.1)
        C0 = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
        ops.gemm_kernel_mode(2, 1)
        try:
            ops.gemm_nt(A, B, C0, M, N, K, L.EPI_BF16)
            for it in range(12):
                noise.normal_()                                   # evict caches, shift timing
                C = torch.zeros_like(C0)
                ops.gemm_nt(A, B, C, M, N, K, L.EPI_BF16)
                assert torch.equal(C, C0), (M, N, K, it)
        finally:
            ops.gemm_kernel_mode(1, 1)
    slab = torch.empty(64 * 1024 * 1024 // 4, device=DEV)
    for M, Ni, Nj in [(8192 + 17, 2304, 768), (16384, 768, 768), (4096, 3456, 1152)]:
        P = _bf(torch.randn(M, Ni, device=DEV)); Q = _bf(torch.randn(M, Nj, device=DEV))
        splits = ops.tn_splits(Ni, Nj, M, slab.numel())
        G0 = torch.zeros(Ni, Nj, device=DEV)
        ops.gemm_tn(P, Q, G0, Ni, Nj, M, splits=splits, slab=slab if splits > 1 else None)
        for it in range(12):
            noise.normal_()
            G = torch.zeros(Ni, Nj, device=DEV)
            ops.gemm_tn(P, Q, G, Ni, Nj, M, splits=splits, slab=slab if splits > 1 else None)
            assert torch.equal(G, G0), (M, Ni, Nj, it)


def test_gemms_on_operands_beyond_4gib():
    """the logit-gradient operand of a large micro-batch exceeds 4 GiB: the 256-kernels re-base their buffer
    descriptors per tile / per step, so 32-bit buffer offsets never limit the operand size"""
    M, V, D = 40000, 64000, 256
    A = torch.empty(M, V, device=DEV, dtype=torch.bfloat16)            # 5.1 GB
    for r in range(0, M, 4000):
        A[r:r + 4000] = _bf(torch.randn(4000, V, device=DEV))
    W = _bf(torch.randn(D, V, device=DEV) * 0.02)
    C = torch.empty(M, D, device=DEV, dtype=torch.bfloat16)
    ops.gemm_nt(A, W, C, M, D, V, L.EPI_BF16)                          # dgrad of the logits: [M, V] x [D, V]^T
    for lo, hi in ((0, 64), (33500, 33700), (M - 100, M)):             # 33554 rows = 4 GiB
        ref = A[lo:hi].float() @ W.float().t()
        assert _rel(C[lo:hi].float(), ref) < 4e-3, (lo, hi)
    X = _bf(torch.randn(M, D, device=DEV))
    G = torch.zeros(V, D, device=DEV)
    ops.gemm_tn(A, X, G, V, D, M)                                      # wgrad of the table: A^T X
    for lo, hi in ((0, 256), (63744, 64000)):
        ref = A[:, lo:hi].float().t() @ X.float()
        assert _rel(G[lo:hi], ref) < 1e-4, (lo, hi)
    del A


# ------------------------------------------------------------------------------------------ attention
def _attn_ref(q, k, v, ks, ke, scale):
    """fp32 reference with the reference's masked_fill(-max) semantics. q:(B,H,Nq,64) k,v:(B,H,Nk,64); ks,ke:(B,Nq)."""
    B, H, Nq, _ = q.shape
    Nk = k.shape[2]
    s = (q @ k.transpose(-1, -2)) * scale
    j = torch.arange(Nk, device=q.device)[None, None, :]
    blocked = ~((j >= ks[:, :, None]) & (j < ke[:, :, None].clamp(max=Nk)))
    s = s.masked_fill(blocked[:, None], -torch.finfo(torch.float32).max)
    p = s.softmax(-1)
    return p @ v


@pytest.mark.parametrize("B,H,Nq,Nk,kind", [(2, 3, 200, 333, "ragged"), (1, 12, 2048, 2048, "blocks"), (2, 2, 30, 30, "pad"),
                                             (1, 2, 256, 512, "full")])
def test_attention_fwd_bwd(B, H, Nq, Nk, kind):
    D = H * 64
    # packed layouts like the engine's: q rows [B,Nq,H,64], kv rows [B,Nk,2,H,64]
    qb = _bf(torch.randn(B, Nq, D, device=DEV))
    kvb = _bf(torch.randn(B, Nk, 2, D, device=DEV))
    ks = torch.zeros(B, Nq, dtype=torch.int32, device=DEV)
    ke = torch.full((B, Nq), Nk, dtype=torch.int32, device=DEV)
    if kind == "ragged":
        ks = torch.randint(0, Nk // 2, (B, Nq), device=DEV, dtype=torch.int32)
        ke = ks + torch.randint(0, Nk // 2, (B, Nq), device=DEV, dtype=torch.int32)   # some empty intervals
        ke[:, 5] = ks[:, 5]                                                          # force an empty one
    elif kind == "blocks":
        bounds = [0, 1009, 1024, 2033, 2048]
        for a, b_ in zip(bounds[:-1], bounds[1:]):
            ks[:, a:b_] = a
            ke[:, a:b_] = b_
    elif kind == "pad":
        ke[:] = 17
    scale = 64 ** -0.5
    q = qb.view(B, Nq, H, 64).permute(0, 2, 1, 3).float().requires_grad_(True)
    k = kvb[:, :, 0].reshape(B, Nk, H, 64).permute(0, 2, 1, 3).float().requires_grad_(True)
    v = kvb[:, :, 1].reshape(B, Nk, H, 64).permute(0, 2, 1, 3).float().requires_grad_(True)
    ref = _attn_ref(q, k, v, ks.long(), ke.long(), scale)                # (B,H,Nq,64)
    o = torch.empty(B, Nq, D, device=DEV, dtype=torch.bfloat16)
    lse = torch.empty(B, H, Nq, device=DEV)
    kp, vp = kvb.data_ptr(), kvb.data_ptr() + D * 2
    ops.attn_fwd(qb.data_ptr(), Nq * D, D, kp, Nk * 2 * D, 2 * D, vp, Nk * 2 * D, 2 * D, o.data_ptr(), Nq * D, D, lse, ks, ke,
                 Nq, 1, B, H, Nq, Nk, scale)
    got = o.view(B, Nq, H, 64).permute(0, 2, 1, 3).float()
    # stated tolerances: forward 4e-3, backward 8e-3 relative L2 against fp32 torch on the same bf16 inputs (bf16 output rounding
    # alone is 2^-9 / sqrt(3) = 1.1e-3; the engine-level bf16-oracle test measures ~2e-3 / ~4e-3 through these kernels), and each
    # value is also held to its recorded baseline + 30 % (conftest.bar)
    tag = f"{B}x{H}x{Nq}x{Nk}.{kind}"
    bar(f"attn_fwd.{tag}", _rel(got, ref), hard=ATT_FWD_TOL)

    do = _bf(torch.randn(B, Nq, D, device=DEV))
    ref.backward(do.view(B, Nq, H, 64).permute(0, 2, 1, 3).float())
    dq = torch.zeros(B, Nq, D, device=DEV, dtype=torch.bfloat16)
    dkv = torch.zeros(B, Nk, 2, D, device=DEV, dtype=torch.bfloat16)
    delta = torch.empty(B, H, Nq, device=DEV)
    ops.attn_bwd(qb.data_ptr(), Nq * D, D, kp, Nk * 2 * D, 2 * D, vp, Nk * 2 * D, 2 * D, o.data_ptr(), Nq * D, D,
                 do.data_ptr(), Nq * D, D, lse, delta, dq.data_ptr(), Nq * D, D, dkv.data_ptr(), Nk * 2 * D, 2 * D,
                 dkv.data_ptr() + D * 2, Nk * 2 * D, 2 * D, ks, ke, Nq, 1, B, H, Nq, Nk, scale)
    gq = dq.view(B, Nq, H, 64).permute(0, 2, 1, 3).float()
    gk = dkv[:, :, 0].reshape(B, Nk, H, 64).permute(0, 2, 1, 3).float()
    gv = dkv[:, :, 1].reshape(B, Nk, H, 64).permute(0, 2, 1, 3).float()
    bar(f"attn_bwd_dq.{tag}", _rel(gq, q.grad), hard=ATT_BWD_TOL)
    bar(f"attn_bwd_dk.{tag}", _rel(gk, k.grad), hard=ATT_BWD_TOL)
    bar(f"attn_bwd_dv.{tag}", _rel(gv, v.grad), hard=ATT_BWD_TOL)


@pytest.mark.parametrize("B,H,N,groups", [
    (2, 12, 2048, [[(0, 1009), (1009, 1009), (2018, 15), (2033, 15)], [(0, 1009), (1009, 1009), (2018, 15), (2033, 15)]]),   # the bench's clip
    (3, 2, 600, [[(0, 130), (130, 0), (130, 300), (430, 7)],       # an empty group, a 163-row tail of padding rows
                 [(0, 64), (64, 128), (192, 192), (384, 216)],      # tile-aligned groups, no tail
                 [(0, 1), (1, 1), (2, 1), (3, 590)]]),              # one-row groups, short tail
    (2, 3, 300, [[(0, 100), (100, 100), (200, 100)], [(0, 100), (100, 100), (200, 100)]]),
])
def test_attention_row_groups(B, H, N, groups):
    """ego_attn_*_d64_seg: block-diagonal self-attention launched by row groups (one interval per workgroup) against fp32 torch
    on the per-row intervals, and against the per-row launches of the same intervals.  Tail rows (behind the last group) keep
    arbitrary per-row intervals - pointing into a group, empty (= uniform attention), anything; sample 1 of the last case is
    flagged seg_bad and carries intervals that are NOT its groups: it must come out as the per-row launch computes it."""
    D = H * 64
    n_seg = len(groups[0])
    qkv = _bf(torch.randn(B, N, 3, D, device=DEV))
    ks = torch.zeros(B, N, dtype=torch.int32, device=DEV)
    ke = torch.zeros(B, N, dtype=torch.int32, device=DEV)
    seg = torch.tensor(groups, dtype=torch.int32, device=DEV)                    # [B, n_seg, 2]
    bad = torch.zeros(B, dtype=torch.int32, device=DEV)
    gen = torch.Generator(device=DEV).manual_seed(5)
    for b in range(B):
        end = 0
        for s0, c in groups[b]:
            ks[b, s0:s0 + c] = s0
            ke[b, s0:s0 + c] = s0 + c
            end = max(end, s0 + c)
        if end < N:                                               # tail rows: some group's interval, an empty one, or a random one
            t = N - end
            pick = torch.randint(0, n_seg + 2, (t,), device=DEV, generator=gen)
            for i in range(t):
                j = int(pick[i])
                if j < n_seg:
                    ks[b, end + i], ke[b, end + i] = groups[b][j][0], groups[b][j][0] + groups[b][j][1]
                elif j == n_seg:
                    ks[b, end + i], ke[b, end + i] = 7, 7
                else:
                    ks[b, end + i], ke[b, end + i] = 3, end
    if B == 2 and N == 300:
        bad[1] = 1
        ks[1] = torch.randint(0, 150, (N,), device=DEV, generator=gen).int()
        ke[1] = ks[1] + torch.randint(0, 150, (N,), device=DEV, generator=gen).int()
    scale = 0.125
    q = qkv[:, :, 0].reshape(B, N, H, 64).permute(0, 2, 1, 3).float().requires_grad_(True)
    k = qkv[:, :, 1].reshape(B, N, H, 64).permute(0, 2, 1, 3).float().requires_grad_(True)
    v = qkv[:, :, 2].reshape(B, N, H, 64).permute(0, 2, 1, 3).float().requires_grad_(True)
    ref = _attn_ref(q, k, v, ks.long(), ke.long(), scale)
    do = _bf(torch.randn(B, N, D, device=DEV))
    ref.backward(do.view(B, N, H, 64).permute(0, 2, 1, 3).float())
    p3 = qkv.data_ptr()

    def run(**kw):
        o = torch.full((B, N, D), 9.0, device=DEV, dtype=torch.bfloat16)
        lse = torch.empty(B, H, N, device=DEV)
        delta = torch.empty(B, H, N, device=DEV)
        dqkv = torch.full((B, N, 3, D), 9.0, device=DEV, dtype=torch.bfloat16)
        g = dqkv.data_ptr()
        ops.attn_fwd(p3, N * 3 * D, 3 * D, p3 + 2 * D, N * 3 * D, 3 * D, p3 + 4 * D, N * 3 * D, 3 * D, o.data_ptr(), N * D, D, lse,
                     ks, ke, N, 1, B, H, N, N, scale, **kw)
        ops.attn_bwd(p3, N * 3 * D, 3 * D, p3 + 2 * D, N * 3 * D, 3 * D, p3 + 4 * D, N * 3 * D, 3 * D, o.data_ptr(), N * D, D,
                     do.data_ptr(), N * D, D, lse, delta, g, N * 3 * D, 3 * D, g + 2 * D, N * 3 * D, 3 * D, g + 4 * D, N * 3 * D, 3 * D,
                     ks, ke, N, 1, B, H, N, N, scale, **kw)
        torch.cuda.synchronize()
        return o, lse, dqkv

    o_r, lse_r, d_r = run()
    o_g, lse_g, d_g = run(seg=seg, seg_bad=bad)
    hv = lambda t: t.reshape(B, N, H, 64).permute(0, 2, 1, 3).float()
    tag = f"{B}x{H}x{N}"
    bar(f"attn_seg_fwd.{tag}", _rel(hv(o_g), ref), hard=ATT_FWD_TOL)
    for i, (nm, gr) in enumerate((("dq", q.grad), ("dk", k.grad), ("dv", v.grad))):
        bar(f"attn_seg_bwd_{nm}.{tag}", _rel(hv(d_g[:, :, i]), gr), hard=ATT_BWD_TOL)
    # against the per-row launches: the same arithmetic up to the order of the key tiles (online softmax / fp32 sums)
    bar(f"attn_seg_vs_rows_o.{tag}", _rel(o_g.float(), o_r.float()), hard=3e-3)
    assert (lse_g - lse_r).abs().max().item() < 1e-3
    bar(f"attn_seg_vs_rows_d.{tag}", _rel(d_g.float(), d_r.float()), hard=5e-3)
    if int(bad.sum()):                       # the flagged sample takes the per-row path itself: bit for bit
        assert torch.equal(o_g[1], o_r[1]) and torch.equal(d_g[1], d_r[1])
    # every row written exactly (no 9.0 left), and two launches agree bit for bit
    assert not (o_g == 9).all(-1).any() and not (d_g == 9).all(-1).any()
    o_g2, _, d_g2 = run(seg=seg, seg_bad=bad)
    assert torch.equal(o_g, o_g2) and torch.equal(d_g, d_g2)


@pytest.mark.parametrize("B,H,Nq,Nk,splits", [(1, 12, 1707, 1707, 3), (1, 12, 1706, 3414, 4), (2, 3, 200, 333, 2), (1, 2, 130, 8534, 8),
                                               (1, 2, 64, 100, 5)])
def test_attention_fwd_split_keys(B, H, Nq, Nk, splits):
    """ego_attn_fwd_d64_split (under-filled grids of the generation path): the keys of every query tile handled by `splits`
    workgroups + a combine kernel - against fp32 torch and against the unsplit launch; per-sample key counts (the encoder's
    padding), runs that see no key at all (more runs than key tiles), LSE as the unsplit kernel writes it."""
    D = H * 64
    qb = _bf(torch.randn(B, Nq, D, device=DEV))
    kvb = _bf(torch.randn(B, Nk, 2, D, device=DEV))
    ksb = torch.zeros(B, dtype=torch.int32, device=DEV)
    keb = torch.tensor([Nk, max(1, Nk // 3)][:B], dtype=torch.int32, device=DEV)
    scale = 0.125
    q = qb.view(B, Nq, H, 64).permute(0, 2, 1, 3).float()
    k = kvb[:, :, 0].reshape(B, Nk, H, 64).permute(0, 2, 1, 3).float()
    v = kvb[:, :, 1].reshape(B, Nk, H, 64).permute(0, 2, 1, 3).float()
    ref = _attn_ref(q, k, v, ksb.long()[:, None].expand(B, Nq), keb.long()[:, None].expand(B, Nq), scale)
    kp, vp = kvb.data_ptr(), kvb.data_ptr() + D * 2
    o0 = torch.empty(B, Nq, D, device=DEV, dtype=torch.bfloat16)
    lse0 = torch.empty(B, H, Nq, device=DEV)
    ops.attn_fwd(qb.data_ptr(), Nq * D, D, kp, Nk * 2 * D, 2 * D, vp, Nk * 2 * D, 2 * D, o0.data_ptr(), Nq * D, D, lse0, ksb, keb, 1, 0,
                 B, H, Nq, Nk, scale)
    ws = torch.empty(ops.attn_fwd_split_floats(B, H, Nq, splits), device=DEV)
    o1 = torch.full((B, Nq, D), 9.0, device=DEV, dtype=torch.bfloat16)
    lse1 = torch.empty(B, H, Nq, device=DEV)
    ops.attn_fwd_split(qb.data_ptr(), Nq * D, D, kp, Nk * 2 * D, 2 * D, vp, Nk * 2 * D, 2 * D, o1.data_ptr(), Nq * D, D, lse1, ksb, keb, 1, 0,
                       B, H, Nq, Nk, scale, splits, ws)
    got = o1.view(B, Nq, H, 64).permute(0, 2, 1, 3).float()
    tag = f"{B}x{H}x{Nq}x{Nk}x{splits}"
    bar(f"attn_split_fwd.{tag}", _rel(got, ref), hard=ATT_FWD_TOL)
    # (both outputs are bf16 roundings of fp32 values that agree to ~1e-6: up to one bf16 ulp apart where a rounding boundary falls
    #  between them; recorded 3.04e-3 on the (1, 12, 1707, 1707, 3) case - held to the recorded value + 15 %)
    bar(f"attn_split_vs_unsplit.{tag}", _rel(o1.float(), o0.float()), hard=4e-3, rel_margin=0.15)
    assert (lse1 - lse0).abs().max().item() < 1e-3
    with pytest.raises(L.EgoHipError):                       # scratch too small
        ops.attn_fwd_split(qb.data_ptr(), Nq * D, D, kp, Nk * 2 * D, 2 * D, vp, Nk * 2 * D, 2 * D, o1.data_ptr(), Nq * D, D, lse1, ksb, keb,
                           1, 0, B, H, Nq, Nk, scale, splits, ws[:-64])


@pytest.mark.parametrize("hd,hdp", [(68, 96), (68, 128), (66, 128), (96, 96), (120, 128)])
@pytest.mark.parametrize("B,H,Nq,Nk,kind", [(2, 3, 200, 333, "ragged"), (1, 5, 1100, 1100, "blocks"), (2, 2, 30, 30, "pad"),
                                             (1, 2, 256, 512, "full"), (3, 2, 100, 160, "sample")])
def test_attention_other_head_dims(hd, hdp, B, H, Nq, Nk, kind):
    """ego_attn_*_hd: heads of dimension hd stored zero-padded to hdp (the registered ego-L has 15 heads of 68) against fp32
    torch on the UNPADDED heads; the gradient's pad columns must come out exactly zero (they are the pad rows' weights' only
    source of gradient)."""
    A = H * hdp
    def padded(*shape):
        t = torch.zeros(*shape, H, hdp, device=DEV)
        t[..., :hd] = torch.randn(*shape, H, hd, device=DEV)
        return _bf(t)
    qb = padded(B, Nq).view(B, Nq, A)
    kvb = padded(B, Nk, 2).view(B, Nk, 2, A)
    ks = torch.zeros(B, Nq, dtype=torch.int32, device=DEV)
    ke = torch.full((B, Nq), Nk, dtype=torch.int32, device=DEV)
    r_bs, r_rs = Nq, 1
    if kind == "ragged":
        ks = torch.randint(0, Nk // 2, (B, Nq), device=DEV, dtype=torch.int32)
        ke = ks + torch.randint(0, Nk // 2, (B, Nq), device=DEV, dtype=torch.int32)
        ke[:, 5] = ks[:, 5]
    elif kind == "blocks":
        bounds = [0, 500, 530, 1070, 1100]
        for a, b_ in zip(bounds[:-1], bounds[1:]):
            ks[:, a:b_] = a
            ke[:, a:b_] = b_
    elif kind == "pad":
        ke[:] = 17
    elif kind == "sample":
        nv = torch.tensor([160, 77, 1], dtype=torch.int32, device=DEV)
        ke = nv[:, None].expand(B, Nq).contiguous()
    scale = hd ** -0.5
    q = qb.view(B, Nq, H, hdp)[..., :hd].permute(0, 2, 1, 3).float().requires_grad_(True)
    k = kvb[:, :, 0].reshape(B, Nk, H, hdp)[..., :hd].permute(0, 2, 1, 3).float().requires_grad_(True)
    v = kvb[:, :, 1].reshape(B, Nk, H, hdp)[..., :hd].permute(0, 2, 1, 3).float().requires_grad_(True)
    ref = _attn_ref(q, k, v, ks.long(), ke.long(), scale)
    if kind == "sample":                     # one interval per sample: row stride 0
        ks_k, ke_k, r_bs, r_rs = torch.zeros(B, dtype=torch.int32, device=DEV), nv, 1, 0
    else:
        ks_k, ke_k = ks, ke
    o = torch.empty(B, Nq, A, device=DEV, dtype=torch.bfloat16)
    o_lo = torch.empty_like(o)
    lse = torch.empty(B, H, Nq, device=DEV)
    kp, vp = kvb.data_ptr(), kvb.data_ptr() + A * 2
    ops.attn_fwd(qb.data_ptr(), Nq * A, A, kp, Nk * 2 * A, 2 * A, vp, Nk * 2 * A, 2 * A, o.data_ptr(), Nq * A, A, lse, ks_k, ke_k,
                 r_bs, r_rs, B, H, Nq, Nk, scale, o_lo=o_lo.data_ptr(), hd_pad=hdp)
    got = o.view(B, Nq, H, hdp).float()
    assert (got[..., hd:] == 0).all()
    assert _rel(got[..., :hd].permute(0, 2, 1, 3), ref) < 1e-2
    full = (o.float() + o_lo.float()).view(B, Nq, H, hdp)[..., :hd].permute(0, 2, 1, 3)
    assert _rel(full, ref) < _rel(got[..., :hd].permute(0, 2, 1, 3), ref)        # the residual adds bits (the rest is P's own bf16 rounding)

    do = padded(B, Nq).view(B, Nq, A)
    ref.backward(do.view(B, Nq, H, hdp)[..., :hd].permute(0, 2, 1, 3).float())
    dq = torch.full((B, Nq, A), 7.0, device=DEV, dtype=torch.bfloat16)
    dkv = torch.full((B, Nk, 2, A), 7.0, device=DEV, dtype=torch.bfloat16)
    delta = torch.empty(B, H, Nq, device=DEV)
    ops.attn_bwd(qb.data_ptr(), Nq * A, A, kp, Nk * 2 * A, 2 * A, vp, Nk * 2 * A, 2 * A, o.data_ptr(), Nq * A, A,
                 do.data_ptr(), Nq * A, A, lse, delta, dq.data_ptr(), Nq * A, A, dkv.data_ptr(), Nk * 2 * A, 2 * A,
                 dkv.data_ptr() + A * 2, Nk * 2 * A, 2 * A, ks_k, ke_k, r_bs, r_rs, B, H, Nq, Nk, scale, o_lo=o_lo.data_ptr(), hd_pad=hdp)
    gq = dq.view(B, Nq, H, hdp).float()
    gk = dkv[:, :, 0].reshape(B, Nk, H, hdp).float()
    gv = dkv[:, :, 1].reshape(B, Nk, H, hdp).float()
    for nm, g_, r_ in (("dq", gq, q.grad), ("dk", gk, k.grad), ("dv", gv, v.grad)):
        assert (g_[..., hd:] == 0).all(), nm
        assert _rel(g_[..., :hd].permute(0, 2, 1, 3), r_) < 2e-2, (nm, _rel(g_[..., :hd].permute(0, 2, 1, 3), r_))
    # the same calls with the head's real dimension given (hd_pad = pitch | hd << 16): contraction steps over columns that are
    # all padding are skipped (68 of 96: 5 of 6 steps) - exact zeros left out, the same bits
    o2, o2_lo, lse2 = torch.empty_like(o), torch.empty_like(o), torch.empty_like(lse)
    ops.attn_fwd(qb.data_ptr(), Nq * A, A, kp, Nk * 2 * A, 2 * A, vp, Nk * 2 * A, 2 * A, o2.data_ptr(), Nq * A, A, lse2, ks_k, ke_k,
                 r_bs, r_rs, B, H, Nq, Nk, scale, o_lo=o2_lo.data_ptr(), hd_pad=hdp, hd=hd)
    assert torch.equal(o2, o) and torch.equal(o2_lo, o_lo) and torch.equal(lse2, lse)
    dq2, dkv2, delta2 = torch.full_like(dq, 7.0), torch.full_like(dkv, 7.0), torch.empty_like(delta)
    ops.attn_bwd(qb.data_ptr(), Nq * A, A, kp, Nk * 2 * A, 2 * A, vp, Nk * 2 * A, 2 * A, o.data_ptr(), Nq * A, A,
                 do.data_ptr(), Nq * A, A, lse, delta2, dq2.data_ptr(), Nq * A, A, dkv2.data_ptr(), Nk * 2 * A, 2 * A,
                 dkv2.data_ptr() + A * 2, Nk * 2 * A, 2 * A, ks_k, ke_k, r_bs, r_rs, B, H, Nq, Nk, scale, o_lo=o_lo.data_ptr(), hd_pad=hdp, hd=hd)
    assert torch.equal(dq2, dq) and torch.equal(dkv2, dkv) and torch.equal(delta2, delta)


def test_attention_per_sample_interval():
    B, H, Nq, Nk = 3, 2, 100, 160
    D = H * 64
    qb = _bf(torch.randn(B, Nq, D, device=DEV))
    kvb = _bf(torch.randn(B, Nk, 2, D, device=DEV))
    nv = torch.tensor([160, 77, 1], dtype=torch.int32, device=DEV)
    zero = torch.zeros(B, dtype=torch.int32, device=DEV)
    o = torch.empty(B, Nq, D, device=DEV, dtype=torch.bfloat16)
    lse = torch.empty(B, H, Nq, device=DEV)
    ops.attn_fwd(qb.data_ptr(), Nq * D, D, kvb.data_ptr(), Nk * 2 * D, 2 * D, kvb.data_ptr() + 2 * D, Nk * 2 * D, 2 * D,
                 o.data_ptr(), Nq * D, D, lse, zero, nv, 1, 0, B, H, Nq, Nk, 0.125)
    q = qb.view(B, Nq, H, 64).permute(0, 2, 1, 3).float()
    k = kvb[:, :, 0].reshape(B, Nk, H, 64).permute(0, 2, 1, 3).float()
    v = kvb[:, :, 1].reshape(B, Nk, H, 64).permute(0, 2, 1, 3).float()
    ref = _attn_ref(q, k, v, zero[:, None].expand(B, Nq).long(), nv[:, None].expand(B, Nq).long(), 0.125)
    assert _rel(o.view(B, Nq, H, 64).permute(0, 2, 1, 3).float(), ref) < 1e-2


# ------------------------------------------------------------------------------------------ layernorm
@pytest.mark.parametrize("rows,D", [(1000, 768), (37, 128), (513, 1152)])
def test_layernorm(rows, D):
    x = torch.randn(rows, D, device=DEV) * 2 + 0.3
    w = torch.randn(D, device=DEV) * 0.1 + 1
    perm = torch.randperm(rows, device=DEV).int()
    perm[3] = -1
    y = torch.zeros(rows, D, device=DEV, dtype=torch.bfloat16)
    mean = torch.empty(rows, device=DEV)
    rstd = torch.empty(rows, device=DEV)
    ops.layernorm_fwd(x, w, y, mean, rstd, out_row=perm)
    xr = x.clone().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    ref = torch.nn.functional.layer_norm(xr, (D,), wr, None, 1e-6)
    keep = perm >= 0
    assert _rel(y[perm[keep].long()].float(), ref[keep]) < 4e-3
    dy = _bf(torch.randn(rows, D, device=DEV))           # stored in the permuted row order
    dy_rows = torch.zeros(rows, D, device=DEV)
    dy_rows[keep] = dy[perm[keep].long()].float()
    ref.backward(dy_rows)
    dx_in = torch.randn(rows, D, device=DEV)
    dx = torch.empty(rows, D, device=DEV)
    dxb = torch.empty(rows, D, device=DEV, dtype=torch.bfloat16)
    dw = torch.zeros(D, device=DEV)
    ops.layernorm_bwd(dy, x, mean, rstd, w, dx, dw, dx_in=dx_in, dx_bf16=dxb, dy_row=perm)
    assert _rel(dx, dx_in + xr.grad) < 1e-5
    assert _rel(dxb.float(), dx_in + xr.grad) < 4e-3
    assert _rel(dw, wr.grad) < 1e-4


# ------------------------------------------------------------------------------------------ swiglu / casts / CE
@pytest.mark.parametrize("rows,D,ld,Lyr", [(1000, 768, 768, 12), (37, 128, 128, 3), (513, 1152, 1152, 24), (130, 1020, 1024, 2), (64, 126, 128, 1)])
def test_layernorm_one_input_many_layers(rows, D, ld, Lyr):
    """ego_layernorm_fwd_multi / _bwd_multi (the decoder's per-layer context_norm of one context tensor) against the chained
    single-layer calls: outputs, statistics and dx bit for bit (same expressions, same order of the layer sum), the weight
    gradients to fp32 round-off (their partial rows are cut differently), padded pitches included; and against fp32 torch."""
    x = torch.zeros(rows, ld, device=DEV)
    x[:, :D] = torch.randn(rows, D, device=DEV) * 2 + 0.3
    ws = [torch.rand(D, device=DEV) + 0.5 for _ in range(Lyr)]
    dys = []
    for _ in range(Lyr):
        d = torch.zeros(rows, ld, device=DEV)
        d[:, :D] = torch.randn(rows, D, device=DEV)
        dys.append(_bf(d))
    # chained single-layer calls (decoder order: last layer first in the backward)
    ys1 = [torch.full((rows, ld), 7.0, device=DEV, dtype=torch.bfloat16) for _ in range(Lyr)]
    mean1, rstd1 = torch.empty(rows, device=DEV), torch.empty(rows, device=DEV)
    for l in range(Lyr):
        ops.layernorm_fwd(x, ws[l], ys1[l], mean1, rstd1, width=D)
    dx1, dxb1 = torch.empty(rows, ld, device=DEV), torch.empty(rows, ld, device=DEV, dtype=torch.bfloat16)
    dws1 = [torch.full((D,), 0.25, device=DEV) for _ in range(Lyr)]
    for k, l in enumerate(reversed(range(Lyr))):
        ops.layernorm_bwd(dys[l], x, mean1, rstd1, ws[l], dx1, dws1[l], dx_in=None if k == 0 else dx1, dx_bf16=dxb1, width=D)
    # fused
    ys2 = [torch.full((rows, ld), 7.0, device=DEV, dtype=torch.bfloat16) for _ in range(Lyr)]
    mean2, rstd2 = torch.empty(rows, device=DEV), torch.empty(rows, device=DEV)
    ops.layernorm_fwd_multi(x, ws, ys2, mean2, rstd2, width=D)
    dx2, dxb2 = torch.full((rows, ld), 9.0, device=DEV), torch.full((rows, ld), 9.0, device=DEV, dtype=torch.bfloat16)
    dws2 = [torch.full((D,), 0.25, device=DEV) for _ in range(Lyr)]
    ops.layernorm_bwd_multi(dys, x, mean2, rstd2, ws, dx2, dws2, dx_bf16=dxb2, width=D)
    torch.cuda.synchronize()
    assert torch.equal(mean1, mean2) and torch.equal(rstd1, rstd2)
    for l in range(Lyr):
        assert torch.equal(ys1[l], ys2[l]), l
        assert _rel(dws2[l] - 0.25, dws1[l] - 0.25) < 1e-5, l
    assert torch.equal(dx1, dx2) and torch.equal(dxb1, dxb2)
    # fp32 torch reference of the sum of the layers' LayerNorm backward
    xr = x[:, :D].clone().requires_grad_(True)
    tot = 0
    for l in range(Lyr):
        yl = torch.nn.functional.layer_norm(xr, (D,), ws[l], None, 1e-6)
        tot = tot + (yl * dys[l][:, :D].float()).sum()
    tot.backward()
    assert _rel(dx2[:, :D], xr.grad) < 1e-5
    assert not dx2[:, D:].any()
    # accumulate on top of an existing gradient (dx_in)
    base = torch.randn(rows, ld, device=DEV)
    dx3 = torch.empty(rows, ld, device=DEV)
    ops.layernorm_bwd_multi(dys, x, mean2, rstd2, ws, dx3, [torch.zeros(D, device=DEV) for _ in range(Lyr)], dx_in=base, width=D)
    assert _rel(dx3[:, :D], (dx2 + base)[:, :D]) < 1e-6


def test_swiglu():
    rows, F = 777, 2048
    ab = _bf(torch.randn(rows, 2 * F, device=DEV))
    h = torch.empty(rows, F, device=DEV, dtype=torch.bfloat16)
    ops.swiglu_fwd(ab, h, rows, F)
    a = ab[:, :F].float().requires_grad_(True)
    b = ab[:, F:].float().requires_grad_(True)
    ref = torch.nn.functional.silu(a) * b
    assert _rel(h.float(), ref) < 6e-3
    dh = _bf(torch.randn(rows, F, device=DEV))
    ref.backward(dh.float())
    dab = torch.empty(rows, 2 * F, device=DEV, dtype=torch.bfloat16)
    ops.swiglu_bwd(ab, dh, dab, rows, F)
    assert _rel(dab[:, :F].float(), a.grad) < 4e-3
    assert _rel(dab[:, F:].float(), b.grad) < 4e-3


def test_cast_weight_pad_and_transpose():
    W = torch.randn(341, 128, device=DEV)
    Wb = torch.full((384, 128), 9.0, device=DEV, dtype=torch.bfloat16)
    Wt = torch.full((128, 384), 9.0, device=DEV, dtype=torch.bfloat16)
    ops.cast_weight(W, Wb, Wt, rows_dst=384)
    assert torch.equal(Wb[:341], _bf(W)) and (Wb[341:] == 0).all()
    assert torch.equal(Wt[:, :341], _bf(W).t()) and (Wt[:, 341:] == 0).all()


def test_cross_entropy():
    n_total, V = 300, 64000
    logits = _bf(torch.randn(n_total, V, device=DEV) * 2)
    tgt = torch.randint(0, V, (n_total,), device=DEV, dtype=torch.int32)
    rng = torch.tensor([40, 200], device=DEV, dtype=torch.int32)
    lse = torch.zeros(n_total, device=DEV)
    nll = torch.zeros(n_total, device=DEV)
    ops.ce_fwd(logits, V, V, tgt, rng, 260, lse, nll)
    lf = logits[40:240].float().requires_grad_(True)
    ref = torch.nn.functional.cross_entropy(lf, tgt[40:240].long(), reduction="none")
    assert _rel(nll[40:240], ref) < 1e-5
    out = torch.zeros(3, device=DEV)
    ranges = torch.tensor([40, 200, 0, 0], device=DEV, dtype=torch.int32)
    ops.loss_finalize(nll, ranges, 2, out)
    assert abs(out[1].item() - ref.mean().item()) < 1e-4 and out[2].item() == 0
    assert abs(out[0].item() - ref.mean().item() / 2) < 1e-4
    (ref.mean() / 2 * 0.5).backward()
    g = torch.tensor([0.5], device=DEV)
    keep = logits.clone()
    ops.ce_bwd(logits, V, V, tgt, rng, 260, lse, g, 2)
    assert _rel(logits[40:240].float(), lf.grad) < 5e-3
    assert torch.equal(logits[:40], keep[:40]) and torch.equal(logits[240:], keep[240:])


@pytest.mark.parametrize("D,ld", [(1020, 1024), (2046, 2048), (126, 128)])
def test_layernorm_on_padded_rows(D, ld):
    """LayerNorm over D columns stored in rows of ld (the registered ego-L: 1020 in 1024; ego-XL: 2046 in 2048, where the last
    16-byte chunk is half pad): statistics over the D, pad columns of the output and of the input gradient written as zeros,
    weight gradient of the D only."""
    rows = 777
    x = torch.zeros(rows, ld, device=DEV); x[:, :D] = torch.randn(rows, D, device=DEV) * 2 + 0.3
    w = torch.zeros(ld, device=DEV); w[:D] = torch.rand(D, device=DEV) + 0.5
    y = torch.full((rows, ld), 9.0, device=DEV, dtype=torch.bfloat16)
    mean, rstd = torch.empty(rows, device=DEV), torch.empty(rows, device=DEV)
    ops.layernorm_fwd(x, w, y, mean, rstd, eps=1e-6, width=D)
    xr = x[:, :D].clone().requires_grad_(True)
    wr = w[:D].clone().requires_grad_(True)
    ref = torch.nn.functional.layer_norm(xr, (D,), wr, None, 1e-6)
    assert _rel(y[:, :D].float(), ref) < 4e-3 and not bool(y[:, D:].any())
    assert _rel(mean, xr.mean(-1)) < 1e-5
    dy = torch.zeros(rows, ld, device=DEV); dy[:, :D] = torch.randn(rows, D, device=DEV)
    dy = _bf(dy)
    ref.backward(dy[:, :D].float())
    dx_in = torch.zeros(rows, ld, device=DEV); dx_in[:, :D] = torch.randn(rows, D, device=DEV)
    dx = torch.full((rows, ld), 9.0, device=DEV)
    dxb = torch.full((rows, ld), 9.0, device=DEV, dtype=torch.bfloat16)
    dw = torch.zeros(ld, device=DEV)
    ops.layernorm_bwd(dy, x, mean, rstd, w, dx, dw, dx_in=dx_in, dx_bf16=dxb, width=D)
    assert _rel(dx[:, :D], xr.grad + dx_in[:, :D]) < 1e-4 and not bool(dx[:, D:].any()) and not bool(dxb[:, D:].any())
    assert _rel(dw[:D], wr.grad) < 1e-4 and not bool(dw[D:].any())


@pytest.mark.parametrize("V", [64000, 256, 65536, 2056])
def test_cross_entropy_one_pass_equals_the_two_calls(V):
    """ego_ce_fwd_bwd (training: forward and backward of the CE in one pass over the logits) writes bit for bit what
    ego_ce_fwd followed by ego_ce_bwd writes; rows outside the range stay untouched; V > 65536 is refused."""
    n_total = 300
    logits = _bf(torch.randn(n_total, V, device=DEV) * 3)
    tgt = torch.randint(0, V, (n_total,), device=DEV, dtype=torch.int32)
    tgt[41], tgt[42] = 0, V - 1
    rng = torch.tensor([40, 200], device=DEV, dtype=torch.int32)
    g = torch.tensor([0.25], device=DEV)
    two = logits.clone()
    lse2, nll2 = torch.zeros(n_total, device=DEV), torch.zeros(n_total, device=DEV)
    ops.ce_fwd(two, V, V, tgt, rng, 260, lse2, nll2)
    ops.ce_bwd(two, V, V, tgt, rng, 260, lse2, g, 4)
    one = logits.clone()
    lse1, nll1 = torch.zeros(n_total, device=DEV), torch.zeros(n_total, device=DEV)
    assert ops.ce_fusable(V)
    ops.ce_fwd_bwd(one, V, V, tgt, rng, 260, lse1, nll1, g, 4)
    assert torch.equal(lse1, lse2) and torch.equal(nll1, nll2)
    assert torch.equal(one, two)
    assert torch.equal(one[:40], logits[:40]) and torch.equal(one[240:], logits[240:])
    assert not ops.ce_fusable(65544)
    with pytest.raises(L.EgoHipError):
        ops.ce_fwd_bwd(one, 65544, 65544, tgt, rng, 1, lse1, nll1, g, 4)


def test_bias_grad_and_cast():
    g = _bf(torch.randn(1000, 768, device=DEV))
    db = torch.zeros(768, device=DEV)
    ops.bias_grad(g, 1000, 768, db)
    assert _rel(db, g.float().sum(0)) < 1e-5
    src = torch.randn(4096, device=DEV)
    dst = torch.empty(4096, device=DEV, dtype=torch.bfloat16)
    ops.cast_f32_bf16(src, dst)
    assert torch.equal(dst, _bf(src))


# ------------------------------------------------------------------------------------------ optimiser
def test_adamw_matches_torch():
    n = 100003
    p0 = torch.randn(n + 1, device=DEV)[:n]          # keep 16-byte alignment of the base
    p0 = torch.randn(n, device=DEV)
    g = torch.randn(n, device=DEV) * 3
    pr = p0.clone().requires_grad_(True)
    opt = torch.optim.AdamW([pr], lr=1e-3, betas=(0.9, 0.95), eps=1e-8, weight_decay=0.05)
    p = p0.clone(); m = torch.zeros(n, device=DEV); v = torch.zeros(n, device=DEV)
    sq = torch.zeros(1, device=DEV, dtype=torch.float64)
    for step in (1, 2, 3):
        gs = g * step
        pr.grad = gs.clone()
        total = torch.nn.utils.clip_grad_norm_([pr], 1.0)
        opt.step()
        sq.zero_()
        graw = gs.clone()
        ops.grad_sqnorm(graw, sq)
        assert abs(math.sqrt(sq.item()) - total.item()) < 1e-4 * total.item()
        ops.adamw_step(p, graw, m, v, 1e-3, 0.05, step, gscale=1.0, max_norm=1.0, sqnorm=sq, zero_grad=True)
        assert (graw == 0).all()
        assert _rel(p, pr.detach()) < 1e-6


def test_attention_kernels_are_deterministic_under_load():
    """same screen for the attention kernels' 3-stage LDS-DMA rings (forward, dQ, dK/dV)"""
    B, H, N = 4, 12, 2048
    D = H * 64
    qkv = _bf(torch.randn(B, N, 3, D, device=DEV))
    do = _bf(torch.randn(B, N, D, device=DEV))
    ks = torch.zeros(B, N, dtype=torch.int32, device=DEV)
    ke = torch.full((B, N), N, dtype=torch.int32, device=DEV)
    ks[:, 1009:] = 1009; ke[:, :1009] = 1009                      # two blocks: partial tiles at the seam
    noise = torch.empty(64 * 1024 * 1024, device=DEV)
    p = qkv.data_ptr()

    def run():
        o = torch.zeros(B, N, D, device=DEV, dtype=torch.bfloat16)
        lse = torch.zeros(B, H, N, device=DEV); delta = torch.zeros(B, H, N, device=DEV)
        dqkv = torch.zeros_like(qkv)
        ops.attn_fwd(p, N * 3 * D, 3 * D, p + 2 * D, N * 3 * D, 3 * D, p + 4 * D, N * 3 * D, 3 * D, o.data_ptr(), N * D, D,
                     lse, ks, ke, N, 1, B, H, N, N, 0.125)
        g = dqkv.data_ptr()
        ops.attn_bwd(p, N * 3 * D, 3 * D, p + 2 * D, N * 3 * D, 3 * D, p + 4 * D, N * 3 * D, 3 * D, o.data_ptr(), N * D, D,
                     do.data_ptr(), N * D, D, lse, delta, g, N * 3 * D, 3 * D, g + 2 * D, N * 3 * D, 3 * D, g + 4 * D,
                     N * 3 * D, 3 * D, ks, ke, N, 1, B, H, N, N, 0.125)
        return o, lse, dqkv

    o0, l0, d0 = run()
    for it in range(8):
        noise.normal_()
        o, l, d = run()
        assert torch.equal(o, o0) and torch.equal(l, l0) and torch.equal(d, d0), it


@pytest.mark.parametrize("M,N,K,epi", [(1000, 768, 768, "bf16"), (512, 2304, 768, "bf16"), (700, 768, 2048, "resid"), (384, 1152, 1152, "bias_resid")])
def test_gemm_nt_fp8_against_dequantised_operands(M, N, K, epi):
    """BASELINE config 5: e4m3 operands with per-row / per-output-channel scales on v_mfma_scale_f32_16x16x128_f8f6f4.
    (1) exact contract: equals an fp32 matmul of the DEQUANTISED operands up to summation order - checks the operand
    layout, the K-tiling in 128-byte steps, the scale application and every epilogue; (2) accuracy: within e4m3's
    quantisation noise of the bf16 product (3 mantissa bits: ~2^-4 / sqrt(3) per operand element, averaging over K)."""
    torch.manual_seed(M + N)
    A = (torch.randn(M, K, device=DEV) * torch.rand(M, 1, device=DEV) * 3).bfloat16()         # rows of different magnitude
    B = (torch.randn(N, K, device=DEV) * 0.05).bfloat16()
    A8 = torch.empty(M, K, device=DEV, dtype=torch.uint8); sa = torch.empty(M, device=DEV)
    B8 = torch.empty(N, K, device=DEV, dtype=torch.uint8); sb = torch.empty(N, device=DEV)
    ops.quant_fp8_rows(A, A8, sa)
    ops.quant_fp8_rows(B, B8, sb)
    # quantiser: scale = amax / 448, values round-to-nearest e4m3
    assert torch.allclose(sa, A.float().abs().amax(1) / 448.0, rtol=1e-6)
    Ad = A8.view(torch.float8_e4m3fn).float() * sa[:, None]
    Bd = B8.view(torch.float8_e4m3fn).float() * sb[:, None]
    assert float((Ad - A.float()).abs().max() / A.float().abs().max()) < 2.0 ** -4
    ref = Ad.double() @ Bd.double().t()
    R = torch.randn(M, N, device=DEV)
    bias = torch.randn(N, device=DEV)
    if epi == "bf16":
        C = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
        ops.gemm_nt_fp8(A8, sa, B8, sb, C, M, N, K, L.EPI_BF16)
        want = ref.float().bfloat16().float()
        tol = 2.0 ** -8
    elif epi == "resid":
        C = torch.empty(M, N, device=DEV)
        ops.gemm_nt_fp8(A8, sa, B8, sb, C, M, N, K, L.EPI_RESID, R=R)
        want = R + ref.float().bfloat16().float()
        tol = 2.0 ** -8
    else:
        C = torch.empty(M, N, device=DEV)
        ops.gemm_nt_fp8(A8, sa, B8, sb, C, M, N, K, L.EPI_BIAS_RESID, R=R, bias=bias)
        want = R + (ref.float() + bias.bfloat16().float()).bfloat16().float()
        tol = 2.0 ** -8
    torch.cuda.synchronize()
    err = (C.float() - want).abs().max() / want.abs().max()
    assert float(err) < tol, float(err)                                  # one bf16 ulp of slack for the output rounding
    exact = (A.double() @ B.double().t()).float()
    got = C.float() - (0 if epi == "bf16" else R) - (bias.bfloat16().float() if epi == "bias_resid" else 0)
    rel = float((got - exact).norm() / exact.norm())
    assert rel < 4e-2, rel                                               # e4m3 quantisation noise of both operands


def test_fused_fc13_gate_fp8_equals_two_call_form():
    M, F, K = 1280, 2048, 768
    torch.manual_seed(3)
    X = torch.randn(M, K, device=DEV).bfloat16()
    W = (torch.randn(2 * F, K, device=DEV) * 0.04).bfloat16()
    X8 = torch.empty(M, K, device=DEV, dtype=torch.uint8); sx = torch.empty(M, device=DEV)
    W8 = torch.empty(2 * F, K, device=DEV, dtype=torch.uint8); sw = torch.empty(2 * F, device=DEV)
    ops.quant_fp8_rows(X, X8, sx); ops.quant_fp8_rows(W, W8, sw)
    ab = torch.empty(M, 2 * F, device=DEV, dtype=torch.bfloat16); h = torch.empty(M, F, device=DEV, dtype=torch.bfloat16)
    ops.gemm_nt_swiglu_fwd_fp8(X8, sx, W8, sw, ab, h, M, F, K)
    ab2 = torch.empty_like(ab); h2 = torch.empty_like(h)
    ops.gemm_nt_fp8(X8, sx, W8, sw, ab2, M, 2 * F, K, L.EPI_BF16)
    ops.swiglu_fwd(ab2, h2, M, F)
    torch.cuda.synchronize()
    assert torch.equal(ab, ab2) and torch.equal(h, h2)
