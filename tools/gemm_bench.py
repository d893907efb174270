"""GEMM microbenchmark on the engine's shapes (random bf16 data, HIP-event timing on the launch stream).
   EGO_GEMM_NT_VARIANT=k python tools/gemm_bench.py"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from egom2p_amd import ops, _lib as L

def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3

def main():
    dev = "cuda"
    res = {}
    nt_shapes = [(32768, 2304, 768), (32768, 768, 768), (32768, 4096, 768), (32768, 768, 2048), (32768, 768, 4096),
                 (16144, 64000, 768), (16144, 768, 64000), (4096, 4096, 4096), (8192, 8192, 8192),
                 (65536, 768, 768), (65536, 2304, 768), (65536, 768, 2048), (65536, 1536, 768), (32288, 64000, 768), (32288, 768, 64000)]
    if os.environ.get('SKIP_NT') == '1': nt_shapes = []
    for M, N, K in nt_shapes:
        A = (torch.rand(M, K, device=dev) * 2 - 1).bfloat16()
        B = (torch.rand(N, K, device=dev) * 2 - 1).bfloat16()
        C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        t = timeit(lambda: ops.gemm_nt(A, B, C, M, N, K, L.EPI_BF16))
        res[f"nt {M}x{N}x{K}"] = round(2.0 * M * N * K / t / 1e12, 1)
        del A, B, C
    for M, N, K in ([] if os.environ.get('SKIP_NT') == '1' else [(32768, 768, 768), (32768, 768, 2048), (65536, 768, 768), (65536, 768, 2048)]):
        A = (torch.rand(M, K, device=dev) * 2 - 1).bfloat16()
        B = (torch.rand(N, K, device=dev) * 2 - 1).bfloat16()
        R = torch.randn(M, N, device=dev)
        C = torch.empty(M, N, device=dev)
        t = timeit(lambda: ops.gemm_nt(A, B, C, M, N, K, L.EPI_RESID, R=R))
        res[f"nt_resid {M}x{N}x{K}"] = round(2.0 * M * N * K / t / 1e12, 1)
        del A, B, C, R
    # the two fused SwiGLU launches (fc1||fc3 + gate; fc2 dgrad + gate backward)
    for M, F, K in ([] if os.environ.get('SKIP_NT') == '1' else [(65536, 2048, 768), (131072, 2048, 768)]):
        X = (torch.rand(M, K, device=dev) * 2 - 1).bfloat16()
        W13 = ((torch.rand(2 * F, K, device=dev) * 2 - 1) * 0.05).bfloat16()
        ab = torch.empty(M, 2 * F, device=dev, dtype=torch.bfloat16)
        h = torch.empty(M, F, device=dev, dtype=torch.bfloat16)
        t = timeit(lambda: ops.gemm_nt_swiglu_fwd(X, W13, ab, h, M, F, K))
        res[f"swiglu_fwd {M}x{F}x{K}"] = round(2.0 * M * 2 * F * K / t / 1e12, 1)
        W2t = ((torch.rand(F, K, device=dev) * 2 - 1) * 0.05).bfloat16()
        dab = torch.empty(M, 2 * F, device=dev, dtype=torch.bfloat16)
        t = timeit(lambda: ops.gemm_nt_swiglu_bwd(X, W2t, ab, dab, M, F, K))
        res[f"swiglu_bwd {M}x{F}x{K}"] = round(2.0 * M * F * K / t / 1e12, 1)
        del X, W13, ab, h, W2t, dab
    if os.environ.get("SKIP_TN") != "1":
        slab = torch.empty(64 * 1024 * 1024 // 4, device=dev)
        for M, Ni, Nj in [(65536, 768, 768), (32768, 2304, 768), (32768, 768, 768), (32768, 4096, 768), (32768, 768, 2048), (65536, 2304, 768), (65536, 4096, 768), (65536, 768, 2048), (65536, 1536, 768), (16144, 64000, 768)]:
            P = (torch.rand(M, Ni, device=dev) * 2 - 1).bfloat16()
            Q = (torch.rand(M, Nj, device=dev) * 2 - 1).bfloat16()
            C = torch.zeros(Ni, Nj, device=dev)
            splits = ops.tn_splits(Ni, Nj, M, slab.numel())
            t = timeit(lambda: ops.gemm_tn(P, Q, C, Ni, Nj, M, splits=splits, slab=slab if splits > 1 else None))
            res[f"tn {M}x{Ni}x{Nj} s{splits}"] = round(2.0 * M * Ni * Nj / t / 1e12, 1)
            del P, Q, C
    print(json.dumps({"variant": os.environ.get("EGO_GEMM_NT_VARIANT", "0"), **res}))

if __name__ == "__main__":
    main()
