"""Attention microbenchmark on the engine's shapes (random bf16 data, HIP-event timing)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from egom2p_amd import ops

def timeit(fn, iters=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3

def main():
    dev = "cuda"; B, H, N = int(os.environ.get("B", 16)), 12, 2048
    D = H * 64
    qkv = (torch.randn(B, N, 3, D, device=dev)).bfloat16()
    o = torch.empty(B, N, D, device=dev, dtype=torch.bfloat16)
    do = torch.randn(B, N, D, device=dev).bfloat16()
    dqkv = torch.empty_like(qkv)
    lse = torch.empty(B, H, N, device=dev); delta = torch.empty(B, H, N, device=dev)
    res = {}
    for kind in os.environ.get("KINDS", "full,half,aligned,blocks").split(","):
        ks = torch.zeros(B, N, dtype=torch.int32, device=dev); ke = torch.full((B, N), N, dtype=torch.int32, device=dev)
        pairs = float(B) * N * N
        if kind == "half":
            ke[:] = 1024; pairs = float(B) * N * 1024
        if kind in ("blocks", "aligned"):
            bounds = [0, 1009, 2018, 2033, 2048] if kind == "blocks" else [0, 1024, 2048]; pairs = 0.0
            for a, b_ in zip(bounds[:-1], bounds[1:]):
                ks[:, a:b_] = a; ke[:, a:b_] = b_; pairs += float(B) * (b_ - a) ** 2
        p = qkv.data_ptr()
        f = lambda: ops.attn_fwd(p, N * 3 * D, 3 * D, p + 2 * D, N * 3 * D, 3 * D, p + 4 * D, N * 3 * D, 3 * D, o.data_ptr(), N * D, D,
                                 lse, ks, ke, N, 1, B, H, N, N, 0.125)
        t = timeit(f)
        res[f"fwd {kind}"] = (round(t * 1e6), round(4 * 64 * H * pairs / t / 1e12, 1))
        g = dqkv.data_ptr()
        b = lambda: ops.attn_bwd(p, N * 3 * D, 3 * D, p + 2 * D, N * 3 * D, 3 * D, p + 4 * D, N * 3 * D, 3 * D, o.data_ptr(), N * D, D,
                                 do.data_ptr(), N * D, D, lse, delta, g, N * 3 * D, 3 * D, g + 2 * D, N * 3 * D, 3 * D, g + 4 * D,
                                 N * 3 * D, 3 * D, ks, ke, N, 1, B, H, N, N, 0.125)
        t = timeit(b)
        res[f"bwd {kind}"] = (round(t * 1e6), round(10 * 64 * H * pairs / t / 1e12, 1))
    print(json.dumps(res))

if __name__ == "__main__":
    main()
