"""wgrad GEMM: MALL-warm (small M, repeated) vs HBM-cold (rotating operand sets) rates."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from egom2p_amd import ops
dev = "cuda"
slab = torch.empty(64 * 1024 * 1024 // 4, device=dev)
res = {}
for (M, Ni, Nj, nbuf) in [(65536, 2304, 768, 1), (65536, 2304, 768, 4), (16384, 2304, 768, 1), (16384, 2304, 768, 16), (65536, 4096, 768, 1), (65536, 4096, 768, 4)]:
    Ps = [(torch.rand(M, Ni, device=dev) * 2 - 1).bfloat16() for _ in range(nbuf)]
    Qs = [(torch.rand(M, Nj, device=dev) * 2 - 1).bfloat16() for _ in range(nbuf)]
    C = torch.zeros(Ni, Nj, device=dev)
    splits = ops.tn_splits(Ni, Nj, M, slab.numel())
    def run(i): ops.gemm_tn(Ps[i % nbuf], Qs[i % nbuf], C, Ni, Nj, M, splits=splits, slab=slab if splits > 1 else None)
    for i in range(8): run(i)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(64): run(i)
    e.record(); torch.cuda.synchronize()
    res[f"tn {M}x{Ni}x{Nj} nbuf{nbuf} s{splits}"] = round(2.0 * M * Ni * Nj * 64 / (s.elapsed_time(e) * 1e-3) / 1e12)
    del Ps, Qs
print(json.dumps(res))
