"""Does a GEMM hold its burst rate?  Same launch repeated for ~2 s, rate per 100-launch window."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from egom2p_amd import ops, _lib as L
M, N, K = [int(x) for x in os.environ.get("SHAPE", "65536,2304,768").split(",")]
dev = "cuda"
A = (torch.rand(M, K, device=dev) * 2 - 1).bfloat16(); B = (torch.rand(N, K, device=dev) * 2 - 1).bfloat16()
nbuf = int(os.environ.get("NBUF", 1))
na, nc = int(os.environ.get("NBUF_A", nbuf)), int(os.environ.get("NBUF_C", nbuf))
Cs = [torch.empty(M, N, device=dev, dtype=torch.bfloat16) for _ in range(nc)]
As = [A] + [A.clone() for _ in range(na - 1)]
res = []
for w in range(int(os.environ.get("WINDOWS", 6))):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(100): ops.gemm_nt(As[i % na], B, Cs[i % nc], M, N, K, L.EPI_BF16)
    e.record(); torch.cuda.synchronize()
    res.append(round(2.0 * M * N * K * 100 / (s.elapsed_time(e) * 1e-3) / 1e12))
print(json.dumps({"shape": [M, N, K], "nbuf_a": na, "nbuf_c": nc, "tflops_per_window": res}))
