"""LayerNorm forward / backward micro-benchmark at the engine's row counts (HIP-event timing, algorithmic GB/s).
   EGOM2P_HIP_LIB=<variant .so> python tools/ln_bench.py"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from egom2p_amd import ops


def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


res = {}
for rows, D in [(131072, 768), (65536, 768), (65536, 1024), (65536, 1152), (131072, 1152), (65536, 1536)]:
    x = torch.randn(rows, D, device="cuda"); w = torch.rand(D, device="cuda") + 0.5
    y = torch.empty(rows, D, device="cuda", dtype=torch.bfloat16)
    mean = torch.empty(rows, device="cuda"); rstd = torch.empty(rows, device="cuda")
    t = timeit(lambda: ops.layernorm_fwd(x, w, y, mean, rstd))
    res[f"fwd {rows}x{D}"] = round(rows * D * 6 / t / 1e9)
    dy = torch.randn(rows, D, device="cuda").bfloat16(); dx = torch.randn(rows, D, device="cuda"); dw = torch.zeros(D, device="cuda")
    dxb = torch.empty(rows, D, device="cuda", dtype=torch.bfloat16)
    t = timeit(lambda: ops.layernorm_bwd(dy, x, mean, rstd, w, dx, dw, dx_in=dx, dx_bf16=dxb))
    res[f"bwd+in+bf16 {rows}x{D}"] = round(rows * D * 16 / t / 1e9)
    t = timeit(lambda: ops.layernorm_bwd(dy, x, mean, rstd, w, dx, dw, dx_in=dx))
    res[f"bwd+in {rows}x{D}"] = round(rows * D * 14 / t / 1e9)
    del x, y, dy, dx, dxb
print(json.dumps(res))
