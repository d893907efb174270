"""Correctness spot check of the NT GEMM entry against torch on shapes that take the large-tile kernels (used with
EGOM2P_HIP_LIB=<variant .so> while experimenting with kernel variants)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from egom2p_amd import ops, _lib as L

torch.manual_seed(0)
worst = 0.0
for M, N, K in [(8200, 2304, 768), (65536 - 3, 768, 768), (4099, 1536, 2048), (16144, 64000, 768)]:
    A = (torch.rand(M, K, device="cuda") * 2 - 1).bfloat16()
    B = (torch.rand(N, K, device="cuda") * 2 - 1).bfloat16()
    C = torch.full((M, N), 7.0, device="cuda", dtype=torch.bfloat16)
    ops.gemm_nt(A, B, C, M, N, K, L.EPI_BF16)
    rows = torch.cat([torch.arange(0, 300), torch.randint(0, M, (600,)), torch.arange(M - 300, M)]).cuda()
    ref = (A[rows].float() @ B.float().t())
    err = ((C[rows].float() - ref).abs().max() / ref.abs().max()).item()
    worst = max(worst, err)
    print(f"nt {M}x{N}x{K}: max err / max |ref| = {err:.2e}")
assert worst < 1e-2, worst
print("ok")
