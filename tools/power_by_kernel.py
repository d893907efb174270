"""Shader clock and socket power per kernel class: each class is looped for ~SECS seconds on the training step's shapes while
rocm-smi is polled (reads only).  Which kernels run at the power cap, which clock higher?

    SECS=8 python tools/power_by_kernel.py > gpurun_out/power_by_kernel.json
"""
import json
import os
import re
import statistics
import subprocess
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from egom2p_amd import _lib as L  # noqa: E402
from egom2p_amd import ops  # noqa: E402

DEV = "cuda"
SECS = float(os.environ.get("SECS", 8))


class Sampler(threading.Thread):
    def __init__(self):
        super().__init__(daemon=True)
        self.rows, self.stop = [], False

    def run(self):
        while not self.stop:
            try:
                out = subprocess.run(["rocm-smi", "-d", "0", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=5).stdout
                s = re.search(r"sclk clock level: \S+ \((\d+)Mhz\)", out)
                p = re.search(r"Power \(W\): ([\d.]+)", out)
                if s and p:
                    self.rows.append((time.time(), int(s.group(1)), float(p.group(1))))
            except Exception:
                pass
            time.sleep(0.3)


def loop(name, fn, flops=0.0, nbytes=0.0):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    smp = Sampler(); smp.start()
    t0 = time.time(); n = 0
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    while time.time() - t0 < SECS:
        for _ in range(20):
            fn()
        n += 20
        torch.cuda.synchronize()
    e.record(); torch.cuda.synchronize()
    smp.stop = True; smp.join()
    ms = s.elapsed_time(e) / n
    rows = [r for r in smp.rows if r[0] - t0 > 1.5]            # after the ramp
    res = {"kernel": name, "us_per_launch": round(ms * 1e3, 1), "tflops": round(flops / ms / 1e9, 1) if flops else None,
           "tb_per_s": round(nbytes / ms / 1e9, 2) if nbytes else None,
           "sclk_mhz_median": statistics.median(r[1] for r in rows) if rows else None,
           "power_w_median": statistics.median(r[2] for r in rows) if rows else None, "samples": len(rows)}
    print(json.dumps(res), flush=True)
    time.sleep(2.0)


def bf(*shape, scale=1.0):
    return ((torch.rand(*shape, device=DEV) * 2 - 1) * scale).bfloat16()


def main():
    R, D, F, H, B, N = 131072, 768, 2048, 12, 64, 2048
    A = bf(R, D); W = bf(3 * D, D, scale=0.05); C = torch.empty(R, 3 * D, device=DEV, dtype=torch.bfloat16)
    loop("gemm_nt qkv 131072x2304x768", lambda: ops.gemm_nt(A, W, C, R, 3 * D, D, L.EPI_BF16), flops=2.0 * R * 3 * D * D)
    A4 = bf(R, 2 * F); W4 = bf(D, 2 * F, scale=0.02); C4 = torch.empty(R, D, device=DEV, dtype=torch.bfloat16)
    loop("gemm_nt dgrad fc13 131072x768x4096", lambda: ops.gemm_nt(A4, W4, C4, R, D, 2 * F, L.EPI_BF16), flops=2.0 * R * D * 2 * F)
    G = torch.zeros(3 * D, D, device=DEV); slab = torch.empty(64 * 1024 * 1024 // 4, device=DEV)
    sp = ops.tn_splits(3 * D, D, R, slab.numel())
    loop("gemm_tn wgrad qkv 2304x768 over 131072 rows", lambda: ops.gemm_tn(C, A, G, 3 * D, D, R, splits=sp, slab=slab if sp > 1 else None),
         flops=2.0 * R * 3 * D * D)
    qkv = bf(B * N, 3 * D); o = torch.empty(B * N, D, device=DEV, dtype=torch.bfloat16); lse = torch.empty(B, H, N, device=DEV)
    zero = torch.zeros(B, dtype=torch.int32, device=DEV); full = torch.full((B,), N, dtype=torch.int32, device=DEV)
    p0 = qkv.data_ptr()

    def fwd():
        ops.attn_fwd(p0, N * 3 * D, 3 * D, p0 + 2 * D, N * 3 * D, 3 * D, p0 + 4 * D, N * 3 * D, 3 * D, o.data_ptr(), N * D, D, lse, zero, full, 1, 0,
                     B, H, N, N, 0.125)
    loop("attn_fwd 64 x 12 x 2048 x 2048", fwd, flops=4.0 * 64 * H * B * N * N)
    do = bf(B * N, D); dqkv = torch.empty(B * N, 3 * D, device=DEV, dtype=torch.bfloat16); delta = torch.empty(B, H, N, device=DEV)
    fwd()
    d0 = dqkv.data_ptr()

    def bwd():
        ops.attn_bwd(p0, N * 3 * D, 3 * D, p0 + 2 * D, N * 3 * D, 3 * D, p0 + 4 * D, N * 3 * D, 3 * D, o.data_ptr(), N * D, D, do.data_ptr(), N * D, D,
                     lse, delta, d0, N * 3 * D, 3 * D, d0 + 2 * D, N * 3 * D, 3 * D, d0 + 4 * D, N * 3 * D, 3 * D, zero, full, 1, 0, B, H, N, N, 0.125)
    loop("attn_bwd (dQ + dK/dV) 64 x 12 x 2048 x 2048", bwd, flops=10.0 * 64 * H * B * N * N)
    x = torch.randn(R, D, device=DEV); w = torch.ones(D, device=DEV); y = torch.empty(R, D, device=DEV, dtype=torch.bfloat16)
    mean, rstd = torch.empty(R, device=DEV), torch.empty(R, device=DEV)
    loop("layernorm_fwd 131072 x 768", lambda: ops.layernorm_fwd(x, w, y, mean, rstd), nbytes=R * D * 6.0)
    dx = torch.zeros(R, D, device=DEV); dxb = torch.empty(R, D, device=DEV, dtype=torch.bfloat16); dw = torch.zeros(D, device=DEV)
    loop("layernorm_bwd 131072 x 768", lambda: ops.layernorm_bwd(y, x, mean, rstd, w, dx, dw, dx_in=dx, dx_bf16=dxb), nbytes=R * D * 16.0)


if __name__ == "__main__":
    main()
