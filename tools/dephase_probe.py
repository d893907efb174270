"""De-phasing the persistent 256 x 256 NT GEMM's workgroups.

All 256 workgroups of a launch start together and run equal tiles, so their epilogue store bursts hit HBM at the same moments.
`ego_gemm_tune(1, d)` starts every other workgroup of an XCD d x ~0.5 us late (results unchanged).  For each training shape and
epilogue class this prints the launch time without and with start delays of about a quarter / half / three quarters of one tile's
time, and what the de-phasing is worth NET of the delay itself - the upper bound of what an idle-free form (unequal first tiles)
could recover is `gross` = the gain with the delay's own cost taken out.      python tools/dephase_probe.py
"""
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from egom2p_amd import _lib as L, ops  # noqa: E402


def time_interleaved(fn, lib, delays, iters=10, rounds=15):
    """median us per launch for each start delay in `delays` (0 = the product), the settings taken in turn inside every round so
    that clock / temperature drift hits them alike (a first, unrecorded round warms up)"""
    ts = {d: [] for d in delays}
    for r in range(rounds + 1):
        for d in delays:
            lib.ego_gemm_tune(1, d)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                fn()
            e1.record()
            torch.cuda.synchronize()
            if r:
                ts[d].append(e0.elapsed_time(e1) / iters * 1e3)
    lib.ego_gemm_tune(1, 0)
    return {d: statistics.median(v) for d, v in ts.items()}


def main():
    dev = "cuda"
    lib = L.load()
    M = int(os.environ.get("M", 131072))
    torch.manual_seed(0)
    bf = lambda *s: (torch.randn(*s, device=dev) * 0.5).bfloat16()
    cases = []
    # (name, epilogue class, N, K, callable factory)
    def plain(N, K, m=M):
        x, w, y = bf(m, K), bf(N, K), torch.empty(m, N, device=dev, dtype=torch.bfloat16)
        return lambda: ops.gemm_nt(x, w, y, m, N, K, L.EPI_BF16), 2.0 * m * N * K, (m + 255) // 256 * ((N + 255) // 256)
    def resid(N, K):
        x, w, r, y = bf(M, K), bf(N, K), torch.randn(M, N, device=dev), torch.empty(M, N, device=dev)
        return lambda: ops.gemm_nt(x, w, y, M, N, K, L.EPI_RESID, R=r), 2.0 * M * N * K, (M // 256) * (N // 256)
    def gate_fwd(F, K):
        x, w, ab, h = bf(M, K), bf(2 * F, K), torch.empty(M, 2 * F, device=dev, dtype=torch.bfloat16), torch.empty(M, F, device=dev, dtype=torch.bfloat16)
        return lambda: ops.gemm_nt_swiglu_fwd(x, w, ab, h, M, F, K), 4.0 * M * F * K, (M // 256) * (F // 128)
    def gate_bwd(F, K):
        dy, w2t, ab, dab = bf(M, K), bf(F, K), bf(M, 2 * F), torch.empty(M, 2 * F, device=dev, dtype=torch.bfloat16)
        return lambda: ops.gemm_nt_swiglu_bwd(dy, w2t, ab, dab, M, F, K), 2.0 * M * F * K, (M // 256) * (F // 256)
    plan = [("qkv <0>", lambda: plain(2304, 768)), ("dgrad fc13 <0>", lambda: plain(768, 4096)), ("proj + residual <1>", lambda: resid(768, 768)),
            ("fc2 + residual <1>", lambda: resid(768, 2048)), ("fc2 dgrad + SwiGLU bwd <2>", lambda: gate_bwd(2048, 768)),
            ("fc1||fc3 + gate <3>", lambda: gate_fwd(2048, 768)), ("logits <0, strips>", lambda: plain(64000, 768, 64576))]
    for name, make in plan:
        fn, flops, tiles = make()
        t0 = time_interleaved(fn, lib, [0], iters=5, rounds=2)[0]
        per_tile = t0 / max(1.0, tiles / 256.0)                      # us of one tile of one workgroup
        ds = sorted({max(1, round(f * per_tile / 0.5)) for f in (0.25, 0.5, 0.75)})
        res = time_interleaved(fn, lib, [0] + ds)
        t0 = res[0]
        line = f"{name:28s} tiles/CU {tiles / 256.0:6.1f}  tile {per_tile:6.1f} us  base {t0:8.1f} us = {flops / t0 / 1e6:7.1f} TF/s |"
        for d in ds:
            t, delay_us = res[d], d * 0.5
            line += f"  d={d:3d} ({delay_us:5.1f} us): {t:8.1f} us net {100 * (t0 / t - 1):+5.1f} % gross {100 * (t0 / max(t - delay_us, 1e-9) - 1):+5.1f} % |"
        print(line, flush=True)
        del fn
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
