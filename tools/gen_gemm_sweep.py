"""Which NT kernel family is fastest on the generation path's row counts?  (config 4)

The rgb -> depth schedule runs its encoder linears on 5120 ... 11948 rows (conditional + unconditional groups of a guided step in
one launch) and its decoder linears on 1707 / 3414 rows.  ego_gemm_nt_bf16 picks a kernel by tile counts that were tuned on the
training shapes (131072 rows).  For every (rows, N, K, epilogue) of the schedule this times the three families - 64 x 64 tiles
(one tile per workgroup, 4-deep ring), 128 x 128 persistent (2 workgroups per CU), 256 x 256 persistent (1 per CU) - by forcing
each through the tuning hooks, and prints the winner beside what the selector picks today.     python tools/gen_gemm_sweep.py
"""
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from egom2p_amd import _lib as L, ops  # noqa: E402


def timeit(fn, iters=20, rounds=5):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / iters * 1e3)
    return statistics.median(ts)


def main():
    dev = "cuda"
    lib = L.load()
    torch.manual_seed(0)
    D, A, F = 768, 768, 2048
    rows_list = [int(x) for x in os.environ.get("ROWS", "1707,3414,5120,6827,8534,10241,11948").split(",")]
    shapes = [("qkv", 3 * A, D, L.EPI_BF16), ("proj+res", D, A, L.EPI_RESID), ("fc2+res", D, F, L.EPI_RESID), ("fc13", 2 * F, D, L.EPI_BF16),
              ("q", A, D, L.EPI_BF16), ("kv", 2 * A, D, L.EPI_BF16), ("logits", 64000, D, L.EPI_BF16)]
    # (ego_gemm_kernel_mode nt256, ego_gemm_small_tiles, ego_gemm_tune key 2): 128x64 / 128x128 = the round-5 low-latency instantiations
    modes = {"64": (0, 1 << 30, 0), "128": (0, 0, 0), "128x64": (0, 0, 1), "128x128": (0, 0, 2), "256": (2, 0, 0), "auto": (1, 400, 0)}
    print(f"{'shape':10s} {'rows':>6s} | " + " ".join(f"{m:>9s}" for m in modes) + " | best")
    for name, N, K, epi in shapes:
        for M in (rows_list if name != "logits" else [1707, 3414]):
            x = (torch.randn(M, K, device=dev) * 0.5).bfloat16()
            w = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
            r = torch.randn(M, N, device=dev) if epi == L.EPI_RESID else None
            y = torch.empty(M, N, device=dev, dtype=torch.float32 if epi == L.EPI_RESID else torch.bfloat16)
            res = {}
            for m, (nt256, small, force) in modes.items():
                lib.ego_gemm_kernel_mode(nt256, 1)
                lib.ego_gemm_small_tiles(small)
                lib.ego_gemm_tune(2, force)
                res[m] = timeit(lambda: ops.gemm_nt(x, w, y, M, N, K, epi, R=r))
            lib.ego_gemm_kernel_mode(1, 1)
            lib.ego_gemm_small_tiles(400)
            lib.ego_gemm_tune(2, 0)
            best = min((m for m in modes if m != "auto"), key=lambda m: res[m])
            print(f"{name:10s} {M:6d} | " + " ".join(f"{res[m]:9.1f}" for m in modes) + f" | {best} ({100 * (res['auto'] / res[best] - 1):+.0f} % vs auto)", flush=True)
            del x, w, y, r
    # fc1||fc3 + gate: the fused 256 x 256 launch (gemm_nt256_kernel<3>) against the plain GEMM (selector's choice) + ego_swiglu_fwd
    print("fc13 + gate: fused launch vs GEMM + swiglu pass (us)")
    for M in rows_list:
        x = (torch.randn(M, D, device=dev) * 0.5).bfloat16()
        w = (torch.randn(2 * F, D, device=dev) * 0.05).bfloat16()
        ab = torch.empty(M, 2 * F, device=dev, dtype=torch.bfloat16)
        h = torch.empty(M, F, device=dev, dtype=torch.bfloat16)
        t_f = timeit(lambda: ops.gemm_nt_swiglu_fwd(x, w, ab, h, M, F, D))

        def two():
            ops.gemm_nt(x, w, ab, M, 2 * F, D, L.EPI_BF16)
            ops.swiglu_fwd(ab, h, M, F)
        t_2 = timeit(two)
        print(f"gate       {M:6d} | fused {t_f:8.1f}   two launches {t_2:8.1f}", flush=True)


if __name__ == "__main__":
    main()
