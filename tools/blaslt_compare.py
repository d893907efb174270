"""How far are the hand-written NT / TN GEMMs from the vendor library on the training shapes?  (round 4 probe)

torch.nn.functional.linear on bf16 operands (hipBLASLt / rocBLAS behind torch) against ego_gemm_nt_bf16 / ego_gemm_tn on the
same operands, plain bf16 epilogue, M = 131072 rows (micro-batch 64).  The library is NOT on the product path; this only
says what a tuned plain GEMM reaches on this chip at these shapes.   python tools/blaslt_compare.py
"""
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402
from egom2p_amd import _lib as L, ops  # noqa: E402


def timeit(fn, iters=10, rounds=5):
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
    M = int(os.environ.get("M", 131072))
    torch.manual_seed(0)
    shapes = [("qkv", 2304, 768), ("kv", 1536, 768), ("proj/q", 768, 768), ("fc13", 4096, 768), ("fc2", 768, 2048),
              ("dgrad qkv", 768, 2304), ("dgrad fc13", 768, 4096), ("dgrad fc2", 2048, 768), ("logits", 64000, 768), ("dgrad logits", 768, 64000)]
    only = [x.strip() for x in os.environ.get("SHAPES", "").split(",") if x.strip()]      # e.g. SHAPES="fc2,dgrad qkv,dgrad fc13"
    lib_only = os.environ.get("LIB_ONLY") == "1"          # under rocprofv3 --kernel-trace: the library's kernels only (tools/blaslt_names.sh)
    for name, N, K in shapes:
        if only and name not in only:
            continue
        m = M if "logits" not in name else 64576
        x = torch.randn(m, K, device=dev).bfloat16()
        w = torch.randn(N, K, device=dev).bfloat16()
        y = torch.empty(m, N, device=dev, dtype=torch.bfloat16)
        t_lib = timeit(lambda: F.linear(x, w))
        t_own = t_lib if lib_only else timeit(lambda: ops.gemm_nt(x, w, y, m, N, K, L.EPI_BF16))
        fl = 2.0 * m * N * K
        print(f"NT {name:13s} M={m} N={N} K={K}: library {t_lib:8.1f} us = {fl / t_lib / 1e6:7.1f} TF/s   own {t_own:8.1f} us = {fl / t_own / 1e6:7.1f} TF/s", flush=True)
        del x, w, y
    if only or lib_only:
        return
    # weight gradients: dW[N, K] = dY[M, N]^T X[M, K]
    for name, N, K in [("wgrad qkv", 2304, 768), ("wgrad fc13", 4096, 768), ("wgrad fc2", 768, 2048), ("wgrad proj", 768, 768)]:
        dy = torch.randn(M, N, device=dev).bfloat16()
        x = torch.randn(M, K, device=dev).bfloat16()
        g = torch.zeros(N, K, device=dev)
        t_lib = timeit(lambda: torch.matmul(dy.t(), x))
        splits = ops.tn_splits(N, K, M, 1 << 26, ranged=False, ldp=N, ldq=K)
        slab = torch.empty(1 << 26, device=dev) if splits > 1 else None
        t_own = timeit(lambda: ops.gemm_tn(dy, x, g, N, K, M, splits=splits, slab=slab, ldp=N, ldq=K, ldc=K))
        fl = 2.0 * M * N * K
        print(f"TN {name:13s} M={M} N={N} K={K}: library {t_lib:8.1f} us = {fl / t_lib / 1e6:7.1f} TF/s   own {t_own:8.1f} us = {fl / t_own / 1e6:7.1f} TF/s (fp32 accumulate into G)", flush=True)


if __name__ == "__main__":
    main()
