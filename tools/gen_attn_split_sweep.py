"""Forward attention on the generation path's shapes (config 4): time of the unsplit launch against ego_attn_fwd_d64_split with
2 ... 8 key runs per query tile (+ the combine kernel), for the encoder self-attention of the conditional passes (N = 5120 / 6827 /
8534 rows, 12 heads), the unconditional ones (1707 / 3414) and the decoder's cross-attention (1707 queries over up to 8534 keys).
    python tools/gen_attn_split_sweep.py"""
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from egom2p_amd import ops  # noqa: E402


def timeit(fn, iters=10, rounds=5):
    for _ in range(2):
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
    dev, H, D = "cuda", 12, 768
    torch.manual_seed(0)
    cases = [(n, n) for n in (1707, 3414, 5120, 6827, 8534)] + [(1707, k) for k in (3414, 5120, 6827, 8534)]
    for Nq, Nk in cases:
        q = (torch.randn(1, Nq, D, device=dev)).bfloat16()
        kv = (torch.randn(1, Nk, 2, D, device=dev)).bfloat16()
        o = torch.empty(1, Nq, D, device=dev, dtype=torch.bfloat16)
        lse = torch.empty(1, H, Nq, device=dev)
        ks = torch.zeros(1, dtype=torch.int32, device=dev)
        ke = torch.full((1,), Nk, dtype=torch.int32, device=dev)
        kp, vp = kv.data_ptr(), kv.data_ptr() + D * 2
        base = H * ((Nq + 127) // 128)
        t1 = timeit(lambda: ops.attn_fwd(q.data_ptr(), Nq * D, D, kp, Nk * 2 * D, 2 * D, vp, Nk * 2 * D, 2 * D, o.data_ptr(), Nq * D, D, lse, ks, ke, 1, 0,
                                         1, H, Nq, Nk, 0.125))
        line = f"Nq {Nq:5d} Nk {Nk:5d} workgroups {base:4d} | unsplit {t1:7.1f} us = {4.0 * 64 * H * Nq * Nk / t1 / 1e6:6.1f} TF/s |"
        for s in (2, 3, 4, 5, 6, 8):
            if (Nk + 63) // 64 < 2 * s:
                continue
            ws = torch.empty(ops.attn_fwd_split_floats(1, H, Nq, s), device=dev)
            t = timeit(lambda: ops.attn_fwd_split(q.data_ptr(), Nq * D, D, kp, Nk * 2 * D, 2 * D, vp, Nk * 2 * D, 2 * D, o.data_ptr(), Nq * D, D, lse, ks, ke, 1, 0,
                                                  1, H, Nq, Nk, 0.125, s, ws))
            line += f" s{s} {t:7.1f}"
            del ws
        print(line, flush=True)


if __name__ == "__main__":
    main()
