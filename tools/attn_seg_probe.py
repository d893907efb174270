"""Where does the block-diagonal decoder self-attention lose its time?  (round 4 probe)

The canonical decoder mask (per-row intervals 1009 / 1009 / 15 / 15) carries 48.5 % of the dense score work but its
launches take 70-75 % of the dense launches' time.  This probe times, in one process and interleaved:

    enc      one interval [0, 2048) per sample, B samples of 2048 rows            (dense, the uniform fast path)
    dec      per-row intervals 1009 / 1009 / 15 / 15                               (the product's decoder launch)
    dec1024  per-row intervals 1024 / 1024                                         (per-row path, tile-aligned segments)
    seg1024  one interval per sample, 2B samples of 1024 rows                      (= dec1024's work on the uniform path)
    seg1009  one interval per sample, 2B samples of 1009 rows                      (= dec's big segments on the uniform path)

    B=64 python tools/attn_seg_probe.py
"""
import ctypes as C
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from egom2p_amd import _lib as L  # noqa: E402


def main():
    lib = L.load()
    dev = "cuda"
    B, H = int(os.environ.get("B", 64)), 12
    rounds, iters = int(os.environ.get("ROUNDS", 5)), int(os.environ.get("ITERS", 4))
    D = H * 64
    st = torch.cuda.current_stream().cuda_stream
    torch.manual_seed(0)

    def case(name):
        if name in ("enc", "dec", "dec1024"):
            Bc, N = B, 2048
        else:
            Bc, N = 2 * B, (1024 if name == "seg1024" else 1009)
        qkv = torch.randn(Bc, N, 3, D, device=dev).bfloat16()
        do = torch.randn(Bc, N, D, device=dev).bfloat16()
        o = torch.empty(Bc, N, D, device=dev, dtype=torch.bfloat16)
        dqkv = torch.empty_like(qkv)
        lse, delta = torch.empty(Bc, H, N, device=dev), torch.empty(Bc, H, N, device=dev)
        if name in ("dec", "dec1024"):
            ks = torch.zeros(Bc, N, dtype=torch.int32, device=dev)
            ke = torch.zeros(Bc, N, dtype=torch.int32, device=dev)
            cuts = [0, 1009, 2018, 2033, 2048] if name == "dec" else [0, 1024, 2048]
            pairs = 0.0
            for a, b_ in zip(cuts[:-1], cuts[1:]):
                ks[:, a:b_] = a; ke[:, a:b_] = b_; pairs += float(Bc) * (b_ - a) ** 2
            r = (ks, ke, N, 1)
        else:
            r = (torch.zeros(Bc, dtype=torch.int32, device=dev), torch.full((Bc,), N, dtype=torch.int32, device=dev), 1, 0)
            pairs = float(Bc) * N * N
        p3, g = qkv.data_ptr(), dqkv.data_ptr()
        fa = (p3, N * 3 * D, 3 * D, p3 + 2 * D, N * 3 * D, 3 * D, p3 + 4 * D, N * 3 * D, 3 * D, o.data_ptr(), N * D, D, None,
              lse.data_ptr(), r[0].data_ptr(), r[1].data_ptr(), r[2], r[3], Bc, H, N, N, 0.125, st)
        ba = (p3, N * 3 * D, 3 * D, p3 + 2 * D, N * 3 * D, 3 * D, p3 + 4 * D, N * 3 * D, 3 * D, o.data_ptr(), N * D, D, None,
              do.data_ptr(), N * D, D, lse.data_ptr(), delta.data_ptr(), g, N * 3 * D, 3 * D, g + 2 * D, N * 3 * D, 3 * D,
              g + 4 * D, N * 3 * D, 3 * D, r[0].data_ptr(), r[1].data_ptr(), r[2], r[3], Bc, H, N, N, 0.125, st)
        keep = (qkv, do, o, dqkv, lse, delta, r)
        return fa, ba, pairs, keep

    names = os.environ.get("CASES", "enc,dec,dec1024,seg1024,seg1009").split(",")
    cases = {n: case(n) for n in names}
    times = {n: {"fwd": [], "bwd": []} for n in names}
    for _ in range(rounds):
        for n in names:
            fa, ba, pairs, _k = cases[n]
            for kind, fn, args in (("fwd", lib.ego_attn_fwd_d64, fa), ("bwd", lib.ego_attn_bwd_d64, ba)):
                assert fn(*args) == 0
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                for _ in range(iters):
                    fn(*args)
                e.record()
                torch.cuda.synchronize()
                times[n][kind].append(s.elapsed_time(e) / iters * 1e3)
    for n in names:
        pairs = cases[n][2]
        f, b = statistics.median(times[n]["fwd"]), statistics.median(times[n]["bwd"])
        print(n, json.dumps({"fwd_us": round(f, 1), "bwd_us": round(b, 1), "fwd_tflops": round(4 * 64 * H * pairs / f / 1e6, 1),
                             "bwd_tflops": round(10 * 64 * H * pairs / b / 1e6, 1), "score_pairs_G": round(pairs / 1e9, 3)}), flush=True)


if __name__ == "__main__":
    main()
