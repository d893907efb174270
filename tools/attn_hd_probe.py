"""Timing of the padded-head attention kernels (ego_attn_*_hd) at the registered ego-L's shapes (round 4 probe).

    B=8 H=16 HD=68 HDP=96 python tools/attn_hd_probe.py          # micro-batch 8 of the registered ego-L (16 x 96 storage)

Cases: enc (one interval per sample, dense), dec (the decoder's per-row intervals 1009 / 1009 / 15 / 15: 48.5 % of the pairs),
per launch: forward, backward (delta + dQ + dK/dV), and TF/s on the EXECUTED (padded-head) MFMA work of the visited tiles.
"""
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from egom2p_amd import ops  # noqa: E402


def main():
    dev = "cuda"
    B, H = int(os.environ.get("B", 8)), int(os.environ.get("H", 16))
    hd, hdp = int(os.environ.get("HD", 68)), int(os.environ.get("HDP", 96))
    N = int(os.environ.get("N", 2048))
    rounds, iters = int(os.environ.get("ROUNDS", 5)), int(os.environ.get("ITERS", 6))
    A = H * hdp
    torch.manual_seed(0)

    def padded(*shape):
        t = torch.zeros(*shape, H, hdp, device=dev)
        t[..., :hd] = torch.randn(*shape, H, hd, device=dev)
        return t.bfloat16()

    qkv = padded(B, N, 3).view(B, N, 3 * A)
    do = padded(B, N).view(B, N, A)
    o = torch.empty(B, N, A, device=dev, dtype=torch.bfloat16)
    dqkv = torch.empty_like(qkv)
    lse, delta = torch.empty(B, H, N, device=dev), torch.empty(B, H, N, device=dev)
    scale = hd ** -0.5

    def intervals(name):
        if name == "enc":
            return torch.zeros(B, dtype=torch.int32, device=dev), torch.full((B,), N, dtype=torch.int32, device=dev), 1, 0, float(B) * N * N
        if name == "half":          # every row attends keys [0, N / 2): the tile range is cut, nothing is masked
            return torch.zeros(B, dtype=torch.int32, device=dev), torch.full((B,), N // 2, dtype=torch.int32, device=dev), 1, 0, float(B) * N * N / 2
        if name == "rows_full":     # per-row intervals that are all [0, N): the per-row path on dense work
            return (torch.zeros(B, N, dtype=torch.int32, device=dev), torch.full((B, N), N, dtype=torch.int32, device=dev), N, 1, float(B) * N * N)
        if name == "rows_masked":   # one row per wave attends key 0 only: every tile of every wave takes the masked path
            ke = torch.full((B, N), N, dtype=torch.int32, device=dev)
            ke[:, ::32] = 1
            return torch.zeros(B, N, dtype=torch.int32, device=dev), ke, N, 1, float(B) * N * N * 31 / 32
        if name == "rows_half":     # per-row intervals that are all [0, N / 2)
            return (torch.zeros(B, N, dtype=torch.int32, device=dev), torch.full((B, N), N // 2, dtype=torch.int32, device=dev), N, 1, float(B) * N * N / 2)
        ks = torch.zeros(B, N, dtype=torch.int32, device=dev)
        ke = torch.zeros(B, N, dtype=torch.int32, device=dev)
        pairs = 0.0
        cuts = {"dec2": [0, N // 2, N], "cut1040": [0, 1040, N], "cut1009": [0, 1009, N], "cut3": [0, 1009, 2018, N]}.get(name, [0, 1009, 2018, 2033, 2048])
        for a, b_ in zip(cuts[:-1], cuts[1:]):
            ks[:, a:b_] = a
            ke[:, a:b_] = b_
            pairs += float(B) * (b_ - a) ** 2
        return ks, ke, N, 1, pairs

    p3, g = qkv.data_ptr(), dqkv.data_ptr()
    res = {}
    for name in os.environ.get("CASES", "enc,dec").split(","):
        ks, ke, r_bs, r_rs, pairs = intervals(name)

        def fwd():
            ops.attn_fwd(p3, N * 3 * A, 3 * A, p3 + 2 * A, N * 3 * A, 3 * A, p3 + 4 * A, N * 3 * A, 3 * A, o.data_ptr(), N * A, A, lse, ks, ke,
                         r_bs, r_rs, B, H, N, N, scale, hd_pad=hdp)

        def bwd():
            ops.attn_bwd(p3, N * 3 * A, 3 * A, p3 + 2 * A, N * 3 * A, 3 * A, p3 + 4 * A, N * 3 * A, 3 * A, o.data_ptr(), N * A, A,
                         do.data_ptr(), N * A, A, lse, delta, g, N * 3 * A, 3 * A, g + 2 * A, N * 3 * A, 3 * A, g + 4 * A, N * 3 * A, 3 * A,
                         ks, ke, r_bs, r_rs, B, H, N, N, scale, hd_pad=hdp)

        out = {}
        for nm, fn, units in (("fwd", fwd, 2), ("bwd", bwd, 7)):
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
            us = statistics.median(ts)
            out[f"{nm}_us"] = round(us, 1)
            out[f"{nm}_tflops_padded"] = round(units * 2.0 * pairs * H * hdp / (us * 1e-6) / 1e12, 1)
        res[name] = out
        print(name, json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
