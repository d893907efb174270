"""Phase probe of the dK / dV kernel: two workgroups share a CU (one wave of each per SIMD) and nothing holds
them in opposite phases of their MFMA / vector-work alternation.  `ego_attn_tune(1, d)` starts the workgroups that are dispatched SECOND on
their CU d x 512 clocks late (one 16-MFMA block of a wave takes ~2200 clocks at two waves per SIMD).  Settings interleaved inside every round;
us per dQ + dK/dV pair at micro-batch B (encoder: one interval per sample; decoder: the block-diagonal mask by row groups is not used here -
per-row intervals, whose workgroups are not equally long).      B=64 python tools/attn_stagger_probe.py"""
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from egom2p_amd import _lib as L, ops  # noqa: E402


def main():
    dev = "cuda"
    lib = L.load()
    B, H, N = int(os.environ.get("B", 64)), 12, 2048
    D = H * 64
    torch.manual_seed(0)
    qkv = torch.randn(B, N, 3, D, device=dev).bfloat16()
    o = torch.empty(B, N, D, device=dev, dtype=torch.bfloat16)
    do = torch.randn(B, N, D, device=dev).bfloat16()
    dqkv = torch.empty_like(qkv)
    lse = torch.empty(B, H, N, device=dev)
    delta = torch.empty(B, H, N, device=dev)
    zero = torch.zeros(B, dtype=torch.int32, device=dev)
    full = torch.full((B,), N, dtype=torch.int32, device=dev)
    p, g = qkv.data_ptr(), dqkv.data_ptr()
    ops.attn_fwd(p, N * 3 * D, 3 * D, p + 2 * D, N * 3 * D, 3 * D, p + 4 * D, N * 3 * D, 3 * D, o.data_ptr(), N * D, D, lse, zero, full, 1, 0, B, H, N, N, 0.125)
    bwd = lambda: ops.attn_bwd(p, N * 3 * D, 3 * D, p + 2 * D, N * 3 * D, 3 * D, p + 4 * D, N * 3 * D, 3 * D, o.data_ptr(), N * D, D,
                               do.data_ptr(), N * D, D, lse, delta, g, N * 3 * D, 3 * D, g + 2 * D, N * 3 * D, 3 * D, g + 4 * D, N * 3 * D, 3 * D,
                               zero, full, 1, 0, B, H, N, N, 0.125)
    delays = [0, 1, 2, 3, 4, 6, 9]
    ts = {d: [] for d in delays}
    ref = None
    for r in range(9):
        for d in delays:
            lib.ego_attn_tune(1, d)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                bwd()
            e1.record()
            torch.cuda.synchronize()
            if r:
                ts[d].append(e0.elapsed_time(e1) / 5 * 1e3)
            if ref is None:
                ref = dqkv.clone()
            else:
                assert torch.equal(ref, dqkv), "the probe changed a result"
    lib.ego_attn_tune(1, 0)
    t0 = statistics.median(ts[0])
    print(f"B {B}: dQ + dK/dV pair, us per launch; start delay of the second workgroup of a CU in units of 512 clocks")
    for d in delays:
        t = statistics.median(ts[d])
        print(f"  delay {d:2d}: {t:8.1f} us  ({100 * (t0 / t - 1):+5.2f} %)")


if __name__ == "__main__":
    main()
