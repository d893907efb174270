"""Can weight-gradient GEMMs use the matrix pipe the attention backward leaves idle?  (round 4)

The dK / dV kernel keeps the MFMA pipe 44-46 % busy; the wgrad GEMMs of the same layer do not depend on it.  Round 1 ran them on
a side stream beside the dgrad / attention chain: 5 % SLOWER - two full-LDS kernels cannot share a CU, they just take turns.
This probe builds the co-resident case properly: a dK / dV launch padded with dynamic LDS so that only ONE of its workgroups
fits a CU (86 KB), beside the 128x128 wgrad kernel (64 KB of LDS) on a second stream, and compares wall times:

    seq        dK/dV (product, 2 workgroups per CU), then the layer's four wgrads on the 256x256 kernel, one stream
    naive      the same two on two streams (round 1's arrangement)
    cores      dK/dV at ONE workgroup per CU on stream A  ||  the wgrads on the 128x128 kernel on stream B
    alone      each side of `cores` by itself

    bash tools/abl_attn.sh build "dkvonly:-DATT_ONLY=2" "dkv1:-DATT_ONLY=2 -DDKV_PAD_LDS=12288"
    B=64 python tools/coresident_probe.py
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
from egom2p_amd import ops  # noqa: E402


def load(path):
    lib = C.CDLL(path)
    fn = lib.ego_attn_bwd_d64
    fn.argtypes = L._SIGS["ego_attn_bwd_d64"]
    fn.restype = C.c_int
    return lib


def main():
    L.load()
    v_only = load(os.path.join(ROOT, "variants", "libego_dkvonly.so"))
    v_one = load(os.path.join(ROOT, "variants", "libego_dkv1.so"))
    dev = "cuda"
    B, H, N = int(os.environ.get("B", 64)), 12, 2048
    D, F = H * 64, 2048
    R = B * N
    rounds = int(os.environ.get("ROUNDS", 7))
    torch.manual_seed(0)
    qkv = torch.randn(B, N, 3, D, device=dev).bfloat16()
    do = torch.randn(B, N, D, device=dev).bfloat16()
    o = torch.randn(B, N, D, device=dev).bfloat16()
    lse, delta = torch.randn(B, H, N, device=dev), torch.randn(B, H, N, device=dev)
    dqkv = torch.empty_like(qkv)
    zero_b = torch.zeros(B, dtype=torch.int32, device=dev)
    nval = torch.full((B,), N, dtype=torch.int32, device=dev)
    p3, g = qkv.data_ptr(), dqkv.data_ptr()
    sA, sB = torch.cuda.Stream(), torch.cuda.Stream()

    def bwd_args(stream):
        return (p3, N * 3 * D, 3 * D, p3 + 2 * D, N * 3 * D, 3 * D, p3 + 4 * D, N * 3 * D, 3 * D, o.data_ptr(), N * D, D, None,
                do.data_ptr(), N * D, D, lse.data_ptr(), delta.data_ptr(), g, N * 3 * D, 3 * D, g + 2 * D, N * 3 * D, 3 * D,
                g + 4 * D, N * 3 * D, 3 * D, zero_b.data_ptr(), nval.data_ptr(), 1, 0, B, H, N, N, 0.125, stream.cuda_stream)

    # the four wgrads of an encoder layer at micro-batch B: dY^T X into fp32 [Ni, Nj]
    shapes = [(D, F), (2 * F, D), (D, D), (3 * D, D)]              # fc2, fc1||fc3, proj, qkv
    ops_ = []
    slab = torch.empty(64 * 1024 * 1024 // 4, device=dev)
    for Ni, Nj in shapes:
        P = torch.randn(R, Ni, device=dev).bfloat16()
        Q = torch.randn(R, Nj, device=dev).bfloat16()
        G = torch.zeros(Ni, Nj, device=dev)
        ops_.append((P, Q, G, Ni, Nj))
    flop_w = sum(2.0 * R * Ni * Nj for _, _, _, Ni, Nj in ops_)

    def wgrads(stream, tn256):
        ops.gemm_kernel_mode(1, 1 if tn256 else 0)
        with torch.cuda.stream(stream):
            for P, Q, G, Ni, Nj in ops_:
                sp = ops.tn_splits(Ni, Nj, R, slab.numel())
                ops.gemm_tn(P, Q, G, Ni, Nj, R, splits=sp, slab=slab if sp > 1 else None)

    def timed(fn):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(torch.cuda.current_stream())
        sA.wait_event(e0); sB.wait_event(e0)
        fn()
        torch.cuda.current_stream().wait_stream(sA); torch.cuda.current_stream().wait_stream(sB)
        e1.record(torch.cuda.current_stream())
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3

    cases = {
        "seq": lambda: (v_only.ego_attn_bwd_d64(*bwd_args(sA)), wgrads(sA, True)),
        "naive": lambda: (v_only.ego_attn_bwd_d64(*bwd_args(sA)), wgrads(sB, True)),
        "cores": lambda: (v_one.ego_attn_bwd_d64(*bwd_args(sA)), wgrads(sB, False)),
        "cores_tn256": lambda: (v_one.ego_attn_bwd_d64(*bwd_args(sA)), wgrads(sB, True)),
        "alone_dkv_2wg": lambda: v_only.ego_attn_bwd_d64(*bwd_args(sA)),
        "alone_dkv_1wg": lambda: v_one.ego_attn_bwd_d64(*bwd_args(sA)),
        "alone_wgrad_tn256": lambda: wgrads(sB, True),
        "alone_wgrad_tn128": lambda: wgrads(sB, False),
    }
    times = {k: [] for k in cases}
    for _ in range(rounds):
        for k, fn in cases.items():
            fn()
            times[k].append(timed(fn))
    ops.gemm_kernel_mode(1, 1)
    out = {k: round(statistics.median(v), 1) for k, v in times.items()}
    out["wgrad_tflop"] = round(flop_w / 1e12, 3)
    print(json.dumps(out))
    print(f"seq {out['seq']} us; naive two streams {out['naive']} ({out['naive'] / out['seq'] - 1:+.1%}); co-resident 1 WG/CU + tn128 "
          f"{out['cores']} ({out['cores'] / out['seq'] - 1:+.1%}); 1 WG/CU + tn256 {out['cores_tn256']} ({out['cores_tn256'] / out['seq'] - 1:+.1%})")


if __name__ == "__main__":
    main()
