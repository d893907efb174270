"""In-process A/B of attention-kernel builds (cdna_hip_programming.md section 5.4 rule 24: variants x rounds interleaved in ONE
process, median and min reported).  Every `variants/libego_*.so` (built by tools/abl_attn.sh) plus the product library is
loaded side by side through ctypes; per case the backward (and forward) of every build runs round-robin, results are checked
against the product build (max |diff| of dQ / dK / dV), times are per launch pair (dQ kernel + dK/dV kernel).

    B=32 ROUNDS=7 python tools/attn_ab.py [name ...]        # names = variants/libego_<name>.so; default: all

Cases = the three attention sites of a training step at the bench shapes (H = 12, d = 64, N = M = 2048):
    enc    encoder self-attention: one interval [0, n_valid) per sample (uniform walk)
    dec    decoder self-attention: block-diagonal per-row intervals 1009 / 1009 / 15 / 15 (the plain entry point: query tiles straddle groups)
    decseg the same mask through the row-group entry points, as the engine launches it (tiles never straddle a group)
    cross  cross-attention: decoder queries x encoder keys, one interval per sample, separate q and kv buffers, O residual
"""
import ctypes as C
import glob
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from egom2p_amd import _lib as L  # noqa: E402


def load(path):
    hip_rt = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
    if os.path.exists(hip_rt):
        C.CDLL(hip_rt, mode=C.RTLD_GLOBAL)
    lib = C.CDLL(path)
    for name in ("ego_attn_fwd_d64", "ego_attn_bwd_d64", "ego_attn_fwd_d64_seg", "ego_attn_bwd_d64_seg"):
        fn = getattr(lib, name)
        fn.argtypes = L._SIGS[name]
        fn.restype = C.c_int
    return lib


def main():
    names = sys.argv[1:]
    libs = {"product": load(L.LIB_PATH)}
    for p in sorted(glob.glob(os.path.join(ROOT, "variants", "libego_*.so"))):
        n = os.path.basename(p)[len("libego_"):-3]
        if not names or n in names:
            libs[n] = load(p)
    dev = "cuda"
    B, H, N = int(os.environ.get("B", 32)), 12, 2048
    rounds, iters = int(os.environ.get("ROUNDS", 7)), int(os.environ.get("ITERS", 5))
    D = H * 64
    torch.manual_seed(0)
    qkv = torch.randn(B, N, 3, D, device=dev).bfloat16()
    q = torch.randn(B, N, D, device=dev).bfloat16()
    kv = torch.randn(B, N, 2, D, device=dev).bfloat16()
    do = torch.randn(B, N, D, device=dev).bfloat16()
    o, olo = torch.empty(B, N, D, device=dev, dtype=torch.bfloat16), torch.empty(B, N, D, device=dev, dtype=torch.bfloat16)
    lse, delta = torch.empty(B, H, N, device=dev), torch.empty(B, H, N, device=dev)
    zero_b = torch.zeros(B, dtype=torch.int32, device=dev)
    nval = torch.full((B,), N, dtype=torch.int32, device=dev)
    ks = torch.zeros(B, N, dtype=torch.int32, device=dev)
    ke = torch.zeros(B, N, dtype=torch.int32, device=dev)
    pairs_dec = 0.0
    for a, b_ in zip([0, 1009, 2018, 2033], [1009, 2018, 2033, 2048]):
        ks[:, a:b_] = a; ke[:, a:b_] = b_; pairs_dec += float(B) * (b_ - a) ** 2
    st = torch.cuda.current_stream().cuda_stream
    p3, pq, pkv = qkv.data_ptr(), q.data_ptr(), kv.data_ptr()

    seg = torch.tensor([[0, 1009], [1009, 1009], [2018, 15], [2033, 15]], dtype=torch.int32, device=dev).repeat(B, 1, 1).contiguous()

    def mk(case):
        if case in ("enc", "dec", "decseg"):
            dqkv = torch.empty_like(qkv)
            g = dqkv.data_ptr()
            r = (zero_b, nval, 1, 0) if case == "enc" else (ks, ke, N, 1)
            fa = (p3, N * 3 * D, 3 * D, p3 + 2 * D, N * 3 * D, 3 * D, p3 + 4 * D, N * 3 * D, 3 * D, o.data_ptr(), N * D, D, None,
                  lse.data_ptr(), r[0].data_ptr(), r[1].data_ptr(), r[2], r[3], B, H, N, N, 0.125, st)
            ba = (p3, N * 3 * D, 3 * D, p3 + 2 * D, N * 3 * D, 3 * D, p3 + 4 * D, N * 3 * D, 3 * D, o.data_ptr(), N * D, D, None,
                  do.data_ptr(), N * D, D, lse.data_ptr(), delta.data_ptr(), g, N * 3 * D, 3 * D, g + 2 * D, N * 3 * D, 3 * D,
                  g + 4 * D, N * 3 * D, 3 * D, r[0].data_ptr(), r[1].data_ptr(), r[2], r[3], B, H, N, N, 0.125, st)
            if case == "decseg":        # (seg, n_seg, seg_bad) sit between the interval strides and B
                fa = fa[:18] + (seg.data_ptr(), 4, None) + fa[18:]
                ba = ba[:31] + (seg.data_ptr(), 4, None) + ba[31:]
            return fa, ba, [dqkv], (float(B) * N * N if case == "enc" else pairs_dec)
        dq, dkv = torch.empty_like(q), torch.empty_like(kv)
        fa = (pq, N * D, D, pkv, N * 2 * D, 2 * D, pkv + 2 * D, N * 2 * D, 2 * D, o.data_ptr(), N * D, D, olo.data_ptr(),
              lse.data_ptr(), zero_b.data_ptr(), nval.data_ptr(), 1, 0, B, H, N, N, 0.125, st)
        ba = (pq, N * D, D, pkv, N * 2 * D, 2 * D, pkv + 2 * D, N * 2 * D, 2 * D, o.data_ptr(), N * D, D, olo.data_ptr(),
              do.data_ptr(), N * D, D, lse.data_ptr(), delta.data_ptr(), dq.data_ptr(), N * D, D, dkv.data_ptr(), N * 2 * D, 2 * D,
              dkv.data_ptr() + 2 * D, N * 2 * D, 2 * D, zero_b.data_ptr(), nval.data_ptr(), 1, 0, B, H, N, N, 0.125, st)
        return fa, ba, [dq, dkv], float(B) * N * N

    out = {}
    for case in os.environ.get("CASES", "enc,dec,decseg,cross").split(","):
        fa, ba, outs, pairs = mk(case)
        fwd_name, bwd_name = ("ego_attn_fwd_d64_seg", "ego_attn_bwd_d64_seg") if case == "decseg" else ("ego_attn_fwd_d64", "ego_attn_bwd_d64")
        assert getattr(libs["product"], fwd_name)(*fa) == 0
        ref, errs = None, {}
        for n, lib in libs.items():
            for t in outs:
                t.zero_()
            assert getattr(lib, bwd_name)(*ba) == 0, n
            torch.cuda.synchronize()
            got = [t.float().clone() for t in outs]
            if ref is None:
                ref = got
            errs[n] = max(float((a - b).abs().max()) for a, b in zip(got, ref))
        times = {n: {"fwd": [], "bwd": []} for n in libs}
        for _ in range(rounds):
            for n, lib in libs.items():
                for kind, fn, args in (("fwd", getattr(lib, fwd_name), fa), ("bwd", getattr(lib, bwd_name), ba)):
                    fn(*args)
                    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    s.record()
                    for _ in range(iters):
                        fn(*args)
                    e.record()
                    torch.cuda.synchronize()
                    times[n][kind].append(s.elapsed_time(e) / iters * 1e3)
        out[case] = {n: {"bwd_us_med": round(statistics.median(t["bwd"]), 1), "bwd_us_min": round(min(t["bwd"]), 1),
                         "bwd_tflops": round(10 * 64 * H * pairs / statistics.median(t["bwd"]) / 1e6, 1),
                         "fwd_us_med": round(statistics.median(t["fwd"]), 1), "max_abs_diff_vs_product": errs[n]}
                     for n, t in times.items()}
        print(case, json.dumps(out[case]), flush=True)
    # one micro-batch of the step at this batch: 12 enc + 12 dec + 12 cross backward launches
    tot = {n: round((out["enc"][n]["bwd_us_med"] + out["decseg"][n]["bwd_us_med"] + out["cross"][n]["bwd_us_med"]) * 12 / 1e3, 2)
           for n in libs} if all(c in out for c in ("enc", "decseg", "cross")) else {}
    print("attn_bwd ms per micro-batch of", B, json.dumps(tot))


if __name__ == "__main__":
    main()
