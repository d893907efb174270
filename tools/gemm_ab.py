"""In-process A/B of GEMM builds on the training step's NT shapes at micro-batch 64 (rule 24: variants x rounds interleaved
in one process).  Loads the product library and every variants/libego_*.so; prints TF/s (median over rounds) per shape.

    ROUNDS=5 python tools/gemm_ab.py [name ...]
    PMC=1 python tools/gemm_ab.py        # one launch per shape and library, in a fixed order (for rocprofv3 --pmc runs)
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

ENTRIES = ("ego_gemm_nt_bf16", "ego_gemm_nt_swiglu_fwd", "ego_gemm_nt_swiglu_bwd")


def load(path):
    hip_rt = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
    if os.path.exists(hip_rt):
        C.CDLL(hip_rt, mode=C.RTLD_GLOBAL)
    lib = C.CDLL(path)
    for name in ENTRIES:
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
    R = int(os.environ.get("ROWS", 131072))
    rounds, iters = int(os.environ.get("ROUNDS", 5)), int(os.environ.get("ITERS", 4))
    pmc = os.environ.get("PMC") == "1"
    st = torch.cuda.current_stream().cuda_stream
    D, F = 768, 2048

    def rnd(*shape, scale=1.0):
        return ((torch.rand(*shape, device=dev) * 2 - 1) * scale).bfloat16()

    cases = []      # (name, flops, fn(lib))
    A768 = rnd(R, D)
    res32 = torch.randn(R, D, device=dev)
    out32 = torch.empty(R, D, device=dev)

    def nt(name, A, N, K, epi=0, Rr=None, Cout=None):
        B = rnd(N, K, scale=0.05)
        Cc = Cout if Cout is not None else torch.empty(A.shape[0], N, device=dev, dtype=torch.bfloat16)
        M = A.shape[0]
        args = (A.data_ptr(), A.stride(0), B.data_ptr(), K, Cc.data_ptr(), Cc.stride(0), None if Rr is None else Rr.data_ptr(),
                0 if Rr is None else Rr.stride(0), None, None, M, N, K, epi, st)
        cases.append((name, 2.0 * M * N * K, lambda lib, a=args, keep=(A, B, Cc, Rr): lib.ego_gemm_nt_bf16(*a)))

    nt("qkv 2304x768", A768, 3 * D, D)
    nt("proj+resid 768x768", A768, D, D, epi=2, Rr=res32, Cout=out32)
    nt("kv 1536x768", A768, 2 * D, D)
    Ah = rnd(R, F)
    nt("fc2+resid 768x2048", Ah, D, F, epi=2, Rr=res32, Cout=out32)
    Aq = rnd(R, 3 * D)
    nt("dgrad qkv 768x2304", Aq, D, 3 * D)
    Aab = rnd(R, 2 * F)
    nt("dgrad fc13 768x4096", Aab, D, 2 * F)
    nt("dgrad proj 768x768", A768, D, D)
    Ay = rnd(64576, D)
    nt("logits 64000x768", Ay, 64000, D)
    # fused gate forward and backward
    W13 = rnd(2 * F, D, scale=0.05)
    ab = torch.empty(R, 2 * F, device=dev, dtype=torch.bfloat16)
    h = torch.empty(R, F, device=dev, dtype=torch.bfloat16)
    a_f = (A768.data_ptr(), D, W13.data_ptr(), D, ab.data_ptr(), 2 * F, h.data_ptr(), F, R, F, D, st)
    cases.append(("fc13+gate 2x2048x768", 2.0 * R * 2 * F * D, lambda lib: lib.ego_gemm_nt_swiglu_fwd(*a_f)))
    W2t = rnd(F, D, scale=0.05)
    dab = torch.empty(R, 2 * F, device=dev, dtype=torch.bfloat16)
    a_b = (A768.data_ptr(), D, W2t.data_ptr(), D, ab.data_ptr(), dab.data_ptr(), 2 * F, R, F, D, st)
    cases.append(("fc2 dgrad+gate bwd 2048x768", 2.0 * R * F * D, lambda lib: lib.ego_gemm_nt_swiglu_bwd(*a_b)))

    out = {}
    for name, flops, fn in cases:
        if pmc:
            for n, lib in libs.items():
                assert fn(lib) == 0, (name, n)
            torch.cuda.synchronize()
            continue
        times = {n: [] for n in libs}
        for _ in range(rounds):
            for n, lib in libs.items():
                assert fn(lib) == 0, (name, n)
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                for _ in range(iters):
                    fn(lib)
                e.record()
                torch.cuda.synchronize()
                times[n].append(s.elapsed_time(e) / iters * 1e-3)
        out[name] = {n: round(flops / statistics.median(t) / 1e12, 1) for n, t in times.items()}
        print(name, json.dumps(out[name]), flush=True)
    if pmc:
        print("order:", [c[0] for c in cases], "x", list(libs))


if __name__ == "__main__":
    main()
