"""fp8 (e4m3) forward GEMMs: the fragment map (round 4: VERDICT r3 item 5, "the e4m3 K-tile takes 1.7 x the bf16 K-tile").

In-process A/B on ego-L's forward linears (65,536 rows): the bf16 kernel, the fp8 kernel with the round-3 fragment map
(variants/libego_fp8old.so: -DFP8_FRAG_OLD=1, 2-way LDS bank conflict on every fragment read) and the product's map.

    SRC=gemm bash tools/abl_attn.sh build "fp8old:-DFP8_FRAG_OLD=1"
    python tools/fp8_frag_probe.py          # PMC=1: one launch each, for rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE ...
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


def main():
    prod = L.load()
    old = C.CDLL(os.path.join(ROOT, "variants", "libego_fp8old.so"))
    for name in ("ego_gemm_nt_fp8", "ego_gemm_nt_bf16"):
        fn = getattr(old, name)
        fn.argtypes = L._SIGS[name]
        fn.restype = C.c_int
    dev = "cuda"
    R = int(os.environ.get("ROWS", 65536))
    rounds, iters = int(os.environ.get("ROUNDS", 5)), int(os.environ.get("ITERS", 4))
    pmc = os.environ.get("PMC") == "1"
    st = torch.cuda.current_stream().cuda_stream
    torch.manual_seed(0)
    for N, K in ((3456, 1152), (1152, 1152), (1152, 3072), (6144, 1152), (2304, 768)):
        A = torch.randn(R, K, device=dev).bfloat16()
        B = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
        Cb = torch.empty(R, N, device=dev, dtype=torch.bfloat16)
        A8, B8 = torch.empty(R, K, device=dev, dtype=torch.uint8), torch.empty(N, K, device=dev, dtype=torch.uint8)
        sa, sb = torch.empty(R, device=dev), torch.empty(N, device=dev)
        ops.quant_fp8_rows(A, A8, sa)
        ops.quant_fp8_rows(B, B8, sb)
        C_new, C_old = torch.empty_like(Cb), torch.empty_like(Cb)
        a_bf = (A.data_ptr(), K, B.data_ptr(), K, Cb.data_ptr(), N, None, 0, None, None, R, N, K, 0, st)
        a_new = (A8.data_ptr(), K, sa.data_ptr(), B8.data_ptr(), K, sb.data_ptr(), C_new.data_ptr(), N, None, 0, None, R, N, K, 0, st)
        a_old = (A8.data_ptr(), K, sa.data_ptr(), B8.data_ptr(), K, sb.data_ptr(), C_old.data_ptr(), N, None, 0, None, R, N, K, 0, st)
        cases = {"bf16": lambda: prod.ego_gemm_nt_bf16(*a_bf), "fp8_r3_map": lambda: old.ego_gemm_nt_fp8(*a_old),
                 "fp8": lambda: prod.ego_gemm_nt_fp8(*a_new)}
        for fn in cases.values():
            assert fn() == 0
        torch.cuda.synchronize()
        if pmc:
            continue
        # the two maps contract the same products in another order: equal up to the fp32 sums inside the MFMA
        d = (C_new.float() - C_old.float()).abs().max().item() / C_old.float().abs().max().item()
        ref = (A8.view(torch.float8_e4m3fn).float() * sa[:, None])[:2048] @ (B8.view(torch.float8_e4m3fn).float() * sb[:, None]).t()
        e = ((C_new[:2048].float() - ref).norm() / ref.norm()).item()
        times = {k: [] for k in cases}
        for _ in range(rounds):
            for k, fn in cases.items():
                fn()
                s_, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s_.record()
                for _ in range(iters):
                    fn()
                e_.record()
                torch.cuda.synchronize()
                times[k].append(s_.elapsed_time(e_) / iters * 1e-3)
        flops = 2.0 * R * N * K
        print(f"{R}x{N}x{K}", json.dumps({k: round(flops / statistics.median(t) / 1e12, 1) for k, t in times.items()}),
              f"max |new - r3 map| / max = {d:.2e}; new vs dequantised fp32 reference rel {e:.2e}", flush=True)


if __name__ == "__main__":
    main()
