"""Summarise rocprofv3 --pmc counter CSVs per kernel class: HBM-side bytes per launch.

usage: python tools/pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json>

FETCH_SIZE / WRITE_SIZE are reported in KiB-like units of 1024 B... (rocprofv3 derives them as
TCC_EA0_RDREQ-based sums / 1024); on gfx950 FETCH_SIZE tallies each 128-B request of a wide streaming read
as 64 B, so it is doubled (MI355X_MICROARCH.md, section HBM).  WRITE_SIZE is exact for 16-B-per-lane stores.
"""
import csv
import json
import re
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egom2p_amd.profiler import kernel_source_sha  # noqa: E402

CLASSES = [("gemm_nt256", r"gemm_nt256_kernel"), ("gemm_nt", r"gemm_nt_kernel"), ("gemm_tn256", r"gemm_tn256_kernel"),
           ("gemm_tn", r"gemm_tn_kernel"), ("attn_fwd", r"attn_fwd_kernel"), ("attn_bwd_dq", r"attn_bwd_dq_kernel"),
           ("attn_bwd_dkv", r"attn_bwd_dkv_kernel"), ("ln_bwd", r"ln_bwd_kernel"), ("ln_fwd", r"ln_fwd_kernel"),
           # round 5: the HBM-bound paths north_star names - front end, loss, fused norms, optimiser
           ("ln_fwd_multi", r"ln_fwd_multi_kernel"), ("ln_bwd_multi", r"ln_bwd_multi_kernel"), ("colsum", r"colsum_kernel"),
           ("ce_fwd_bwd", r"ce_fwd_bwd_kernel"), ("embed", r"::embed_kernel"), ("compact", r"compact_kernel"),
           ("embed_tables", r"embed_tables_kernel"), ("embed_sums", r"embed_sums_kernel"), ("adamw", r"adamw_kernel"),
           ("cast_w", r"cast_w_kernel"), ("sqnorm", r"sqnorm_kernel"), ("tn_reduce", r"tn_reduce_kernel"),
           ("bias_grad", r"bias_grad_kernel"), ("loss_perm", r"loss_perm_kernel"), ("swiglu_fwd", r"swiglu_fwd_kernel"),
           ("swiglu_bwd", r"swiglu_bwd_kernel")]


def load(path, counter):
    per = defaultdict(lambda: [0.0, set()])
    with open(path) as f:
        for row in csv.DictReader(f):
            if row.get("Counter_Name") != counter:
                continue
            name = row["Kernel_Name"]
            for cls, pat in CLASSES:
                if re.search(pat, name):
                    per[cls][0] += float(row["Counter_Value"])
                    per[cls][1].add(row["Dispatch_Id"])
                    break
    return {k: (v[0], len(v[1])) for k, v in per.items()}


def main():
    fetch = load(sys.argv[1], "FETCH_SIZE")
    write = load(sys.argv[2], "WRITE_SIZE")
    out = {}
    for cls in fetch:
        f_kb, n = fetch[cls]
        w_kb, n2 = write.get(cls, (0.0, n))
        out[cls] = {"launches": n, "fetch_bytes_per_launch": 2.0 * f_kb * 1024 / n, "write_bytes_per_launch": w_kb * 1024 / max(n2, 1),
                    "traffic_bytes_per_launch": 2.0 * f_kb * 1024 / n + w_kb * 1024 / max(n2, 1)}
    # C-ABI level classes (what bench.py's KernelTimer brackets): both device kernels of one entry point together
    abi = {}
    # (the first listed kernel gives the call count of a multi-kernel entry point; gemm_nt / gemm_tn: either kernel is a call.
    #  The HBM-bound entries carry bench.py's hbm_paths names; colsum_kernel's bytes are shared by the LayerNorm / embedding
    #  backward launches and stay a class of their own: ~0.3 % of ln_bwd's bytes)
    for entry, parts in (("gemm_nt", ("gemm_nt", "gemm_nt256")), ("gemm_tn", ("gemm_tn", "gemm_tn256")),
                         ("attn_bwd", ("attn_bwd_dq", "attn_bwd_dkv")), ("attn_fwd", ("attn_fwd",)),
                         ("layernorm_fwd", ("ln_fwd",)), ("layernorm_bwd", ("ln_bwd",)), ("layernorm_fwd_multi", ("ln_fwd_multi",)),
                         ("layernorm_bwd_multi", ("ln_bwd_multi",)), ("ce_fwd_bwd", ("ce_fwd_bwd",)), ("other:embed_fwd", ("embed",)),
                         ("other:compact", ("compact",)), ("other:embed_bwd", ("embed_sums", "embed_tables")),
                         ("other:adamw_step", ("adamw",)), ("other:cast_weight", ("cast_w",)), ("other:grad_sqnorm", ("sqnorm",)),
                         ("other:bias_grad", ("bias_grad",)), ("other:loss_perm", ("loss_perm",))):
        have = [out[k] for k in parts if k in out]
        if not have:
            continue
        n = sum(h["launches"] for h in have) if entry in ("gemm_nt", "gemm_tn") else have[0]["launches"]
        tot = sum(h["traffic_bytes_per_launch"] * h["launches"] for h in have)
        abi[entry] = {"launches": n, "traffic_bytes_per_launch": tot / n}
    meta = {"note": "HBM-side bytes per launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes); FETCH_SIZE doubled "
                    "per the gfx950 correction (MI355X_MICROARCH.md, HBM); counter units KiB -> bytes.  The correction is calibrated for "
                    "wide (16 B per lane) streaming reads, which is what the row kernels issue; FETCH_SIZE counts requests leaving the "
                    "L2s, Infinity-Cache hits included: bytes a 256 MB cache serves (weights, a tensor the previous kernel just wrote) "
                    "are in `traffic` but never reached HBM",
            "command": "rocprofv3 --pmc <FETCH_SIZE|WRITE_SIZE> --output-format csv -- python3 bench.py --clips-per-gpu MB --micro-batch MB --steps 1 "
                       "--warmup 0 --no-cpu-baseline --no-kernel-profile",
            "micro_batch": int(os.environ.get("EGOM2P_PMC_MICRO_BATCH", "64")), "kernel_src_sha": kernel_source_sha(), "kernels": out, "abi": abi}
    json.dump(meta, open(sys.argv[3], "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
