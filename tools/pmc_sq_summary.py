"""Per kernel class: MFMA-pipe utilisation and where the waves' cycles go, from one rocprofv3 --pmc pass (tools/pmc_sq.sh).

  mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel cycles), kernel cycles = GRBM_GUI_ACTIVE / 8 (rocprofv3 sums the
  8 XCDs; MI355X_MICROARCH.md, DVFS give-back) - the share of the launch during which the MFMA pipes were executing;
  parked / stalled / issuing = SQ_WAIT_ANY / SQ_WAIT_INST_ANY / SQ_ACTIVE_INST_ANY over SQ_WAVE_CYCLES (disjoint, quad-cycles);
  lds_conflict = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE.
usage: python tools/pmc_sq_summary.py <counter_collection.csv> <out.json>
"""
import csv
import json
import os
import re
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egom2p_amd.profiler import kernel_source_sha  # noqa: E402

CLASSES = [("gemm_nt256<0>", r"gemm_nt256_kernel<0"), ("gemm_nt256<1>", r"gemm_nt256_kernel<1"), ("gemm_nt256<2>", r"gemm_nt256_kernel<2"),
           ("gemm_nt256<3>", r"gemm_nt256_kernel<3"), ("gemm_tn256", r"gemm_tn256_kernel"), ("attn_fwd", r"attn_fwd_kernel"),
           ("attn_bwd_dq", r"attn_bwd_dq_kernel"), ("attn_bwd_dkv", r"attn_bwd_dkv_kernel"),
           ("hd_fwd", r"hd_fwd_kernel"), ("hd_dq", r"hd_dq_kernel"), ("hd_dkv", r"hd_dkv_kernel")]


def main():
    acc = defaultdict(lambda: defaultdict(float))
    disp = defaultdict(set)
    with open(sys.argv[1]) as f:
        for row in csv.DictReader(f):
            for cls, pat in CLASSES:
                if re.search(pat, row["Kernel_Name"]):
                    acc[cls][row["Counter_Name"]] += float(row["Counter_Value"])
                    disp[cls].add(row["Dispatch_Id"])
                    break
    out = {}
    for cls, c in acc.items():
        cycles = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
        wave = max(c.get("SQ_WAVE_CYCLES", 0.0), 1.0)
        out[cls] = {"launches": len(disp[cls]),
                    "mfma_busy": round(c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / max(1024.0 * cycles, 1.0), 4),
                    "kernel_cycles_per_launch": round(cycles / max(len(disp[cls]), 1)),
                    "waves_parked": round(c.get("SQ_WAIT_ANY", 0.0) / wave, 4),
                    "waves_issue_stalled": round(c.get("SQ_WAIT_INST_ANY", 0.0) / wave, 4),
                    "waves_issuing": round(c.get("SQ_ACTIVE_INST_ANY", 0.0) / wave, 4),
                    "lds_conflict": round(c.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(c.get("SQ_LDS_IDX_ACTIVE", 0.0), 1.0), 4)}
    meta = {"note": __doc__.split("usage")[0].strip(), "kernel_src_sha": kernel_source_sha(),
            "micro_batch": int(os.environ.get("MB", "64")), "kernels": out}
    json.dump(meta, open(sys.argv[2], "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
