"""Per kernel class: MFMA-pipe utilisation and where the waves' cycles go, from the rocprofv3 --pmc passes of tools/pmc_sq.sh.

  pass 1: mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel cycles), kernel cycles = GRBM_GUI_ACTIVE / 8 (rocprofv3
  sums the 8 XCDs; MI355X_MICROARCH.md, DVFS give-back) - the share of the launch during which the MFMA pipes were executing;
  parked / stalled / issuing = SQ_WAIT_ANY / SQ_WAIT_INST_ANY / SQ_ACTIVE_INST_ANY over SQ_WAVE_CYCLES (disjoint, quad-cycles);
  lds_conflict = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE.
  pass 2 (own SQ_WAVE_CYCLES): active_valu / active_lds / active_vmem / active_scalar / active_misc = SQ_ACTIVE_INST_* over
  SQ_WAVE_CYCLES (the unit that executes while a wave is "issuing"; MFMAs count as VALU), stalled_on_lds = SQ_WAIT_INST_LDS
  over SQ_WAVE_CYCLES (the LDS-issue sub-bucket of waves_issue_stalled, NOT lgkmcnt waits), salu_cycles likewise.
  pass 3: instructions per launch by unit, VALU instructions per MFMA, and valu_mfma_coexec = SQ_VALU_MFMA_COEXEC_CYCLES /
  SQ_BUSY_CYCLES.
usage: python tools/pmc_sq_summary.py <out.json> <counter_collection.csv> [<counter_collection.csv> ...]
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


def r4(x):
    return round(x, 4)


def main():
    out = {}
    for path in sys.argv[2:]:
        acc = defaultdict(lambda: defaultdict(float))
        disp = defaultdict(set)
        with open(path) as f:
            for row in csv.DictReader(f):
                for cls, pat in CLASSES:
                    if re.search(pat, row["Kernel_Name"]):
                        acc[cls][row["Counter_Name"]] += float(row["Counter_Value"])
                        disp[cls].add(row["Dispatch_Id"])
                        break
        for cls, c in acc.items():
            o = out.setdefault(cls, {"launches": len(disp[cls])})
            n = max(len(disp[cls]), 1)
            wave = max(c.get("SQ_WAVE_CYCLES", 0.0), 1.0)
            if "SQ_VALU_MFMA_BUSY_CYCLES" in c:
                cycles = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
                o.update(mfma_busy=r4(c["SQ_VALU_MFMA_BUSY_CYCLES"] / max(1024.0 * cycles, 1.0)),
                         kernel_cycles_per_launch=round(cycles / n),
                         waves_parked=r4(c.get("SQ_WAIT_ANY", 0.0) / wave),
                         waves_issue_stalled=r4(c.get("SQ_WAIT_INST_ANY", 0.0) / wave),
                         waves_issuing=r4(c.get("SQ_ACTIVE_INST_ANY", 0.0) / wave),
                         lds_conflict=r4(c.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(c.get("SQ_LDS_IDX_ACTIVE", 0.0), 1.0)))
            if "SQ_ACTIVE_INST_VALU" in c:
                o.update(active_valu=r4(c["SQ_ACTIVE_INST_VALU"] / wave), active_lds=r4(c.get("SQ_ACTIVE_INST_LDS", 0.0) / wave),
                         active_vmem=r4(c.get("SQ_ACTIVE_INST_VMEM", 0.0) / wave), active_scalar=r4(c.get("SQ_ACTIVE_INST_SCA", 0.0) / wave),
                         active_misc=r4(c.get("SQ_ACTIVE_INST_MISC", 0.0) / wave), stalled_on_lds=r4(c.get("SQ_WAIT_INST_LDS", 0.0) / wave),
                         salu_cycles=r4(c.get("SQ_INST_CYCLES_SALU", 0.0) / wave))
            if "SQ_INSTS_VALU" in c:
                mf = max(c.get("SQ_INSTS_MFMA", 0.0), 1.0)
                o.update(insts_per_launch={k[9:].lower(): round(c[k] / n) for k in c if k.startswith("SQ_INSTS_")},
                         valu_per_mfma=r4((c["SQ_INSTS_VALU"] - c.get("SQ_INSTS_MFMA", 0.0)) / mf),
                         lds_per_mfma=r4(c.get("SQ_INSTS_LDS", 0.0) / mf), salu_per_mfma=r4(c.get("SQ_INSTS_SALU", 0.0) / mf),
                         trans_per_mfma=r4(c.get("SQ_INSTS_VALU_TRANS_F32", 0.0) / mf),
                         valu_mfma_coexec=r4(c.get("SQ_VALU_MFMA_COEXEC_CYCLES", 0.0) / max(c.get("SQ_BUSY_CYCLES", 0.0), 1.0)))
    meta = {"note": __doc__.split("usage")[0].strip(), "kernel_src_sha": kernel_source_sha(),
            "micro_batch": int(os.environ.get("MB", "64")), "kernels": out}
    json.dump(meta, open(sys.argv[1], "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
