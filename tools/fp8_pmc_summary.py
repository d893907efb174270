"""Summarise `rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE` around
`PMC=1 python tools/fp8_frag_probe.py`: per shape the three GEMM launches in order (bf16, fp8 with the round-3 fragment map, fp8).
usage: python tools/fp8_pmc_summary.py <counter_collection.csv>"""
import csv
import sys
from collections import OrderedDict, defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
disp = OrderedDict()
for r in rows:
    if "gemm_nt256_kernel" not in r["Kernel_Name"]:
        continue
    d = disp.setdefault(int(r["Dispatch_Id"]), defaultdict(float))
    d[r["Counter_Name"]] += float(r["Counter_Value"])
    d["_name"] = r["Kernel_Name"][:60]
labels = ["bf16", "fp8_r3_map", "fp8"]
for i, (k, c) in enumerate(sorted(disp.items())):
    cyc = c["GRBM_GUI_ACTIVE"] / 8.0
    print(f"shape {i // 3} {labels[i % 3]:11s} kernel cycles {cyc:10.0f}  mfma_busy {c['SQ_VALU_MFMA_BUSY_CYCLES'] / max(1024 * cyc, 1):.3f}  "
          f"lds conflict cycles / lds active {c['SQ_LDS_BANK_CONFLICT'] / max(c['SQ_LDS_IDX_ACTIVE'], 1):.3f}  ({c['_name']})")
