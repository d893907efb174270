"""Print per-kernel sums of SQ counters from a rocprofv3 --pmc counter_collection.csv (one row per dispatch x counter)."""
import csv, sys, re
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(float)); n = defaultdict(set)
for path in sys.argv[1:]:
    for row in csv.DictReader(open(path)):
        m = re.search(r"(attn_\w+_kernel|gemm_\w+_kernel)", row["Kernel_Name"])
        if not m: continue
        acc[m.group(1)][row["Counter_Name"]] += float(row["Counter_Value"]); n[m.group(1)].add(row["Dispatch_Id"])
for k, c in acc.items():
    print(k, "dispatches", len(n[k]))
    for name, v in sorted(c.items()):
        print(f"   {name:32s} {v / len(n[k]):16.0f}")
