#!/bin/bash
# Which kernels does the vendor library (hipBLASLt / rocBLAS behind torch.nn.functional.linear) run on the long-K NT shapes where it
# is 6 - 12 % ahead of gemm_nt256_kernel?  The Tensile kernel names encode macro-tile (MT), MFMA instruction
# (MI), wave grouping, direct-to-LDS, prefetch depths ...  One rocprofv3 --kernel-trace pass (no counters) around the library
# calls only.  Run ON THE GPU BOX from the repo root:   bash tools/blaslt_names.sh r05   -> gpurun_out/<tag>_vendor_gemm_kernels.log
TAG=${1:-r05}
REPO=/root/repo
OUT=$REPO/gpurun_out/prof_${TAG}_blaslt
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export LIB_ONLY=1 SHAPES="${SHAPES:-qkv,fc2,dgrad qkv,dgrad fc13,logits}"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/tools/blaslt_compare.py > $OUT/run.log 2>&1 || { tail -20 $OUT/run.log; exit 1; }
S=$(find $OUT/trace -name '*kernel_stats.csv' | head -1)
cd $REPO
{
  echo "# vendor-library kernels on the training shapes (M = 131072; logits M = 64576), rocprofv3 --kernel-trace --stats around"
  echo "# LIB_ONLY=1 SHAPES='$SHAPES' python3 tools/blaslt_compare.py; the run's own TF/s lines first, then name, calls, average ns"
  grep "^NT" $OUT/run.log
  python3 - "$S" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"])):
    n = r["Name"]
    if "Cijk" in n or "gemm" in n.lower() or "blas" in n.lower():
        print(f'{r["Calls"]:>6} calls  avg {float(r["AverageNs"]):10.0f} ns  {n}')
PY
} > gpurun_out/${TAG}_vendor_gemm_kernels.log
rm -rf $OUT/trace
echo "[blaslt_names] wrote gpurun_out/${TAG}_vendor_gemm_kernels.log"
