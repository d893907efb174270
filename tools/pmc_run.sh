#!/bin/bash
# HBM-side traffic (rocprofv3 PMC, separate FETCH_SIZE / WRITE_SIZE passes, no tracing domains beside them) and the
# kernel-trace statistics of one micro-batch (MB, default 64 = bench.py's default) step of bench.py.  Run ON THE GPU BOX from the repo root:
#     bash tools/pmc_run.sh r02_a
# writes gpurun_out/<tag>_pmc.json (copy to profiles/<tag>_pmc.json and profiles/pmc_latest.json) and
# gpurun_out/<tag>_kernel_stats.csv (copy to profiles/<tag>_bench_kernel_stats.csv).
set -e
TAG=${1:-r02}
REPO=/root/repo
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
MB=${MB:-64}
export EGOM2P_PMC_MICRO_BATCH=$MB
ARGS="--clips-per-gpu $MB --micro-batch $MB --steps 1 --warmup 0 --no-cpu-baseline --no-kernel-profile --no-extras"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $REPO/bench.py $ARGS > $OUT/fetch.log 2>&1
echo "[pmc_run] FETCH_SIZE pass done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $REPO/bench.py $ARGS > $OUT/write.log 2>&1
echo "[pmc_run] WRITE_SIZE pass done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/bench.py --clips-per-gpu $MB --micro-batch $MB --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-profile --no-extras > $OUT/trace.log 2>&1
echo "[pmc_run] kernel-trace pass done"
F=$(find $OUT/fetch -name '*counter_collection.csv' | head -1)
W=$(find $OUT/write -name '*counter_collection.csv' | head -1)
S=$(find $OUT/trace -name '*kernel_stats.csv' | head -1)
cd $REPO
python3 tools/pmc_summary.py "$F" "$W" gpurun_out/${TAG}_pmc.json > $OUT/summary.log
cp "$S" gpurun_out/${TAG}_kernel_stats.csv
# the per-dispatch CSVs are large: keep only the summaries
rm -rf $OUT/fetch $OUT/write $OUT/trace
echo "[pmc_run] wrote gpurun_out/${TAG}_pmc.json gpurun_out/${TAG}_kernel_stats.csv"
