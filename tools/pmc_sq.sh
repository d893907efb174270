#!/bin/bash
# Shader-side counters (rocprofv3 PMC, one pass, no tracing domains beside it) of one micro-batch step of bench.py: MFMA-pipe
# busy cycles, wave cycles split into parked / issue-stalled / issuing, LDS bank conflicts - per kernel class.
# Run ON THE GPU BOX from the repo root:   bash tools/pmc_sq.sh r02_e     -> gpurun_out/<tag>_sq.json
set -e
TAG=${1:-r02}
REPO=/root/repo
OUT=$REPO/gpurun_out/prof_${TAG}_sq
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
MB=${MB:-64}
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE \
    --output-format csv -d $OUT/sq -- python3 $REPO/bench.py --clips-per-gpu $MB --micro-batch $MB --steps 1 --warmup 0 --no-cpu-baseline --no-kernel-profile --no-extras > $OUT/sq.log 2>&1
F=$(find $OUT/sq -name '*counter_collection.csv' | head -1)
cd $REPO
python3 tools/pmc_sq_summary.py "$F" gpurun_out/${TAG}_sq.json
rm -rf $OUT/sq
echo "[pmc_sq] wrote gpurun_out/${TAG}_sq.json"
