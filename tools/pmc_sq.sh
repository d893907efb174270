#!/bin/bash
# Shader-side counters (rocprofv3 PMC, no tracing domains beside them) of one micro-batch step of bench.py, per kernel class:
#   pass 1  MFMA-pipe busy cycles, wave cycles split into parked / issue-stalled / issuing, LDS bank conflicts
#   pass 2  (round 5) WHAT the issue-stalled / issuing cycles are: cycles with a VALU / LDS / VMEM / scalar
#           instruction active, the LDS sub-bucket of the issue stalls, SALU instruction cycles
#   pass 3  instruction counts (VALU, MFMA, LDS, SALU, VMEM, transcendental) and VALU || MFMA co-execution cycles
# Passes 2 and 3 may be skipped with SQ_PASSES=1; a pass whose counter set the profiler refuses is reported and left out.
# Run ON THE GPU BOX from the repo root:   bash tools/pmc_sq.sh r05_a     -> gpurun_out/<tag>_sq.json
TAG=${1:-r05}
REPO=/root/repo
OUT=$REPO/gpurun_out/prof_${TAG}_sq
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export MB=${MB:-64}
PASSES=${SQ_PASSES:-3}
ARGS="--clips-per-gpu $MB --micro-batch $MB --steps 1 --warmup 0 --no-cpu-baseline --no-kernel-profile --no-extras"
P1="SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"
P2="SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU"
P3="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_VALU_TRANS_F32 SQ_VALU_MFMA_COEXEC_CYCLES SQ_BUSY_CYCLES"
FILES=""
for i in 1 2 3; do
    [ $i -le $PASSES ] || continue
    eval "CNT=\$P$i"
    if rocprofv3 --pmc $CNT --output-format csv -d $OUT/sq$i -- python3 $REPO/bench.py $ARGS > $OUT/sq$i.log 2>&1; then
        F=$(find $OUT/sq$i -name '*counter_collection.csv' | head -1)
        [ -n "$F" ] && FILES="$FILES $F"
        echo "[pmc_sq] pass $i done"
    else
        echo "[pmc_sq] pass $i FAILED (see $OUT/sq$i.log)"; tail -5 $OUT/sq$i.log
    fi
done
cd $REPO
python3 tools/pmc_sq_summary.py gpurun_out/${TAG}_sq.json $FILES
rm -rf $OUT/sq1 $OUT/sq2 $OUT/sq3
echo "[pmc_sq] wrote gpurun_out/${TAG}_sq.json"
