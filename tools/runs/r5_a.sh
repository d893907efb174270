#!/bin/bash
# round 5, call A: full GPU suite in baseline-recording mode, vendor library kernel names, baseline bench of this box
cd /root/repo && mkdir -p gpurun_out
rm -f gpurun_out/bars.jsonl
EGOM2P_RECORD_BARS=/root/repo/gpurun_out/bars.jsonl timeout -k 10 900 python -m pytest tests -q -m gpu -p no:cacheprovider > gpurun_out/r5a_tests.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r5a_tests.log
tail -15 gpurun_out/r5a_tests.log
bash tools/blaslt_names.sh r05 2>&1 | tail -3
timeout -k 10 400 python bench.py --steps 5 --warmup 2 > gpurun_out/r5a_bench.json 2> gpurun_out/r5a_bench.err
echo "bench rc=$?"; cut -c1-600 gpurun_out/r5a_bench.json
