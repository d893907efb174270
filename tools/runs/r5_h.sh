#!/bin/bash
# round 5, call H: full GPU suite on the current tree (baseline-recording mode), then bench + counter passes
mkdir -p gpurun_out; rm -f gpurun_out/bars.jsonl
EGOM2P_RECORD_BARS=/root/repo/gpurun_out/bars.jsonl timeout -k 10 1000 python -m pytest tests -q -m gpu -p no:cacheprovider > gpurun_out/r5h_tests.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r5h_tests.log; tail -8 gpurun_out/r5h_tests.log | cut -c1-300
