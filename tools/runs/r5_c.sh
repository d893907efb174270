#!/bin/bash
# round 5, call C: GEMM tests after the probe hook, the de-phasing probe, kernel trace of the rgb -> depth generation path
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest -q -p no:cacheprovider tests/test_kernels_gpu.py -k "gemm" > gpurun_out/r5c_tests.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r5c_tests.log; tail -3 gpurun_out/r5c_tests.log
timeout -k 10 500 python tools/dephase_probe.py > gpurun_out/r05_dephase_probe.log 2>&1; echo "probe rc=$?"; cat gpurun_out/r05_dephase_probe.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/prof_r05_eval -- python3 /root/repo/eval_model_rgb2depth.py --bench 5 > /root/repo/gpurun_out/r05_eval_trace.log 2>&1
echo "eval trace rc=$?"; tail -3 /root/repo/gpurun_out/r05_eval_trace.log | cut -c1-300
S=$(find /root/repo/gpurun_out/prof_r05_eval -name '*kernel_stats.csv' | head -1); cp "$S" /root/repo/gpurun_out/r05_eval_rgb2depth_kernel_stats.csv
rm -rf /root/repo/gpurun_out/prof_r05_eval
python3 eval_model_rgb2depth.py --bench 5 2>&1 | tail -2
