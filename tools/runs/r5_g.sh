#!/bin/bash
# round 5, call G: generation path with independent launches on a second stream (A/B), split rule, fused gate from 3000 rows
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest -q -p no:cacheprovider -x tests/test_generate_gpu.py tests/test_robustness_gpu.py > gpurun_out/r5g_tests.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r5g_tests.log; tail -6 gpurun_out/r5g_tests.log | cut -c1-300
for ov in 1 0; do
  echo "EGOM2P_GEN_OVERLAP=$ov"
  EGOM2P_GEN_OVERLAP=$ov python eval_model_rgb2depth.py --bench 8 2>&1 | grep metric | cut -c1-330
  EGOM2P_GEN_OVERLAP=$ov python eval_model_rgb2depth.py --bench 3 --batch 8 2>&1 | grep metric | cut -c1-330
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/prof_r05_eval -- python3 /root/repo/eval_model_rgb2depth.py --bench 5 > /root/repo/gpurun_out/r05_eval_trace.log 2>&1
S=$(find /root/repo/gpurun_out/prof_r05_eval -name '*kernel_stats.csv' | head -1); cp "$S" /root/repo/gpurun_out/r05_eval_rgb2depth_kernel_stats.csv
rm -rf /root/repo/gpurun_out/prof_r05_eval
