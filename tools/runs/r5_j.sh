#!/bin/bash
# round 5, call J: dK/dV phase probe; generation tests + bench on the re-sized split scratch; yaml4 preset; final generation trace
mkdir -p gpurun_out
timeout -k 10 200 python tools/attn_stagger_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_attn_stagger_probe.log
timeout -k 10 400 python -m pytest -q -p no:cacheprovider -x tests/test_generate_gpu.py tests/test_kernels_gpu.py -k "not other_head" > gpurun_out/r5j_tests.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r5j_tests.log; tail -4 gpurun_out/r5j_tests.log | cut -c1-300
python eval_model_rgb2depth.py --bench 8 2>&1 | grep metric | cut -c1-330
python eval_model_rgb2depth.py --bench 3 --batch 8 2>&1 | grep metric | cut -c1-330
timeout -k 10 200 python bench.py --preset yaml4 --steps 20 --warmup 3 --no-cpu-baseline --no-extras --no-kernel-profile > gpurun_out/r05_yaml4_preset_bench.json 2>/dev/null; cut -c1-260 gpurun_out/r05_yaml4_preset_bench.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/prof_r05_eval -- python3 /root/repo/eval_model_rgb2depth.py --bench 5 > /root/repo/gpurun_out/r05_eval_trace.log 2>&1
S=$(find /root/repo/gpurun_out/prof_r05_eval -name '*kernel_stats.csv' | head -1); cp "$S" /root/repo/gpurun_out/r05_eval_rgb2depth_kernel_stats.csv
rm -rf /root/repo/gpurun_out/prof_r05_eval
