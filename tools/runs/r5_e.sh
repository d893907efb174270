#!/bin/bash
# round 5, call E: low-latency NT tile instantiations (parity + sweep), LayerNorm bandwidth by row width
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest -q -p no:cacheprovider tests/test_kernels_gpu.py -k "small_grid" > gpurun_out/r5e_tests.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r5e_tests.log; tail -4 gpurun_out/r5e_tests.log
timeout -k 10 400 python tools/gen_gemm_sweep.py > gpurun_out/r05_gen_gemm_sweep.log 2>&1; grep -v amdgpu.ids gpurun_out/r05_gen_gemm_sweep.log
timeout -k 10 200 python tools/ln_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_ln_bench.log
