#!/bin/bash
# round 5, call F: LayerNorm rewrite + NT selector: kernel tests, LN bandwidth, GEMM sweep with the new selector, generation bench, ego-L bench
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest -q -p no:cacheprovider tests/test_kernels_gpu.py tests/test_generate_gpu.py "tests/test_engine_gpu.py::test_engine_matches_reference[b2]" "tests/test_engine_gpu.py::test_engine_matches_reference[L1020]" "tests/test_engine_gpu.py::test_engine_matches_reference[XL2046]" "tests/test_engine_gpu.py::test_engine_matches_reference[L2]" tests/test_robustness_gpu.py > gpurun_out/r5f_tests.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r5f_tests.log; tail -6 gpurun_out/r5f_tests.log | cut -c1-300
timeout -k 10 200 python tools/ln_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_ln_bench_after.log
timeout -k 10 400 python tools/gen_gemm_sweep.py > gpurun_out/r05_gen_gemm_sweep_after.log 2>&1; grep -v amdgpu.ids gpurun_out/r05_gen_gemm_sweep_after.log | grep -E "auto|gate|fc13|logits"
python eval_model_rgb2depth.py --bench 5 2>&1 | grep metric | cut -c1-400
python eval_model_rgb2depth.py --bench 3 --batch 8 2>&1 | grep metric | cut -c1-400
timeout -k 10 300 python bench.py --model ego_L_1152 --clips-per-gpu 64 --micro-batch 32 --steps 3 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/r5f_egoL.json 2> gpurun_out/r5f_egoL.err
python - <<'PY'
import json
r = json.load(open("gpurun_out/r5f_egoL.json"))
print("ego-L bf16", round(r["clips_per_s"], 2), {k: round(v["ms"], 1) for k, v in r["kernel_breakdown"].items() if v["ms"] > 2})
PY
