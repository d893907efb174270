#!/bin/bash
# round 5, call B: the new paths (register tokens, top-k, per-tensor AdamW step, regrouped buckets), then the PMC / SQ counter passes
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest -q -p no:cacheprovider tests/test_model_api_gpu.py tests/test_robustness_gpu.py tests/test_dp_gpu.py tests/test_frontend_gpu.py \
    "tests/test_engine_gpu.py::test_engine_matches_reference[b2_reg4]" "tests/test_engine_gpu.py::test_engine_matches_reference[tiny_pad]" \
    "tests/test_engine_gpu.py::test_engine_matches_reference[b2_ragged]" \
    "tests/test_engine_bf16_oracle_gpu.py::test_engine_matches_bf16_mode_oracle[b2_reg4]" \
    tests/test_generate_gpu.py > gpurun_out/r5b_tests.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r5b_tests.log
tail -25 gpurun_out/r5b_tests.log
bash tools/pmc_run.sh r05_b 2>&1 | tail -4 && bash tools/pmc_sq.sh r05_b 2>&1 | grep "pmc_sq"
