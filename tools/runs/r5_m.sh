#!/bin/bash
# round 5, call M: the tests touched after the last full run
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest -q -p no:cacheprovider tests/test_model_api_gpu.py tests/test_generate_gpu.py -k "adamw or whole_schedule or sampler" > gpurun_out/r5m_tests.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r5m_tests.log; tail -4 gpurun_out/r5m_tests.log | cut -c1-300
