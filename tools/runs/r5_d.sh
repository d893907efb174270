#!/bin/bash
# round 5, call D: fp8 dgrad parity + ego-L steps bf16 / fp8 fwd / fp8 fwd+dgrad, de-phasing probe (interleaved), generation sweeps
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest -q -p no:cacheprovider -s tests/test_engine_gpu.py -k "fp8" tests/test_generate_gpu.py -k "fp8 or sampler" > gpurun_out/r5d_tests.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r5d_tests.log; grep -E "fp8 |passed|failed|rc=" gpurun_out/r5d_tests.log | cut -c1-300 | tail -12
for mode in "" "--fp8" "--fp8 --fp8-bwd"; do
  timeout -k 10 300 python bench.py --model ego_L_1152 --clips-per-gpu 64 --micro-batch 32 --steps 3 --warmup 1 --no-cpu-baseline --no-extras $mode > "gpurun_out/r5d_egoL${mode// /_}.json" 2> gpurun_out/r5d_egoL.err || tail -5 gpurun_out/r5d_egoL.err
  python - "gpurun_out/r5d_egoL${mode// /_}.json" <<'PY'
import json, sys
r = json.load(open(sys.argv[1]))
kb = r["kernel_breakdown"]
print(sys.argv[1], round(r["clips_per_s"], 2), "clips/s", {k: round(v["ms"], 1) for k, v in kb.items() if v["ms"] > 2})
PY
done
timeout -k 10 300 python tools/dephase_probe.py > gpurun_out/r05_dephase_probe.log 2>&1; cat gpurun_out/r05_dephase_probe.log | grep -v amdgpu.ids
timeout -k 10 300 python tools/gen_gemm_sweep.py > gpurun_out/r05_gen_gemm_sweep.log 2>&1; grep -v amdgpu.ids gpurun_out/r05_gen_gemm_sweep.log
timeout -k 10 300 python tools/gen_attn_split_sweep.py > gpurun_out/r05_gen_attn_split_sweep.log 2>&1; grep -v amdgpu.ids gpurun_out/r05_gen_attn_split_sweep.log
