#!/bin/bash
# round 5, call K2: counter passes on the end-of-round kernel sources -> pmc_latest.json, then the default bench line; registered ego-L
mkdir -p gpurun_out
bash tools/pmc_run.sh r05_d 2>&1 | tail -1 && cp gpurun_out/r05_d_pmc.json profiles/pmc_latest.json
bash tools/pmc_sq.sh r05_d 2>&1 | grep "wrote"
timeout -k 10 900 python bench.py > gpurun_out/r05_d_bench.json 2> gpurun_out/r05_d_bench.err
echo "bench rc=$?"; cut -c1-330 gpurun_out/r05_d_bench.json
timeout -k 10 300 python bench.py --model egom2p_large_24e_24d_swiglu_nobias --clips-per-gpu 64 --micro-batch 32 --steps 2 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/r05_egoL1020_mb32_bench.json 2>/dev/null; cut -c1-300 gpurun_out/r05_egoL1020_mb32_bench.json
