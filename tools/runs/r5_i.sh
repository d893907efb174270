#!/bin/bash
# round 5, call I: counter passes on the current kernel sources, then the default bench line (reads the fresh pmc_latest.json)
mkdir -p gpurun_out
bash tools/pmc_run.sh r05_c 2>&1 | tail -2 && cp gpurun_out/r05_c_pmc.json profiles/pmc_latest.json
bash tools/pmc_sq.sh r05_c 2>&1 | grep "pmc_sq"
timeout -k 10 900 python bench.py --steps 5 --warmup 2 > gpurun_out/r05_c_bench.json 2> gpurun_out/r05_c_bench.err
echo "bench rc=$?"; tail -3 gpurun_out/r05_c_bench.err; cut -c1-300 gpurun_out/r05_c_bench.json
