#!/bin/bash
# round 5, call L: the two-rank self-launch rehearsal of bench.py on the end-of-round tree (gloo, both ranks on the box's one GPU: not a performance number)
mkdir -p gpurun_out
EGOM2P_DIST_BACKEND=gloo EGOM2P_ONE_DEVICE=1 timeout -k 10 400 python3 bench.py --gpus 2 --steps 2 --warmup 1 --clips-per-gpu 64 --micro-batch 32 --no-cpu-baseline --no-extras --no-kernel-profile > gpurun_out/r05_bench_2rank_selflaunch_rehearsal.json 2> gpurun_out/r5l.err; echo "rc=$?"; cut -c1-400 gpurun_out/r05_bench_2rank_selflaunch_rehearsal.json
EGOM2P_DIST_BACKEND=gloo EGOM2P_ONE_DEVICE=1 timeout -k 10 400 python3 bench.py --gpus 2 --steps 2 --warmup 1 --clips-per-gpu 64 --micro-batch 32 --no-cpu-baseline --no-extras --no-kernel-profile --dp-algo rs_ag --sparse-tables on > gpurun_out/r05_bench_2rank_selflaunch_rehearsal_rs_ag_sparse.json 2>> gpurun_out/r5l.err; echo "rc=$?"; cut -c1-400 gpurun_out/r05_bench_2rank_selflaunch_rehearsal_rs_ag_sparse.json
tail -3 gpurun_out/r5l.err | cut -c1-300
