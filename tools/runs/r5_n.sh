#!/bin/bash
# round 5, call N: dQ kernel prologue (DMA first, all row loads in one batch) - A/B against the previous build + attention parity
mkdir -p gpurun_out
timeout -k 10 300 python3 tools/attn_ab.py > gpurun_out/r5n_ab.log 2> gpurun_out/r5n_ab.err; echo "ab rc=$?"; cat gpurun_out/r5n_ab.log
timeout -k 10 500 python3 -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "attn or attention" > gpurun_out/r5n_tests.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r5n_tests.log
