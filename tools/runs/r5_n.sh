#!/bin/bash
# round 5, call N: attention kernels compiled without SLP packing of f32 VALU (v_pk_mul_f32 / v_pk_add_f32 beside MFMAs) - A/B against the product build
mkdir -p gpurun_out
timeout -k 10 300 python3 tools/attn_ab.py > gpurun_out/r5n_ab.log 2> gpurun_out/r5n_ab.err; echo "ab rc=$?"; cat gpurun_out/r5n_ab.log
