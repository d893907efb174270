#!/bin/bash
# round 5, call K1: the full GPU suite (recorded baselines enforced) and smoke() on the end-of-round tree
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -x -q -m gpu -p no:cacheprovider > gpurun_out/r5k_tests.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r5k_tests.log; tail -5 gpurun_out/r5k_tests.log | cut -c1-300
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
