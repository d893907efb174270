#!/bin/bash
# round 5, call O: the whole GPU suite + smoke on the final tree (after the late attention experiments were reverted)
mkdir -p gpurun_out
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > gpurun_out/r5o_tests.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r5o_tests.log
timeout -k 10 200 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r5o_smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/r5o_smoke.log
