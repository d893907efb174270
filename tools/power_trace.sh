#!/bin/bash
# Shader clock / socket power of GPU 0 sampled beside a bench.py run (rocm-smi reads only; no settings are changed).
#   bash tools/power_trace.sh <tag>      -> gpurun_out/<tag>_power_trace.txt, gpurun_out/<tag>_power_bench.json
TAG=${1:-r03}
mkdir -p gpurun_out
python bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-kernel-profile > gpurun_out/${TAG}_power_bench.json 2> gpurun_out/${TAG}_power_bench.err &
BPID=$!
: > gpurun_out/${TAG}_power_trace.txt
while kill -0 $BPID 2>/dev/null; do
    echo "t=$(date +%s.%N)" >> gpurun_out/${TAG}_power_trace.txt
    timeout 5 rocm-smi -d 0 --showclocks --showpower --showtemp --showuse 2>/dev/null | grep -E "sclk|mclk|fclk|Power|Temperature \(Sensor (edge|junction|memory)|GPU use" >> gpurun_out/${TAG}_power_trace.txt
    sleep 0.5
done
wait $BPID
echo "bench exit $?"
