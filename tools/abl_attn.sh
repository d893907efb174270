#!/bin/bash
# Build timing-only variants of the attention kernels into variants/ (git-ignored .so files that travel with gpurun).
#   bash tools/abl_attn.sh build "<name>:<flags>" ...      e.g. "dq:-DATT_ONLY=1"   (SRC=gemm selects gemm.hip instead of attention.hip)
#   bash tools/abl_attn.sh run                              (on the GPU box) runs BENCH (default tools/attn_bench.py) with every variants/*.so
set -e
cd "$(dirname "$0")/.."
if [ "$1" = build ]; then
    shift
    make -C egom2p_amd/csrc -j8 > /dev/null
    for spec in "$@"; do
        name=${spec%%:*}; flags=${spec#*:}
        /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iegom2p_amd/csrc -Iinclude -Wno-unused-result $flags \
            -c egom2p_amd/csrc/${SRC:-attention}.hip -o build/abl_$name.o
        objs=$(ls build/csrc/*.o | grep -v ${SRC:-attention}.o)
        /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o variants/libego_$name.so build/abl_$name.o $objs
        echo "built variants/libego_$name.so ($flags)"
    done
else
    for lib in variants/libego_*.so; do
        echo -n "$(basename $lib .so) "
        EGOM2P_HIP_LIB=$PWD/$lib B=${B:-32} KINDS=${KINDS:-full,blocks} python ${BENCH:-tools/attn_bench.py} 2>/dev/null | tail -1
    done
fi
