// Probe (round 4): how many bytes per second do the 256 CUs of an MI355X pull out of their L2s - into LDS by LDS-DMA
// (global_load_lds, 16 B per lane), or into registers (global_load_dwordx4) - when nothing else runs?  The working set is small
// enough to stay in every XCD's 4 MB L2 (2 MB; then 16 MB = Infinity Cache, 512 MB = HBM), every workgroup walks it with 64 - 128 KB
// in flight.  The NT GEMM's 256 x 256 tile needs 1 byte from L2 per 128 flop: this rate x 128 is its ceiling whatever the MFMA pipes do
// (DESIGN.md section 4e (46)).
//   hipcc --offload-arch=gfx950 -O3 -o l2_stream_probe tools/probes/l2_stream_probe.cpp && ./l2_stream_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__device__ __forceinline__ void glds16(const void* gsrc, void* lds_dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// MODE 0: LDS-DMA, MODE 1: loads into registers.  Each wave moves 1 KB per instruction; a workgroup keeps PIECES x WAVES KB in flight.
template <int MODE, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void stream(const char* src, long ws_bytes, int iters, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int PIECES = 64 / WAVES;                  // 64 KB in flight per workgroup
    // every workgroup starts somewhere else in the window and walks it in 64 KB steps
    long off = ((long)blockIdx.x * 65536) % ws_bytes;
    unsigned acc = 0;
    for (int it = 0; it < iters; ++it) {
        const char* base = src + off + wave * (PIECES * 1024) + lane * 16;
        if (MODE == 0) {
            // two halves of LDS alternate: the next 64 KB are issued before the previous 64 KB are waited for
            char* dst = smem + (it & 1) * 65536 + wave * (PIECES * 1024);
#pragma unroll
            for (int j = 0; j < PIECES; ++j) glds16(base + j * 1024, dst + j * 1024);
            if (PIECES == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        } else {
            u32x4 v[PIECES];
#pragma unroll
            for (int j = 0; j < PIECES; ++j) v[j] = *(const u32x4*)(base + j * 1024);
#pragma unroll
            for (int j = 0; j < PIECES; ++j) acc += v[j][0] ^ v[j][3];
        }
        off += 65536;
        if (off >= ws_bytes) off -= ws_bytes;
    }
    if (MODE == 0) acc = *(unsigned*)(smem + threadIdx.x * 4);
    if (acc == 0x12345678u) sink[0] = acc;              // (keeps the loads alive)
}

template <int MODE, int WAVES>
void run(const char* name, const char* buf, long ws_bytes, unsigned* sink, int wgs_per_cu) {
    const int iters = 2000, blocks = 256 * wgs_per_cu;
    hipFuncSetAttribute((const void*)stream<MODE, WAVES>, hipFuncAttributeMaxDynamicSharedMemorySize, MODE == 0 ? 131072 : 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((stream<MODE, WAVES>), dim3(blocks), dim3(WAVES * 64), MODE == 0 ? 131072 : 1024, 0, buf, ws_bytes, 50, sink);
    hipEventRecord(e0);
    hipLaunchKernelGGL((stream<MODE, WAVES>), dim3(blocks), dim3(WAVES * 64), MODE == 0 ? 131072 : 1024, 0, buf, ws_bytes, iters, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)blocks * iters * 65536.0;
    printf("%-44s window %5ld KB, %d WG/CU x %d waves: %7.2f TB/s\n", name, ws_bytes >> 10, wgs_per_cu, WAVES, bytes / (ms * 1e-3) / 1e12);
    fflush(stdout);
}

int main() {
    char* buf; unsigned* sink;
    const long cap = 1L << 30;
    hipMalloc(&buf, cap); hipMemset(buf, 1, cap); hipMalloc(&sink, 64);
    for (long ws : {2L << 20, 16L << 20, 512L << 20}) {      // L2-resident, Infinity-Cache-resident, HBM
        run<0, 4>("LDS-DMA (global_load_lds, 16 B per lane)", buf, ws, sink, 1);
        run<0, 8>("LDS-DMA (global_load_lds, 16 B per lane)", buf, ws, sink, 1);
        run<1, 4>("registers (global_load_dwordx4)", buf, ws, sink, 2);
        run<1, 8>("registers (global_load_dwordx4)", buf, ws, sink, 2);
        run<1, 8>("registers (global_load_dwordx4)", buf, ws, sink, 4);
    }
    return 0;
}
