// Probe: what the MFMA pipes of an MI355X sustain with NO memory traffic, and at which shader clock.
// Every wave runs a register-only loop of independent v_mfma_f32_16x16x32_bf16 (or 32x32x16) chains (inline asm with VGPR
// accumulators: through the builtin hipcc parks them in AGPRs and copies them every iteration); s_memtime (shader
// clock) against s_memrealtime (100 MHz) gives the clock the chip actually holds under that load.  The result calibrates the
// "peak" of the GEMM rooflines in DESIGN.md: fraction of the 2.5 PFLOP/s datasheet figure vs fraction of what the silicon
// delivers at its sustained clock.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_peak_probe tools/probes/mfma_peak_probe.cpp && ./mfma_peak_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int KIND>   // 0: 16x16x32, 1: 32x32x16
__global__ __launch_bounds__(256) void burn(float* out, unsigned long long* clk, int iters) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (threadIdx.x + i)); b[i] = (__bf16)(0.002f * (threadIdx.x - i)); }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    float acc_out = 0.f;
    if constexpr (KIND == 0) {
        f32x4 c[8];
        for (int j = 0; j < 8; ++j) c[j] = f32x4{0, 0, 0, 0};
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int j = 0; j < 8; ++j) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c[j]) : "v"(a), "v"(b));
        for (int j = 0; j < 8; ++j) acc_out += c[j][0] + c[j][3];
    } else {
        f32x16 c[4];
        for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) c[j][i] = 0.f;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int j = 0; j < 4; ++j) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c[j]) : "v"(a), "v"(b));
        for (int j = 0; j < 4; ++j) acc_out += c[j][0] + c[j][15];
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc_out;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int KIND>
void run(const char* name, int waves_per_simd, int iters) {
    const int blocks = 256 * waves_per_simd;          // 256 CUs x (4 waves per block = 1 wave per SIMD per block)
    float* out; unsigned long long* clk;
    hipMalloc(&out, blocks * 256 * 4); hipMalloc(&clk, blocks * 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(burn<KIND>, dim3(blocks), dim3(256), 0, 0, out, clk, iters / 10);      // warm-up
    hipEventRecord(e0);
    hipLaunchKernelGGL(burn<KIND>, dim3(blocks), dim3(256), 0, 0, out, clk, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(2 * blocks);
    hipMemcpy(h.data(), clk, blocks * 16, hipMemcpyDeviceToHost);
    double cyc = 0, real = 0;
    for (int i = 0; i < blocks; ++i) { cyc += h[2 * i]; real += h[2 * i + 1]; }
    const double mfma_per_wave = (double)iters * (KIND == 0 ? 8 : 4);
    const double flop = mfma_per_wave * (KIND == 0 ? 16384.0 : 32768.0) * blocks * 4;
    const double ghz = cyc / real * 0.1;                                                    // memrealtime ticks at 100 MHz
    printf("%s, %d wave(s)/SIMD: %.1f TFLOP/s (event time %.3f ms), shader clock %.3f GHz, %.2f cycles per MFMA per SIMD\n", name,
           waves_per_simd, flop / (ms * 1e-3) / 1e12, ms, ghz, (cyc / blocks) / (mfma_per_wave * waves_per_simd));
    hipFree(out); hipFree(clk);
}

int main() {
    for (int w = 1; w <= 2; ++w) {
        run<0>("v_mfma_f32_16x16x32_bf16", w, 400000);
        run<1>("v_mfma_f32_32x32x16_bf16", w, 200000);
    }
    return 0;
}
