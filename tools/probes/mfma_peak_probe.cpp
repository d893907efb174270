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
typedef int v8i __attribute__((ext_vector_type(8)));

// KIND 0: bf16 16x16x32, 1: bf16 32x32x16, 2: block-scaled e4m3 16x16x128 (v_mfma_scale_f32_16x16x128_f8f6f4, scales 1.0: what
// the fp8 forward GEMM issues), 3: plain e4m3 16x16x32 (v_mfma_f32_16x16x32_fp8_fp8).  Round 4: "the e4m3
// K-tile takes 1.7 x the bf16 K-tile's time at equal bytes and MFMA cycles" - is it cycles, or the clock the chip holds?
template <int KIND>
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
    } else if constexpr (KIND == 2) {
        v8i a8, b8;                                   // random e4m3 bytes (exponent field kept off 1111: finite)
        for (int i = 0; i < 8; ++i) {
            a8[i] = (int)((threadIdx.x * 2654435761u + i * 40503u) & 0x77777777u | 0x08080808u);
            b8[i] = (int)((threadIdx.x * 2246822519u + i * 3266489917u) & 0xf7777777u | 0x08080808u);
        }
        f32x4 c[8];
        for (int j = 0; j < 8; ++j) c[j] = f32x4{0, 0, 0, 0};
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int j = 0; j < 8; ++j)
                asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,0,0]" : "+v"(c[j]) : "v"(a8), "v"(b8), "v"(127), "v"(127));
        for (int j = 0; j < 8; ++j) acc_out += c[j][0] + c[j][3];
    } else if constexpr (KIND == 3) {
        long a8 = (long)((threadIdx.x * 2654435761ull * 40503ull) & 0x7777777777777777ull | 0x0808080808080808ull);
        long b8 = (long)((threadIdx.x * 2246822519ull * 3266489917ull) & 0xf777777777777777ull | 0x0808080808080808ull);
        f32x4 c[8];
        for (int j = 0; j < 8; ++j) c[j] = f32x4{0, 0, 0, 0};
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int j = 0; j < 8; ++j) asm volatile("v_mfma_f32_16x16x32_fp8_fp8 %0, %1, %2, %0" : "+v"(c[j]) : "v"(a8), "v"(b8));
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
    const double mfma_per_wave = (double)iters * (KIND == 1 ? 4 : 8);
    const double flop_per_mfma = KIND == 0 ? 16384.0 : KIND == 1 ? 32768.0 : KIND == 2 ? 65536.0 : 16384.0;     // 2 m n k
    const double flop = mfma_per_wave * flop_per_mfma * blocks * 4;
    const double ghz = cyc / real * 0.1;                                                    // memrealtime ticks at 100 MHz
    printf("%s, %d wave(s)/SIMD: %.1f TFLOP/s (event time %.3f ms), shader clock %.3f GHz, %.2f cycles per MFMA per SIMD\n", name,
           waves_per_simd, flop / (ms * 1e-3) / 1e12, ms, ghz, (cyc / blocks) / (mfma_per_wave * waves_per_simd));
    fflush(stdout);
    hipFree(out); hipFree(clk);
}

int main() {
    for (int w = 1; w <= 2; ++w) {
        run<0>("v_mfma_f32_16x16x32_bf16", w, 400000);
        run<1>("v_mfma_f32_32x32x16_bf16", w, 200000);
        run<2>("v_mfma_scale_f32_16x16x128_f8f6f4 (e4m3, scale 1.0)", w, 200000);
        run<3>("v_mfma_f32_16x16x32_fp8_fp8", w, 400000);
    }
    return 0;
}
