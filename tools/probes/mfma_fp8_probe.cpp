// Probe: operand layout and scale semantics of v_mfma_scale_f32_16x16x128_f8f6f4 with e4m3 operands on gfx950.
// Hypothesis (by analogy with 16x16x32 bf16): lane l holds A[row l&15][k = 32*(l>>4) + j], j = 0..31 in its 8 VGPRs (byte j of the
// 32-byte fragment), B likewise with col l&15; the lane's scale byte (E8M0, 127 = 1.0) applies to its own 32-element block.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
#include <vector>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));

__global__ void k(const uint8_t* A, const uint8_t* B, float* C, int sa, int sb) {
    const int l = threadIdx.x;
    v8i a, b;
    const uint8_t* ap = A + (l & 15) * 128 + 32 * (l >> 4);
    const uint8_t* bp = B + (l & 15) * 128 + 32 * (l >> 4);
    for (int i = 0; i < 8; ++i) { a[i] = *(const int*)(ap + 4 * i); b[i] = *(const int*)(bp + 4 * i); }
    v4f c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, sa, 0, sb);
    for (int r = 0; r < 4; ++r) C[((l >> 4) * 4 + r) * 16 + (l & 15)] = c[r];      // row = (l>>4)*4 + r (from A), col = l & 15 (from B)
}

static float e4m3(uint8_t v) {
    int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
    float f = e == 0 ? ldexpf(m / 8.f, -6) : ldexpf(1.f + m / 8.f, e - 7);
    return s ? -f : f;
}
int main() {
    std::vector<uint8_t> A(16 * 128), B(16 * 128);
    uint32_t x = 12345;
    auto rnd = [&]() { x = x * 1664525u + 1013904223u; return x >> 8; };
    for (auto& v : A) { v = rnd() & 0x7f; if ((v & 0x78) == 0x78) v &= 0x3f; if (rnd() & 1) v |= 0x80; }   // finite e4m3
    for (auto& v : B) { v = rnd() & 0x7f; if ((v & 0x78) == 0x78) v &= 0x3f; if (rnd() & 1) v |= 0x80; }
    uint8_t *dA, *dB; float* dC;
    hipMalloc(&dA, A.size()); hipMalloc(&dB, B.size()); hipMalloc(&dC, 256 * 4);
    hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size(), hipMemcpyHostToDevice);
    for (int cfg = 0; cfg < 3; ++cfg) {
        const int sa = cfg == 0 ? 127 : cfg == 1 ? 128 : 127, sb = cfg == 2 ? 126 : 127;
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC, sa, sb);
        std::vector<float> C(256);
        hipMemcpy(C.data(), dC, 1024, hipMemcpyDeviceToHost);
        const double mul = ldexp(1.0, (sa - 127) + (sb - 127));
        double worst = 0, ref00 = 0;
        for (int m = 0; m < 16; ++m) for (int n = 0; n < 16; ++n) {
            double r = 0;
            for (int kk = 0; kk < 128; ++kk) r += (double)e4m3(A[m * 128 + kk]) * e4m3(B[n * 128 + kk]);
            r *= mul;
            if (m == 0 && n == 0) ref00 = r;
            worst = fmax(worst, fabs(C[m * 16 + n] - r) / fmax(1.0, fabs(r)));
        }
        printf("scale_a %d scale_b %d: C[0][0] %.4f ref %.4f   worst rel err (row-from-A/col-from-B layout) %.3e\n", sa, sb, C[0], ref00, worst);
        // transposed hypothesis
        double worst_t = 0;
        for (int m = 0; m < 16; ++m) for (int n = 0; n < 16; ++n) {
            double r = 0;
            for (int kk = 0; kk < 128; ++kk) r += (double)e4m3(A[m * 128 + kk]) * e4m3(B[n * 128 + kk]);
            worst_t = fmax(worst_t, fabs(C[n * 16 + m] - r * mul) / fmax(1.0, fabs(r * mul)));
        }
        printf("      transposed-output hypothesis worst %.3e\n", worst_t);
    }
    return 0;
}
