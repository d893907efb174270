// Optimiser-side kernels over the flat fp32 parameter / gradient / moment buffers (gfx950, HBM-bound):
// global gradient sum of squares, and AdamW with the clip coefficient and the data-parallel 1/world
// factor folded in (no separate unscale / clip / zero_grad passes).
//
// Replaces: torch.nn.utils.clip_grad_norm_ (egom2p/utils/native_scaler.py:33) and torch.optim.AdamW
// created at egom2p/utils/optim_factory.py:226 (betas 0.9/0.95, eps 1e-8, decoupled weight decay).
#include "common.h"
#include "egom2p_hip.h"
#include <math.h>

namespace {

__global__ __launch_bounds__(256) void sqnorm_kernel(const float* __restrict__ g, long n4, long n, double* __restrict__ out,
                                                     double* __restrict__ part) {
    __shared__ float red[4];
    float s = 0.f;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const f32x4 v = *(const f32x4*)(g + i * 4);
        s += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0)
        for (long i = n4 * 4; i < n; ++i) s += g[i] * g[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const double t = (double)(red[0] + red[1] + red[2] + red[3]);
        if (part) part[blockIdx.x] = t;        // summed in block order by sqnorm_finish_kernel: bitwise reproducible
        else atomicAdd(out, t);
    }
}

__global__ __launch_bounds__(64) void sqnorm_finish_kernel(const double* __restrict__ part, int nparts, double* __restrict__ out) {
    if (threadIdx.x != 0) return;
    double s = 0.0;
    for (int i = 0; i < nparts; ++i) s += part[i];
    *out += s;
}

// p, g, m, v: flat fp32 of length n.  sqnorm: device double, sum of squares of the RAW grads.
// effective grad = g * gscale * clip,  clip = min(1, max_norm / (sqrt(sqnorm) * gscale + 1e-6)) (max_norm <= 0: no clip)
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, long n, float lr, float wd, float b1, float b2,
                                                    float eps, float bc1, float bc2_sqrt, float gscale, float max_norm,
                                                    const double* __restrict__ sqnorm, int zero_grad) {
    float coef = gscale;
    if (max_norm > 0.f && sqnorm) {
        const float total = (float)sqrt(*sqnorm) * gscale;
        coef *= fminf(1.f, max_norm / (total + 1e-6f));
    }
    const float step = lr / bc1, decay = 1.f - lr * wd;
    const long n4 = n >> 2;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        f32x4 pp = *(f32x4*)(p + i * 4), gg = *(f32x4*)(g + i * 4), mm = *(f32x4*)(m + i * 4), vv = *(f32x4*)(v + i * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float ge = gg[e] * coef;
            pp[e] *= decay;
            mm[e] = b1 * mm[e] + (1.f - b1) * ge;
            vv[e] = b2 * vv[e] + (1.f - b2) * ge * ge;
            pp[e] -= step * (mm[e] / (sqrtf(vv[e]) / bc2_sqrt + eps));
        }
        *(f32x4*)(p + i * 4) = pp; *(f32x4*)(m + i * 4) = mm; *(f32x4*)(v + i * 4) = vv;
        if (zero_grad) *(f32x4*)(g + i * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        for (long i = n4 * 4; i < n; ++i) {
            const float ge = g[i] * coef;
            float pe = p[i] * decay;
            const float me = b1 * m[i] + (1.f - b1) * ge, ve = b2 * v[i] + (1.f - b2) * ge * ge;
            pe -= step * (me / (sqrtf(ve) / bc2_sqrt + eps));
            p[i] = pe; m[i] = me; v[i] = ve;
            if (zero_grad) g[i] = 0.f;
        }
    }
}

}  // namespace

extern "C" int ego_abi_version(void) { return EGO_ABI_VERSION; }

extern "C" int ego_grad_sqnorm(const float* g, long n, double* out, double* work, hipStream_t stream) {
    if (n <= 0) return EGO_OK;
    if (((uintptr_t)g) % 16) return EGO_ERR_ARG;
    const long n4 = n / 4;
    int blocks = (int)((n4 + 255) / 256 < EGO_SQNORM_WORK ? (n4 + 255) / 256 : EGO_SQNORM_WORK);
    if (blocks < 1) blocks = 1;
    EGO_LAUNCH(sqnorm_kernel, dim3(blocks), dim3(256), 0, stream, g, n4, n, out, work);
    LAUNCH_CHECK();
    if (work) {
        EGO_LAUNCH(sqnorm_finish_kernel, dim3(1), dim3(64), 0, stream, (const double*)work, blocks, out);
        LAUNCH_CHECK();
    }
    return EGO_OK;
}

extern "C" int ego_adamw_step(float* p, float* g, float* m, float* v, long n, float lr, float wd, float beta1, float beta2,
                              float eps, int step, float gscale, float max_norm, const double* sqnorm, int zero_grad,
                              hipStream_t stream) {
    if (n <= 0) return EGO_OK;
    if (step < 1 || ((uintptr_t)p) % 16 || ((uintptr_t)g) % 16 || ((uintptr_t)m) % 16 || ((uintptr_t)v) % 16) return EGO_ERR_ARG;
    const float bc1 = (float)(1.0 - pow((double)beta1, (double)step));
    const float bc2s = (float)sqrt(1.0 - pow((double)beta2, (double)step));
    const long n4 = n / 4;
    const int blocks = (int)((n4 + 255) / 256 < 4096 ? (n4 + 255) / 256 : 4096);
    EGO_LAUNCH(adamw_kernel, dim3(blocks < 1 ? 1 : blocks), dim3(256), 0, stream, p, g, m, v, n, lr, wd, beta1, beta2, eps,
                       bc1, bc2s, gscale, max_norm, sqnorm, zero_grad);
    LAUNCH_CHECK();
    return EGO_OK;
}
