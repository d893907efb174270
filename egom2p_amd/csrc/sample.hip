// Token sampling for the ROAR / classifier-free-guidance generation path (gfx950).
//
// Replaces, per decoded row, the chain of the reference's GenerationSampler
// (egom2p/models/generate.py): CFG mix `uncond + (cond - uncond) * s` (:805), nucleus filtering
// `top_k_top_p_filtering` (:332-359, a full descending sort + cumsum + argsort + gather over V = 64000),
// `softmax(filtered / temperature)` and `torch.multinomial` (:361-371).
//
// One 1024-thread workgroup per row keeps the whole row in registers (<= 64 logits per lane):
//   * nucleus set without sorting: token j is kept iff the probability mass of strictly larger logits
//     is <= top_p (the reference keeps sorted tokens while the cumulative mass *before* them is <= top_p);
//     the cut is found by a 32-step binary search on the order-preserving integer image of the logits.
//     Tokens with exactly equal logits are kept or dropped together (the reference's sort breaks such
//     ties arbitrarily) - the only semantic difference.
//   * sample by inverse CDF of softmax(kept / T) with a caller-supplied uniform number per row
//     (explicit RNG, so a captured graph replays deterministically).
#include "common.h"
#include "egom2p_hip.h"

namespace {

constexpr int SMP_THREADS = 1024;
constexpr int SMP_MAXE = 64;            // V <= 65536

__device__ __forceinline__ unsigned f2key(float f) {      // order-preserving float -> uint
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// block-wide sum / max over 1024 threads (16 waves); result broadcast to all threads
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < SMP_THREADS / 64; ++i) t += red[i];
    return t;
}
__device__ __forceinline__ float block_max(float v, float* red) {
    v = wave_max(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = red[0];
#pragma unroll
    for (int i = 1; i < SMP_THREADS / 64; ++i) t = fmaxf(t, red[i]);
    return t;
}

__global__ __launch_bounds__(SMP_THREADS) void sample_kernel(const bf16_t* __restrict__ cond, const bf16_t* __restrict__ uncond,
                                                             long ld, int V, float cfg, float top_p, float temperature,
                                                             const float* __restrict__ uniforms, int* __restrict__ out_tok,
                                                             float* __restrict__ out_prob) {
    __shared__ float red[SMP_THREADS / 64];
    __shared__ float wsum[SMP_THREADS / 64];
    __shared__ int s_pick[2];
    const long row = blockIdx.x;
    const int tid = threadIdx.x;
    const bf16_t* c = cond + row * ld;
    const bf16_t* u = uncond ? uncond + row * ld : nullptr;

    float l[SMP_MAXE];
    float mx = -3.0e38f;
#pragma unroll
    for (int k = 0; k < SMP_MAXE; ++k) {
        const int i = tid + SMP_THREADS * k;
        float v = -3.0e38f;
        if (i < V) {
            const float cv = bf16_to_f32(c[i]);
            // explicit roundings (no fma contraction): bit-identical to torch's uncond + (cond - uncond) * s
            v = u ? __fadd_rn(bf16_to_f32(u[i]), __fmul_rn(__fsub_rn(cv, bf16_to_f32(u[i])), cfg)) : cv;
            mx = fmaxf(mx, v);
        }
        l[k] = v;
    }
    mx = block_max(mx, red);

    // arg-max (lowest index among equal maxima): the answer at temperature 0 and the fallback below
    int amax;
    {
        int best = 0x7fffffff;
#pragma unroll
        for (int k = 0; k < SMP_MAXE; ++k) {
            const int i = tid + SMP_THREADS * k;
            if (i < V && l[k] == mx) best = min(best, i);
        }
        best = wave_min_i(best);
        __syncthreads();
        if ((tid & 63) == 0) ((int*)red)[tid >> 6] = best;
        __syncthreads();
        amax = 0x7fffffff;
#pragma unroll
        for (int i = 0; i < SMP_THREADS / 64; ++i) amax = min(amax, ((int*)red)[i]);
        __syncthreads();
    }
    if (temperature <= 1e-10f) {            // np.isclose(temperature, 0) branch of sample_tokens (:362-365)
        if (tid == 0) { out_tok[row] = amax; if (out_prob) out_prob[row] = 1.f; }
        return;
    }

    // ---- nucleus cut: smallest key K with mass{key > K} <= top_p * Z
    unsigned cut = 0u;                       // keep everything
    if (top_p > 0.f) {
        float z = 0.f;
#pragma unroll
        for (int k = 0; k < SMP_MAXE; ++k) z += (tid + SMP_THREADS * k < V) ? __expf(l[k] - mx) : 0.f;
        z = block_sum(z, red);
        const float budget = top_p * z;
        unsigned lo = 0u, hi = f2key(mx);    // invariant: mass{> hi} <= budget ; answer in [lo, hi]
        for (int it = 0; it < 32 && lo < hi; ++it) {
            const unsigned mid = lo + ((hi - lo) >> 1);
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < SMP_MAXE; ++k)
                s += (tid + SMP_THREADS * k < V && f2key(l[k]) > mid) ? __expf(l[k] - mx) : 0.f;
            s = block_sum(s, red);
            if (s <= budget) hi = mid; else lo = mid + 1;
        }
        cut = hi;
    }

    // ---- sample from softmax(kept / T) by inverse CDF (thread-major element order)
    const float invT = 1.f / temperature;
    float q[SMP_MAXE];
    float mine = 0.f;
#pragma unroll
    for (int k = 0; k < SMP_MAXE; ++k) {
        const bool keep = (tid + SMP_THREADS * k < V) && f2key(l[k]) >= cut;
        q[k] = keep ? __expf((l[k] - mx) * invT) : 0.f;
        mine += q[k];
    }
    // inclusive scan of per-thread masses
    float incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const float t = __shfl_up(incl, o, 64);
        if ((tid & 63) >= o) incl += t;
    }
    __syncthreads();
    if ((tid & 63) == 63) wsum[tid >> 6] = incl;
    __syncthreads();
    float base = 0.f, total = 0.f;
#pragma unroll
    for (int i = 0; i < SMP_THREADS / 64; ++i) {
        if (i < (tid >> 6)) base += wsum[i];
        total += wsum[i];
    }
    const float target = fminf(uniforms[row], 0.99999994f) * total;
    const float hi_c = base + incl, lo_c = hi_c - mine;
    if (tid == 0) { s_pick[0] = -1; s_pick[1] = 0; }
    __syncthreads();
    // the thread whose [lo_c, hi_c) interval contains the target (first such thread wins; mass > 0)
    if (mine > 0.f && target >= lo_c && target < hi_c) {
        float run = lo_c;
        int pick = -1;
        float pq = 0.f;
#pragma unroll
        for (int k = 0; k < SMP_MAXE; ++k) {
            if (pick < 0 && q[k] > 0.f) {
                run += q[k];
                if (target < run || k == SMP_MAXE - 1) { pick = tid + SMP_THREADS * k; pq = q[k]; }
            }
        }
        if (pick < 0) {                       // numerical edge: take this thread's last kept element
#pragma unroll
            for (int k = 0; k < SMP_MAXE; ++k) if (q[k] > 0.f) { pick = tid + SMP_THREADS * k; pq = q[k]; }
        }
        if (atomicCAS(&s_pick[0], -1, pick) == -1) s_pick[1] = __float_as_int(pq / total);
    }
    __syncthreads();
    if (tid == 0) {
        // no interval matched only through float round-off at an interval seam: take the arg-max (always kept)
        const int pick = s_pick[0];
        out_tok[row] = pick >= 0 ? pick : amax;
        if (out_prob) out_prob[row] = pick >= 0 ? __int_as_float(s_pick[1]) : 0.f;
    }
}

}  // namespace

extern "C" int ego_sample_cfg_topp(const void* cond, const void* uncond, long ld, int V, float cfg_scale, float top_p,
                                   float temperature, const float* uniforms, int* out_tokens, float* out_prob, int rows,
                                   hipStream_t stream) {
    if (rows <= 0) return EGO_OK;
    if (V <= 0 || V > SMP_THREADS * SMP_MAXE || !cond || !uniforms || !out_tokens) return EGO_ERR_ARG;
    EGO_LAUNCH(sample_kernel, dim3(rows), dim3(SMP_THREADS), 0, stream, (const bf16_t*)cond, (const bf16_t*)uncond, ld, V,
               cfg_scale, top_p, temperature, uniforms, out_tokens, out_prob);
    LAUNCH_CHECK();
    return EGO_OK;
}
