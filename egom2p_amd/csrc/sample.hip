// Token sampling for the ROAR / classifier-free-guidance generation path (gfx950).
//
// Replaces, per decoded row, the chain of the reference's GenerationSampler
// (egom2p/models/generate.py): CFG mix `uncond + (cond - uncond) * s` (:805), nucleus filtering
// `top_k_top_p_filtering` (:332-359: top-k by torch.topk, then a full descending sort + cumsum + argsort + gather over V = 64000),
// `softmax(filtered / temperature)` and `torch.multinomial` (:361-371).
//
// One 1024-thread workgroup per row.  The row (2 x 128 KB of bf16 logits at V = 64000) is read three times - the first
// time from HBM, then from L2 - by 16-byte loads, so that only ONE float per token has to live in registers (64 per lane):
//   pass 1  mixed logit -> row maximum and arg-max;
//   pass 2  p = exp(mixed - max) kept in registers -> Z, then the nucleus cut WITHOUT sorting: token j is kept iff the
//           probability mass of strictly larger tokens is <= top_p (the reference keeps sorted tokens while the cumulative
//           mass *before* them is <= top_p); the cut is a <= 30-step binary search on the bit pattern of p (p >= 0: the
//           integer order of the bits is the order of the values) whose step costs compare / select / add per token - no
//           exponential inside the search (round 3 recomputed exp() for all 64 tokens of a lane in each of 32 steps and
//           searched on the logit: 2.0 ms per 1707-row launch, 13 % of a rgb -> depth clip at batch 1).  Tokens whose p is
//           equal (equal logits, or logits closer than one fp32 ulp of p) are kept or dropped together - the reference's
//           sort breaks such ties arbitrarily: the only semantic difference;
//   pass 3  q = exp((mixed - max) / T) of the kept tokens overwrites p; the sample is the inverse CDF of q at a
//           caller-supplied uniform number per row (explicit RNG, so a captured graph replays deterministically).
// Element order of the CDF: lane-major (lane t owns chunks t, t + 1024, ... of 8 consecutive tokens).
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

// slot k = 8 m + e of lane t is token 8 (t + 1024 m) + e (VEC: rows 16-byte aligned, V % 8 == 0) or token t + 1024 k
template <bool VEC>
__device__ __forceinline__ int slot_token(int tid, int k) { return VEC ? 8 * (tid + SMP_THREADS * (k >> 3)) + (k & 7) : tid + SMP_THREADS * k; }

// mixed logits of chunk m (slots 8m .. 8m+7) of this lane; tokens beyond V come out as -3e38.  Branch-free: a chunk beyond V
// loads the row's last chunk and is overwritten afterwards (divergent branches around the loads cost hipcc ~300 spilled SGPRs).
template <bool VEC, bool CFG>
__device__ __forceinline__ void load_chunk(const bf16_t* __restrict__ c, const bf16_t* __restrict__ u, int V, float cfg, int tid, int m,
                                           float (&v)[8]) {
    if (VEC) {
        const int i0 = 8 * (tid + SMP_THREADS * m);
        const int j0 = min(i0, V - 8);
        const u32x4 cw = *(const u32x4*)(c + j0);
        u32x4 uw = {0u, 0u, 0u, 0u};
        if (CFG) uw = *(const u32x4*)(u + j0);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float cv = __uint_as_float((e & 1) ? (cw[e >> 1] & 0xffff0000u) : (cw[e >> 1] << 16));
            const float uv = __uint_as_float((e & 1) ? (uw[e >> 1] & 0xffff0000u) : (uw[e >> 1] << 16));
            // explicit roundings (no fma contraction): bit-identical to torch's uncond + (cond - uncond) * s
            const float x = CFG ? __fadd_rn(uv, __fmul_rn(__fsub_rn(cv, uv), cfg)) : cv;
            v[e] = (i0 < V) ? x : -3.0e38f;
        }
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int i = tid + SMP_THREADS * (8 * m + e);
            const int j = min(i, V - 1);
            const float cv = bf16_to_f32(c[j]);
            const float uv = CFG ? bf16_to_f32(u[j]) : 0.f;
            const float x = CFG ? __fadd_rn(uv, __fmul_rn(__fsub_rn(cv, uv), cfg)) : cv;
            v[e] = (i < V) ? x : -3.0e38f;
        }
    }
}

// TOPK: its own instantiations - the search loop and its compares cost the top-p-only kernel (config 4) 100 spilled registers
// when they share one (0.49 -> 1.22 ms per 1707-row launch, rocprofv3 profiles/r05_eval_rgb2depth_kernel_stats.csv history)
template <bool VEC, bool CFG, bool TOPK>
__global__ __launch_bounds__(SMP_THREADS) void sample_kernel(const bf16_t* __restrict__ cond, const bf16_t* __restrict__ uncond,
                                                             long ld, int V, float cfg, float top_p, int top_k, float temperature,
                                                             const float* __restrict__ uniforms, int* __restrict__ out_tok,
                                                             float* __restrict__ out_prob) {
    __shared__ float red[SMP_THREADS / 64];
    __shared__ float wsum[SMP_THREADS / 64];
    __shared__ int s_pick[2];
    const long row = blockIdx.x;
    const int tid = threadIdx.x;
    const bf16_t* c = cond + row * ld;
    const bf16_t* u = CFG ? uncond + row * ld : nullptr;

    // ---- pass 1: maximum and arg-max (lowest index among equal maxima: the answer at temperature 0, the fallback below)
    float mx = -3.0e38f;
    int best = 0x7fffffff;
#pragma unroll
    for (int m = 0; m < SMP_MAXE / 8; ++m) {
        float v[8];
        load_chunk<VEC, CFG>(c, u, V, cfg, tid, m, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            // (slots are visited in increasing token order: strict > keeps the lowest index; a slot beyond V holds -3e38)
            if (v[e] > mx) { mx = v[e]; best = slot_token<VEC>(tid, 8 * m + e); }
        }
    }
    const float lane_mx = mx;
    mx = block_max(mx, red);
    int amax;
    {
        best = (lane_mx == mx) ? best : 0x7fffffff;
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

    float p[SMP_MAXE];
    // ---- top-k (top_k_top_p_filtering, generate.py:335-345): remove every token whose mixed logit is below the k-th largest one
    // (`logits < topk(logits, k)[0][..., -1]`: tokens that tie with the k-th stay).  The k-th largest is found without sorting by a
    // 32-step binary search on the order-preserving integer key of the logit: the largest key T with count{key >= T} >= k; the
    // registers that hold p afterwards hold the mixed logits during the search.  kth = 0 keeps everything.
    unsigned kth = 0u;
    if constexpr (TOPK) {
#pragma unroll
        for (int m = 0; m < SMP_MAXE / 8; ++m) {
            float v[8];
            load_chunk<VEC, CFG>(c, u, V, cfg, tid, m, v);
#pragma unroll
            for (int e = 0; e < 8; ++e) p[8 * m + e] = v[e];    // (a slot beyond V holds -3e38: below every real logit)
        }
        unsigned lo = 0u, hi = 0xffffffffu;                     // invariant: count{key >= lo} >= k
        while (lo < hi) {
            const unsigned mid = lo + ((hi - lo) >> 1) + ((hi - lo) & 1u);     // upper middle: the loop ends at the LARGEST such key
            float c0 = 0.f, c1 = 0.f;                           // counts <= 65536: exact in fp32
#pragma unroll
            for (int k = 0; k < SMP_MAXE; k += 2) {
                c0 += (f2key(p[k]) >= mid) ? 1.f : 0.f;
                c1 += (f2key(p[k + 1]) >= mid) ? 1.f : 0.f;
            }
            const float cnt = block_sum(c0 + c1, red);
            if (cnt >= (float)top_k) lo = mid; else hi = mid - 1u;
        }
        kth = lo;
    }

    // ---- pass 2: p = exp(mixed - max) (0 beyond V and for tokens the top-k filter removed: the nucleus is taken over the
    // renormalised survivors, as softmax of the filtered logits does) and the nucleus cut: smallest bit pattern K with
    // mass{p > K} <= top_p * Z
    unsigned cut = 0u;                       // keep everything
    if (top_p > 0.f) {
        float z = 0.f;
#pragma unroll
        for (int m = 0; m < SMP_MAXE / 8; ++m) {
            float v[8];
            load_chunk<VEC, CFG>(c, u, V, cfg, tid, m, v);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                p[8 * m + e] = (!TOPK || f2key(v[e]) >= kth) ? __expf(v[e] - mx) : 0.f;       // a slot beyond V: exp(-3e38 - max) = 0
                z += p[8 * m + e];
            }
        }
        z = block_sum(z, red);
        const float budget = top_p * z;
        unsigned lo = 0u, hi = __float_as_uint(1.0f);    // invariant: mass{p > hi} <= budget ; answer in [lo, hi]
        while (lo < hi) {
            const unsigned mid = lo + ((hi - lo) >> 1);
            const float t = __uint_as_float(mid);
            float s0 = 0.f, s1 = 0.f;
#pragma unroll
            for (int k = 0; k < SMP_MAXE; k += 2) {
                s0 += (p[k] > t) ? p[k] : 0.f;
                s1 += (p[k + 1] > t) ? p[k + 1] : 0.f;
            }
            const float s = block_sum(s0 + s1, red);
            if (s <= budget) hi = mid; else lo = mid + 1;
        }
        cut = hi;
    }

    // ---- pass 3: q = exp((mixed - max) / T) of the kept tokens (in p's registers); sample by inverse CDF, lane-major order
    const float invT = 1.f / temperature;
    float mine = 0.f;
#pragma unroll
    for (int m = 0; m < SMP_MAXE / 8; ++m) {
        float v[8];
        load_chunk<VEC, CFG>(c, u, V, cfg, tid, m, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = 8 * m + e;
            const bool keep = (cut == 0u || __float_as_uint(p[k]) >= cut) && (!TOPK || f2key(v[e]) >= kth);   // (a slot beyond V: q = exp(-inf) = 0 either way)
            p[k] = keep ? __expf((v[e] - mx) * invT) : 0.f;
            mine += p[k];
        }
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
            if (pick < 0 && p[k] > 0.f) {
                run += p[k];
                if (target < run || k == SMP_MAXE - 1) { pick = slot_token<VEC>(tid, k); pq = p[k]; }
            }
        }
        if (pick < 0) {                       // numerical edge: take this thread's last kept element
#pragma unroll
            for (int k = 0; k < SMP_MAXE; ++k) if (p[k] > 0.f) { pick = slot_token<VEC>(tid, k); pq = p[k]; }
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

extern "C" int ego_sample_cfg_topp(const void* cond, const void* uncond, long ld, int V, float cfg_scale, float top_p, int top_k,
                                   float temperature, const float* uniforms, int* out_tokens, float* out_prob, int rows,
                                   hipStream_t stream) {
    if (rows <= 0) return EGO_OK;
    if (V <= 0 || V > SMP_THREADS * SMP_MAXE || !cond || !uniforms || !out_tokens || top_k < 0) return EGO_ERR_ARG;
    const bool vec = V % 8 == 0 && V >= 8 && ld % 8 == 0 && ((((uintptr_t)cond) | ((uintptr_t)uncond)) & 15) == 0;
    const bool topk = top_k > 0 && top_k < V;            // k >= V removes nothing
#define SMP_GO(VEC, CFG, TOPK)                                                                                                   \
    EGO_LAUNCH((sample_kernel<VEC, CFG, TOPK>), dim3(rows), dim3(SMP_THREADS), 0, stream, (const bf16_t*)cond, (const bf16_t*)uncond, ld, V, \
               cfg_scale, top_p, top_k, temperature, uniforms, out_tokens, out_prob)
#define SMP_GO2(VEC, CFG) do { if (topk) { SMP_GO(VEC, CFG, true); } else { SMP_GO(VEC, CFG, false); } } while (0)
    if (vec && uncond) SMP_GO2(true, true);
    else if (vec) SMP_GO2(true, false);
    else if (uncond) SMP_GO2(false, true);
    else SMP_GO2(false, false);
#undef SMP_GO2
#undef SMP_GO
    LAUNCH_CHECK();
    return EGO_OK;
}
