// On-device synthesis of the `mod_dict` input contract: token ids, input / target
// masks and the decoder attention-mask marker of one modality for B clips, bit-identical to the host generator
// `egom2p_amd/synth.py:make_clip_batch` (which the goldens were made with), so training and bench steps can draw fresh
// clips without a dataloader or a host-to-device copy.
//
// The reference builds the same structure on CPU workers (`egom2p/data/masking.py:236-266`): per modality a random
// permutation of the positions, the first k_in of it become encoder inputs (mask False), the next k_tgt decoder
// targets, and `decoder_attention_mask` carries k_tgt at the first target position.  Here the permutation is the
// stable argsort of a counter-based hash (splitmix64 of key + i * golden), so rank[i] - the position of i in that
// order - decides the masks; ranks come from an all-pairs count over the hashes held in LDS (n <= 8192, one
// workgroup per (clip, modality): 26 M compares for a video modality, microseconds on 1024 lanes).
#include "common.h"
#include "egom2p_hip.h"
#include <stdint.h>

namespace {

constexpr int SYN_MAX_N = 8192;

__device__ __forceinline__ uint64_t splitmix(uint64_t key, uint64_t i) {
    uint64_t x = i * 0x9E3779B97F4A7C15ull + key;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

__global__ __launch_bounds__(1024) void clip_synth_kernel(const uint64_t* __restrict__ key_ids, const uint64_t* __restrict__ key_perm,
                                                          const int* __restrict__ k_in, const int* __restrict__ k_tgt, int n, int vocab,
                                                          long* __restrict__ ids, unsigned char* __restrict__ in_mask,
                                                          unsigned char* __restrict__ tg_mask, int* __restrict__ dam) {
    __shared__ uint64_t h[SYN_MAX_N];
    __shared__ int first;
    const int b = blockIdx.x;
    const uint64_t kp = key_perm[b], ki = key_ids[b];
    const int kin = k_in[b], ktg = k_tgt[b];
    if (threadIdx.x == 0) first = n;
    for (int i = threadIdx.x; i < n; i += blockDim.x) h[i] = splitmix(kp, (uint64_t)i);
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const uint64_t hi = h[i];
        int rank = 0;                                   // stable argsort position of i
        for (int j = 0; j < n; ++j) {
            const uint64_t hj = h[j];
            rank += (hj < hi) || (hj == hi && j < i);
        }
        const bool is_in = rank < kin, is_tg = rank >= kin && rank < kin + ktg;
        const long o = (long)b * n + i;
        if (ids) ids[o] = (long)((splitmix(ki, (uint64_t)i) >> 11) % (uint64_t)vocab);
        in_mask[o] = is_in ? 0 : 1;
        tg_mask[o] = is_tg ? 0 : 1;
        dam[o] = 0;
        if (is_tg) atomicMin(&first, i);
    }
    __syncthreads();
    // k_tgt at the first target position (position 0 when the clip has no target in this modality: value 0 anyway)
    if (threadIdx.x == 0) dam[(long)b * n + (first < n ? first : 0)] = ktg;
}

// ---------------------------------------------------------------------------------------------
// Token budgets: `UnifiedMasking.input_token_budget` / `target_token_budget` of the
// reference (egom2p/data/masking.py:181-234, called at :530-541) on the device - one thread per clip:
//   mixture component ~ multinomial(sampling_weights); N_in, N_tgt ~ randint(range)           (:530-536)
//   budget = floor(Dirichlet(alpha) * N); the N - sum(budget) leftover tokens go, one each, to the arg-max modality of
//   one more Dirichlet draw (so modalities with alpha ~ 0 are not topped up); clamp to max_tokens (targets: to what
//   the inputs left over, for the non-sequence modalities); redraw while a modality is below min_tokens (<= max_tries).
// Randomness: a counter-based stream per clip (splitmix64 of the clip key and a running counter) - Gamma variates by
// Marsaglia-Tsang (alpha < 1 boosted with U^(1/alpha)), kept in the log domain so alpha = 0.01 does not underflow.
// ---------------------------------------------------------------------------------------------
struct BudgetArgs {
    ego_budget_desc d;
    const uint64_t* keys;
    int B;
    int* k_in; int* k_tgt;      // [n_mods][B]
};

struct Rng {
    uint64_t key, ctr;
    __device__ float uni() {                               // (0, 1)
        const uint64_t x = splitmix(key, ctr++);
        return ((float)(x >> 40) + 0.5f) * (1.0f / 16777216.0f);
    }
    __device__ float normal() {
        const float u1 = uni(), u2 = uni();
        return sqrtf(-2.f * logf(u1)) * cosf(6.28318530718f * u2);
    }
    // log of a Gamma(alpha, 1) variate
    __device__ float log_gamma(float alpha) {
        const float a = alpha < 1.f ? alpha + 1.f : alpha;
        const float dd = a - 1.f / 3.f, c = 1.f / sqrtf(9.f * dd);
        float g = dd;
        for (int it = 0; it < 64; ++it) {
            const float x = normal(), t = 1.f + c * x;
            if (t <= 0.f) continue;
            const float v = t * t * t, u = uni();
            if (logf(u) < 0.5f * x * x + dd - dd * v + dd * logf(v)) { g = dd * v; break; }
        }
        float lg = logf(g);
        if (alpha < 1.f) lg += logf(uni()) / alpha;
        return lg;
    }
    __device__ void dirichlet(const float* alpha, int n, float* p) {
        float lg[EGO_MAX_MODS], mx = -3.0e38f, sum = 0.f;
        for (int i = 0; i < n; ++i) { lg[i] = log_gamma(alpha[i]); mx = fmaxf(mx, lg[i]); }
        for (int i = 0; i < n; ++i) { p[i] = expf(lg[i] - mx); sum += p[i]; }
        // torch's Dirichlet.sample() (masking.py:192) clamps every float32 component to [FLT_MIN, 1 - 2^-24] (ATen
        // _s_dirichlet): a one-hot draw is never exactly 1, floor(p * N) then leaves a token over - part of the reference's law
        for (int i = 0; i < n; ++i) p[i] = fminf(fmaxf(p[i] / sum, 1.17549435e-38f), 0.99999994f);
    }
};

__device__ void draw_budget(Rng& r, const float* alpha, int n, int total, const int* cap, const int* min_tokens, int max_tries, int* out) {
    for (int tr = 0; tr < max_tries; ++tr) {
        float p[EGO_MAX_MODS];
        r.dirichlet(alpha, n, p);
        int sum = 0;
        for (int i = 0; i < n; ++i) { out[i] = (int)floorf(p[i] * (float)total); sum += out[i]; }
        for (int left = total - sum; left > 0; --left) {                      // masking.py:193-196
            r.dirichlet(alpha, n, p);
            int am = 0;
            for (int i = 1; i < n; ++i) if (p[i] > p[am]) am = i;
            out[am] += 1;
        }
        bool ok = true;
        for (int i = 0; i < n; ++i) { out[i] = min(out[i], cap[i]); ok &= out[i] >= min_tokens[i]; }
        if (ok) return;
    }
}

__global__ void budget_kernel(BudgetArgs a) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= a.B) return;
    const ego_budget_desc& d = a.d;
    Rng r{a.keys[b], 0};
    // mixture component (:530) and the token counts (:535-536: random.randint is inclusive on both ends)
    float wsum = 0.f;
    for (int j = 0; j < d.n_mix; ++j) wsum += d.mix_weight[j];
    float u = r.uni() * wsum;
    int mix = d.n_mix - 1;
    for (int j = 0; j < d.n_mix; ++j) { if (u < d.mix_weight[j]) { mix = j; break; } u -= d.mix_weight[j]; }
    const int n_in = d.n_in_lo + (int)(r.uni() * (float)(d.n_in_hi - d.n_in_lo + 1));
    const int n_tg = d.n_tgt_lo + (int)(r.uni() * (float)(d.n_tgt_hi - d.n_tgt_lo + 1));
    int kin[EGO_MAX_MODS], ktg[EGO_MAX_MODS], rem[EGO_MAX_MODS];
    draw_budget(r, d.in_alpha[mix], d.n_mods, min(n_in, d.n_in_hi), d.max_tokens, d.min_tokens, d.max_tries, kin);
    for (int i = 0; i < d.n_mods; ++i)                                        // masking.py:217-218
        rem[i] = max(d.min_tokens[i], d.not_seq[i] ? d.max_tokens[i] - kin[i] : d.max_tokens[i]);
    draw_budget(r, d.tgt_alpha[mix], d.n_mods, min(n_tg, d.n_tgt_hi), rem, d.min_tokens, d.max_tries, ktg);
    for (int i = 0; i < d.n_mods; ++i) { a.k_in[i * a.B + b] = kin[i]; a.k_tgt[i * a.B + b] = ktg[i]; }
}

}  // namespace

extern "C" int ego_budget_dirichlet(const ego_budget_desc* d, const void* clip_keys, int B, int* k_in, int* k_tgt, hipStream_t stream) {
    if (B <= 0) return EGO_OK;
    if (!d || d->n_mods <= 0 || d->n_mods > EGO_MAX_MODS || d->n_mix <= 0 || d->n_mix > EGO_MAX_MIX || d->max_tries <= 0 ||
        d->n_in_lo > d->n_in_hi || d->n_tgt_lo > d->n_tgt_hi || d->n_in_lo < 0 || d->n_tgt_lo < 0)
        return EGO_ERR_ARG;
    for (int j = 0; j < d->n_mix; ++j)
        for (int i = 0; i < d->n_mods; ++i)
            if (!(d->in_alpha[j][i] > 0.f) || !(d->tgt_alpha[j][i] > 0.f)) return EGO_ERR_ARG;
    BudgetArgs a{*d, (const uint64_t*)clip_keys, B, k_in, k_tgt};
    EGO_LAUNCH(budget_kernel, dim3((B + 63) / 64), dim3(64), 0, stream, a);
    LAUNCH_CHECK();
    return EGO_OK;
}

extern "C" int ego_clip_synth(const void* key_ids, const void* key_perm, const int* k_in, const int* k_tgt, int B, int n, int vocab,
                              long* ids, void* input_mask, void* target_mask, int* dam, hipStream_t stream) {
    if (B <= 0) return EGO_OK;
    if (n <= 0 || n > SYN_MAX_N || (ids && vocab <= 0)) return EGO_ERR_ARG;
    EGO_LAUNCH(clip_synth_kernel, dim3(B), dim3(1024), 0, stream, (const uint64_t*)key_ids, (const uint64_t*)key_perm, k_in, k_tgt, n,
               vocab, ids, (unsigned char*)input_mask, (unsigned char*)target_mask, dam);
    LAUNCH_CHECK();
    return EGO_OK;
}
