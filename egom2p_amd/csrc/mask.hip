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
        ids[o] = (long)((splitmix(ki, (uint64_t)i) >> 11) % (uint64_t)vocab);
        in_mask[o] = is_in ? 0 : 1;
        tg_mask[o] = is_tg ? 0 : 1;
        dam[o] = 0;
        if (is_tg) atomicMin(&first, i);
    }
    __syncthreads();
    // k_tgt at the first target position (position 0 when the clip has no target in this modality: value 0 anyway)
    if (threadIdx.x == 0) dam[(long)b * n + (first < n ? first : 0)] = ktg;
}

}  // namespace

extern "C" int ego_clip_synth(const void* key_ids, const void* key_perm, const int* k_in, const int* k_tgt, int B, int n, int vocab,
                              long* ids, void* input_mask, void* target_mask, int* dam, hipStream_t stream) {
    if (B <= 0) return EGO_OK;
    if (n <= 0 || n > SYN_MAX_N || vocab <= 0) return EGO_ERR_ARG;
    EGO_LAUNCH(clip_synth_kernel, dim3(B), dim3(1024), 0, stream, (const uint64_t*)key_ids, (const uint64_t*)key_perm, k_in, k_tgt, n,
               vocab, ids, (unsigned char*)input_mask, (unsigned char*)target_mask, dam);
    LAUNCH_CHECK();
    return EGO_OK;
}
