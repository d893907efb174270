// Shared device helpers for the EgoM2P gfx950 kernels.  CDNA4 only: wave = 64 lanes,
// MFMA bf16 tiles, 160 KiB LDS per CU.  No portability layers on purpose.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef unsigned short bf16_t;                                   // raw bf16 bits in memory
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));       // one MFMA A/B fragment (4 VGPRs)
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));         // 16x16 accumulator
typedef float f32x16 __attribute__((ext_vector_type(16)));       // 32x32 accumulator
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

#define EGO_OK 0
#define EGO_ERR_ARG 1
#define EGO_ERR_LAUNCH 2

// clear any stale (non-sticky) error left by an earlier, unrelated HIP call, then launch
#define EGO_LAUNCH(...)                  \
    do {                                 \
        (void)hipGetLastError();         \
        hipLaunchKernelGGL(__VA_ARGS__); \
    } while (0)

#define LAUNCH_CHECK()                                        \
    do {                                                      \
        hipError_t e__ = hipGetLastError();                   \
        if (e__ != hipSuccess) return EGO_ERR_LAUNCH;         \
    } while (0)

__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((unsigned)v) << 16); }

// round-to-nearest-even; a plain cast lowers to v_cvt_pk_bf16_f32 on gfx950 and keeps NaNs NaN
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
    __bf16 b = (__bf16)f;
    return __builtin_bit_cast(bf16_t, b);
}
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
    return (unsigned)f32_to_bf16(lo) | ((unsigned)f32_to_bf16(hi) << 16);
}
__device__ __forceinline__ float round_bf16(float f) { return bf16_to_f32(f32_to_bf16(f)); }
__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + __expf(-x)); }

// Wave-wide reductions on the VALU: four DPP steps (quad swaps, half-row mirror, row mirror) leave every lane of a 16-lane
// row with the row's result, four v_readlane + scalar-operand ops join the rows.  (__shfl_xor compiles to ds_bpermute_b32:
// six dependent LDS round trips per reduction, each behind an lgkmcnt(0) - on the critical path of every LayerNorm row and of
// the attention kernels' prologues.)  The result is wave-uniform.
#define EGO_DPP_QUAD_X1 0xB1       // quad_perm [1,0,3,2]
#define EGO_DPP_QUAD_X2 0x4E       // quad_perm [2,3,0,1]
#define EGO_DPP_HALF_MIRROR 0x141  // row_half_mirror
#define EGO_DPP_MIRROR 0x140       // row_mirror
#define EGO_DPP_F(v, ctrl) __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(v), ctrl, 0xF, 0xF, true))
#define EGO_RL_F(v, l) __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), l))
__device__ __forceinline__ float wave_sum(float v) {
    v += EGO_DPP_F(v, EGO_DPP_QUAD_X1);
    v += EGO_DPP_F(v, EGO_DPP_QUAD_X2);
    v += EGO_DPP_F(v, EGO_DPP_HALF_MIRROR);
    v += EGO_DPP_F(v, EGO_DPP_MIRROR);
    return (EGO_RL_F(v, 0) + EGO_RL_F(v, 16)) + (EGO_RL_F(v, 32) + EGO_RL_F(v, 48));
}
__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, EGO_DPP_F(v, EGO_DPP_QUAD_X1));
    v = fmaxf(v, EGO_DPP_F(v, EGO_DPP_QUAD_X2));
    v = fmaxf(v, EGO_DPP_F(v, EGO_DPP_HALF_MIRROR));
    v = fmaxf(v, EGO_DPP_F(v, EGO_DPP_MIRROR));
    return fmaxf(fmaxf(EGO_RL_F(v, 0), EGO_RL_F(v, 16)), fmaxf(EGO_RL_F(v, 32), EGO_RL_F(v, 48)));
}
// (the integer reductions stay on __shfl_xor: they only run in kernel prologues, and with wave-uniform DPP / readlane results
// hipcc's register allocation of the attention forward kernel went from 144 VGPRs to 168 + 96 B of scratch)
__device__ __forceinline__ int wave_min_i(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ int wave_max_i(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o, 64));
    return v;
}

// max of a value with the one held by lane ^ 32 (the other half-wave), on the VALU: v_permlane32_swap exchanges the upper
// half of its first operand with the lower half of its second, so {r0, r1} = {own, other} in lanes 0-31 and {other, own} in
// lanes 32-63.  (__shfl_xor(x, 32) compiles to ds_bpermute_b32: an LDS round trip plus an lgkmcnt(0) that also waits for
// every LDS read the wave has in flight.)
__device__ __forceinline__ float xhalf_max(float x) {
    const unsigned u = __float_as_uint(x);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float xhalf_sum(float x) {
    const unsigned u = __float_as_uint(x);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// LDS transposed read: per 16-lane group a 4(row) x 16(col) block of 16-bit elements; lane i of
// the group receives column i of the 4 rows.  Lane 4q+p supplies the address of row q, cols 4p..4p+3.
__device__ __forceinline__ s16x4 lds_read_tr16(const void* lds_addr) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds_addr));
}

// workgroup barrier that orders LDS traffic only: __syncthreads() also drains vmcnt (its fence covers global
// memory), which would wait for every in-flight LDS-DMA / global store instead of the counted ones
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// ds_read_b64_tr_b16 as inline asm: hipcc does not see an LDS read, so it neither drains in-flight LDS-DMA in front
// of it (it does for the builtin) nor tracks its completion - the caller waits (s_waitcnt lgkmcnt) before the use,
// through an asm statement that names the results so that no consumer (or register copy) can be scheduled above it.
template <int OFF>
__device__ __forceinline__ s16x4 tr_read(unsigned lds_addr) {
    s16x4 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(lds_addr), "n"(OFF) : "memory");
    return v;
}
__device__ __forceinline__ bf16x8 join8(s16x4 lo, s16x4 hi) {
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}
template <int N>   // wait until at most N LDS operations are outstanding; the 8 listed results are final afterwards
__device__ __forceinline__ void lgkm_wait_tied(s16x4 (&v)[4][2]) {
    asm volatile("s_waitcnt lgkmcnt(%8)"
                 : "+v"(v[0][0]), "+v"(v[0][1]), "+v"(v[1][0]), "+v"(v[1][1]), "+v"(v[2][0]), "+v"(v[2][1]), "+v"(v[3][0]), "+v"(v[3][1])
                 : "n"(N) : "memory");
}

// 16 B per lane straight from global memory into LDS (LDS-DMA): dst = wave-uniform base + lane * 16.
// Counts in vmcnt; any swizzle has to be applied on the SOURCE address.
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

// XCD-aware block remap (bijective for any grid): blocks b, b+8 share an XCD under round-robin
// dispatch, so give each XCD a contiguous run of tile ids (speed only, never correctness).
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, x = bid & 7, i = bid >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
}

// Attention kernels (attention.hip, attention_hd.hip): workgroup id -> ((batch, head) pair, tile).  The hardware deals consecutive workgroup ids round-robin over the 8
// XCDs and the resident workgroups are a contiguous id range.  B * H is a multiple of 8 for the model's shapes, so
// both orders below keep every tile of one pair on ONE XCD (with the tile index fastest over the whole grid all 8
// L2s re-fetched each K / V stream: 2x slower on the block-diagonal decoder mask).
//   * walk = true (one interval per batch row: encoder self-attention, cross-attention - every tile costs the
//     same): consecutive ids of an XCD walk the tiles of ONE pair, so only 96 / tiles pairs are alive per L2 and
//     their K / V (Q / dO) streams stay in its 4 MB.  rocprofv3 FETCH_SIZE at B 32: 2.55 GB -> 0.35 GB per forward
//     launch, 6-7 % less time (the streams came from HBM: qkv of one micro-batch is 300 MB, beyond the 256 MB MALL).
//   * walk = false (per-row intervals, block-diagonal decoder mask: tile costs differ up to 2x): pair index
//     fastest; the tile walk ran 5-25 % slower there.
__device__ __forceinline__ void pair_tile(int id, int pairs, int tiles, bool walk, int& pair, int& tile) {
    if (walk && (pairs & 7) == 0) {
        const int x = id & 7, j = id >> 3;
        pair = (j / tiles) * 8 + x;
        tile = j % tiles;
    } else {
        pair = id % pairs;
        tile = id / pairs;
    }
}

// Ordered column sums of per-workgroup partial rows (rowops.hip): dst.p[c / seg][c % seg] += sum_r parts[r][c], rows summed
// in index order, so gradients reduced this way are bitwise reproducible (no float atomics).  `parts` must have room for
// colsum_work_floats(n, W) floats (the partial rows followed by the intermediate levels).
#ifndef EGO_MAX_MODS
#define EGO_MAX_MODS 8
#endif
constexpr int COLSUM_MAX_DST = 32;      // gradient vectors behind one slab of partial rows (embedding: modalities + 1; fused LayerNorm: layers)
struct ColsumDst { float* p[COLSUM_MAX_DST]; int seg; };
int colsum_launch(float* parts, long n, int W, const ColsumDst& dst, hipStream_t stream);
long colsum_work_floats(long n, int W);
