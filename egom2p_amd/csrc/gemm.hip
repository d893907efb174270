// bf16 MFMA GEMMs for the EgoM2P hot path (gfx950).
//
//   gemm_nt : C[M,N] = A[M,K] . B[N,K]^T        forward linears (B = nn.Linear weight [out,in]) and
//                                               dgrad (B = pre-transposed weight W^T [in,out])
//   gemm_tn : C[Ni,Nj] += P[M,Ni]^T . Q[M,Nj]   wgrad (P = dY, Q = layer input), contraction over rows
//
// Both: 128x128 output tile per 256-thread workgroup (4 waves, 2x2, 64x64 per wave as 4x4
// v_mfma_f32_16x16x32_bf16 tiles), 64-deep contraction steps, two LDS stages filled by LDS-DMA
// (global_load_lds_dwordx4: no VGPR round trip, no ds_write), the load of step t+1 in flight under the
// MFMAs of step t, one barrier per step.  LDS images are XOR-swizzled - on the DMA *source* address,
// because the DMA destination is lane-linear - so that the ds_read_b128 row reads (nt) and the
// ds_read_b64_tr_b16 transposed reads (tn) are bank-conflict free.  The accumulator is produced
// transposed (operands swapped in the MFMA) so each lane owns 4 consecutive output columns.
//
// Replaces: torch.nn.functional.linear under autocast(bf16) at every call site of
// egom2p/models/egom2p_utils.py:141-169,180-203,215-242 and decoder_embeddings.py:372-383,489-500,
// and their autograd backward (mm dgrad / wgrad).
#include "common.h"
#include "egom2p_hip.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;        // one operand tile (bf16) = 16 KiB
constexpr int STAGE_BYTES = 2 * TILE_BYTES;    // A tile + B tile = 32 KiB
constexpr int GEMM_LDS = 2 * STAGE_BYTES;      // 64 KiB (tn kernel)
constexpr int EPI_BYTES = 16 * 1024;           // nt kernel: epilogue scratch behind the two stages
constexpr int NT_LDS = GEMM_LDS + EPI_BYTES;   // 80 KiB -> exactly 2 workgroups per CU (160 KiB)

struct NTArgs {
    const bf16_t* A; long lda;
    const bf16_t* B; long ldb;
    void* C; long ldc;
    const float* R; long ldr;
    const float* bias;
    const int* m_range;
    int M, N, K, epi;
    const bf16_t* X; long ldx;      // EPI_SWIGLU_BWD: the saved SwiGLU pre-activations ab [M, 2N]
    bf16_t* H; long ldh;            // EPI_SWIGLU_FWD: the gate output h [M, N]
    const float* sa; const float* sb;   // fp8 operands: per-row scales of A [M] and of B [N rows] (C = sa[m] * sb[n] * acc)
    int dephase;                        // timing probe (ego_gemm_tune key 1): every other workgroup of an XCD starts this many ~0.5 us late
};
constexpr int EPI_SWIGLU_FWD = 101;   // internal: C = ab [M, 2N] and H = bf16(bf16(silu(a)) * b) [M, N] from one 256 x (128 a + 128 b) tile
constexpr int EPI_SWIGLU_BWD = 100;   // internal: C(bf16)[M, 2N] = SwiGLU backward of (acc rounded to bf16) against X

// ---------------------------------------------------------------------------------------------
// NT kernel, persistent.  Each workgroup walks a strided list of 128x128 output tiles; the K loop runs
// as ONE pipeline across tile boundaries (the first K-step of the next tile is loaded under the last
// K-step of the current one), so only the very first load of a workgroup is exposed.  That matters
// here: most of the model's GEMMs have K = 768 = 12 steps.  Tile ownership is XCD-aware: the 64
// workgroups that share an XCD's L2 work on 64 consecutive tiles (same A row panel / neighbouring B
// panels) - speed only, never correctness.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(NTArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;

    int M = p.M;
    long moff = 0;
    if (p.m_range) { moff = p.m_range[0]; M = min(M, p.m_range[1]); }
    if (M <= 0) return;
    const int tiles_n = (p.N + BN - 1) / BN;
    const int ntiles = ((M + BM - 1) / BM) * tiles_n;       // tiles that exist (M may come from the device)
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int nslot = (gridDim.x + 7 - xcd) >> 3;           // workgroups with this blockIdx & 7
    const int per_xcd = (ntiles + 7) >> 3;
    const int t_hi = min(ntiles, (xcd + 1) * per_xcd);
    int tile = xcd * per_xcd + slot;
    if (tile >= t_hi) return;
    const int nt = p.K / BK;

    // Staging map: wave instruction (wave*4 + j) writes 1 KiB = tile rows 8(wave*4+j) .. +7, 128 B each,
    // lane-linear; slot s of row r holds global chunk s ^ (r & 7).  Rows beyond M / N are clamped to the
    // last valid row: they only feed output rows / columns that are never stored.
    int lr[4], lc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        lr[j] = 8 * (wave * 4 + j) + (lane >> 3);
        lc[j] = ((lane & 7) ^ (lr[j] & 7)) * 8;
    }
    auto stage = [&](int s, int tl, int k0) {
        const int row0 = (tl / tiles_n) * BM, col0 = (tl % tiles_n) * BN;
        char* sa = smem + s * STAGE_BYTES + wave * 4096;
        char* sb = sa + TILE_BYTES;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            glds16(p.A + (moff + min(row0 + lr[j], M - 1)) * p.lda + lc[j] + k0, sa + j * 1024);
            glds16(p.B + (long)min(col0 + lr[j], p.N - 1) * p.ldb + lc[j] + k0, sb + j * 1024);
        }
    };

    stage(0, tile, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_barrier();
    int cur = 0;
    bool pre = false;          // K-step 1 of this tile was already issued behind the previous tile's last barrier
    bool pre_counted = false;  // ... and exactly 8 younger store instructions of this wave sit behind it
    while (true) {
        f32x4 acc[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int next_tile = tile + nslot;
        const bool more = next_tile < t_hi;
        for (int kt = 0; kt < nt; ++kt) {
            if (kt == 0 && pre) { /* in flight */ }
            else if (kt + 1 < nt) stage(cur ^ 1, tile, (kt + 1) * BK);
            else if (more) stage(cur ^ 1, next_tile, 0);
            const char* sa = smem + cur * STAGE_BYTES;
            const char* sb = sa + TILE_BYTES;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bf16x8 af[4], bfr[4];
                const int c = ks * 4 + (lane >> 4);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int ar = wm * 64 + i * 16 + (lane & 15);
                    af[i] = *(const bf16x8*)(sa + ar * 128 + ((c ^ (ar & 7)) << 4));
                    const int br = wn * 64 + i * 16 + (lane & 15);
                    bfr[i] = *(const bf16x8*)(sb + br * 128 + ((c ^ (br & 7)) << 4));
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        // operands swapped: D[row = n (4 regs)][col = m (lane&15)]
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
            }
            // the stage issued for the next step must have landed.  vmcnt is in issue order and counts stores
            // too: after a prefetched step the 8 epilogue stores of this wave are YOUNGER than the DMA we need,
            // so they may stay in flight (their completion latency was the largest epilogue cost).
            if (kt == 0 && pre_counted) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            lds_barrier();
            cur ^= 1;
        }
        pre = false;
        pre_counted = false;

        // Epilogue through LDS.  The next tile's first K-step already sits in stage `cur`; stage `cur ^ 1` is
        // free: the accumulators are transposed through it so that every wave store instruction writes whole
        // contiguous rows (4 rows x 256 B) instead of 16 scattered 32-byte pieces - on the K = 768 shapes the
        // scattered form cost ~30 % of the kernel.  16-byte chunk c of tile row r sits at chunk c ^ (r & 15).
        const int row0 = (tile / tiles_n) * BM, col0 = (tile % tiles_n) * BN;
        if (p.epi == EGO_EPI_BF16) {
            // stage `cur` holds the next tile's K-step 0; put its K-step 1 in flight NOW (stage cur^1 is free)
            // and run the epilogue through the dedicated scratch, two 64-row halves of 16 KiB
            if (more && nt >= 2) {
                stage(cur ^ 1, next_tile, BK);
                pre = true;
                pre_counted = (row0 + BM <= M) && (col0 + BN <= p.N);     // all 8 stores below really issue
            }
            char* ebuf = smem + GEMM_LDS;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                if (wm == half) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int ml = i * 16 + (lane & 15);
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int slot = wn * 16 + j * 4 + (lane >> 4);      // 8-byte slot (4 bf16) in the row
                            const f32x4 v = acc[i][j];
                            u32x2 o = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
                            *(u32x2*)(ebuf + ml * 256 + (((slot >> 1) ^ (ml & 15)) << 4) + (slot & 1) * 8) = o;
                        }
                    }
                }
                lds_barrier();
#pragma unroll
                for (int ps = 0; ps < 4; ++ps) {
                    const int r = ps * 16 + (tid >> 4), c = tid & 15;
                    const u32x4 v = *(const u32x4*)(ebuf + r * 256 + ((c ^ (r & 15)) << 4));
                    const int gm = row0 + half * 64 + r, gn = col0 + c * 8;
                    if (gm < M && gn < p.N) __builtin_nontemporal_store(v, (u32x4*)((bf16_t*)p.C + (moff + gm) * p.ldc + gn));
                }
                lds_barrier();
            }
        } else {
            char* ebuf = smem + (cur ^ 1) * STAGE_BYTES;
#pragma unroll
            for (int half = 0; half < 2; ++half) {                          // 64 rows x 128 fp32 = 32 KiB per pass
                if (wm == half) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int ml = i * 16 + (lane & 15);
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int c = wn * 16 + j * 4 + (lane >> 4);    // 16-byte chunk (4 fp32) in the row
                            *(f32x4*)(ebuf + ml * 512 + ((c ^ (ml & 15)) << 4)) = acc[i][j];
                        }
                    }
                }
                lds_barrier();
#pragma unroll
                for (int ps = 0; ps < 8; ++ps) {
                    const int r = ps * 8 + (tid >> 5), c = tid & 31;
                    f32x4 v = *(const f32x4*)(ebuf + r * 512 + ((c ^ (r & 15)) << 4));
                    const int gm = row0 + half * 64 + r, gn = col0 + c * 4;
                    if (gm < M && gn < p.N) {
                        const long mrow = moff + gm;
                        if (p.epi == EGO_EPI_RESID) {
                            const f32x4 rr = *(const f32x4*)(p.R + mrow * p.ldr + gn);
                            v = f32x4{rr[0] + round_bf16(v[0]), rr[1] + round_bf16(v[1]), rr[2] + round_bf16(v[2]), rr[3] + round_bf16(v[3])};
                        } else if (p.epi == EGO_EPI_BIAS_RESID) {
                            const f32x4 rr = *(const f32x4*)(p.R + mrow * p.ldr + gn);
                            const f32x4 b = *(const f32x4*)(p.bias + gn);
                            v = f32x4{rr[0] + round_bf16(v[0] + round_bf16(b[0])), rr[1] + round_bf16(v[1] + round_bf16(b[1])),
                                      rr[2] + round_bf16(v[2] + round_bf16(b[2])), rr[3] + round_bf16(v[3] + round_bf16(b[3]))};
                        }
                        *(f32x4*)((float*)p.C + mrow * p.ldc + gn) = v;
                    }
                }
                lds_barrier();
            }
        }
        if (!more) break;
        tile = next_tile;
    }
}

// ---------------------------------------------------------------------------------------------
// NT kernel for SMALL grids (round 4): 64x64 output tile per workgroup, one tile per workgroup, 4-deep LDS-DMA ring.
//
// The generation path (BASELINE config 4: rgb -> depth, batch 1) runs its decoder linears on 1707 rows: a 1707 x 768
// output is 84 tiles of 128 x 128 on a 256-CU chip, and a launch then takes as long as ONE tile - 12 to 32 K-steps, each a
// full DMA round trip in the two-stage kernel above (24 - 65 us per launch, rocprofv3: profiles/r04_eval_rgb2depth_*).
// Such a launch is bound by the latency of one workgroup's K loop, not by MFMA or HBM rate, so this kernel (a) cuts the
// output into 64 x 64 tiles - four times the workgroups, a quarter of the work each - and (b) keeps three K-steps in
// flight (counted vmcnt, one barrier per step), so that a step costs its LDS reads + 8 MFMAs per wave rather than a DMA
// round trip.  Same operand layout and swizzle as gemm_nt_kernel (64-row images); epilogues straight from the accumulator
// registers (a lane owns 4 consecutive output columns: 8- / 16-byte stores), bit-identical results to the other NT kernels
// up to the fp32 summation order inside the MFMA chain (same K order: identical here).
// ---------------------------------------------------------------------------------------------
// Round 5: the same kernel for other tile shapes (template: TM x TN output tile, NS ring slots).  The generation path's ENCODER
// linears run on 5120 ... 11948 rows: 180 - 1300 tiles of 128 x 128, i.e. one to three rounds of the persistent two-stage kernel
// above, whose tile takes 12 K-steps x one exposed DMA round trip (~1 us each: 24 - 38 us per launch, rocprofv3
// profiles/r05_eval_rgb2depth_before_kernel_stats.csv).  A 128 x 64 tile on a 3-deep ring (72 KiB: two workgroups per CU) keeps
// two K-steps in flight, its K-step costs LDS reads + 16 MFMAs per wave, and the grid has twice the workgroups to fill rounds with.
template <int TM, int TN, int NS>
struct NTL {
    static constexpr int PA = TM / 32, PB = TN / 32;          // LDS-DMA pieces (8 rows x 128 B) per wave, K-step and operand
    static constexpr int TA = TM * 128, TB = TN * 128;        // operand tiles: rows x 64 bf16
    static constexpr int STAGE = TA + TB;
    static constexpr int LDS = NS * STAGE;
    static constexpr int PER = PA + PB;                       // DMA instructions of a wave per K-step
};
constexpr int NT64_LDS = NTL<64, 64, 4>::LDS;        // 64 KiB: two workgroups per CU
constexpr int NTL128x64_LDS = NTL<128, 64, 3>::LDS;  // 72 KiB: two workgroups per CU
constexpr int NTL128x128_LDS = NTL<128, 128, 3>::LDS;   // 96 KiB: one workgroup per CU

template <int N>
__device__ __forceinline__ void vm_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int TM, int TN, int NS>
__global__ __launch_bounds__(256, (NTL<TM, TN, NS>::LDS <= 80 * 1024 ? 2 : 1)) void gemm_ntl_kernel(NTArgs p) {
    using T = NTL<TM, TN, NS>;
    constexpr int MI = TM / 32, NI = TN / 32;                 // 16 x 16 accumulator blocks of a wave: (TM / 2) x (TN / 2) outputs
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    int M = p.M;
    long moff = 0;
    if (p.m_range) { moff = p.m_range[0]; M = min(M, p.m_range[1]); }
    const int tiles_n = (p.N + TN - 1) / TN;
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int row0 = (tile / tiles_n) * TM, col0 = (tile % tiles_n) * TN;
    if (row0 >= M) return;
    const int nt = p.K / BK;

    // staging: wave instruction (P * wave + j) of an operand writes tile rows 8 (P wave + j) .. +7 (1 KiB, lane-linear);
    // slot s of row r holds global chunk s ^ (r & 7).  Rows beyond M / N are clamped (they feed outputs that are never stored).
    const bf16_t* ga[T::PA];
    const bf16_t* gb[T::PB];
#pragma unroll
    for (int j = 0; j < T::PA; ++j) {
        const int r = 8 * (T::PA * wave + j) + (lane >> 3);
        ga[j] = p.A + (moff + min(row0 + r, M - 1)) * p.lda + ((lane & 7) ^ (r & 7)) * 8;
    }
#pragma unroll
    for (int j = 0; j < T::PB; ++j) {
        const int r = 8 * (T::PB * wave + j) + (lane >> 3);
        gb[j] = p.B + (long)min(col0 + r, p.N - 1) * p.ldb + ((lane & 7) ^ (r & 7)) * 8;
    }
    auto stage = [&](int s, int k0) {
        char* sa = smem + s * T::STAGE + wave * T::PA * 1024;
        char* sb = smem + s * T::STAGE + T::TA + wave * T::PB * 1024;
#pragma unroll
        for (int j = 0; j < T::PA; ++j) glds16(ga[j] + k0, sa + j * 1024);
#pragma unroll
        for (int j = 0; j < T::PB; ++j) glds16(gb[j] + k0, sb + j * 1024);
    };
#pragma unroll
    for (int s = 0; s < NS - 1; ++s)
        if (s < nt) stage(s, s * BK);

    f32x4 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    int cur = 0;
    for (int kt = 0; kt < nt; ++kt) {
        // K-step kt must have landed; the (at most NS - 2) younger steps in flight stay out: T::PER DMA instructions per wave and step
        const int younger = min(nt - 1 - kt, NS - 2);
        if (NS >= 4 && younger >= 2) vm_wait<2 * T::PER>();
        else if (younger >= 1) vm_wait<T::PER>();
        else vm_wait<0>();
        lds_barrier();                                   // ... for every wave; and every wave is done reading step kt - 1
        if (kt + NS - 1 < nt) stage(cur == 0 ? NS - 1 : cur - 1, (kt + NS - 1) * BK);
        const char* sa = smem + cur * T::STAGE;
        const char* sb = sa + T::TA;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 af[MI], bfr[NI];
            const int c = ks * 4 + (lane >> 4);
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const int ar = wm * (TM / 2) + i * 16 + (lane & 15);
                af[i] = *(const bf16x8*)(sa + ar * 128 + ((c ^ (ar & 7)) << 4));
            }
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                const int br = wn * (TN / 2) + j * 16 + (lane & 15);
                bfr[j] = *(const bf16x8*)(sb + br * 128 + ((c ^ (br & 7)) << 4));
            }
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    // operands swapped: D[row = n (4 regs)][col = m (lane & 15)]
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
        }
        cur = (cur + 1 == NS) ? 0 : cur + 1;
    }

#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int gm = row0 + wm * (TM / 2) + i * 16 + (lane & 15);
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int gn = col0 + wn * (TN / 2) + j * 16 + 4 * (lane >> 4);
            if (gm >= M || gn >= p.N) continue;
            const long mrow = moff + gm;
            f32x4 v = acc[i][j];
            if (p.epi == EGO_EPI_BF16) {
                *(u32x2*)((bf16_t*)p.C + mrow * p.ldc + gn) = u32x2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
                continue;
            }
            if (p.epi == EGO_EPI_RESID) {
                const f32x4 rr = *(const f32x4*)(p.R + mrow * p.ldr + gn);
                v = f32x4{rr[0] + round_bf16(v[0]), rr[1] + round_bf16(v[1]), rr[2] + round_bf16(v[2]), rr[3] + round_bf16(v[3])};
            } else if (p.epi == EGO_EPI_BIAS_RESID) {
                const f32x4 rr = *(const f32x4*)(p.R + mrow * p.ldr + gn);
                const f32x4 b = *(const f32x4*)(p.bias + gn);
                v = f32x4{rr[0] + round_bf16(v[0] + round_bf16(b[0])), rr[1] + round_bf16(v[1] + round_bf16(b[1])),
                          rr[2] + round_bf16(v[2] + round_bf16(b[2])), rr[3] + round_bf16(v[3] + round_bf16(b[3]))};
            }
            *(f32x4*)((float*)p.C + mrow * p.ldc + gn) = v;
        }
    }
}
// the round-4 name of the 64 x 64 instantiation (profiles, tools)
#define gemm_nt64_kernel (gemm_ntl_kernel<64, 64, 4>)

// ---------------------------------------------------------------------------------------------
// NT kernel, 256x256 tile, 8 waves, staggered half-phases (for the large GEMMs).
//
// Waves 0-3 (group 0, output rows 0-127) and 4-7 (group 1, rows 128-255) share the four SIMDs pairwise
// (wave i and i+4 sit on one SIMD).  Each K-tile (64 deep) is 4 phases, one 64x32 quadrant of the wave's
// 128x64 output per phase: a LOAD segment (ds_read_b128 of the fragments the quadrant needs, plus this
// wave's share of the LDS-DMA for the NEXT K-tile) and a COMPUTE segment (16 MFMAs), each closed by a
// workgroup barrier.  Group 1 runs one barrier behind group 0, so on every SIMD one wave is in its MFMA
// segment while its partner reads LDS / issues DMA: the matrix pipe is fed in every half-phase slot.
//   slot 2q   : G0 LOAD(q)      G1 COMPUTE(q-1)
//   slot 2q+1 : G0 COMPUTE(q)   G1 LOAD(q)
// Hazards (two 64-KiB stages, DMA one K-tile ahead):
//   RAW  every wave drains its own DMA (vmcnt(0)) before the barrier that closes slot 7; the first read
//        of the new tile (G0, slot 0) is behind that barrier.
//   WAR  the DMA for tile t+1 targets the stage last read in tile t-1; G1's last reads of it (slot 7)
//        are consumed at the head of slot 0 of tile t, so G0 issues DMA only from LOAD(1) on (slot 2),
//        G1 from its LOAD(0) (slot 1): both behind a barrier that follows those reads.
// ---------------------------------------------------------------------------------------------
constexpr int T2_BYTES = 256 * 128;            // operand tile: 256 rows x 64 bf16
constexpr int S2_BYTES = 2 * T2_BYTES;         // stage = A + B = 64 KiB
constexpr int NT2_LDS = 2 * S2_BYTES;          // 128 KiB: one workgroup per CU

__device__ __forceinline__ void bar_pinned() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
}

// Persistent: a workgroup walks tiles id, id + gridDim.x, ... as ONE K-tile stream (stream index g runs over the
// K-tiles of all its tiles).  A is the cold operand (activations / gradients of 100-500 MB, read from HBM), B the
// weights (L2-resident): A gets a THREE-deep ring and is fetched two K-tiles ahead, B a two-deep ring one ahead
// (3 x 32 + 2 x 32 KiB = all 160 KiB of LDS).  With A one tile ahead the kernel ran at 1150 TF/s on MALL-warm A and
// 950 on cold A - the engine's case.  Per wave and K-step the four B pieces are issued before the four A pieces, so
// "at most 4 outstanding" (vmcnt(4)) means B(g+1) and the older A(g+1) have landed while A(g+2) stays in flight.
// The bf16 epilogue runs in two 128-row passes through the A and B stages the last K-step left free, and its stores
// drain under the next tile.  Operands arrive by buffer loads (LDS-DMA): lane offsets are loop-invariant, tile and K
// position are the scalar offset, rows past M / N read as zeros.  K >= 128.
// cache policy of the operand DMAs (aux of buffer_load ... lds: 0 default, 2 = nt: streamed, evict-first in L2)
#ifndef NT256_STRIPS
#define NT256_STRIPS 1      // column-strip tile walk for wide outputs (0: row-major everywhere, the round-2 walk)
#endif
#ifndef NT256_A_AUX
#define NT256_A_AUX 0
#endif
#ifndef NT256_B_AUX
#define NT256_B_AUX 0
#endif
constexpr int RA_BYTES = 256 * 128;                // one ring slot of either operand: 256 rows x 64 bf16 = 32 KiB
constexpr int NT3_LDS = 5 * RA_BYTES;              // A slots 0..2, B slots 3..4

// FP8: operands are OCP e4m3 bytes with one fp32 scale per row (A) / per output channel (B).  A 128-byte LDS row then holds
// 128 k-values instead of 64, everything about the staging (LDS-DMA pieces, swizzle, rings, K-tiles of 128 BYTES) is
// unchanged, and one v_mfma_scale_f32_16x16x128_f8f6f4 (block scales fixed to 1.0: E8M0 127) replaces the two 16x16x32
// bf16 MFMAs of a K-tile at the same cycles - twice the contraction per tile.  Lane l supplies row l & 15, k-bytes
// 32 (l >> 4) .. +31 of either operand (probed with exact e4m3 data: tools/probes/mfma_fp8_probe.cpp); the accumulator
// layout is the bf16 one, and the row / column scales are applied when the epilogue reads the accumulators.
typedef int i32x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ i32x8 cat_frag(bf16x8 lo, bf16x8 hi) {
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    const i32x4 a = __builtin_bit_cast(i32x4, lo), b = __builtin_bit_cast(i32x4, hi);
    return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
}

// STRIPS: the column-strip tile walk of very wide outputs (its own instantiation: the narrow shapes keep the row-major code as is)
template <int EK, bool FP8 = false, bool STRIPS = false>   // epilogue class: 0 bf16, 1 fp32 family, 2 fused SwiGLU backward, 3 fused SwiGLU forward (separate register allocations)
__global__ __launch_bounds__(512, 2) void gemm_nt256_kernel(NTArgs p) {
    constexpr int ESZ = FP8 ? 1 : 2;                           // bytes per operand element
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, wc = wave & 3;              // grp = row half (and stagger group), wc = 64-col strip

    int M = p.M;
    long moff = 0;
    if (p.m_range) { moff = p.m_range[0]; M = min(M, p.m_range[1]); }
    if (M <= 0) return;
    // EK 3: a tile is 128 columns of the a half plus the matching 128 columns of the b half of fc1||fc3 (p.N = F)
    const int tiles_n = EK == 3 ? p.N / 128 : (p.N + 255) / 256;
    const int tile_w = EK == 3 ? 128 : 256;
    // M may come from the device (row range of one modality): spread the tiles that really exist over the XCDs
    const int ntiles = ((M + 255) / 256) * tiles_n;
    if ((int)blockIdx.x >= ntiles) return;
    const int nt = p.K * ESZ / 128;                            // K-tiles of 128 bytes
    const int G = gridDim.x;
    // De-phasing probe: all persistent workgroups start together and run equal tiles, so
    // their epilogue store bursts hit HBM at the same moments; with p.dephase > 0 every other workgroup of an XCD starts late.
    // 0 (the product setting) compiles to one scalar compare; results are identical either way.
    if (p.dephase > 0 && ((blockIdx.x >> 3) & 1))
        for (int i = 0; i < p.dephase; ++i) __builtin_amdgcn_s_sleep(16);          // 16 x 64 clocks ~ 0.5 us at 2 GHz

    // A: one descriptor per 256-row tile (based at the tile's first row, sized to its valid rows), so the 32-bit buffer
    // offsets never see more than 256 rows - activations / logit gradients larger than 4 GiB are fine
    auto a_rsrc = [&](int row0) {
        return __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p.A + (moff + row0) * p.lda * ESZ), 0,
                                                 (int)(unsigned)((long)(min(256, M - row0) - 1) * p.lda * ESZ + (long)p.K * ESZ), 0x00020000);
    };
    const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)p.B, 0, (int)(unsigned)((long)((EK == 3 ? 2 * p.N : p.N) - 1) * p.ldb * ESZ + (long)p.K * ESZ), 0x00020000);
    // wave instruction (wave*4 + j) fills tile rows 8(wave*4+j)..+7 (1 KiB), swizzle on the source
    // (piece j of a wave starts 8 rows after piece j - 1: one lane offset per operand, the rest goes into the scalar offset)
    unsigned a_off0, b_off0;
    {
        const int r = 8 * (wave * 4) + (lane >> 3), c = ((lane & 7) ^ (r & 7)) * 16;
        a_off0 = (unsigned)(r * (int)p.lda * ESZ + c);
        b_off0 = (unsigned)((EK == 3 && r >= 128 ? p.N - 128 + r : r) * (int)p.ldb * ESZ + c);
    }
    const unsigned a_step = (unsigned)(8 * (int)p.lda * ESZ), b_step = (unsigned)(8 * (int)p.ldb * ESZ);
    // Tile walk.  xcd_remap gives every XCD a contiguous run of tile indices t, and the 32 workgroups of an XCD work on 32
    // consecutive t at a time.  Outputs of up to 31 column tiles: t runs row-major over the whole width - the 32 tiles are a
    // few A row panels times all column tiles.  The 64,000-token logits (250 column tiles): row-major puts all 32 on ONE A
    // panel with 32 different B panels (33 operand panels per 32 tiles, and the 98 MB table re-read for each of the 252 row
    // panels: 24.7 GB per launch); there t walks down column STRIPS of about 8 tiles - 32 tiles = 4 A panels x 8 B panels,
    // the strip's B panels stay the same from one round to the next, and what is re-read per strip is the 99 MB of A
    // (Infinity-Cache resident).  Measured (tools/gemm_ab.py): logits 1005 -> 1102 TF/s; at 16 column tiles (fc1||fc3) strips
    // are 2.5 % SLOWER (A, 201 MB, is then streamed from HBM once per strip), hence the threshold.
    const int rows_t = (M + 255) / 256;
    const int nstrips = STRIPS ? (tiles_n + 7) / 8 : 1;
    const int w0 = tiles_n / nstrips, wide = tiles_n % nstrips;   // the first `wide` strips are w0 + 1 tiles wide
    const int big = wide * (w0 + 1) * rows_t;
    auto tile_origin = [&](int id, int& row0, int& col0) {
        const int t = xcd_remap(id, ntiles);
        int tr, tc;
        if constexpr (STRIPS) {
            int rem, w, c0;
            if (t < big) { const int per = (w0 + 1) * rows_t, sidx = t / per; rem = t - sidx * per; w = w0 + 1; c0 = sidx * (w0 + 1); }
            else { const int per = w0 * rows_t, u = t - big, sidx = u / per; rem = u - sidx * per; w = w0; c0 = wide * (w0 + 1) + sidx * w0; }
            tr = rem / w; tc = c0 + rem % w;
        } else {
            tr = t / tiles_n; tc = t % tiles_n;
        }
        row0 = tr * 256; col0 = tc * tile_w;
    };
    // fetch cursors: the K-tile of the stream that the next A / B DMA brings in (scalar state)
    int idA = blockIdx.x, ktA = 0, rA, cA_unused, slotA = 0; bool moreA = true;
    int idB = blockIdx.x, ktB = 0, rB_unused, cB, slotB = 0; bool moreB = true;
    tile_origin(idA, rA, cA_unused);
    tile_origin(idB, rB_unused, cB);
    unsigned a_so = 0, b_so = 0;
    __amdgpu_buffer_rsrc_t ars = a_rsrc(rA);
    auto cursorA = [&]() { a_so = (unsigned)(ktA * BK * 2); };      // scalar offset of the cursor's K-tile inside its tile rows
    auto advanceA = [&]() {
        slotA = slotA == 2 ? 0 : slotA + 1;
        if (++ktA == nt) {
            ktA = 0; idA += G;
            if (idA < ntiles) { tile_origin(idA, rA, cA_unused); ars = a_rsrc(rA); } else moreA = false;
        }
    };
    auto cursorB = [&]() { b_so = (unsigned)cB * (unsigned)(p.ldb * ESZ) + (unsigned)(ktB * BK * 2); };
    auto advanceB = [&]() {
        slotB ^= 1;
        if (++ktB == nt) { ktB = 0; idB += G; if (idB < ntiles) tile_origin(idB, rB_unused, cB); else moreB = false; }
    };
    auto dmaA = [&](int slot, int j) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ars, (__attribute__((address_space(3))) void*)(smem + slot * RA_BYTES + (wave * 4 + j) * 1024),
                                                 16, a_off0, (int)(a_so + j * a_step), 0, NT256_A_AUX);
    };
    auto dmaB = [&](int slot, int j) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(brs, (__attribute__((address_space(3))) void*)(smem + (3 + slot) * RA_BYTES + (wave * 4 + j) * 1024),
                                                 16, b_off0, (int)(b_so + j * b_step), 0, NT256_B_AUX);
    };
    // fragment byte offsets inside an operand tile: row = base16 + (lane & 15) (base16 multiple of 16, so
    // row & 7 == lane & 7), chunk = ks * 4 + (lane >> 4), slot = chunk ^ (lane & 7)
    // bf16: the lane's 16-byte chunk of k-step ks.  fp8: the SAME two chunks - lane group g = lane >> 4 feeds the MFMA with
    // k-bytes [16 g, 16 g + 16) and [64 + 16 g, 64 + 16 g + 16) of its row instead of the contiguous [32 g, 32 g + 32).  Both
    // operands use the same map, so the contraction still covers every k exactly once (the block scales are all 1.0), and the
    // two reads keep the bank pattern the swizzle was built for.  Round 3 read chunks 2 g + ks: lanes 20-27 of the first
    // ds_read_b128 group then hit exactly the banks of lanes 0-3 / 12-15 - a 2-way conflict on EVERY fragment read, which is
    // what made an e4m3 K-tile take 1.7 x a bf16 K-tile (-DFP8_FRAG_OLD=1 rebuilds that map for the A/B).
#ifndef FP8_FRAG_OLD
#define FP8_FRAG_OLD 0
#endif
    int foff[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
        foff[ks] = (lane & 15) * 128 + ((((FP8 && FP8_FRAG_OLD) ? 2 * (lane >> 4) + ks : ks * 4 + (lane >> 4)) ^ (lane & 7)) << 4);
    const int a_base = grp * 128 * 128, b_base = wc * 64 * 128;

    int id = blockIdx.x, row0, col0;
    tile_origin(id, row0, col0);
    // prologue: A(0), B(0), then A(1); the first two must have landed
    cursorA();
#pragma unroll
    for (int j = 0; j < 4; ++j) dmaA(slotA, j);
    advanceA();
    cursorB();
#pragma unroll
    for (int j = 0; j < 4; ++j) dmaB(slotB, j);
    advanceB();
    if (moreA) {                                // (false only for a single tile with a single K-tile)
        cursorA();
#pragma unroll
        for (int j = 0; j < 4; ++j) dmaA(slotA, j);
        advanceA();
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    bar_pinned();
    int curA = 0, curB = 0;                                   // ring slots of the K-tile about to be consumed

#define LOAD_A(QM)                                                                                         \
    _Pragma("unroll") for (int mi = 0; mi < 4; ++mi) _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)      \
        af[mi][ks] = *(const bf16x8*)(sa + a_base + ((QM) * 4 + mi) * 2048 + foff[ks]);
#define LOAD_B(QN)                                                                                         \
    _Pragma("unroll") for (int ni = 0; ni < 2; ++ni) _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)      \
        bq[ni][ks] = *(const bf16x8*)(sb + b_base + ((QN) * 2 + ni) * 2048 + foff[ks]);
#define COMPUTE(QM, QN)                                                                                    \
    __builtin_amdgcn_s_setprio(1);                                                                         \
    if constexpr (FP8) {                                                                                   \
        _Pragma("unroll") for (int mi = 0; mi < 4; ++mi) _Pragma("unroll") for (int ni = 0; ni < 2; ++ni)  \
            acc[(QM) * 4 + mi][(QN) * 2 + ni] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(          \
                cat_frag(bq[ni][0], bq[ni][1]), cat_frag(af[mi][0], af[mi][1]),                            \
                acc[(QM) * 4 + mi][(QN) * 2 + ni], 0, 0, 0, 127, 0, 127);                                  \
    } else {                                                                                               \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) _Pragma("unroll") for (int mi = 0; mi < 4; ++mi)  \
            _Pragma("unroll") for (int ni = 0; ni < 2; ++ni)                                               \
                acc[(QM) * 4 + mi][(QN) * 2 + ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(               \
                    bq[ni][ks], af[mi][ks], acc[(QM) * 4 + mi][(QN) * 2 + ni], 0, 0, 0);                   \
    }                                                                                                      \
    __builtin_amdgcn_s_setprio(0);

    while (true) {
        f32x4 acc[8][4];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        bf16x8 af[4][2], bq[2][2];

        if (grp == 1) bar_pinned();                              // stagger: group 1 runs one barrier behind
        for (int kt = 0; kt < nt; ++kt) {
            const char* sa = smem + curA * RA_BYTES;
            const char* sb = smem + (3 + curB) * RA_BYTES;
            // this K-step fetches B(g+1) into the B slot consumed in step g-1 and A(g+2) into the A slot consumed in g-1
            // (cursors: A two K-tiles ahead of the stream, B one); scalar offsets and slots are fixed for the whole step
            const bool fa = moreA, fb = moreB;
            const int sA = slotA, sB = slotB;
            if (fa) cursorA();
            if (fb) cursorB();
            // ---- phase 0: quadrant (0,0)
            LOAD_A(0) LOAD_B(0)
            // (the 8 DMA pieces of a wave are spread 2 / 2 / 2 / 2 over the four load phases: 3 + 3 + 2 + 0 made the fullest load
            //  phase outlast the other group's 32 MFMAs - a piece costs 60 - 185 issue cycles; group 0 may not touch the B slot before
            //  phase 1 (group 1, one barrier behind, still reads it in the previous step's last phase), the A slot is free at once)
            if (grp == 1) { if (fb) { dmaB(sB, 0); dmaB(sB, 1); } } else if (fa) { dmaA(sA, 0); dmaA(sA, 1); }
            bar_pinned();
            COMPUTE(0, 0)
            bar_pinned();
            // ---- phase 1: quadrant (0,1)
            LOAD_B(1)
            if (fb) { if (grp == 1) { dmaB(sB, 2); dmaB(sB, 3); } else { dmaB(sB, 0); dmaB(sB, 1); } }
            bar_pinned();
            COMPUTE(0, 1)
            bar_pinned();
            // ---- phase 2: quadrant (1,1)
            LOAD_A(1)
            if (grp == 1) { if (fa) { dmaA(sA, 0); dmaA(sA, 1); } } else if (fb) { dmaB(sB, 2); dmaB(sB, 3); }
            bar_pinned();
            COMPUTE(1, 1)
            bar_pinned();
            // ---- phase 3: quadrant (1,0)
            LOAD_B(0)
            if (fa) { dmaA(sA, 2); dmaA(sA, 3); }
            // G1: B(g+1) and the older A(g+1) landed before slot 7 closes; its 4 pieces of A(g+2) (the youngest) may stay in flight
            if (grp == 1) { if (fa) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
            bar_pinned();
            COMPUTE(1, 0)
            // G0 issued A(g+2) pieces 0, 1 first, then B(g+1), then A(g+2) pieces 2, 3: everything but the last two has landed
            if (grp == 0) { if (fa) asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
            bar_pinned();
            if (fa) advanceA();
            if (fb) advanceB();
            curA = curA == 2 ? 0 : curA + 1;
            curB ^= 1;
        }
        if (grp == 0) bar_pinned();                              // match group 1's extra barrier: groups aligned again
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

        // epilogue: the A and B slots consumed by the last K-step are free (the next DMA into them is issued in the next
        // tile's first K-step, behind the barriers below).  Two passes of 128 rows; image rows 0-63 in the A slot, 64-127
        // in the B slot, [row][512 B], 16-byte chunk c of row r at chunk c ^ (r & 15): whole-row coalesced stores.
        char* ea = smem + (curA == 0 ? 2 : curA - 1) * RA_BYTES;
        char* ebb = smem + (3 + (curB ^ 1)) * RA_BYTES;
        if constexpr (FP8) {
            // C = sa[m] * sb[n] * acc: this lane's 8 rows (i) and 4 x 4 columns (j) of the tile.  All 12 scale loads are
            // issued together and waited for once (24 registers - the operand fragments are dead here): one global-memory
            // round trip per tile instead of four serialised ones (each wait also drains the next tile's operand DMA)
            float ra[8];
            f32x4 cb[4];
#pragma unroll
            for (int i = 0; i < 8; ++i) ra[i] = p.sa[moff + min(row0 + grp * 128 + i * 16 + (lane & 15), M - 1)];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int tc = wc * 64 + j * 16 + (lane >> 4) * 4;                         // first of 4 tile columns
                const int bn = EK == 3 ? (tc < 128 ? col0 + tc : p.N + col0 + tc - 128) : min(col0 + tc, p.N - 4);
                cb[j] = *(const f32x4*)(p.sb + bn);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i][j] = acc[i][j] * (cb[j] * ra[i]);
        }
        if constexpr (EK == 0) {
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                if (grp == half) {
    #pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const int ml = i * 16 + (lane & 15);
                        char* eb = (i < 4 ? ea : ebb) + (ml & 63) * 512;
    #pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int slot = wc * 16 + j * 4 + (lane >> 4);              // 8-byte slot (4 bf16) in the row
                            const f32x4 v = acc[i][j];
                            u32x2 o = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
                            *(u32x2*)(eb + (((slot >> 1) ^ (ml & 15)) << 4) + (slot & 1) * 8) = o;
                        }
                    }
                }
                lds_barrier();
    #pragma unroll
                for (int ps = 0; ps < 8; ++ps) {
                    const int r = ps * 16 + (tid >> 5), c = tid & 31;
                    const u32x4 v = *(const u32x4*)((ps < 4 ? ea : ebb) + (r & 63) * 512 + ((c ^ (r & 15)) << 4));
                    const int gm = row0 + half * 128 + r, gn = col0 + c * 8;
                    if (gm < M && gn < p.N) __builtin_nontemporal_store(v, (u32x4*)((bf16_t*)p.C + (moff + gm) * p.ldc + gn));
                }
                lds_barrier();
            }
        } else if constexpr (EK == 3) {
            // tile columns 0-127 are a[:, col0 ..], columns 128-255 are b[:, col0 ..]: both halves of ab are stored (the
            // backward needs them) together with h = bf16(bf16(silu(a)) * b) (ego_swiglu_fwd's arithmetic on the bf16 ab)
            bf16_t* Cb = (bf16_t*)p.C + moff * p.ldc;
            bf16_t* Hb = p.H + moff * p.ldh;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                if (grp == half) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const int ml = i * 16 + (lane & 15);
                        char* eb = (i < 4 ? ea : ebb) + (ml & 63) * 512;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int slot = wc * 16 + j * 4 + (lane >> 4);
                            const f32x4 v = acc[i][j];
                            u32x2 o = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
                            *(u32x2*)(eb + (((slot >> 1) ^ (ml & 15)) << 4) + (slot & 1) * 8) = o;
                        }
                    }
                }
                lds_barrier();
#pragma unroll
                for (int ps = 0; ps < 4; ++ps) {
                    const int r = ps * 32 + (tid >> 4), c = tid & 15;
                    const char* rowp = (ps < 2 ? ea : ebb) + (r & 63) * 512;
                    const u32x4 av = *(const u32x4*)(rowp + ((c ^ (r & 15)) << 4));
                    const u32x4 bv = *(const u32x4*)(rowp + (((c + 16) ^ (r & 15)) << 4));
                    u32x4 hv;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float a0 = bf16_to_f32(av[e] & 0xffff), a1 = bf16_to_f32(av[e] >> 16);
                        const float b0 = bf16_to_f32(bv[e] & 0xffff), b1 = bf16_to_f32(bv[e] >> 16);
                        hv[e] = pack_bf16x2(round_bf16(a0 * sigmoidf_(a0)) * b0, round_bf16(a1 * sigmoidf_(a1)) * b1);
                    }
                    const int gm = row0 + half * 128 + r, gn = col0 + c * 8;
                    if (gm < M) {
                        __builtin_nontemporal_store(av, (u32x4*)(Cb + (long)gm * p.ldc + gn));
                        __builtin_nontemporal_store(bv, (u32x4*)(Cb + (long)gm * p.ldc + p.N + gn));
                        __builtin_nontemporal_store(hv, (u32x4*)(Hb + (long)gm * p.ldh + gn));
                    }
                }
                lds_barrier();
            }
        } else if constexpr (EK == 2) {
            // The tile is dh = dY W2 (never stored): da = dh b s (1 + a (1 - s)), db = dh a s with s = sigmoid(a), written
            // to dab[:, n] and dab[:, N + n] (ego_swiglu_bwd's arithmetic on the bf16-rounded dh, bit for bit).  Two passes
            // of 128 rows like the bf16 epilogue.
            const bf16_t* Xb = p.X + moff * p.ldx;
            bf16_t* Yb = (bf16_t*)p.C + moff * p.ldc;
            // groups of 2 row slices (8 groups per tile, 4 per pass)
            auto load_ab = [&](int g, u32x4 (&av)[2], u32x4 (&bv)[2]) {
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int gm = row0 + (g >> 2) * 128 + ((g & 3) * 2 + k) * 16 + (tid >> 5), gn = col0 + (tid & 31) * 8;
                    const u32x4 z = {0u, 0u, 0u, 0u};
                    const bool ok = g < 8 && gm < M;
                    av[k] = ok ? __builtin_nontemporal_load((const u32x4*)(Xb + (long)gm * p.ldx + gn)) : z;
                    bv[k] = ok ? __builtin_nontemporal_load((const u32x4*)(Xb + (long)gm * p.ldx + p.N + gn)) : z;
                }
            };
            auto process = [&](int g, const u32x4 (&av)[2], const u32x4 (&bv)[2]) {
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int ps = (g & 3) * 2 + k;
                    const int r = ps * 16 + (tid >> 5), c = tid & 31;
                    const u32x4 gq = *(const u32x4*)((ps < 4 ? ea : ebb) + (r & 63) * 512 + ((c ^ (r & 15)) << 4));
                    const int gm = row0 + (g >> 2) * 128 + r, gn = col0 + c * 8;
                    u32x4 oa, ob;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float a2[2] = {bf16_to_f32(av[k][e] & 0xffff), bf16_to_f32(av[k][e] >> 16)};
                        const float b2[2] = {bf16_to_f32(bv[k][e] & 0xffff), bf16_to_f32(bv[k][e] >> 16)};
                        const float g2[2] = {bf16_to_f32(gq[e] & 0xffff), bf16_to_f32(gq[e] >> 16)};
                        float da[2], db[2];
#pragma unroll
                        for (int t = 0; t < 2; ++t) {
                            const float sg = sigmoidf_(a2[t]);
                            da[t] = g2[t] * b2[t] * sg * (1.f + a2[t] * (1.f - sg));
                            db[t] = g2[t] * a2[t] * sg;
                        }
                        oa[e] = pack_bf16x2(da[0], da[1]);
                        ob[e] = pack_bf16x2(db[0], db[1]);
                    }
                    if (gm < M) {
                        __builtin_nontemporal_store(oa, (u32x4*)(Yb + (long)gm * p.ldc + gn));
                        __builtin_nontemporal_store(ob, (u32x4*)(Yb + (long)gm * p.ldc + p.N + gn));
                    }
                }
            };
            // The accumulators are rounded to bf16 first (64 registers instead of 128): that frees the room to have the a / b
            // rows of half the tile in flight at any time (4 groups x 4 loads of 16 bytes per lane = 128 KiB per CU) - the
            // epilogue is HBM-latency-bound at one workgroup per CU, so its time is the number of dependent round trips
            u32x2 pk[8][4];
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f32x4 v = acc[i][j];
                    pk[i][j] = u32x2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
                }
            u32x4 av[4][2], bv[4][2];                  // groups g and g + 4 share a register set
#pragma unroll
            for (int g = 0; g < 4; ++g) load_ab(g, av[g], bv[g]);
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                if (grp == half) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const int ml = i * 16 + (lane & 15);
                        char* eb = (i < 4 ? ea : ebb) + (ml & 63) * 512;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int slot = wc * 16 + j * 4 + (lane >> 4);
                            *(u32x2*)(eb + (((slot >> 1) ^ (ml & 15)) << 4) + (slot & 1) * 8) = pk[i][j];
                        }
                    }
                }
                lds_barrier();
#pragma unroll
                for (int gg = 0; gg < 4; ++gg) {
                    const int g = half * 4 + gg;
                    process(g, av[gg], bv[gg]);
                    if (half == 0) load_ab(g + 4, av[gg], bv[gg]);
                }
                lds_barrier();
            }
        } else {
            // fp32 outputs (plain / + residual / + bias + residual): four passes of 64 rows, image rows 0-31 in the A slot,
            // 32-63 in the B slot, [row][1024 B].  The residual rows of the NEXT pass are already in flight while a pass
            // transposes and stores (the fragment registers are dead here, so two 8 x 16-byte sets fit).
            const bool has_r = (p.epi == EGO_EPI_RESID || p.epi == EGO_EPI_BIAS_RESID);
            f32x4 bb = {0.f, 0.f, 0.f, 0.f};
            if (p.epi == EGO_EPI_BIAS_RESID) {
                const int gn = col0 + (tid & 63) * 4;
                if (gn < p.N) { const f32x4 b = *(const f32x4*)(p.bias + gn); bb = f32x4{round_bf16(b[0]), round_bf16(b[1]), round_bf16(b[2]), round_bf16(b[3])}; }
            }
            // residual rows: 2 x 4 row groups per pass; each group is re-loaded for the next pass right after its use,
            // so the loads fly under the rest of this pass and the next transpose
            auto load_r = [&](int q, int hb, f32x4 (&rr)[4]) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int gm = row0 + q * 64 + (hb * 4 + k) * 8 + (tid >> 6), gn = col0 + (tid & 63) * 4;
                    rr[k] = (has_r && q < 4 && gm < M && gn < p.N) ? __builtin_nontemporal_load((const f32x4*)(p.R + (moff + gm) * p.ldr + gn))
                                                                    : f32x4{0.f, 0.f, 0.f, 0.f};
                }
            };
            f32x4 rlo[4], rhi[4];
            load_r(0, 0, rlo);
            load_r(0, 1, rhi);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (grp == (q >> 1)) {
#pragma unroll
                    for (int ii = 0; ii < 4; ++ii) {
                        const int i = (q & 1) * 4 + ii;
                        const int ml = ii * 16 + (lane & 15);                        // row inside the 64-row pass
                        char* eb = (ii < 2 ? ea : ebb) + (ml & 31) * 1024;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int c = wc * 16 + j * 4 + (lane >> 4);             // 16-byte chunk (4 fp32), 64 per row
                            *(f32x4*)(eb + ((c ^ (ml & 15)) << 4)) = acc[i][j];
                        }
                    }
                }
                lds_barrier();
#pragma unroll
                for (int hb = 0; hb < 2; ++hb) {
                    f32x4 (&rr)[4] = hb == 0 ? rlo : rhi;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int r = (hb * 4 + k) * 8 + (tid >> 6), c = tid & 63;
                        f32x4 v = *(const f32x4*)((hb == 0 ? ea : ebb) + (r & 31) * 1024 + ((c ^ (r & 15)) << 4));
                        const int gm = row0 + q * 64 + r, gn = col0 + c * 4;
                        if (gm < M && gn < p.N) {
                            if (has_r) {
                                const f32x4 x = rr[k];
                                v = f32x4{x[0] + round_bf16(v[0] + bb[0]), x[1] + round_bf16(v[1] + bb[1]),
                                          x[2] + round_bf16(v[2] + bb[2]), x[3] + round_bf16(v[3] + bb[3])};
                            }
                            *(f32x4*)((float*)p.C + (moff + gm) * p.ldc + gn) = v;
                        }
                    }
                    if (q < 3) load_r(q + 1, hb, rr);
                }
                lds_barrier();
            }
        }
        id += G;
        if (id >= ntiles) break;
        tile_origin(id, row0, col0);
    }
#undef LOAD_A
#undef LOAD_B
#undef COMPUTE
}

// ---------------------------------------------------------------------------------------------
// TN kernel (wgrad)
// ---------------------------------------------------------------------------------------------
struct TNArgs {
    const bf16_t* P; long ldp;
    const bf16_t* Q; long ldq;
    float* C0; float* C1; long ldc;
    float* slab;               // [splits][Ni][Nj] partials when splits > 1
    const int* m_range;
    int split_row, rows0, rows1;
    int Ni, Nj, M, splits;
};

// 32-byte chunk swizzle for the [64 m][128 col] LDS images: the 8 rows one 32-lane half touches in
// a transposed read ({0..3, 8..11} + 16h, or +4) must land on 8 different 32-byte bank groups.
__device__ __forceinline__ int tn_f(int r) { return (r & 3) | (((r >> 3) & 1) << 2); }

// ds_read_b64_tr_b16 as inline asm: hipcc orders its own builtin form behind `vmcnt(0)` whenever an
// LDS-DMA is in flight (it cannot prove the DMA targets the other stage), which serialises load and
// compute; the asm form is invisible to that pass.  Its completion is therefore OUR job: every use is
// behind an explicit `s_waitcnt lgkmcnt(0)` + sched_barrier (cdna_hip_programming.md 5.7 item 1, rule 18).

template <int S>   // LDS stage
__device__ __forceinline__ void tn_compute(const unsigned (&pb)[4], const unsigned (&qb)[4], f32x4 (&acc)[4][4]) {
    constexpr int O = S * STAGE_BYTES;
    s16x4 p0[4][2], q0[4][2], p1[4][2], q1[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {          // k-step 0: tile rows 0..31
        p0[i][0] = tr_read<O>(pb[i]);              p0[i][1] = tr_read<O + 1024>(pb[i]);
        q0[i][0] = tr_read<O + TILE_BYTES>(qb[i]); q0[i][1] = tr_read<O + TILE_BYTES + 1024>(qb[i]);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 4; ++i) {          // k-step 1: tile rows 32..63, in flight under the MFMAs of k-step 0
        p1[i][0] = tr_read<O + 8192>(pb[i]);              p1[i][1] = tr_read<O + 8192 + 1024>(pb[i]);
        q1[i][0] = tr_read<O + TILE_BYTES + 8192>(qb[i]); q1[i][1] = tr_read<O + TILE_BYTES + 8192 + 1024>(qb[i]);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            // D[row = j (4 regs)][col = i (lane&15)]
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(join8(q0[j][0], q0[j][1]), join8(p0[i][0], p0[i][1]), acc[i][j], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(join8(q1[j][0], q1[j][1]), join8(p1[i][0], p1[i][1]), acc[i][j], 0, 0, 0);
}

__global__ __launch_bounds__(256, 2) void gemm_tn_kernel(TNArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wi = wave >> 1, wj = wave & 1;

    int M = p.M;
    long moff = 0;
    if (p.m_range) { moff = p.m_range[0]; M = min(M, p.m_range[1]); }
    const int tiles_j = p.Nj / BN, tiles_i = p.Ni / BM;
    const int ntile = tiles_i * tiles_j;
    const int bid = blockIdx.x;
    const int split = bid / ntile;
    const int t = xcd_remap(bid % ntile, ntile);
    const int i0 = (t / tiles_j) * BM, j0 = (t % tiles_j) * BN;

    // contraction range of this split, in 64-row steps
    const int nsteps = (M + BK - 1) / BK;
    const int per = (nsteps + p.splits - 1) / p.splits;
    const int st0 = split * per, st1 = min(nsteps, st0 + per);

    const bf16_t* P = p.P + moff * p.ldp + i0;
    const bf16_t* Q = p.Q + moff * p.ldq + j0;

    // Staging by LDS-DMA: one wave instruction = 1 KiB = 4 tile rows x 256 B, lane-linear, 32-byte chunk
    // swizzle applied to the SOURCE address.  A 64-row step that crosses M (rows there must contribute
    // zero - clamping is not enough) is staged through registers with zero fill: at most one step.
    long p_off[4], q_off[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = 4 * (wave * 4 + j) + (lane >> 4), s16 = lane & 15;
        const int c = (((s16 >> 1) ^ tn_f(r)) << 1) | (s16 & 1);
        p_off[j] = (long)r * p.ldp + c * 8;
        q_off[j] = (long)r * p.ldq + c * 8;
    }
    auto stage_dma = [&](int s, int m0) {
        char* sp = smem + s * STAGE_BYTES + wave * 4096;
        char* sq = sp + TILE_BYTES;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            glds16(P + (long)m0 * p.ldp + p_off[j], sp + j * 1024);
            glds16(Q + (long)m0 * p.ldq + q_off[j], sq + j * 1024);
        }
    };
    auto stage_regs = [&](int s, int m0) {
        char* sp = smem + s * STAGE_BYTES;
        char* sq = sp + TILE_BYTES;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int q = tid + 256 * i, r = q >> 4, c = q & 15;
            const int m = m0 + r;
            const u32x4 z = {0u, 0u, 0u, 0u};
            const u32x4 vp = (m < M) ? *(const u32x4*)(P + (long)m * p.ldp + c * 8) : z;
            const u32x4 vq = (m < M) ? *(const u32x4*)(Q + (long)m * p.ldq + c * 8) : z;
            const int off = r * 256 + ((((c >> 1) ^ tn_f(r)) << 5) | ((c & 1) << 4));
            *(u32x4*)(sp + off) = vp;
            *(u32x4*)(sq + off) = vq;
        }
    };

    // per-lane LDS byte addresses of the transposed-read blocks (k-step 0, rows 8g+q; +1024 B = rows +4,
    // +8192 B = k-step 1, + stage / operand offsets are immediates): lane 4q+p of a 16-lane group supplies
    // row q, columns 4p..4p+3 of a 4x16 block
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    const int g = lane >> 4, tq = (lane & 15) >> 2, tp = lane & 3;
    const int r0 = 8 * g + tq;
    unsigned pb[4], qb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        pb[i] = lds0 + r0 * 256 + (((wi * 4 + i) ^ tn_f(r0)) << 5) + tp * 8;
        qb[i] = lds0 + r0 * 256 + (((wj * 4 + i) ^ tn_f(r0)) << 5) + tp * 8;
    }

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // full 64-row steps: LDS-DMA, double buffered, the load of step t+1 in flight under the MFMAs of step t
    const int st_full = min(st1, M / BK);             // steps [st0, st_full) have all 64 rows valid
    if (st0 < st_full) {
        stage_dma(0, st0 * BK);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int st = st0; st < st_full; st += 2) {
            if (st + 1 < st_full) stage_dma(1, (st + 1) * BK);
            tn_compute<0>(pb, qb, acc);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (st + 1 >= st_full) break;
            if (st + 2 < st_full) stage_dma(0, (st + 2) * BK);
            tn_compute<1>(pb, qb, acc);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
    }
    // the (at most one) step that crosses M: zero-filled through registers
    for (int st = max(st0, st_full); st < st1; ++st) {
        stage_regs(0, st * BK);
        __syncthreads();
        tn_compute<0>(pb, qb, acc);
        __syncthreads();
    }

#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int gi = i0 + wi * 64 + i * 16 + (lane & 15);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int gj = j0 + wj * 64 + j * 16 + (lane >> 4) * 4;
            const f32x4 v = acc[i][j];
            if (gi >= p.Ni || gj >= p.Nj) continue;          // half-empty last tile (its operand columns held other rows' data)
            if (p.splits > 1) {
                *(f32x4*)(p.slab + ((long)split * p.Ni + gi) * p.Nj + gj) = v;
            } else {
                float* dst;
                if (gi < p.split_row) { if (gi >= p.rows0) continue; dst = p.C0 + (long)gi * p.ldc + gj; }
                else { if (gi - p.split_row >= p.rows1) continue; dst = p.C1 + (long)(gi - p.split_row) * p.ldc + gj; }
                f32x4 o = *(f32x4*)dst;
                o += v;
                *(f32x4*)dst = o;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// TN kernel, 256x256 tile, 8 waves, staggered half-phases (large wgrads; see gemm_nt256_kernel for the
// schedule and its hazard analysis - identical here).  Operand tiles are [64 m][256 cols] (512-B rows),
// every fragment comes from two ds_read_b64_tr_b16 (inline asm, so hipcc does not drain the LDS-DMA in
// front of them); their completion is awaited explicitly after the barrier that opens each COMPUTE segment.
// A partial last step and device-side row ranges are handled by the buffer bounds (zero fill).
// ---------------------------------------------------------------------------------------------
template <int OFF>
__device__ __forceinline__ bf16x8 tr_pair(unsigned a) {     // rows r0..r0+3 and r0+4..r0+7 of one 16-column block
    return join8(tr_read<OFF>(a), tr_read<OFF + 4 * 512>(a));
}

__global__ __launch_bounds__(512, 2) void gemm_tn256_kernel(TNArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, wc = wave & 3;

    int M = p.M;
    long moff = 0;
    if (p.m_range) { moff = p.m_range[0]; M = min(M, p.m_range[1]); }
    const int tiles_j = (p.Nj + 255) / 256, ntile = ((p.Ni + 255) / 256) * tiles_j;      // a last tile may be half empty
    // Workgroups b, b + 8, .. share an XCD (round-robin dispatch): each XCD takes a CONTIGUOUS run of (split, tile) pairs,
    // so the tiles of one split - which stream the same 64-row steps of P and Q - run side by side on one L2 and those
    // rows are fetched from HBM once per XCD-resident split instead of once per tile (speed only, never correctness).
#ifndef TN_MAP
#define TN_MAP 1
#endif
    const int lin = TN_MAP ? xcd_remap(blockIdx.x, gridDim.x) : (int)blockIdx.x;
    const int split = lin / ntile;
    const int t = TN_MAP ? lin % ntile : xcd_remap(lin % ntile, ntile);
    const int i0 = (t / tiles_j) * 256, j0 = (t % tiles_j) * 256;
    const int nsteps = (M + BK - 1) / BK;
    const int per = (nsteps + p.splits - 1) / p.splits;
    const int st0 = split * per, st1 = min(nsteps, st0 + per);

    // LDS-DMA by buffer loads.  Every 64-row step gets its own pair of descriptors, based at the step's first row and
    // sized to its valid rows: the rows of a last partial step read as zeros (they must contribute nothing - no tail
    // path), and the 32-bit buffer offsets only ever span 64 rows (operands beyond 4 GiB, e.g. the logit gradients, are
    // fine).  The lane's piece offsets are loop-invariant.  Wave instruction (wave*4 + j) fills tile rows 2(wave*4+j), +1
    // (512 B each); 32-byte chunk swizzle on the source.
    const bf16_t* Pb = p.P + moff * p.ldp + i0;
    const bf16_t* Qb = p.Q + moff * p.ldq + j0;
    unsigned p_off[4], q_off[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = 2 * (wave * 4 + j) + (lane >> 5), s16 = lane & 31;
        const int c = (((s16 >> 1) ^ tn_f(r)) << 1) | (s16 & 1);
        p_off[j] = (unsigned)(r * p.ldp * 2 + c * 16);
        q_off[j] = (unsigned)(r * p.ldq * 2 + c * 16);
    }
    // Q (the saved forward activation, HBM-cold) sits in a 3-deep ring and is fetched two steps ahead, P (the gradient,
    // written just before) in a 2-deep ring one step ahead: 3 x 32 + 2 x 32 KiB = the whole LDS.  Per wave and step the
    // four P pieces are issued before the four Q pieces, so vmcnt(4) means P(g+1) and the older Q(g+1) have landed.
    auto dmaP = [&](int slot, int m0, int j) {
        const int rows = min(BK, M - m0);
        const __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(Pb + (long)m0 * p.ldp), 0, (int)(unsigned)((long)(rows - 1) * p.ldp * 2 + 2 * min(256, p.Ni - i0)), 0x00020000);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(prs, (__attribute__((address_space(3))) void*)(smem + (3 + slot) * RA_BYTES + (wave * 4 + j) * 1024),
                                                 16, p_off[j], 0, 0, 0);
    };
    auto dmaQ = [&](int slot, int m0, int j) {
        const int rows = min(BK, M - m0);
        const __amdgpu_buffer_rsrc_t qrs = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(Qb + (long)m0 * p.ldq), 0, (int)(unsigned)((long)(rows - 1) * p.ldq * 2 + 2 * min(256, p.Nj - j0)), 0x00020000);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(qrs, (__attribute__((address_space(3))) void*)(smem + slot * RA_BYTES + (wave * 4 + j) * 1024),
                                                 16, q_off[j], 0, 0, 0);
    };
    // transposed-read addresses (slot 0 of each ring): lane 4q+p of a 16-lane group supplies row q, columns 4p..4p+3 of a
    // 4x16 block; 16-column block ci of row r sits at 32-byte chunk ci ^ tn_f(r), and tn_f is the same for r0, r0+4, r0+32
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    const int g = lane >> 4, tq = (lane & 15) >> 2, tp = lane & 3;
    const int r0 = 8 * g + tq, f = tn_f(r0);
    unsigned pa[8], qa[4];
#pragma unroll
    for (int i = 0; i < 8; ++i) pa[i] = lds0 + 3 * RA_BYTES + r0 * 512 + ((grp * 8 + (i ^ f)) << 5) + tp * 8;
#pragma unroll
    for (int i = 0; i < 4; ++i) qa[i] = lds0 + r0 * 512 + (((wc * 4 + i) ^ f) << 5) + tp * 8;

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 pf[4][2], qf[2][2];

    if (st0 < st1) {
        // prologue: Q(0), P(0), then Q(1); the first two must have landed
#pragma unroll
        for (int j = 0; j < 4; ++j) dmaQ(0, st0 * BK, j);
#pragma unroll
        for (int j = 0; j < 4; ++j) dmaP(0, st0 * BK, j);
        if (st0 + 1 < st1) {
#pragma unroll
            for (int j = 0; j < 4; ++j) dmaQ(1, (st0 + 1) * BK, j);
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        bar_pinned();
        if (grp == 1) bar_pinned();

#define TLOAD_P(QM, SO)                                                                                  \
    _Pragma("unroll") for (int mi = 0; mi < 4; ++mi) {                                                   \
        pf[mi][0] = tr_pair<0>(pa[(QM) * 4 + mi] + (SO));                                                \
        pf[mi][1] = tr_pair<32 * 512>(pa[(QM) * 4 + mi] + (SO));                                         \
    }
#define TLOAD_Q(QV, QN, SO)                                                                              \
    _Pragma("unroll") for (int ni = 0; ni < 2; ++ni) {                                                   \
        QV[ni][0] = tr_pair<0>(qa[(QN) * 2 + ni] + (SO));                                                \
        QV[ni][1] = tr_pair<32 * 512>(qa[(QN) * 2 + ni] + (SO));                                         \
    }
#define TCOMPUTE(QV, QM, QN)                                                                             \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                   \
    __builtin_amdgcn_sched_barrier(0);                                                                   \
    __builtin_amdgcn_s_setprio(1);                                                                       \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) _Pragma("unroll") for (int mi = 0; mi < 4; ++mi)    \
        _Pragma("unroll") for (int ni = 0; ni < 2; ++ni)                                                 \
            acc[(QM) * 4 + mi][(QN) * 2 + ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(                 \
                QV[ni][ks], pf[mi][ks], acc[(QM) * 4 + mi][(QN) * 2 + ni], 0, 0, 0);                     \
    __builtin_amdgcn_s_setprio(0);

        // the Q fragments of column half 0 serve quadrants (0,0) and (1,0): they stay in registers (q0) from phase 0 to
        // phase 3 instead of being read from LDS twice
        bf16x8 q0[2][2];
        int sq = 0, spn = 1, sqn = 2;          // Q slot of this step; P slot / Q slot the DMA of this step fills
        for (int st = st0; st < st1; ++st) {
            const unsigned soP = (unsigned)(spn ^ 1) * RA_BYTES, soQ = (unsigned)sq * RA_BYTES;
            const int m1 = (st + 1) * BK, m2 = (st + 2) * BK;
            const bool fp = st + 1 < st1, fq = st + 2 < st1;
            TLOAD_P(0, soP) TLOAD_Q(q0, 0, soQ)
            // (DMA pieces per load phase: all four of P(g+1) - the 2-deep ring, needed next step - in the first phase a group may
            //  touch that slot, then Q(g+2) 2 + 2: 4 / 2 / 2 / 0 for group 1, 0 / 4 / 2 / 2 for group 0.  +1 ... 2 % over 3 / 3 / 2 / 0;
            //  the 2 / 2 / 2 / 2 spread that helps gemm_nt256_kernel measured 2.7 % SLOWER here: profiles/r04_gemm_dma_schedules.log)
            if (grp == 1 && fp) { dmaP(spn, m1, 0); dmaP(spn, m1, 1); dmaP(spn, m1, 2); dmaP(spn, m1, 3); }
            bar_pinned();
            TCOMPUTE(q0, 0, 0)
            bar_pinned();
            TLOAD_Q(qf, 1, soQ)
            if (grp == 1) { if (fq) { dmaQ(sqn, m2, 0); dmaQ(sqn, m2, 1); } }
            else if (fp) { dmaP(spn, m1, 0); dmaP(spn, m1, 1); dmaP(spn, m1, 2); dmaP(spn, m1, 3); }
            bar_pinned();
            TCOMPUTE(qf, 0, 1)
            bar_pinned();
            TLOAD_P(1, soP)
            if (fq) { if (grp == 1) { dmaQ(sqn, m2, 2); dmaQ(sqn, m2, 3); } else { dmaQ(sqn, m2, 0); dmaQ(sqn, m2, 1); } }
            bar_pinned();
            TCOMPUTE(qf, 1, 1)
            bar_pinned();
            if (grp == 0 && fq) { dmaQ(sqn, m2, 2); dmaQ(sqn, m2, 3); }
            if (grp == 1) { if (fq) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
            bar_pinned();
            TCOMPUTE(q0, 1, 0)
            if (grp == 0) { if (fq) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
            bar_pinned();
            sq = sq == 2 ? 0 : sq + 1;
            sqn = sqn == 2 ? 0 : sqn + 1;
            spn ^= 1;
        }
#undef TLOAD_P
#undef TLOAD_Q
#undef TCOMPUTE
        if (grp == 0) bar_pinned();
    }

    // epilogue: lane owns output row gi (i index), 4 consecutive columns gj..gj+3
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int gi = i0 + grp * 128 + i * 16 + (lane & 15);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int gj = j0 + wc * 64 + j * 16 + (lane >> 4) * 4;
            const f32x4 v = acc[i][j];
            if (gi >= p.Ni || gj >= p.Nj) continue;          // half-empty last tile (its operand columns held other rows' data)
            if (p.splits > 1) {
                *(f32x4*)(p.slab + ((long)split * p.Ni + gi) * p.Nj + gj) = v;
            } else {
                float* dst;
                if (gi < p.split_row) { if (gi >= p.rows0) continue; dst = p.C0 + (long)gi * p.ldc + gj; }
                else { if (gi - p.split_row >= p.rows1) continue; dst = p.C1 + (long)(gi - p.split_row) * p.ldc + gj; }
                f32x4 o = *(f32x4*)dst;
                o += v;
                *(f32x4*)dst = o;
            }
        }
    }
}

// C[row][:] += sum_s slab[s][row][:]   (deterministic split-K combine)
__global__ void tn_reduce_kernel(const float* slab, float* C0, float* C1, long ldc, int split_row, int rows0,
                                 int rows1, int Ni, int Nj, int splits) {
    const int nj4 = Nj >> 2;
    const long total = (long)Ni * nj4;
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int gi = (int)(idx / nj4), gj = (int)(idx % nj4) * 4;
        float* dst;
        if (gi < split_row) { if (gi >= rows0) continue; dst = C0 + (long)gi * ldc + gj; }
        else { if (gi - split_row >= rows1) continue; dst = C1 + (long)(gi - split_row) * ldc + gj; }
        f32x4 o = *(f32x4*)dst;
        for (int s = 0; s < splits; ++s) o += *(const f32x4*)(slab + ((long)s * Ni + gi) * Nj + gj);
        *(f32x4*)dst = o;
    }
}

constexpr int NT_WGS = 512;                       // persistent 128x128 grid: 2 workgroups (80 KiB LDS each) per CU x 256 CUs
constexpr long TN256_MIN_AREA = 512L * 1024L;     // smallest Ni*Nj sent to the 256x256 wgrad kernel
// The ONLY process-wide state of the library: the tile-family selector of ego_gemm_kernel_mode (a test / tuning hook:
// 1 = by shape, 0 = 128x128 kernels only, 2 = 256x256 wherever legal) and the one-time "LDS attributes set" latch.
int g_nt256 = 1;
int g_tn256 = 1;
int g_nt64_tiles = 400;      // NT launches of at most this many 128x128 tiles run on 64x64 tiles instead (0 = never; ego_gemm_small_tiles)
int g_nt_dephase = 0;        // timing probe of the persistent 256 x 256 NT kernel (ego_gemm_tune key 1); 0 = product behaviour
int g_ntl_force = 0;         // probe (ego_gemm_tune key 2): 1 = every non-256 NT launch on 128 x 64 tiles, 2 = on 128 x 128 (3-deep ring)
int g_ntl_tiles = 0;         // NT launches of at most this many 128 x 128 tiles (and more than the 64 x 64 threshold) take the 128 x 64 low-latency kernel
bool g_attr_done = false;
void ensure_attrs() {
    if (g_attr_done) return;
    (void)hipFuncSetAttribute((const void*)gemm_nt_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, NT_LDS);
    (void)hipFuncSetAttribute((const void*)gemm_nt64_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, NT64_LDS);
    (void)hipFuncSetAttribute((const void*)gemm_ntl_kernel<128, 64, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, NTL128x64_LDS);
    (void)hipFuncSetAttribute((const void*)gemm_ntl_kernel<128, 128, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, NTL128x128_LDS);
    (void)hipFuncSetAttribute((const void*)gemm_tn_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS);
    (void)hipFuncSetAttribute((const void*)gemm_nt256_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, NT3_LDS);
    (void)hipFuncSetAttribute((const void*)gemm_nt256_kernel<0, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, NT3_LDS);
    (void)hipFuncSetAttribute((const void*)gemm_nt256_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, NT3_LDS);
    (void)hipFuncSetAttribute((const void*)gemm_nt256_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, NT3_LDS);
    (void)hipFuncSetAttribute((const void*)gemm_nt256_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, NT3_LDS);
    (void)hipFuncSetAttribute((const void*)gemm_nt256_kernel<0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, NT3_LDS);
    (void)hipFuncSetAttribute((const void*)gemm_nt256_kernel<1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, NT3_LDS);
    (void)hipFuncSetAttribute((const void*)gemm_nt256_kernel<3, true>, hipFuncAttributeMaxDynamicSharedMemorySize, NT3_LDS);
    (void)hipFuncSetAttribute((const void*)gemm_tn256_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, NT3_LDS);
    g_attr_done = true;
}

bool tn256_legal(int Ni, int Nj, long ldp, long ldq) {
    return (Ni % 128 == 0) && (Nj % 128 == 0) && 64L * ldp * 2 < 0x7ff00000L && 64L * ldq * 2 < 0x7ff00000L;
}

}  // namespace

extern "C" int ego_gemm_kernel_mode(int nt256, int tn256) {
    // which tile family the two GEMM entries may pick: 1 = by shape (default), 0 = 128x128 kernels only, 2 = 256x256
    // wherever legal.  Results are the same up to fp32 summation order; used by the full-size cross-check test.
    if (nt256 < 0 || nt256 > 2 || tn256 < 0 || tn256 > 2) return EGO_ERR_ARG;
    ensure_attrs();
    g_nt256 = nt256;
    g_tn256 = tn256;
    return EGO_OK;
}

extern "C" int ego_gemm_small_tiles(int max_tiles128) {
    // tuning / test hook like ego_gemm_kernel_mode: NT launches of at most `max_tiles128` 128x128 tiles use the 64x64-tile
    // small-grid kernel (0 = never).  Returns the previous value.
    if (max_tiles128 < 0) return g_nt64_tiles;
    const int old = g_nt64_tiles;
    g_nt64_tiles = max_tiles128;
    return old;
}

extern "C" int ego_gemm_tune(int key, int value) {
    // probe hook (no reference counterpart; results never change): key 1 = start delay, in units of ~0.5 us, of every other
    // persistent 256 x 256 NT workgroup of an XCD.  Returns the previous value, value < 0 only queries.
    // key 2 = force the low-latency tile family on every launch the 256 x 256 kernel does not take (1: 128 x 64, 2: 128 x 128);
    // key 3 = largest 128 x 128 tile count sent to the 128 x 64 low-latency kernel (the product threshold).
    int* v = key == 1 ? &g_nt_dephase : key == 2 ? &g_ntl_force : key == 3 ? &g_ntl_tiles : nullptr;
    if (!v) return -1;
    const int old = *v;
    if (value >= 0) *v = value;
    return old;
}

extern "C" int ego_gemm_nt_bf16(const void* A, long lda, const void* B, long ldb, void* C, long ldc,
                                const float* R, long ldr, const float* bias, const int* m_range,
                                int M, int N, int K, int epi, hipStream_t stream) {
    if (M <= 0 || N <= 0) return EGO_OK;
    if (K <= 0 || K % BK || N % 8 || lda % 8 || ldb % 8 || ldc % 4 || epi < 0 || epi > EGO_EPI_BIAS_RESID) return EGO_ERR_ARG;
    if (epi == EGO_EPI_BF16 && ldc % 8) return EGO_ERR_ARG;
    if ((epi == EGO_EPI_RESID || epi == EGO_EPI_BIAS_RESID) && (!R || ldr % 4)) return EGO_ERR_ARG;
    if (epi == EGO_EPI_BIAS_RESID && !bias) return EGO_ERR_ARG;
    ensure_attrs();
    NTArgs a{(const bf16_t*)A, lda, (const bf16_t*)B, ldb, C, ldc, R, ldr, bias, m_range, M, N, K, epi, nullptr, 0, nullptr, 0, nullptr, nullptr, g_nt_dephase};
    const int tiles256 = ((M + 255) / 256) * ((N + 255) / 256);
    // The persistent 256x256 kernel runs one workgroup per CU.  Measured on MI355X (tools/gemm_bench.py, EGO_GEMM_NT256=2
    // forces it): it wins when the tiles fill the 256 CUs for about three rounds or more, and for deep K already from a
    // partly filled single round on (dgrad of the logits: 1030-1110 vs 800-950 TF/s); few tiles with K = 768 and the
    // residual epilogue stay on the 128x128 kernel (two workgroups per CU hide each other's epilogue).
    const bool legal256 = N % 128 == 0 && K >= 2 * BK && 256L * lda * 2 < 0x7ff00000L && (long)N * ldb * 2 < 0xfff00000L;
    const bool big = legal256 && (g_nt256 == 2 || (g_nt256 == 1 && (tiles256 >= 640 || (K >= 2048 && tiles256 >= 160))));
    if (big) {
        if (epi == EGO_EPI_BF16 && NT256_STRIPS && (N + 255) / 256 >= 32) {              // the 64,000-token logits: column-strip walk
            EGO_LAUNCH((gemm_nt256_kernel<0, false, true>), dim3(tiles256 < 256 ? tiles256 : 256), dim3(512), NT3_LDS, stream, a);
        } else if (epi == EGO_EPI_BF16) { EGO_LAUNCH(gemm_nt256_kernel<0>, dim3(tiles256 < 256 ? tiles256 : 256), dim3(512), NT3_LDS, stream, a); }
        else { EGO_LAUNCH(gemm_nt256_kernel<1>, dim3(tiles256 < 256 ? tiles256 : 256), dim3(512), NT3_LDS, stream, a); }
        LAUNCH_CHECK();
        return EGO_OK;
    }
    const int tiles = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
    // Grids of one to a few rounds (the generation path: 1707 ... 11948 rows): which family is fastest was measured per shape
    // (tools/gen_gemm_sweep.py, profiles/r05_gen_gemm_sweep_*.log) and follows three rules - all families give the same bits:
    //   * narrow outputs (N <= 768: q, proj) and the fp32 / residual epilogues: 64 x 64 tiles up to ~100 tiles of 128 x 128, then
    //     128 x 64 tiles on the 3-deep ring up to ~520 (their register epilogues store whole 16-byte fp32 pieces; the two-stage
    //     128 x 128 kernel pays an exposed DMA round trip per K-step and a slow ragged last row tile: 36 vs 25 us at 8534 rows);
    //   * wider bf16 outputs: 128 x 64 tiles up to ~330 tiles of 128 x 128 (kv at <= 3414 rows, qkv at 1707);
    //   * beyond that the launch time is (rounds) x (time of one tile) for either persistent kernel - a 256 x 256 tile takes about
    //     1.85 x a 128 x 128 tile of the two-stage kernel at two workgroups per CU - so the 256 x 256 kernel is taken when
    //     1.85 ceil(tiles256 / 256) <= ceil(tiles128 / 512): e.g. fc1||fc3 at 3414 rows 224 tiles = ONE round (25 vs 29 us), at
    //     5120 rows 320 tiles = two rounds against three (45 vs 41 us: the 128 x 128 kernel stays).
    // Device-side row ranges (m_range: the host M is only an upper bound) keep the round-4 choice.
    if (g_nt256 == 1 && g_ntl_force == 0 && g_ntl_tiles == 0 && !m_range && g_nt64_tiles == 400) {
        const bool narrow = N <= 768 || epi != EGO_EPI_BF16;
        int pick = 0;                                   // 0: round-4 rule below, 1: 64 x 64, 2: 128 x 64, 3: 256 x 256
        if (narrow) pick = tiles <= 100 ? 1 : tiles <= 520 ? 2 : 0;
        else if (tiles <= 330) pick = 2;
        else if (legal256 && 185 * ((tiles256 + 255) / 256) <= 100 * ((tiles + 511) / 512)) pick = 3;
        if (pick == 1) {
            EGO_LAUNCH(gemm_nt64_kernel, dim3(((M + 63) / 64) * ((N + 63) / 64)), dim3(256), NT64_LDS, stream, a);
            LAUNCH_CHECK();
            return EGO_OK;
        }
        if (pick == 2) {
            EGO_LAUNCH((gemm_ntl_kernel<128, 64, 3>), dim3(((M + 127) / 128) * ((N + 63) / 64)), dim3(256), NTL128x64_LDS, stream, a);
            LAUNCH_CHECK();
            return EGO_OK;
        }
        if (pick == 3) {
            if (epi == EGO_EPI_BF16) { EGO_LAUNCH(gemm_nt256_kernel<0>, dim3(tiles256 < 256 ? tiles256 : 256), dim3(512), NT3_LDS, stream, a); }
            else { EGO_LAUNCH(gemm_nt256_kernel<1>, dim3(tiles256 < 256 ? tiles256 : 256), dim3(512), NT3_LDS, stream, a); }
            LAUNCH_CHECK();
            return EGO_OK;
        }
    }
    if (g_ntl_force == 2) {
        EGO_LAUNCH((gemm_ntl_kernel<128, 128, 3>), dim3(tiles), dim3(256), NTL128x128_LDS, stream, a);
        LAUNCH_CHECK();
        return EGO_OK;
    }
    if (g_ntl_force == 1 || (tiles > g_nt64_tiles && tiles <= g_ntl_tiles)) {
        // one to a few rounds of 128 x 128 tiles (the generation path's encoder linears): 128 x 64 tiles on a 3-deep ring
        EGO_LAUNCH((gemm_ntl_kernel<128, 64, 3>), dim3(((M + 127) / 128) * ((N + 63) / 64)), dim3(256), NTL128x64_LDS, stream, a);
        LAUNCH_CHECK();
        return EGO_OK;
    }
    // under-filled grids (the generation path's 1707-row linears): 64x64 tiles, four times the workgroups (gemm_nt64_kernel).
    // With a device-side row range the host M is only an upper bound: the grid is sized for it, surplus workgroups exit.
    if (g_nt64_tiles > 0 && tiles <= g_nt64_tiles) {
        const int tiles64 = ((M + 63) / 64) * ((N + 63) / 64);
        EGO_LAUNCH(gemm_nt64_kernel, dim3(tiles64), dim3(256), NT64_LDS, stream, a);
        LAUNCH_CHECK();
        return EGO_OK;
    }
    EGO_LAUNCH(gemm_nt_kernel, dim3(tiles < NT_WGS ? tiles : NT_WGS), dim3(256), NT_LDS, stream, a);
    LAUNCH_CHECK();
    return EGO_OK;
}

extern "C" int ego_gemm_nt_swiglu_fwd(const void* X, long ldx, const void* W13, long ldw, void* ab, long ld_ab, void* h, long ld_h,
                                      int M, int F, int K, hipStream_t stream) {
    if (M <= 0) return EGO_OK;
    // only the persistent 256x256 kernel carries this epilogue; the caller falls back to gemm + ego_swiglu_fwd otherwise
    if (F % 128 || K % BK || K < 2 * BK || ldx % 8 || ldw % 8 || ld_ab % 8 || ld_h % 8 || ld_ab < 2L * F || ld_h < F) return EGO_ERR_ARG;
    if (256L * ldx * 2 >= 0x7ff00000L || 2L * F * ldw * 2 >= 0xfff00000L) return EGO_ERR_ARG;
    ensure_attrs();
    NTArgs a{(const bf16_t*)X, ldx, (const bf16_t*)W13, ldw, ab, ld_ab, nullptr, 0, nullptr, nullptr, M, F, K, EPI_SWIGLU_FWD,
             nullptr, 0, (bf16_t*)h, ld_h, nullptr, nullptr, g_nt_dephase};
    const int tiles = ((M + 255) / 256) * (F / 128);
    EGO_LAUNCH(gemm_nt256_kernel<3>, dim3(tiles < 256 ? tiles : 256), dim3(512), NT3_LDS, stream, a);
    LAUNCH_CHECK();
    return EGO_OK;
}

// ---- fp8 (OCP e4m3) forward linears: BASELINE config 5 ("bf16 + fp8 MFMA GEMMs") --------------------------------------
extern "C" int ego_gemm_nt_fp8(const void* A8, long lda, const float* sa, const void* B8, long ldb, const float* sb, void* C, long ldc,
                               const float* R, long ldr, const float* bias, int M, int N, int K, int epi, hipStream_t stream) {
    if (M <= 0 || N <= 0) return EGO_OK;
    // one kernel family only (persistent 256x256): K in whole 128-byte tiles, at least two of them
    if (K < 256 || K % 128 || N % 128 || lda % 16 || ldb % 16 || ldc % 4 || epi < 0 || epi > EGO_EPI_BIAS_RESID || !sa || !sb) return EGO_ERR_ARG;
    if (epi == EGO_EPI_BF16 && ldc % 8) return EGO_ERR_ARG;
    if ((epi == EGO_EPI_RESID || epi == EGO_EPI_BIAS_RESID) && (!R || ldr % 4)) return EGO_ERR_ARG;
    if (epi == EGO_EPI_BIAS_RESID && !bias) return EGO_ERR_ARG;
    if (256L * lda >= 0x7ff00000L || (long)N * ldb >= 0xfff00000L) return EGO_ERR_ARG;
    ensure_attrs();
    NTArgs a{(const bf16_t*)A8, lda, (const bf16_t*)B8, ldb, C, ldc, R, ldr, bias, nullptr, M, N, K, epi, nullptr, 0, nullptr, 0, sa, sb, g_nt_dephase};
    const int tiles256 = ((M + 255) / 256) * ((N + 255) / 256);
    if (epi == EGO_EPI_BF16) { EGO_LAUNCH((gemm_nt256_kernel<0, true>), dim3(tiles256 < 256 ? tiles256 : 256), dim3(512), NT3_LDS, stream, a); }
    else { EGO_LAUNCH((gemm_nt256_kernel<1, true>), dim3(tiles256 < 256 ? tiles256 : 256), dim3(512), NT3_LDS, stream, a); }
    LAUNCH_CHECK();
    return EGO_OK;
}

extern "C" int ego_gemm_nt_swiglu_fwd_fp8(const void* X8, long ldx, const float* sx, const void* W13_8, long ldw, const float* sw,
                                          void* ab, long ld_ab, void* h, long ld_h, int M, int F, int K, hipStream_t stream) {
    if (M <= 0) return EGO_OK;
    if (F % 128 || K % 128 || K < 256 || ldx % 16 || ldw % 16 || ld_ab % 8 || ld_h % 8 || ld_ab < 2L * F || ld_h < F || !sx || !sw) return EGO_ERR_ARG;
    if (256L * ldx >= 0x7ff00000L || 2L * F * ldw >= 0xfff00000L) return EGO_ERR_ARG;
    ensure_attrs();
    NTArgs a{(const bf16_t*)X8, ldx, (const bf16_t*)W13_8, ldw, ab, ld_ab, nullptr, 0, nullptr, nullptr, M, F, K, EPI_SWIGLU_FWD,
             nullptr, 0, (bf16_t*)h, ld_h, sx, sw, g_nt_dephase};
    const int tiles = ((M + 255) / 256) * (F / 128);
    EGO_LAUNCH((gemm_nt256_kernel<3, true>), dim3(tiles < 256 ? tiles : 256), dim3(512), NT3_LDS, stream, a);
    LAUNCH_CHECK();
    return EGO_OK;
}

extern "C" int ego_gemm_nt_swiglu_bwd(const void* dY, long ldy, const void* W2t, long ldw, const void* ab, void* dab, long ld_ab,
                                      int M, int F, int K, hipStream_t stream) {
    if (M <= 0) return EGO_OK;
    // only the persistent 256x256 kernel carries this epilogue; the caller falls back to gemm + ego_swiglu_bwd otherwise
    if (F % 256 || K % BK || K < 2 * BK || ldy % 8 || ldw % 8 || ld_ab % 8 || ld_ab < 2L * F) return EGO_ERR_ARG;
    if (256L * ldy * 2 >= 0x7ff00000L || (long)F * ldw * 2 >= 0xfff00000L) return EGO_ERR_ARG;
    ensure_attrs();
    NTArgs a{(const bf16_t*)dY, ldy, (const bf16_t*)W2t, ldw, dab, ld_ab, nullptr, 0, nullptr, nullptr, M, F, K, EPI_SWIGLU_BWD,
             (const bf16_t*)ab, ld_ab, nullptr, 0, nullptr, nullptr, g_nt_dephase};
    const int tiles256 = ((M + 255) / 256) * (F / 256);
    EGO_LAUNCH(gemm_nt256_kernel<2>, dim3(tiles256 < 256 ? tiles256 : 256), dim3(512), NT3_LDS, stream, a);
    LAUNCH_CHECK();
    return EGO_OK;
}

extern "C" int ego_gemm_tn_plan(int Ni, int Nj, int M, long ldp, long ldq, long slab_elems, int ranged) {
    // Split-K factor ego_gemm_tn_bf16 should be called with: one full round of workgroups and no more (each split
    // writes an fp32 [Ni, Nj] slab).  Device-side row ranges are not split (their length is unknown on the host).
    if (ranged || Ni <= 0 || Nj <= 0 || M <= 0) return 1;
    const long steps = (M + BK - 1) / BK, cap = slab_elems / ((long)Ni * Nj);
    long s;
    if (g_tn256 != 0 && tn256_legal(Ni, Nj, ldp, ldq) && (long)Ni * Nj >= TN256_MIN_AREA)
        s = 256 / (((Ni + 255) / 256) * ((Nj + 255) / 256));               // 256x256 kernel: one workgroup per CU
    else
        s = NT_WGS / (((Ni + BM - 1) / BM) * ((Nj + BN - 1) / BN));        // 128x128 kernel: two per CU
    s = s < steps ? s : steps;
    s = s < cap ? s : cap;
    return (int)(s < 1 ? 1 : s);
}

extern "C" int ego_gemm_tn_bf16(const void* P, long ldp, const void* Q, long ldq, float* C0, float* C1, long ldc,
                                int split_row, int rows0, int rows1, const int* m_range, int Ni, int Nj, int M,
                                int splits, float* slab, hipStream_t stream) {
    if (M <= 0) return EGO_OK;
    if (Ni % BM || Nj % BN || ldp % 8 || ldq % 8 || ldc % 4 || splits < 1) return EGO_ERR_ARG;
    if (splits > 1 && !slab) return EGO_ERR_ARG;
    if (!C1) { split_row = Ni; rows1 = 0; }
    ensure_attrs();
    TNArgs a{(const bf16_t*)P, ldp, (const bf16_t*)Q, ldq, C0, C1, ldc, slab, m_range, split_row, rows0, rows1, Ni, Nj, M, splits};
    // 256x256 staggered kernel, one workgroup per CU
    const bool legal256 = tn256_legal(Ni, Nj, ldp, ldq);
    const int tiles256 = ((Ni + 255) / 256) * ((Nj + 255) / 256);
    if (legal256 && (g_tn256 == 2 || (g_tn256 == 1 && tiles256 * splits >= 128 && (long)Ni * Nj >= TN256_MIN_AREA))) {
        EGO_LAUNCH(gemm_tn256_kernel, dim3(tiles256 * splits), dim3(512), NT3_LDS, stream, a);
    } else {
        const int tiles = (Ni / BM) * (Nj / BN);
        EGO_LAUNCH(gemm_tn_kernel, dim3(tiles * splits), dim3(256), GEMM_LDS, stream, a);
    }
    LAUNCH_CHECK();
    if (splits > 1) {
        const long total = (long)Ni * (Nj / 4);
        const int blocks = (int)min((long)2048, (total + 255) / 256);
        EGO_LAUNCH(tn_reduce_kernel, dim3(blocks), dim3(256), 0, stream, slab, C0, C1, ldc, split_row, rows0, rows1, Ni, Nj, splits);
        LAUNCH_CHECK();
    }
    return EGO_OK;
}
