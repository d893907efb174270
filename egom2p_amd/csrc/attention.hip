// Fused attention for head_dim 64 on gfx950 (flash-style, nothing N x N ever reaches HBM).
//
// Replaces Attention.forward / CrossAttention.forward of the reference
// (egom2p/models/egom2p_utils.py:185-205, 222-244): (q k^T) * d^-0.5 -> masked_fill -> softmax -> @ v,
// and their autograd backward.  The reference's boolean masks (encoder key padding,
// egom2p_model.py:389-394; decoder block-diagonal-by-modality mask, :446-481) are passed as one
// allowed key interval [ks, ke) per query row (never materialised).  A row whose interval is empty
// reproduces the reference's masked_fill(-finfo.max)+softmax result: uniform attention over all keys
// (implemented as score scale 0 over [0, Nk)).
//
// Layout choices (32x32x16 bf16 MFMA, wave = 64):
//   * forward / dQ: one wave owns 32 query rows; S^T = K.Q^T is computed with the key index in the
//     accumulator rows and the query on the lane, so the softmax statistics are per-lane scalars and
//     the S^T accumulator registers are directly the B operand of O^T = V^T.P^T (no LDS round trip).
//   * dK/dV: one wave owns 32 keys (key on the lane); P and dS accumulators are directly the
//     B operands of dV^T = dO^T.P and dK^T = Q^T.dS.  dQ is produced by its own query-major kernel,
//     so there are no atomics and the result is bitwise reproducible.
//   * K/V (or Q/dO) tiles of 64 rows x 64 dims are staged through LDS in one XOR-swizzled image that is
//     bank-conflict free for both the ds_read_b128 row reads and the ds_read_b64_tr_b16 transposed reads.
#include "common.h"
#include "egom2p_hip.h"
#include <limits.h>

namespace {

constexpr float LOG2E = 1.4426950408889634f;
constexpr float NEG_BIG = -1.0e30f;
constexpr int TILE_BYTES = 64 * 128;  // 64 rows x 64 bf16

struct AttnArgs {
    const bf16_t* Q; const bf16_t* K; const bf16_t* V;
    long q_bs, q_rs, k_bs, k_rs, v_bs, v_rs;
    bf16_t* O; long o_bs, o_rs;
    bf16_t* Olo;                // optional, laid out like O: bf16(o - bf16(o)), the rounding residual of the stored output
    float* LSE;                 // [B,H,Nq] MINUS the log2-domain log-sum-exp of the scaled scores (the backward's accumulator init)
    const int* ks; const int* ke;  // interval of query row (b,q) at [b*r_bs + q*r_rs]
    long r_bs, r_rs;
    // row groups (self-attention under a block-diagonal mask; optional): seg[b][g] = (first row, row count) of group g of
    // sample b, groups ascending and disjoint; every row of a group attends exactly the group's own row range.  Rows behind
    // the last group (and all rows of a sample with seg_bad[b] != 0) take the per-row intervals ks / ke.
    const int* seg; const int* seg_bad; int n_seg;
    // forward only, optional: the keys of a query tile cut into `kv_splits` runs handled by separate workgroups; each writes its
    // unnormalised O (fp32 [split][B, H, Nq][64]) and (m, l) (fp32 [split][B, H, Nq][2]) to `split_ws`, attn_combine_kernel joins them
    int kv_splits; float* split_o; float* split_ml;
    int B, H, Nq, Nk;
    float scale;
    // backward only
    const bf16_t* dO; long do_bs, do_rs;
    const float* DELTA;         // [B,H,Nq]  MINUS rowsum(dO o O) (read by the dK/dV kernel)
    float* DELTA_OUT;           // same buffer, written by the dQ kernel
    bf16_t* dQ; long dq_bs, dq_rs;
    bf16_t* dK; long dk_bs, dk_rs;
    bf16_t* dV; long dv_bs, dv_rs;
    int stagger;                // probe (ego_attn_tune key 1): start delay, in 512-clock units, of every other 256-workgroup wave of the grid
};
int g_attn_stagger = 0;         // probe state behind ego_attn_tune (0 = product behaviour); results never depend on it

// (pair_tile: workgroup id -> ((batch, head) pair, tile), XCD-aware - common.h)

// Row-group launches: tile index t of sample b -> the group it belongs to.  Groups are cut into 128-row tiles of their own
// (a tile never straddles two groups, so all its rows share one interval: the uniform fast path of the kernels); what is left
// of the tile index space covers the rows behind the last group ("tail": per-row intervals), tile by tile.  Returns false when
// the tile index is beyond the sample's tiles.  g0 / g1 = row range of the group (or of the tail), t = tile inside it.
struct GroupTile { int g0, g1, t; bool uniform; int tail0; };
__device__ __forceinline__ bool group_tile(const AttnArgs& p, int b, int tile, int n_rows, GroupTile& g) {
    int tail0 = 0;
    bool found = false;
    g.uniform = false;
    if (!(p.seg_bad && p.seg_bad[b])) {
        for (int i = 0; i < p.n_seg; ++i) {
            const int s0 = p.seg[((long)b * p.n_seg + i) * 2], c = p.seg[((long)b * p.n_seg + i) * 2 + 1];
            const int nt = (c + 127) >> 7;
            if (!found) {
                if (tile < nt) { found = true; g.g0 = s0; g.g1 = s0 + c; g.t = tile; g.uniform = true; }
                else tile -= nt;
            }
            tail0 = max(tail0, s0 + c);
        }
    }
    g.tail0 = min(tail0, n_rows);
    if (found) return true;
    if (tile >= ((n_rows - g.tail0 + 127) >> 7)) return false;
    g.g0 = g.tail0; g.g1 = n_rows; g.t = tile;
    return true;
}

// byte offset of 16-byte chunk c16 (0..7) of row r in the swizzled [64][64] bf16 image
__device__ __forceinline__ int att_off(int r, int c16) {
    const int v = (((r >> 1) & 1) << 2) | ((r >> 2) & 3);
    return r * 128 + ((c16 ^ v) << 4);
}

// MFMA 32x32x16 operand whose 32 rows are tile rows rb*32..+31 and whose k index is the tile column
// (k-step s covers columns 16s..16s+15): lane (row l&31, half l>>5) reads 8 consecutive columns.
__device__ __forceinline__ bf16x8 row_frag(const char* tile, int rb, int s, int lane) {
    return *(const bf16x8*)(tile + att_off(rb * 32 + (lane & 31), 2 * s + (lane >> 5)));
}

// MFMA 32x32x16 operand whose 32 rows are tile COLUMNS db*32..+31 and whose k index is the tile row,
// in the permuted order an accumulator tile presents its rows (k-step sp covers tile rows 16sp..16sp+15;
// element j of lane half h is row 16sp + 8(j>>2) + 4h + (j&3)).
__device__ __forceinline__ bf16x8 tr_frag(const char* tile, int db, int sp, int lane) {
    const int g = lane >> 4, t = lane & 15, q = t >> 2, p = t & 3, h = g >> 1;
    const int c16 = 4 * db + 2 * (g & 1) + (p >> 1), sub = (p & 1) * 8;
    const int r0 = 16 * sp + 4 * h + q, r1 = r0 + 8;
    s16x4 lo = lds_read_tr16(tile + att_off(r0, c16) + sub);
    s16x4 hi = lds_read_tr16(tile + att_off(r1, c16) + sub);
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}

// The same operand through inline-asm reads: per lane four base addresses [db][lo/hi row] (the swizzle term does not
// depend on sp), k-step sp and the tile's position are immediates.  tr_issue starts the 8 reads of k-steps SP0 and
// SP0+1 (both d blocks); the caller waits with lgkm_wait_tied before joining the halves.
struct TrAddr { unsigned a[2][2]; };
__device__ __forceinline__ TrAddr tr_addr(unsigned lds_base, int lane) {
    const int g = lane >> 4, t = lane & 15, q = t >> 2, p = t & 3, h = g >> 1;
    TrAddr r;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int w = 0; w < 2; ++w)
            r.a[db][w] = lds_base + att_off(4 * h + q + 8 * w, 4 * db + 2 * (g & 1) + (p >> 1)) + (p & 1) * 8;
    return r;
}
template <int TOFF, int SP0>
__device__ __forceinline__ void tr_issue(const TrAddr& A, unsigned stage_off, s16x4 (&v)[4][2]) {
#pragma unroll
    for (int db = 0; db < 2; ++db) {
        v[db][0] = tr_read<TOFF + SP0 * 2048>(A.a[db][0] + stage_off);
        v[db][1] = tr_read<TOFF + SP0 * 2048>(A.a[db][1] + stage_off);
        v[2 + db][0] = tr_read<TOFF + (SP0 + 1) * 2048>(A.a[db][0] + stage_off);
        v[2 + db][1] = tr_read<TOFF + (SP0 + 1) * 2048>(A.a[db][1] + stage_off);
    }
}

__device__ __forceinline__ bf16x8 pack8(const f32x16& a, int x) {
    bf16x8 r;
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = (__bf16)a[8 * x + j];
    return r;
}

// accumulator row index (0..31) of register r for lane half h
__device__ __forceinline__ int acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

#ifndef ATT_WIDE_STORE
#define ATT_WIDE_STORE 1
#endif
// Two 8-dim groups (g0, g1) of a row, 4 dims each held by the lane and 4 by its partner lane ^ 32 (same row, other half):
// one v_permlane32_swap per dword leaves the lower-half lane with all 8 dims of g0 and the upper-half lane with all 8 of
// g1 - one 16-byte store per lane instead of two 8-byte ones (the store tail is issue-bound: half the instructions).
__device__ __forceinline__ void store_pair16(bf16_t* row, int g0, u32x2 a, u32x2 b, int hh) {
    const auto r0 = __builtin_amdgcn_permlane32_swap(a[0], b[0], false, false);
    const auto r1 = __builtin_amdgcn_permlane32_swap(a[1], b[1], false, false);
    const u32x4 v = {r0[0], r1[0], r0[1], r1[1]};
    *(u32x4*)(row + 8 * (g0 + hh)) = v;
}

// store a transposed 64(d) x 32(rows on lanes) accumulator pair as bf16 rows [row][64 d] (rows 16-byte aligned)
__device__ __forceinline__ void store_rows_bf16(bf16_t* dst, const f32x16 (&t)[2], float mul, int hh) {
#pragma unroll
    for (int db = 0; db < 2; ++db) {
        u32x2 o[4];
#pragma unroll
        for (int g = 0; g < 4; ++g)
            o[g] = u32x2{pack_bf16x2(t[db][4 * g] * mul, t[db][4 * g + 1] * mul), pack_bf16x2(t[db][4 * g + 2] * mul, t[db][4 * g + 3] * mul)};
#if ATT_WIDE_STORE
        store_pair16(dst + db * 32, 0, o[0], o[1], hh);
        store_pair16(dst + db * 32, 2, o[2], o[3], hh);
#else
#pragma unroll
        for (int g = 0; g < 4; ++g) *(u32x2*)(dst + db * 32 + 8 * g + 4 * hh) = o[g];
#endif
    }
}

// K / V (or Q / dO) tiles go straight from global memory into LDS (LDS-DMA, no VGPR staging): wave instruction i of
// wave w fills tile rows 8(2w+i)..+7 - lane l lands in row +(l>>3), 16-byte slot l&7, so it fetches the logical chunk
// that the swizzle keeps in that slot.
// with buffer loads: the (batch, head) slice is a raw buffer whose size ends with row nrows-1 (rows past it read as
// zeros, like the reference's padding), the lane's two piece offsets are loop-invariant VGPRs, the tile's position
// is the scalar offset and the LDS destination sits in M0 - no vector ALU work per tile.
struct DmaOff { unsigned o[2]; };     // byte offsets of this lane's two pieces inside a 64-row tile
__device__ __forceinline__ DmaOff dma_off(long rs, int wave, int lane) {
    DmaOff d;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = 8 * (2 * wave + i) + (lane >> 3);
        d.o[i] = (unsigned)(r * (int)rs * 2 + (((lane & 7) ^ ((((r >> 1) & 1) << 2) | ((r >> 2) & 3))) << 4));
    }
    return d;
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t slice_rsrc(const bf16_t* base, long rs, int nrows) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)((long)(nrows - 1) * rs * 2 + 128), 0x00020000);
}
__device__ __forceinline__ void dma_tile64(__amdgpu_buffer_rsrc_t rsrc, long rs, const DmaOff& off, int row0, char* tile, int wave) {
    const int soff = row0 * (int)rs * 2;
#pragma unroll
    for (int i = 0; i < 2; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(tile + (2 * wave + i) * 1024), 16,
                                                 off.o[i], soff, 0, 0);
}

// the same rows as two bf16 numbers per element: hi = bf16(x), lo = bf16(x - hi) (x to ~16 significant bits)
__device__ __forceinline__ void store_rows_bf16_hilo(bf16_t* hi, bf16_t* lo, const f32x16 (&t)[2], float mul, int hh) {
#pragma unroll
    for (int db = 0; db < 2; ++db) {
        u32x2 o[4], q[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float x[4], r[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) { x[e] = t[db][4 * g + e] * mul; r[e] = x[e] - round_bf16(x[e]); }
            o[g] = u32x2{pack_bf16x2(x[0], x[1]), pack_bf16x2(x[2], x[3])};
            q[g] = u32x2{pack_bf16x2(r[0], r[1]), pack_bf16x2(r[2], r[3])};
        }
#if ATT_WIDE_STORE
        store_pair16(hi + db * 32, 0, o[0], o[1], hh); store_pair16(hi + db * 32, 2, o[2], o[3], hh);
        store_pair16(lo + db * 32, 0, q[0], q[1], hh); store_pair16(lo + db * 32, 2, q[2], q[3], hh);
#else
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            *(u32x2*)(hi + db * 32 + 8 * g + 4 * hh) = o[g];
            *(u32x2*)(lo + db * 32 + 8 * g + 4 * hh) = q[g];
        }
#endif
    }
}

// ---------------------------------------------------------------------------------------------
// What a query-major workgroup (forward, dQ: 4 waves x 32 query rows) works on: its (batch, head), its rows, their key
// interval(s) and the range of 64-key tiles it has to visit.  Three cases:
//   * one interval per sample (r_rs == 0: encoder self-attention, cross-attention) and row-group tiles (group_tile): every row
//     of the workgroup has the same [ks, ke) - no reductions (24 dependent ds_bpermute round trips per wave otherwise, ~8 % of
//     a workgroup's life); a group's key tiles start at the group's first key (kbase), so only its last tile is partial;
//   * per-row intervals: wave / workgroup extremes by reductions; rows whose interval is empty attend every key with score
//     scale 0 (the reference's masked_fill(-finfo.max) + softmax = uniform attention).
// `scratch`: 64 bytes of LDS.  Returns false (for the whole workgroup) when the tile does not exist.
// ---------------------------------------------------------------------------------------------
struct QTile { int h, b, q0, qhi, qrow, ks, ke, kbase, w_ksmax, w_kemin, kt0, kt1, split; bool flat; };
__device__ __forceinline__ bool q_tile(const AttnArgs& p, char* scratch, QTile& q) {
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int splits = p.kv_splits > 1 ? p.kv_splits : 1;
    const int tiles = ((p.Nq + 127) >> 7) + (p.seg ? p.n_seg + 1 : 0);
    int pair, tile;
    pair_tile(blockIdx.x, p.B * p.H, tiles * splits, p.r_rs == 0 || p.seg != nullptr, pair, tile);
    q.split = tile % splits; tile /= splits;
    q.h = pair % p.H; q.b = pair / p.H;
    bool uniform = p.r_rs == 0;
    int g0 = 0;
    q.qhi = p.Nq;
    if (p.seg) {
        GroupTile g;
        if (!group_tile(p, q.b, tile, p.Nq, g)) return false;
        g0 = g.g0; q.qhi = g.g1; tile = g.t; uniform = g.uniform;
    } else if (tile * 128 >= p.Nq) {
        return false;
    }
    q.q0 = g0 + tile * 128 + wave * 32;
    q.qrow = min(q.q0 + (lane & 31), q.qhi - 1);
    int ks, ke;
    if (p.seg && uniform) { ks = g0; ke = min(q.qhi, p.Nk); }
    else { ks = p.ks[q.b * p.r_bs + q.qrow * p.r_rs]; ke = min(p.ke[q.b * p.r_bs + q.qrow * p.r_rs], p.Nk); }
    q.flat = ke <= ks;
    if (q.flat) { ks = 0; ke = p.Nk; }
    int kmin = ks, kmax = ke;
    q.w_ksmax = ks; q.w_kemin = ke;
    if (!uniform) {
        const int w_lo = wave_min_i(ks), w_hi = wave_max_i(ke);
        q.w_ksmax = wave_max_i(ks); q.w_kemin = wave_min_i(ke);
        int* rng = (int*)scratch;
        if (lane == 0) { rng[wave] = w_lo; rng[4 + wave] = w_hi; }
        __syncthreads();
        kmin = min(min(rng[0], rng[1]), min(rng[2], rng[3]));
        kmax = max(max(rng[4], rng[5]), max(rng[6], rng[7]));
    }
    q.ks = ks; q.ke = ke;
    // (readfirstlane: these feed the scalar offsets of the LDS-DMA - as VGPR values hipcc wraps every DMA in a waterfall loop)
    kmin = __builtin_amdgcn_readfirstlane(kmin); kmax = __builtin_amdgcn_readfirstlane(kmax);
    q.kbase = (p.seg && uniform) ? kmin : 0;
    q.kt0 = (kmin - q.kbase) >> 6; q.kt1 = (kmax - q.kbase + 63) >> 6;
    if (splits > 1) {                                   // this workgroup's run of the key tiles (may be empty)
        const int per = (q.kt1 - q.kt0 + splits - 1) / splits;
        const int a = q.kt0 + q.split * per;
        q.kt1 = min(q.kt1, a + per); q.kt0 = min(a, q.kt1);
    }
    return true;
}

// ---------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int FWD_STAGES = 3;
constexpr float FWD_TAU = 8.f;      // the softmax reference moves when a score exceeds it by more than this (exp2 domain)
__global__ __launch_bounds__(256, 3) void attn_fwd_kernel(AttnArgs p) {
    __shared__ __attribute__((aligned(16))) char smem[FWD_STAGES * 2 * TILE_BYTES + 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    QTile qt;
    if (!q_tile(p, smem + FWD_STAGES * 2 * TILE_BYTES, qt)) return;
    const int h = qt.h, b = qt.b, q0 = qt.q0, qhi = qt.qhi, qrow = qt.qrow, ks = qt.ks, ke = qt.ke, kbase = qt.kbase;
    const int w_ksmax = qt.w_ksmax, w_kemin = qt.w_kemin, kt0 = qt.kt0, kt1 = qt.kt1;
    const int ql = lane & 31, hh = lane >> 5;
    const float sc = qt.flat ? 0.f : p.scale * LOG2E;

    const bf16_t* Qp = p.Q + (long)b * p.q_bs + (long)qrow * p.q_rs + h * 64;
    // Q is pre-scaled by scale * log2(e) (as in the backward kernels), so the scores leave the MFMA in the exp2 domain
    bf16x8 qf[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const bf16x8 qr = *(const bf16x8*)(Qp + 16 * s + 8 * hh);
#pragma unroll
        for (int e = 0; e < 8; ++e) qf[s][e] = (__bf16)((float)qr[e] * sc);
    }

    const bf16_t* Kb = p.K + (long)b * p.k_bs + h * 64;
    const bf16_t* Vb = p.V + (long)b * p.v_bs + h * 64;
    const DmaOff kvoff = dma_off(p.k_rs, wave, lane);               // k_rs == v_rs (checked by the launcher)
    const __amdgpu_buffer_rsrc_t krs = slice_rsrc(Kb, p.k_rs, p.Nk), vrs = slice_rsrc(Vb, p.v_rs, p.Nk);
    auto dma_tile = [&](int kt, int s) {
        dma_tile64(krs, p.k_rs, kvoff, kbase + kt * 64, smem + s * 2 * TILE_BYTES, wave);
        dma_tile64(vrs, p.v_rs, kvoff, kbase + kt * 64, smem + s * 2 * TILE_BYTES + TILE_BYTES, wave);
    };

    f32x16 ot[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) { ot[0][i] = 0.f; ot[1][i] = 0.f; }
    float m = NEG_BIG, l = 0.f;
    bool seeded = false;                 // every lane of the wave has a finite reference m
    const TrAddr tra = tr_addr((unsigned)(size_t)(__attribute__((address_space(3))) char*)smem, lane);

    // K / V ring of FWD_STAGES tiles, filled two tiles ahead; every tile is 4 DMA instructions of this wave, so
    // "at most 4 outstanding" means the older tile has landed.
    if (kt0 < kt1) dma_tile(kt0, 0);
    if (kt0 + 1 < kt1) {
        dma_tile(kt0 + 1, 1);
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    lds_barrier();
    int s_ = 0;
    for (int kt = kt0; kt < kt1; ++kt) {
        const bool ahead = kt + 2 < kt1;
        if (ahead) dma_tile(kt + 2, s_ >= 1 ? s_ - 1 : 2);      // that stage was last read before the previous barrier
        const char* Kt = smem + s_ * 2 * TILE_BYTES;
        const char* Vt = Kt + TILE_BYTES;

        // Online softmax with a LAZY reference: `m` is the value subtracted from the scores, not necessarily their running
        // maximum.  On a tile that lies inside every lane's interval, once every lane has a finite reference, -m is the
        // INITIAL accumulator of the S^T chain (the subtraction is free) and p = exp2(acc) directly; the reference only moves
        // when some score exceeds it by more than FWD_TAU (p <= 2^FWD_TAU, far from any overflow) - after the first tiles
        // that is rare.  This kernel is VALU-bound at head dim 64: 32 exp + ~48 other vector instructions per 16 MFMAs.
        const bool full = (kbase + kt * 64 >= w_ksmax) && (kbase + kt * 64 + 64 <= w_kemin);
        const bool fast = full && seeded;
        const float ini = fast ? -m : 0.f;
        f32x16 st[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int i = 0; i < 16; ++i) st[kb][i] = ini;
#pragma unroll
            for (int s = 0; s < 4; ++s)
                st[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(Kt, kb, s, lane), qf[s], st[kb], 0, 0, 0);
        }
        s16x4 va[4][2], vb[4][2];
        tr_issue<TILE_BYTES, 0>(tra, s_ * 2 * TILE_BYTES, va);
        float mx = NEG_BIG;
        if (full) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) mx = fmaxf(mx, st[kb][r]);
        } else {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int kidx = kbase + kt * 64 + kb * 32 + acc_row(r, hh);
                    const float v = (kidx >= ks && kidx < ke) ? st[kb][r] : NEG_BIG;
                    st[kb][r] = v;
                    mx = fmaxf(mx, v);
                }
        }
        if (fast) {
            // scores are relative to m already; move the reference only where they outgrow it.  The two half-waves hold
            // the two key halves of a query's tile: they only have to agree on the maximum when the reference moves (rare)
            if (__any(mx > FWD_TAU)) {
                mx = xhalf_max(mx);
                const float d = fmaxf(mx, 0.f);
                const float alpha = __builtin_amdgcn_exp2f(-d);
                m += d;
                l *= alpha;
#pragma unroll
                for (int i = 0; i < 16; ++i) { ot[0][i] *= alpha; ot[1][i] *= alpha; st[0][i] -= d; st[1][i] -= d; }
            }
        } else {
            mx = xhalf_max(mx);
            const float mnew = fmaxf(m, mx);
            if (__any(mnew > m)) {
                const float alpha = __builtin_amdgcn_exp2f(m - mnew);
                l *= alpha;
#pragma unroll
                for (int i = 0; i < 16; ++i) { ot[0][i] *= alpha; ot[1][i] *= alpha; }
            }
            m = mnew;
            // masked scores (and rows that have seen no key yet) must give p = 0: exp2(-1e30) = 0
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) st[kb][r] = (st[kb][r] <= NEG_BIG) ? NEG_BIG : st[kb][r] - mnew;
            seeded = __all(m > NEG_BIG);
        }
        f32x2 rs2 = {0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const f32x2 pv = {__builtin_amdgcn_exp2f(st[kb][r]), __builtin_amdgcn_exp2f(st[kb][r + 1])};
                rs2 += pv;
                st[kb][r] = pv[0]; st[kb][r + 1] = pv[1];
            }
        l += rs2[0] + rs2[1];
        // O^T += V^T P^T: V fragments by asm transposed reads (k-steps 0,1 were started before the softmax)
        tr_issue<TILE_BYTES, 2>(tra, s_ * 2 * TILE_BYTES, vb);
        lgkm_wait_tied<8>(va);
#pragma unroll
        for (int x = 0; x < 2; ++x) {
            const bf16x8 pf = pack8(st[0], x);
#pragma unroll
            for (int db = 0; db < 2; ++db)
                ot[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(join8(va[2 * x + db][0], va[2 * x + db][1]), pf, ot[db], 0, 0, 0);
        }
        lgkm_wait_tied<0>(vb);
#pragma unroll
        for (int x = 0; x < 2; ++x) {
            const bf16x8 pf = pack8(st[1], x);
#pragma unroll
            for (int db = 0; db < 2; ++db)
                ot[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(join8(vb[2 * x + db][0], vb[2 * x + db][1]), pf, ot[db], 0, 0, 0);
        }
        if (ahead) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        lds_barrier();
        s_ = (s_ == 2) ? 0 : s_ + 1;
    }

    const float lt = xhalf_sum(l);
    if (p.kv_splits > 1) {
        // partial result of this run of keys: O^T unnormalised (relative to the run's own reference m), m and l per row
        if (q0 + ql < qhi) {
            const long row = ((long)b * p.H + h) * p.Nq + qrow;
            const long all = (long)p.B * p.H * p.Nq;
            float* po = p.split_o + ((long)qt.split * all + row) * 64;
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *(f32x4*)(po + db * 32 + 8 * g + 4 * hh) = f32x4{ot[db][4 * g], ot[db][4 * g + 1], ot[db][4 * g + 2], ot[db][4 * g + 3]};
            if (hh == 0) { float* pm = p.split_ml + ((long)qt.split * all + row) * 2; pm[0] = m; pm[1] = lt; }
        }
        return;
    }
    const float inv = lt > 0.f ? 1.f / lt : 0.f;
    if (q0 + ql < qhi) {
        const long oo = (long)b * p.o_bs + (long)qrow * p.o_rs + h * 64;
        if (p.Olo) store_rows_bf16_hilo(p.O + oo, p.Olo + oo, ot, inv, hh);
        else store_rows_bf16(p.O + oo, ot, inv, hh);
        if (hh == 0) p.LSE[((long)b * p.H + h) * p.Nq + qrow] = -(m + __builtin_amdgcn_logf(lt));  // v_log_f32 = log2; stored NEGATED
    }
}

// Joins the kv_splits partial results of a forward launch: per (b, h, row) M = max m_s, w_s = exp2(m_s - M), O = sum w_s O_s /
// sum w_s l_s.  A run that saw no key has m = -1e30, l = 0 (weight 0).  16 threads x 4 dims per row, 16 rows per workgroup.
__global__ __launch_bounds__(256) void attn_combine_kernel(AttnArgs p) {
    const long all = (long)p.B * p.H * p.Nq;
    const long row = (long)blockIdx.x * 16 + (threadIdx.x >> 4);
    if (row >= all) return;
    const int d4 = (threadIdx.x & 15) * 4;
    float M = NEG_BIG;
    for (int s = 0; s < p.kv_splits; ++s) M = fmaxf(M, p.split_ml[((long)s * all + row) * 2]);
    f32x4 o = {0.f, 0.f, 0.f, 0.f};
    float L = 0.f;
    for (int s = 0; s < p.kv_splits; ++s) {
        const float* ml = p.split_ml + ((long)s * all + row) * 2;
        const float w = ml[1] > 0.f ? __builtin_amdgcn_exp2f(ml[0] - M) : 0.f;
        L += w * ml[1];
        const f32x4 v = *(const f32x4*)(p.split_o + ((long)s * all + row) * 64 + d4);
        o += v * w;
    }
    const float inv = L > 0.f ? 1.f / L : 0.f;
    const int q = (int)(row % p.Nq);
    const long bh = row / p.Nq;
    const int h = (int)(bh % p.H);
    const long b = bh / p.H;
    bf16_t* dst = p.O + b * p.o_bs + (long)q * p.o_rs + h * 64 + d4;
    *(u32x2*)dst = u32x2{pack_bf16x2(o[0] * inv, o[1] * inv), pack_bf16x2(o[2] * inv, o[3] * inv)};
    if (p.LSE && d4 == 0) p.LSE[row] = -(M + __builtin_amdgcn_logf(L));
}

// ---------------------------------------------------------------------------------------------
// backward, query-major: dQ (also forms delta = rowsum(dO o O) and stores -delta for the dK / dV kernel, which runs behind
// this one on the stream).  Two waves per SIMD: Q is pre-scaled by scale * log2(e) and the row constants -LSE2 / -delta
// sit in two 16-register blocks that serve as the INITIAL accumulators of every S' / dP' chain, so the element-wise part
// is p = exp2(S'), dS' = p * dP' (the factor `scale` goes into the final dQ store); both 32-key halves of a tile have
// their S' / dP' chains issued before the element-wise work of the first one starts.  (The three-waves-per-SIMD form
// with fma(S, sc, -LSE2) / fma(dP, scale, -delta * scale) per element ran 4-7 % slower: these kernels are bound by
// vector-instruction issue, not by occupancy.)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_kernel(AttnArgs p) {
    __shared__ __attribute__((aligned(16))) char smem[FWD_STAGES * 2 * TILE_BYTES + 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    QTile qt;
    if (!q_tile(p, smem + FWD_STAGES * 2 * TILE_BYTES, qt)) return;
    const int h = qt.h, b = qt.b, q0 = qt.q0, qhi = qt.qhi, qrow = qt.qrow, ks = qt.ks, ke = qt.ke, kbase = qt.kbase;
    const int w_ksmax = qt.w_ksmax, w_kemin = qt.w_kemin, kt0 = qt.kt0, kt1 = qt.kt1;
    const int ql = lane & 31, hh = lane >> 5;
    // empty interval: q' = 0 -> p = exp2(-LSE2) = 1 / Nk, dQ = 0
    const float sc = qt.flat ? 0.f : p.scale * LOG2E, gsc = qt.flat ? 0.f : p.scale;

    const bf16_t* Qp = p.Q + (long)b * p.q_bs + (long)qrow * p.q_rs + h * 64;
    const bf16_t* Gp = p.dO + (long)b * p.do_bs + (long)qrow * p.do_rs + h * 64;
    bf16x8 qf[4], gf[4];
    float delta = 0.f;
    {
        const bf16_t* Op = p.O + (long)b * p.o_bs + (long)qrow * p.o_rs + h * 64;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const bf16x8 qr = *(const bf16x8*)(Qp + 16 * s + 8 * hh);
            gf[s] = *(const bf16x8*)(Gp + 16 * s + 8 * hh);
            const bf16x8 of = *(const bf16x8*)(Op + 16 * s + 8 * hh);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                qf[s][e] = (__bf16)((float)qr[e] * sc);
                delta += (float)of[e] * (float)gf[s][e];
            }
            // delta multiplies EVERY key's dS: the 2^-9 rounding of the stored O shows up in dQ as (delta error) x
            // (attention-weighted mean of K), against a true dQ that only sees K's deviations from that mean.  With the
            // rounding residual of O (written by the forward) delta is good to ~2^-17.
            if (p.Olo) {
                const bf16x8 ol = *(const bf16x8*)(p.Olo + (Op - p.O) + 16 * s + 8 * hh);
#pragma unroll
                for (int e = 0; e < 8; ++e) delta += (float)ol[e] * (float)gf[s][e];
            }
        }
        delta = xhalf_sum(delta);
        if (hh == 0 && q0 + ql < qhi) p.DELTA_OUT[((long)b * p.H + h) * p.Nq + qrow] = -delta;
    }
    const float nlse2 = p.LSE[((long)b * p.H + h) * p.Nq + qrow];       // -LSE2
    f32x16 c_lse, c_del;
#pragma unroll
    for (int i = 0; i < 16; ++i) { c_lse[i] = nlse2; c_del[i] = -delta; }

    const bf16_t* Kb = p.K + (long)b * p.k_bs + h * 64;
    const bf16_t* Vb = p.V + (long)b * p.v_bs + h * 64;
    const DmaOff kvoff = dma_off(p.k_rs, wave, lane);               // k_rs == v_rs (checked by the launcher)
    const __amdgpu_buffer_rsrc_t krs = slice_rsrc(Kb, p.k_rs, p.Nk), vrs = slice_rsrc(Vb, p.v_rs, p.Nk);
    auto dma_tile = [&](int kt, int s) {
        dma_tile64(krs, p.k_rs, kvoff, kbase + kt * 64, smem + s * 2 * TILE_BYTES, wave);
        dma_tile64(vrs, p.v_rs, kvoff, kbase + kt * 64, smem + s * 2 * TILE_BYTES + TILE_BYTES, wave);
    };

    f32x16 dqt[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) { dqt[0][i] = 0.f; dqt[1][i] = 0.f; }
    const TrAddr tra = tr_addr((unsigned)(size_t)(__attribute__((address_space(3))) char*)smem, lane);
    if (kt0 < kt1) dma_tile(kt0, 0);
    if (kt0 + 1 < kt1) {
        dma_tile(kt0 + 1, 1);
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    lds_barrier();
#pragma unroll
    for (int s = 0; s < 4; ++s) asm volatile("" :: "v"(qf[s]), "v"(gf[s]));
    int s_ = 0;
    for (int kt = kt0; kt < kt1; ++kt) {
        const bool ahead = kt + 2 < kt1;
        if (ahead) dma_tile(kt + 2, s_ >= 1 ? s_ - 1 : 2);      // that stage was last read before the previous barrier
        const char* Kt = smem + s_ * 2 * TILE_BYTES;
        const char* Vt = Kt + TILE_BYTES;
        const bool full = (kbase + kt * 64 >= w_ksmax) && (kbase + kt * 64 + 64 <= w_kemin);
        f32x16 st[2], dp[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            st[kb] = c_lse; dp[kb] = c_del;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                st[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(Kt, kb, s, lane), qf[s], st[kb], 0, 0, 0);
                dp[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(Vt, kb, s, lane), gf[s], dp[kb], 0, 0, 0);
            }
        }
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            s16x4 kfr[4][2];
            if (kb == 0) tr_issue<0, 0>(tra, s_ * 2 * TILE_BYTES, kfr); else tr_issue<0, 2>(tra, s_ * 2 * TILE_BYTES, kfr);
            if (full) {
#pragma unroll
                for (int r = 0; r < 16; ++r) st[kb][r] = __builtin_amdgcn_exp2f(st[kb][r]) * dp[kb][r];
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int kidx = kbase + kt * 64 + kb * 32 + acc_row(r, hh);
                    const bool ok = (kidx >= ks && kidx < ke);
                    st[kb][r] = ok ? __builtin_amdgcn_exp2f(st[kb][r]) * dp[kb][r] : 0.f;
                }
            }
            lgkm_wait_tied<0>(kfr);
#pragma unroll
            for (int x = 0; x < 2; ++x) {
                const bf16x8 dsf = pack8(st[kb], x);
#pragma unroll
                for (int db = 0; db < 2; ++db)
                    dqt[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(join8(kfr[2 * x + db][0], kfr[2 * x + db][1]), dsf, dqt[db], 0, 0, 0);
            }
        }
        if (ahead) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        lds_barrier();
        s_ = (s_ == 2) ? 0 : s_ + 1;
    }
    if (q0 + ql < qhi)
        store_rows_bf16(p.dQ + (long)b * p.dq_bs + (long)qrow * p.dq_rs + h * 64, dqt, gsc, hh);
}

// ---------------------------------------------------------------------------------------------
// backward, key-major: dK, dV
//
// One workgroup = 4 waves = 128 keys of one (batch, head); a wave keeps dK^T and dV^T of its 32 keys in 64 accumulator
// registers and sweeps the q tiles (64 rows = two 32-row blocks) that can see them.  Per block: S' = Q.K'^T - LSE2 and
// dP' = dO.V^T - delta leave their MFMA chains ready-made (K' = K * scale * log2 e in the wave's registers; the row
// constants -LSE2 / -delta are the chains' INITIAL accumulators, read from LDS in the accumulator's row order), so the
// element-wise work is p = exp2(S') and dS' = p * dP': two VALU instructions per score (the factor `scale` of dS is
// applied once, when dK is stored).
//
// Every LDS read of the loop is issued by hand (inline asm, counted waits): the fragments and row constants of a block
// are fetched while the PREVIOUS block's dV / dK MFMAs run, so a block's first MFMA never waits on an LDS round trip
// (compiler-placed reads were issued two at a time right in front of their MFMA: the S / dP chain ran at LDS latency).
// To make that prefetch legal across q tiles the "next tile has landed" barrier sits in the MIDDLE of an iteration:
//     block 0 of tile t | wait DMA(t+1), barrier, issue DMA(t+R-1) | block 1 of tile t (prefetching block 0 of t+1)
// R = DKV_STAGES ring slots of {Q tile, dO tile, 64 x {-LSE2, -delta, ks, ke}}; a tile is 5 LDS-DMA pieces per wave
// (2 Q + 2 dO + one of the four row-constant arrays), so "at most 5 outstanding" = the next tile has landed.
// ---------------------------------------------------------------------------------------------
constexpr int DKV_STAGES = 4;
constexpr int AUX_OFF = DKV_STAGES * 2 * TILE_BYTES;   // per stage: nlse2[64] ndelta[64] ks[64] ke[64] = 1 KiB
constexpr int AGG_OFF = AUX_OFF + DKV_STAGES * 1024;   // per q tile {min ks, max ks, min ke, max ke} (per-row masks only)
constexpr int DKV_MAX_QTILES = 512;
constexpr int DKV_LDS = AGG_OFF + DKV_MAX_QTILES * 16;

template <int OFF>
__device__ __forceinline__ bf16x8 lds_rd128(unsigned addr) {
    bf16x8 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
    return v;
}
template <int OFF>
__device__ __forceinline__ f32x4 lds_rd128f(unsigned addr) {
    f32x4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
    return v;
}

// Per 32-row q block 16 LDS reads feed the S' / dP' chains: the Q / dO row fragments of the four k-steps and the
// -LSE2 / -delta row constants (accumulator order: registers 4g..4g+3 = rows 8g + 4hh ..+3).  A wave issues in order and
// an MFMA group holds it for 32 cycles per MFMA, so a read is only hidden by MFMAs issued AFTER it: all 16 go out in
// front of the PREVIOUS block's dV / dK group.
struct DkvBlk { bf16x8 rq[4], rg[4]; f32x4 cs[4], cd[4]; };

template <int QB>
__device__ __forceinline__ void dkv_issue(DkvBlk& k, const unsigned (&rfa)[4], unsigned auxa) {
    k.cs[0] = lds_rd128f<QB * 128>(auxa);      k.cs[1] = lds_rd128f<QB * 128 + 32>(auxa);
    k.cs[2] = lds_rd128f<QB * 128 + 64>(auxa); k.cs[3] = lds_rd128f<QB * 128 + 96>(auxa);
#pragma unroll
    for (int s = 0; s < 4; ++s) k.rq[s] = lds_rd128<QB * 4096>(rfa[s]);
    k.cd[0] = lds_rd128f<256 + QB * 128>(auxa);      k.cd[1] = lds_rd128f<256 + QB * 128 + 32>(auxa);
    k.cd[2] = lds_rd128f<256 + QB * 128 + 64>(auxa); k.cd[3] = lds_rd128f<256 + QB * 128 + 96>(auxa);
#pragma unroll
    for (int s = 0; s < 4; ++s) k.rg[s] = lds_rd128<TILE_BYTES + QB * 4096>(rfa[s]);
}
__device__ __forceinline__ void dkv_wait(DkvBlk& k) {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(k.rq[0]), "+v"(k.rq[1]), "+v"(k.rq[2]), "+v"(k.rq[3]), "+v"(k.rg[0]), "+v"(k.rg[1]), "+v"(k.rg[2]), "+v"(k.rg[3]),
                   "+v"(k.cs[0]), "+v"(k.cs[1]), "+v"(k.cs[2]), "+v"(k.cs[3]), "+v"(k.cd[0]), "+v"(k.cd[1]), "+v"(k.cd[2]), "+v"(k.cd[3])
                 :: "memory");
}
__device__ __forceinline__ f32x16 cat16(const f32x4 (&c)[4]) {
    typedef float f32x8 __attribute__((ext_vector_type(8)));
    const f32x8 lo = __builtin_shufflevector(c[0], c[1], 0, 1, 2, 3, 4, 5, 6, 7), hi = __builtin_shufflevector(c[2], c[3], 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
}

__global__ __launch_bounds__(256, 2) void attn_bwd_dkv_kernel(AttnArgs p) {
    __shared__ __attribute__((aligned(16))) char smem[DKV_LDS];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int pair, tile;
    pair_tile(blockIdx.x, p.B * p.H, ((p.Nk + 127) >> 7) + (p.seg ? p.n_seg + 1 : 0), p.r_rs == 0 || p.seg != nullptr, pair, tile);
    const int h = pair % p.H, b = pair / p.H;
    // Row-group launches (block-diagonal self-attention): the keys of a group are cut into 128-key tiles of their own; a
    // group's workgroup sweeps (A) the group's own rows - they all see exactly the group's keys: no per-row masks except on the
    // last, partial row tile - and then (F) the rows behind the last group ("tail": padding rows, whose per-row intervals may
    // point anywhere) through the general path.  Tail keys are swept by tail rows only.  Samples flagged seg_bad take the
    // per-row path below on all their rows, exactly like a launch without groups.
    bool grouped = false;
    int kg0 = 0, kg1 = p.Nk, tail0 = p.Nq, nA = 0;
    if (p.seg) {
        GroupTile g;
        if (!group_tile(p, b, tile, p.Nk, g)) return;
        tile = g.t;
        grouped = !(p.seg_bad && p.seg_bad[b]);
        if (grouped) {
            kg0 = g.g0; kg1 = g.g1; tail0 = g.tail0;
            nA = g.uniform ? (kg1 - kg0 + 63) >> 6 : 0;
        }
    } else if (tile * 128 >= p.Nk) {
        return;
    }
    // Phase probe: two workgroups share a CU, one wave of each per SIMD, nothing holds them in opposite
    // phases of their MFMA / vector-work alternation.  With p.stagger > 0 the workgroups dispatched as the SECOND on their CU (ids
    // 256 .. 511 of every 512) start p.stagger x 512 clocks late; equal-length workgroups keep that offset through the launch.
    if (p.stagger > 0 && ((blockIdx.x >> 8) & 1))
        for (int i = 0; i < p.stagger; ++i) __builtin_amdgcn_s_sleep(8);
    const int kw0 = kg0 + tile * 128 + wave * 32;
    const int kl = lane & 31, hh = lane >> 5;
    const int kidx = kw0 + kl;
    const int krow = min(kidx, p.Nk - 1);

    const bf16_t* Kp = p.K + (long)b * p.k_bs + (long)krow * p.k_rs + h * 64;
    const bf16_t* Vp = p.V + (long)b * p.v_bs + (long)krow * p.v_rs + h * 64;
    // K pre-scaled by scale * log2(e) (one more bf16 rounding of an operand that already is bf16)
    const float c_sc = p.scale * LOG2E;
    bf16x8 kf[4], vf[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const bf16x8 kr = *(const bf16x8*)(Kp + 16 * s + 8 * hh);
#pragma unroll
        for (int e = 0; e < 8; ++e) kf[s][e] = (__bf16)((float)kr[e] * c_sc);
        vf[s] = *(const bf16x8*)(Vp + 16 * s + 8 * hh);
    }

    const bf16_t* Qb = p.Q + (long)b * p.q_bs + h * 64;
    const bf16_t* Gb = p.dO + (long)b * p.do_bs + h * 64;
    const int* KSb = p.ks + b * p.r_bs;
    const int* KEb = p.ke + b * p.r_bs;
    const bool per_row = p.r_rs != 0;

    // tiles arrive by LDS-DMA: Q / dO through dma_tile64, and wave w copies row-constant array w of the tile's 64 rows
    // (4 bytes per lane; rows past Nq read as zeros through the descriptor bounds)
    const DmaOff qoff = dma_off(p.q_rs, wave, lane), goff = dma_off(p.do_rs, wave, lane);
    const __amdgpu_buffer_rsrc_t qrs = slice_rsrc(Qb, p.q_rs, p.Nq), grs = slice_rsrc(Gb, p.do_rs, p.Nq);
    const void* abase = wave == 0 ? (const void*)(p.LSE + ((long)b * p.H + h) * p.Nq)
                      : wave == 1 ? (const void*)(p.DELTA + ((long)b * p.H + h) * p.Nq)
                      : wave == 2 ? (const void*)KSb : (const void*)KEb;
    const bool a_row = wave < 2 || per_row;                         // array indexed by the q row (else one value per sample)
    const __amdgpu_buffer_rsrc_t ars = __builtin_amdgcn_make_buffer_rsrc((void*)abase, 0, a_row ? p.Nq * 4 : 4, 0x00020000);
    const unsigned aoff = a_row ? lane * 4 : 0;
    // first row of the q tile with index qt of this workgroup's sweep (grouped: own rows first, then the tail rows)
    auto tile_row = [&](int qt) { return !grouped ? qt * 64 : (qt < nA ? kg0 + qt * 64 : tail0 + (qt - nA) * 64); };
    auto load_tile = [&](int qt, int s) {
        const int r0 = tile_row(qt);
        dma_tile64(qrs, p.q_rs, qoff, r0, smem + s * 2 * TILE_BYTES, wave);
        dma_tile64(grs, p.do_rs, goff, r0, smem + s * 2 * TILE_BYTES + TILE_BYTES, wave);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ars, (__attribute__((address_space(3))) void*)(smem + AUX_OFF + s * 1024 + wave * 256), 4,
                                                 aoff, a_row ? r0 * 4 : 0, 0, 0);
    };

    f32x16 dkt[2], dvt[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) { dkt[0][i] = 0.f; dkt[1][i] = 0.f; dvt[0][i] = 0.f; dvt[1][i] = 0.f; }
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    const TrAddr tra = tr_addr(lds0, lane);
    // row-fragment addresses (stage 0, block 0, Q tile): row lane & 31, 16-byte chunk (2s + hh) ^ swizzle(row)
    unsigned rfb[4];
    {
        const int r = lane & 31, v = (((r >> 1) & 1) << 2) | ((r >> 2) & 3);
#pragma unroll
        for (int s = 0; s < 4; ++s) rfb[s] = lds0 + r * 128 + (((2 * s + hh) ^ v) << 4);
    }
    const unsigned auxb = lds0 + AUX_OFF + 16 * hh;

    // Range of q tiles that can touch this workgroup's 128 keys, and (per-row masks) per-tile interval summaries.
    int qt0, nqt;
    int* agg = (int*)(smem + AGG_OFF);
    const int nq_tiles = (p.Nq + 63) >> 6;
    int u_ks = 0, u_ke = 0;                 // the sample's interval when it is uniform
    if (grouped) {
        qt0 = 0; nqt = nA + ((p.Nq - tail0 + 63) >> 6);
    } else if (!per_row) {
        u_ks = KSb[0]; u_ke = min(KEb[0], p.Nk);
        if (u_ke <= u_ks) { u_ks = -1; u_ke = p.Nk; }               // empty interval: uniform attention, zero score scale
        const bool touch = max(u_ks, 0) < tile * 128 + 128 && u_ke > tile * 128;
        qt0 = 0; nqt = touch ? nq_tiles : 0;
    } else {
        // One pass over the intervals of this batch row: per 64-row q tile the four extremes {min ks, max ks, min ke,
        // max ke} go to LDS (flat rows count as ks = -1 / ke = Nk, rows past Nq as ks = INT_MAX / ke = 0).  They give (a)
        // the range of q tiles that can touch this workgroup's 128 keys - block-diagonal decoder masks and padded encoder
        // keys leave most (q tile, key block) pairs empty - and (b) the per-tile "skip" / "every row sees all my keys"
        // decisions of each wave as scalar compares (no wave reductions per tile).
        for (int base = wave * 64; base < nq_tiles * 64; base += 256) {
            const int r = base + lane;
            int a = INT_MAX, e = 0;
            if (r < p.Nq) {
                a = KSb[r * p.r_rs]; e = min(KEb[r * p.r_rs], p.Nk);
                if (e <= a) { a = -1; e = p.Nk; }
            }
            const int a_min = wave_min_i(a), a_max = wave_max_i(a), e_min = wave_min_i(e), e_max = wave_max_i(e);
            if (lane == 0) { int* g4 = agg + 4 * (base >> 6); g4[0] = a_min; g4[1] = a_max; g4[2] = e_min; g4[3] = e_max; }
        }
        __syncthreads();
        int q_first = INT_MAX, q_last = -1;
        const int kb0 = tile * 128, kb1 = kb0 + 128;
        for (int t = tid; t < nq_tiles; t += 256)
            if (max(agg[4 * t], 0) < kb1 && agg[4 * t + 3] > kb0) { q_first = min(q_first, t); q_last = max(q_last, t); }
        q_first = wave_min_i(q_first); q_last = wave_max_i(q_last);
        int* red = (int*)(smem + AUX_OFF);
        if (lane == 0) { red[wave] = q_first; red[4 + wave] = q_last; }
        __syncthreads();
        q_first = min(min(red[0], red[1]), min(red[2], red[3]));
        q_last = max(max(red[4], red[5]), max(red[6], red[7]));
        __syncthreads();
        qt0 = (q_last < 0) ? 0 : q_first;
        nqt = (q_last < 0) ? 0 : q_last + 1;
    }

    // ring prologue: tiles qt0 .. qt0 + R - 2 in flight, the first one landed
    constexpr int R = DKV_STAGES;
#pragma unroll
    for (int i = 0; i < R - 1; ++i) if (qt0 + i < nqt) load_tile(qt0 + i, i);
    {
        const int stay = min(nqt - qt0, R - 1) - 1;                 // tiles issued, minus the first one (must land now)
        if (stay >= 3) asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
        else if (stay == 2) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
        else if (stay == 1) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    lds_barrier();
    // The K / V fragments were fetched by plain global loads at kernel entry: naming them here keeps hipcc's wait for
    // them (a vmcnt(0), which would also drain the LDS-DMA of the tiles ahead) out of the loop.
#pragma unroll
    for (int s = 0; s < 4; ++s) asm volatile("" :: "v"(kf[s]), "v"(vf[s]));

    // q tiles that can touch THIS wave's 32 keys: [wa, wb) inside [qt0, nqt).  Outside it the wave only keeps the ring
    // going (its DMA pieces, the barriers); inside it every tile is computed (a tile in between that does not look at
    // these keys comes out as all-zero P through the general path).  Keeping the "skip" decision out of the compute
    // loop matters: with it inside, hipcc kept a second copy of the 64 dK / dV accumulators across the branch and spilled.
    int wa = nqt, wb = qt0;
    if (grouped) {
        if (kw0 < kg1) { wa = qt0; wb = nqt; }                      // (a wave whose 32 keys lie behind the group's end idles)
    } else if (!per_row) {
        if (u_ke > kw0 && max(u_ks, 0) < kw0 + 32) { wa = qt0; wb = nqt; }
    } else {
        for (int t0 = qt0; t0 < nqt; t0 += 64) {
            const int t = t0 + lane;
            const bool hit = t < nqt && agg[4 * t + 3] > kw0 && max(agg[4 * t], 0) < kw0 + 32;
            const unsigned long long m = __ballot(hit);
            if (m) {
                wa = min(wa, t0 + (int)__builtin_ctzll(m));
                wb = max(wb, t0 + 64 - (int)__builtin_clzll(m));
            }
        }
        wa = __builtin_amdgcn_readfirstlane(wa); wb = __builtin_amdgcn_readfirstlane(wb);
    }
    if (wa >= wb) { wa = nqt; wb = nqt; }

    DkvBlk blk;
    int s_ = 0;
    unsigned rfa[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) rfa[s] = rfb[s];
    // middle of an iteration: tile qt + 1 must have landed for everybody; then the slot of tile qt - 1 is refilled
    auto ring_step = [&](int qt) {
        const int left = nqt - 1 - qt;                          // tiles after this one
        // in flight here: tiles qt+1 .. qt+R-2 (those that exist); tile qt+1 must land, the younger ones may stay out
        const int stay = min(left, R - 2) - 1;
        if (stay >= 2) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
        else if (stay == 1) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (qt + R - 1 < nqt) load_tile(qt + R - 1, (s_ + R - 1) % R);
    };
    for (int qt = qt0; qt < wa; ++qt) {                          // nothing to compute yet
        ring_step(qt);
        s_ = (s_ + 1 == R) ? 0 : s_ + 1;
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) rfa[s] = rfb[s] + s_ * 2 * TILE_BYTES;
    if (wa < wb) dkv_issue<0>(blk, rfa, auxb + s_ * 1024);

    // One 32-row block.  Order of issue: [this block's 16 reads have landed] the 16 transposed reads of dO / Q | S' and dP'
    // chains (8 MFMAs; the transposed reads land under them) | exp2 / multiply / pack | the NEXT block's 16 reads | dV^T /
    // dK^T chains (8 MFMAs; the next block's reads land under them).  NQB: the next block's index inside its tile; NRF /
    // NAUX: its fragment / constant addresses.  (The timing-only ablation branches of rounds 2 / 3 are gone from the product
    // kernel: tools/probes/attn_bwd_dkv_ablation.md says where to find them.)
    s16x4 gfr[4][2], qfr[4][2];
#define DKV_BLOCK(QB, NQB, NRF, NAUX)                                                                                    \
    {                                                                                                                    \
        dkv_wait(blk);                                                                                                   \
        tr_issue<TILE_BYTES, 2 * QB>(tra, s_ * 2 * TILE_BYTES, gfr);                                                     \
        tr_issue<0, 2 * QB>(tra, s_ * 2 * TILE_BYTES, qfr);                                                              \
        f32x16 st = cat16(blk.cs), dp = cat16(blk.cd);                                                                   \
        if (cls == CLS_LANE) {           /* one interval for the whole tile: keys outside it start from -inf -> p = 0 */ \
            _Pragma("unroll") for (int r = 0; r < 16; ++r) st[r] = lane_ok ? st[r] : -__builtin_inff();                  \
        }                                                                                                                \
        _Pragma("unroll") for (int s = 0; s < 4; ++s) {                                                                  \
            st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(blk.rq[s], kf[s], st, 0, 0, 0);                                 \
            dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(blk.rg[s], vf[s], dp, 0, 0, 0);                                 \
        }                                                                                                                \
        if (cls != CLS_GENERAL) {                                                                                        \
            _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                                             \
                const float pe = __builtin_amdgcn_exp2f(st[r]);                                                          \
                st[r] = pe;                                                                                              \
                dp[r] = pe * dp[r];                                                                                      \
            }                                                                                                            \
        } else {              /* rows with different intervals, empty intervals, rows past the tile's last row */        \
            const float* af = (const float*)(smem + AUX_OFF + s_ * 1024);                                                \
            const int* ai = (const int*)(af + 128);                                                                      \
            _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                                              \
                const int qb0 = QB * 32 + 8 * g + 4 * hh;                                                                \
                _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                                          \
                    const int r = 4 * g + e;                                                                             \
                    int rks = ai[per_row ? qb0 + e : 0], rke = min(ai[64 + (per_row ? qb0 + e : 0)], p.Nk);              \
                    const bool flat = rke <= rks;                          /* empty interval: p = 1 / Nk, dS = 0 */      \
                    if (flat) { rks = 0; rke = p.Nk; }                                                                   \
                    const bool ok = (t_row0 + qb0 + e < t_rowhi) && (kidx >= rks) && (kidx < rke);                       \
                    const float pe = ok ? __builtin_amdgcn_exp2f(flat ? af[qb0 + e] : st[r]) : 0.f;                      \
                    st[r] = pe;                                                                                          \
                    dp[r] = flat ? 0.f : pe * dp[r];                                                                     \
                }                                                                                                        \
            }                                                                                                            \
        }                                                                                                                \
        const bf16x8 pf0 = pack8(st, 0), pf1 = pack8(st, 1), ds0 = pack8(dp, 0), ds1 = pack8(dp, 1);                     \
        lgkm_wait_tied<0>(gfr);                                                                                          \
        lgkm_wait_tied<0>(qfr);                                                                                          \
        dkv_issue<NQB>(blk, NRF, NAUX);                                                                                  \
        _Pragma("unroll") for (int db = 0; db < 2; ++db) {                                                               \
            dvt[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(join8(gfr[db][0], gfr[db][1]), pf0, dvt[db], 0, 0, 0);     \
            dkt[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(join8(qfr[db][0], qfr[db][1]), ds0, dkt[db], 0, 0, 0);     \
        }                                                                                                                \
        _Pragma("unroll") for (int db = 0; db < 2; ++db) {                                                               \
            dvt[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(join8(gfr[2 + db][0], gfr[2 + db][1]), pf1, dvt[db], 0, 0, 0); \
            dkt[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(join8(qfr[2 + db][0], qfr[2 + db][1]), ds1, dkt[db], 0, 0, 0); \
        }                                                                                                                \
    }

    constexpr int CLS_FULL = 1, CLS_LANE = 2, CLS_GENERAL = 3;
    for (int qt = wa; qt < wb; ++qt) {
        // What this q tile is for this wave's 32 keys (scalar decisions): every row sees all of them (full); all 64 rows
        // share ONE interval (per-lane mask); anything else (per-element masks).
        // t_ks / t_ks2 = min / max of the rows' interval starts, t_ke2 / t_ke = min / max of their ends;
        // t_row0 / t_rowhi: the tile's first row and the end of the rows it may use
        int t_ks, t_ks2, t_ke, t_ke2;
        const int t_row0 = tile_row(qt);
        int t_rowhi = p.Nq;
        if (grouped) {
            // own rows: all of them see exactly [kg0, kg1) - unless the tile runs over the group's end (per-element masks:
            // the rows behind it belong to other groups); tail rows: per-element masks
            const bool inside = qt < nA && t_row0 + 64 <= kg1;
            t_ks = inside ? kg0 : -1; t_ks2 = kg0; t_ke = kg1; t_ke2 = kg1;
            if (qt < nA) t_rowhi = kg1;
        } else if (per_row) {
            // (flat rows carry ks = -1, rows beyond Nq carry ks = INT_MAX: both force the general path)
            t_ks = __builtin_amdgcn_readfirstlane(agg[4 * qt]); t_ks2 = __builtin_amdgcn_readfirstlane(agg[4 * qt + 1]);
            t_ke2 = __builtin_amdgcn_readfirstlane(agg[4 * qt + 2]); t_ke = __builtin_amdgcn_readfirstlane(agg[4 * qt + 3]);
        } else {
            const bool whole = qt * 64 + 64 <= p.Nq;
            t_ks = u_ks; t_ks2 = whole ? u_ks : INT_MAX; t_ke = u_ke; t_ke2 = whole ? u_ke : 0;
        }
        int cls;
        if (t_ks >= 0 && t_ks2 <= kw0 && t_ke2 >= kw0 + 32) cls = CLS_FULL;
        else if (t_ks >= 0 && t_ks == t_ks2 && t_ke == t_ke2) cls = CLS_LANE;
        else cls = CLS_GENERAL;
        const bool lane_ok = kidx >= t_ks && kidx < t_ke;
        DKV_BLOCK(0, 1, rfa, auxb + s_ * 1024)
        ring_step(qt);
        // block 1 prefetches block 0 of the next tile (after the last tile: a harmless read of a stale slot)
        const int sn = (s_ + 1 == R) ? 0 : s_ + 1;
        unsigned rfn[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) rfn[s] = rfb[s] + sn * 2 * TILE_BYTES;
        DKV_BLOCK(1, 0, rfn, auxb + sn * 1024)
        s_ = sn;
#pragma unroll
        for (int s = 0; s < 4; ++s) rfa[s] = rfn[s];
    }
#undef DKV_BLOCK
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // the last (unused) prefetch
    for (int qt = wb; qt < nqt; ++qt) {                          // nothing left to compute
        ring_step(qt);
        s_ = (s_ + 1 == R) ? 0 : s_ + 1;
    }
    if (kidx < kg1) {
        store_rows_bf16(p.dK + (long)b * p.dk_bs + (long)krow * p.dk_rs + h * 64, dkt, p.scale, hh);
        store_rows_bf16(p.dV + (long)b * p.dv_bs + (long)krow * p.dv_rs + h * 64, dvt, 1.f, hh);
    }
}

bool check(const AttnArgs& a) {
    // the K and V slices share their lane offsets in the LDS-DMA (packed kv / qkv rows): one row stride for both
    return a.k_rs == a.v_rs && a.B > 0 && a.H > 0 && a.Nq > 0 && a.Nk > 0 && a.q_rs % 8 == 0 && a.k_rs % 8 == 0 && a.v_rs % 8 == 0 &&
           a.q_bs % 8 == 0 && a.k_bs % 8 == 0 && a.v_bs % 8 == 0;
}

bool check_seg(const AttnArgs& a) {
    // row groups: self-attention on one row axis, per-row intervals present for the rows outside the groups
    return !a.seg || (a.n_seg > 0 && a.n_seg <= EGO_MAX_MODS && a.Nq == a.Nk && a.r_rs == 1 && a.r_bs == a.Nq);
}

}  // namespace

extern "C" int ego_attn_tune(int key, int value) {
    // probe hook (timing only; no reference counterpart): key 1 = start delay (x 512 clocks) of the dK / dV workgroups that are the
    // second on their CU.  Returns the previous value; value < 0 only queries; unknown key: -1.
    if (key != 1) return -1;
    const int old = g_attn_stagger;
    if (value >= 0) g_attn_stagger = value;
    return old;
}

extern "C" int ego_attn_fwd_d64_seg(const void* Q, long q_bs, long q_rs, const void* K, long k_bs, long k_rs,
                                    const void* V, long v_bs, long v_rs, void* O, long o_bs, long o_rs, void* O_lo, float* LSE,
                                    const int* ks, const int* ke, long r_bs, long r_rs, const int* seg, int n_seg, const int* seg_bad,
                                    int B, int H, int Nq, int Nk, float scale, hipStream_t stream) {
    AttnArgs a{};
    a.Q = (const bf16_t*)Q; a.K = (const bf16_t*)K; a.V = (const bf16_t*)V;
    a.q_bs = q_bs; a.q_rs = q_rs; a.k_bs = k_bs; a.k_rs = k_rs; a.v_bs = v_bs; a.v_rs = v_rs;
    a.O = (bf16_t*)O; a.o_bs = o_bs; a.o_rs = o_rs; a.Olo = (bf16_t*)O_lo; a.LSE = LSE; a.ks = ks; a.ke = ke; a.r_bs = r_bs; a.r_rs = r_rs;
    a.seg = seg; a.seg_bad = seg ? seg_bad : nullptr; a.n_seg = seg ? n_seg : 0;
    a.B = B; a.H = H; a.Nq = Nq; a.Nk = Nk; a.scale = scale;
    if (B == 0 || Nq == 0) return EGO_OK;
    if (!check(a) || !check_seg(a) || o_rs % 8 || o_bs % 8 || (((uintptr_t)O) & 15) || (((uintptr_t)O_lo) & 15)) return EGO_ERR_ARG;    // 16-byte output rows
    EGO_LAUNCH(attn_fwd_kernel, dim3(B * H * ((Nq + 127) / 128 + (a.seg ? a.n_seg + 1 : 0))), dim3(256), 0, stream, a);
    LAUNCH_CHECK();
    return EGO_OK;
}

extern "C" long ego_attn_fwd_split_floats(int B, int H, int Nq, int kv_splits) {
    return kv_splits > 1 ? (long)kv_splits * B * H * Nq * 66 : 0;
}

extern "C" int ego_attn_fwd_d64_split(const void* Q, long q_bs, long q_rs, const void* K, long k_bs, long k_rs,
                                      const void* V, long v_bs, long v_rs, void* O, long o_bs, long o_rs, float* LSE,
                                      const int* ks, const int* ke, long r_bs, long r_rs, int B, int H, int Nq, int Nk, float scale,
                                      int kv_splits, float* ws, long ws_floats, hipStream_t stream) {
    if (kv_splits <= 1)
        return ego_attn_fwd_d64_seg(Q, q_bs, q_rs, K, k_bs, k_rs, V, v_bs, v_rs, O, o_bs, o_rs, nullptr, LSE, ks, ke, r_bs, r_rs, nullptr, 0,
                                    nullptr, B, H, Nq, Nk, scale, stream);
    AttnArgs a{};
    a.Q = (const bf16_t*)Q; a.K = (const bf16_t*)K; a.V = (const bf16_t*)V;
    a.q_bs = q_bs; a.q_rs = q_rs; a.k_bs = k_bs; a.k_rs = k_rs; a.v_bs = v_bs; a.v_rs = v_rs;
    a.O = (bf16_t*)O; a.o_bs = o_bs; a.o_rs = o_rs; a.LSE = LSE; a.ks = ks; a.ke = ke; a.r_bs = r_bs; a.r_rs = r_rs;
    a.B = B; a.H = H; a.Nq = Nq; a.Nk = Nk; a.scale = scale;
    if (B == 0 || Nq == 0) return EGO_OK;
    if (!check(a) || o_rs % 8 || o_bs % 8 || (((uintptr_t)O) & 15) || kv_splits > 16 || !ws || (((uintptr_t)ws) & 15) ||
        ws_floats < ego_attn_fwd_split_floats(B, H, Nq, kv_splits)) return EGO_ERR_ARG;
    a.kv_splits = kv_splits;
    a.split_o = ws;
    a.split_ml = ws + (long)kv_splits * B * H * Nq * 64;
    EGO_LAUNCH(attn_fwd_kernel, dim3(B * H * ((Nq + 127) / 128) * kv_splits), dim3(256), 0, stream, a);
    LAUNCH_CHECK();
    const long rows = (long)B * H * Nq;
    EGO_LAUNCH(attn_combine_kernel, dim3((unsigned)((rows + 15) / 16)), dim3(256), 0, stream, a);
    LAUNCH_CHECK();
    return EGO_OK;
}

extern "C" int ego_attn_fwd_d64(const void* Q, long q_bs, long q_rs, const void* K, long k_bs, long k_rs,
                                const void* V, long v_bs, long v_rs, void* O, long o_bs, long o_rs, void* O_lo, float* LSE,
                                const int* ks, const int* ke, long r_bs, long r_rs, int B, int H, int Nq, int Nk,
                                float scale, hipStream_t stream) {
    return ego_attn_fwd_d64_seg(Q, q_bs, q_rs, K, k_bs, k_rs, V, v_bs, v_rs, O, o_bs, o_rs, O_lo, LSE, ks, ke, r_bs, r_rs, nullptr, 0, nullptr,
                                B, H, Nq, Nk, scale, stream);
}

extern "C" int ego_attn_bwd_d64_seg(const void* Q, long q_bs, long q_rs, const void* K, long k_bs, long k_rs,
                                    const void* V, long v_bs, long v_rs, const void* O, long o_bs, long o_rs, const void* O_lo,
                                    const void* dO, long do_bs, long do_rs, const float* LSE, float* DELTA,
                                    void* dQ, long dq_bs, long dq_rs, void* dK, long dk_bs, long dk_rs,
                                    void* dV, long dv_bs, long dv_rs, const int* ks, const int* ke, long r_bs, long r_rs,
                                    const int* seg, int n_seg, const int* seg_bad,
                                    int B, int H, int Nq, int Nk, float scale, hipStream_t stream) {
    AttnArgs a{};
    a.Q = (const bf16_t*)Q; a.K = (const bf16_t*)K; a.V = (const bf16_t*)V;
    a.q_bs = q_bs; a.q_rs = q_rs; a.k_bs = k_bs; a.k_rs = k_rs; a.v_bs = v_bs; a.v_rs = v_rs;
    a.LSE = (float*)LSE; a.ks = ks; a.ke = ke; a.r_bs = r_bs; a.r_rs = r_rs;
    a.seg = seg; a.seg_bad = seg ? seg_bad : nullptr; a.n_seg = seg ? n_seg : 0;
    a.B = B; a.H = H; a.Nq = Nq; a.Nk = Nk; a.scale = scale;
    a.O = (bf16_t*)O; a.o_bs = o_bs; a.o_rs = o_rs; a.Olo = (bf16_t*)O_lo;
    a.dO = (const bf16_t*)dO; a.do_bs = do_bs; a.do_rs = do_rs; a.DELTA = DELTA; a.DELTA_OUT = DELTA;
    a.dQ = (bf16_t*)dQ; a.dq_bs = dq_bs; a.dq_rs = dq_rs;
    a.dK = (bf16_t*)dK; a.dk_bs = dk_bs; a.dk_rs = dk_rs;
    a.dV = (bf16_t*)dV; a.dv_bs = dv_bs; a.dv_rs = dv_rs;
    a.stagger = g_attn_stagger;
    if (B == 0 || Nq == 0) return EGO_OK;
    if (!check(a) || !check_seg(a) || do_rs % 8 || do_bs % 8 || dq_rs % 8 || dk_rs % 8 || dv_rs % 8 || dq_bs % 8 || dk_bs % 8 || dv_bs % 8 ||
        o_rs % 4 || o_bs % 4 || ((((uintptr_t)dQ) | ((uintptr_t)dK) | ((uintptr_t)dV)) & 15)) return EGO_ERR_ARG;              // 16-byte gradient rows
    if (Nq > DKV_MAX_QTILES * 64) return EGO_ERR_ARG;          // per-q-tile interval summaries live in LDS
    const int extra = a.seg ? a.n_seg + 1 : 0;
    // timing-only probe builds (tools/abl_attn.sh; never the product): ATT_ONLY 1 = dQ kernel only, 2 = dK / dV kernel only;
    // DKV_PAD_LDS = extra dynamic LDS bytes for the dK / dV launch (> 2.2 KB: only ONE workgroup fits a CU - the co-residence
    // experiment of DESIGN section 4e)
#ifndef ATT_ONLY
#define ATT_ONLY 0
#endif
#ifndef DKV_PAD_LDS
#define DKV_PAD_LDS 0
#endif
    if (ATT_ONLY != 2) { EGO_LAUNCH(attn_bwd_dq_kernel, dim3(B * H * ((Nq + 127) / 128 + extra)), dim3(256), 0, stream, a); }
    LAUNCH_CHECK();
    if (ATT_ONLY != 1) { EGO_LAUNCH(attn_bwd_dkv_kernel, dim3(B * H * ((Nk + 127) / 128 + extra)), dim3(256), DKV_PAD_LDS, stream, a); }
    LAUNCH_CHECK();
    return EGO_OK;
}

extern "C" int ego_attn_bwd_d64(const void* Q, long q_bs, long q_rs, const void* K, long k_bs, long k_rs,
                                const void* V, long v_bs, long v_rs, const void* O, long o_bs, long o_rs, const void* O_lo,
                                const void* dO, long do_bs, long do_rs, const float* LSE, float* DELTA,
                                void* dQ, long dq_bs, long dq_rs, void* dK, long dk_bs, long dk_rs,
                                void* dV, long dv_bs, long dv_rs, const int* ks, const int* ke, long r_bs, long r_rs,
                                int B, int H, int Nq, int Nk, float scale, hipStream_t stream) {
    return ego_attn_bwd_d64_seg(Q, q_bs, q_rs, K, k_bs, k_rs, V, v_bs, v_rs, O, o_bs, o_rs, O_lo, dO, do_bs, do_rs, LSE, DELTA,
                                dQ, dq_bs, dq_rs, dK, dk_bs, dk_rs, dV, dv_bs, dv_rs, ks, ke, r_bs, r_rs, nullptr, 0, nullptr,
                                B, H, Nq, Nk, scale, stream);
}
