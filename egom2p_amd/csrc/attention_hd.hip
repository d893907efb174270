// Attention for head dimensions other than 64 - the registered ego-L (egom2p_model.py:1080-1092: D = 1020, 15 heads of 68).
// A head is stored padded with zero columns to HDP = 96 or 128 (MFMA 32x32x16: the contraction runs in steps of 16, the
// output in blocks of 32), so the scores and the outputs are those of the unpadded head.  Same semantics as the d64
// kernels (attention.hip): one [ks, ke) key interval per query row (or per sample), an empty interval = uniform attention
// over all Nk keys with a zero score scale, rows / keys past Nq / Nk do not exist.
//
// These are the PARITY kernels of a configuration nobody trains at scale (only the 400 M checkpoint is released): same
// math and the same operand layouts as the d64 family - S^T = K Q'^T with the query on the lane, per-lane online softmax,
// O^T = V^T P^T on key-permuted fragments (transposed operands by ds_read_b64_tr_b16 on the row-major tile) - but tiles are
// staged through LDS by plain loads (two LDS stages, the tile after the next one in flight in registers; no LDS-DMA ring,
// no lazy softmax reference); key / query tiles outside every row's interval are skipped.  The throughput shapes (head
// dim 64) never come here.
#include "common.h"
#include "egom2p_hip.h"

namespace {

constexpr float LOG2E = 1.4426950408889634f;
constexpr float NEG_BIG = -1.0e30f;

struct HdArgs {
    const bf16_t* Q; const bf16_t* K; const bf16_t* V;
    long q_bs, q_rs, k_bs, k_rs, v_bs, v_rs;
    bf16_t* O; long o_bs, o_rs;
    bf16_t* Olo;
    float* LSE;                 // [B,H,Nq] log2-domain log-sum-exp of the scaled scores
    const int* ks; const int* ke;
    long r_bs, r_rs;
    int B, H, Nq, Nk;
    float scale;
    const bf16_t* dO; long do_bs, do_rs;
    float* DELTA;               // [B,H,Nq] rowsum(dO o O)
    bf16_t* dQ; long dq_bs, dq_rs;
    bf16_t* dK; long dk_bs, dk_rs;
    bf16_t* dV; long dv_bs, dv_rs;
};

__device__ __forceinline__ int acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

__device__ __forceinline__ bf16x8 pack8(const f32x16& a, int x) {
    bf16x8 r;
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = (__bf16)a[8 * x + j];
    return r;
}

// LDS tile of 32 rows (keys or queries): [32][HDP] bf16, row pitch HDP * 2 + 16 bytes.  A lane reads its row's 16-byte
// fragment per k-step (row-major operand) or, for the transposed operand, two 4-row x 16-column blocks through
// ds_read_b64_tr_b16 (lane i of a 16-lane group receives column i of the block's 4 rows).
template <int HDP> struct Tile {
    static constexpr int NPITCH = HDP * 2 + 16;
    static constexpr int NBYTES = 32 * NPITCH;
    static constexpr int CHUNKS = 32 * (HDP / 8);              // 16-byte chunks of a tile
    static constexpr int PER = (CHUNKS + 255) / 256;           // chunks per thread
};
template <int HDP> struct Pre { u32x4 v[Tile<HDP>::PER]; };   // one tile's rows on their way from global memory to LDS

// rows [row0, row0 + 32) of a [nrows, *] bf16 matrix (row stride rs elements); rows past nrows read as zeros
template <int HDP>
__device__ __forceinline__ void fetch(Pre<HDP>& pre, const bf16_t* base, long rs, int row0, int nrows, int tid) {
    typedef Tile<HDP> T;
#pragma unroll
    for (int i = 0; i < T::PER; ++i) {
        const int c = tid + 256 * i, row = c / (HDP / 8), ch = c % (HDP / 8);
        pre.v[i] = u32x4{0u, 0u, 0u, 0u};
        if (c < T::CHUNKS && row0 + row < nrows) pre.v[i] = *(const u32x4*)(base + (long)(row0 + row) * rs + ch * 8);
    }
}
template <int HDP>
__device__ __forceinline__ void stash(const Pre<HDP>& pre, char* nat, int tid) {
    typedef Tile<HDP> T;
#pragma unroll
    for (int i = 0; i < T::PER; ++i) {
        const int c = tid + 256 * i, row = c / (HDP / 8), ch = c % (HDP / 8);
        if (c < T::CHUNKS) *(u32x4*)(nat + row * T::NPITCH + ch * 16) = pre.v[i];
    }
}

// Fragment of the TRANSPOSED tile for output block db, 16-row step x: dims row 32 db + (lane & 31), tile rows
// {16x + 4hh .. +3} and {16x + 8 + 4hh .. +3} - the row permutation pack8() gives the B operand.  Group g = lane >> 4 covers
// dims 32 db + 16 (g & 1) .. +15 for half hh = g >> 1; lane 4q + p of a group supplies the address of tile row q, dims 4p .. 4p+3.
template <int HDP>
__device__ __forceinline__ bf16x8 trn_frag(const char* nat, int db, int x, int lane) {
    typedef Tile<HDP> T;
    const int g = lane >> 4, t = lane & 15, q = t >> 2, pp = t & 3;
    const char* a0 = nat + (16 * x + 4 * (g >> 1) + q) * T::NPITCH + (32 * db + 16 * (g & 1) + 4 * pp) * 2;
    return join8(lds_read_tr16(a0), lds_read_tr16(a0 + 8 * T::NPITCH));
}
template <int HDP>
__device__ __forceinline__ bf16x8 nat_frag(const char* nat, int s, int lane) {
    typedef Tile<HDP> T;
    return *(const bf16x8*)(nat + (lane & 31) * T::NPITCH + (16 * s + 8 * (lane >> 5)) * 2);
}

// transposed accumulators [HDP dims][32 rows on lanes] -> bf16 rows
template <int HDP>
__device__ __forceinline__ void store_rows(bf16_t* dst, bf16_t* lo, const f32x16 (&t)[HDP / 32], float mul, int hh) {
#pragma unroll
    for (int db = 0; db < HDP / 32; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int d0 = db * 32 + 8 * g + 4 * hh;
            float x[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) x[e] = t[db][4 * g + e] * mul;
            const u32x2 o = {pack_bf16x2(x[0], x[1]), pack_bf16x2(x[2], x[3])};
            *(u32x2*)(dst + d0) = o;
            if (lo) {
                const u32x2 q = {pack_bf16x2(x[0] - round_bf16(x[0]), x[1] - round_bf16(x[1])),
                                 pack_bf16x2(x[2] - round_bf16(x[2]), x[3] - round_bf16(x[3]))};
                *(u32x2*)(lo + d0) = q;
            }
        }
}

// Key tiles [kt0, kt1) that at least one of the workgroup's 128 query rows attends: the union of the rows' intervals (an empty
// interval has already become [0, Nk)).  A tile outside every row's interval contributes exact zeros (p = 0, alpha = 1), so
// skipping it leaves the results bit for bit - the decoder's block-diagonal self-attention visits ~half of the tiles.
__device__ __forceinline__ void key_tile_range(int ks, int ke, int nkt, int lane, int wave, int* w_lo, int* w_hi, int& kt0, int& kt1) {
    const int lo = wave_min_i(ks), hi = wave_max_i(ke);
    if (lane == 0) { w_lo[wave] = lo; w_hi[wave] = hi; }
    __syncthreads();
    kt0 = min(max(min(min(w_lo[0], w_lo[1]), min(w_lo[2], w_lo[3])), 0) >> 5, nkt);
    kt1 = min((max(max(w_hi[0], w_hi[1]), max(w_hi[2], w_hi[3])) + 31) >> 5, nkt);
}

// ---------------------------------------------------------------------------------------------
// forward: workgroup = 128 queries of one (batch, head), wave = 32 of them (query on the lane)
// ---------------------------------------------------------------------------------------------
template <int HDP, int KS>
__global__ __launch_bounds__(256, 2) void hd_fwd_kernel(HdArgs p) {
    typedef Tile<HDP> T;
    constexpr int DB = HDP / 32;        // KS: 16-wide contraction steps over the head dimension that hold non-zero columns
    __shared__ __attribute__((aligned(16))) char Kn[2][T::NBYTES];
    __shared__ __attribute__((aligned(16))) char Vn[2][T::NBYTES];
    __shared__ int w_lo[4], w_hi[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int pair, tile;
    pair_tile(blockIdx.x, p.B * p.H, (p.Nq + 127) >> 7, p.r_rs == 0, pair, tile);      // every tile of a pair on one XCD (common.h)
    const int h = pair % p.H, b = pair / p.H;
    const int q0 = tile * 128 + wave * 32, ql = lane & 31, hh = lane >> 5;
    const int qrow = min(q0 + ql, p.Nq - 1);
    int ks = p.ks[b * p.r_bs + qrow * p.r_rs], ke = min(p.ke[b * p.r_bs + qrow * p.r_rs], p.Nk);
    float sc = p.scale * LOG2E;
    if (ke <= ks) { ks = 0; ke = p.Nk; sc = 0.f; }
    int kt0, kt1;
    key_tile_range(ks, ke, (p.Nk + 31) >> 5, lane, wave, w_lo, w_hi, kt0, kt1);

    const bf16_t* Qp = p.Q + (long)b * p.q_bs + (long)qrow * p.q_rs + h * HDP;
    bf16x8 qf[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const bf16x8 qr = *(const bf16x8*)(Qp + 16 * s + 8 * hh);
#pragma unroll
        for (int e = 0; e < 8; ++e) qf[s][e] = (__bf16)((float)qr[e] * sc);
    }
    const bf16_t* Kb = p.K + (long)b * p.k_bs + h * HDP;
    const bf16_t* Vb = p.V + (long)b * p.v_bs + h * HDP;

    f32x16 ot[DB];
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int i = 0; i < 16; ++i) ot[db][i] = 0.f;
    float m = NEG_BIG, l = 0.f;
    // two LDS stages: tile kt is computed from stage kt & 1 while tile kt + 1 (fetched into registers one iteration earlier)
    // is written into the other stage and tile kt + 2 leaves global memory - one barrier per tile.  (Fetching two tiles ahead
    // with a second register set was 14 % slower: the loop is bound by instruction issue, not by load latency.)
    Pre<HDP> pkA, pvA;
    fetch<HDP>(pkA, Kb, p.k_rs, kt0 * 32, p.Nk, tid);
    fetch<HDP>(pvA, Vb, p.v_rs, kt0 * 32, p.Nk, tid);
    stash<HDP>(pkA, Kn[0], tid);
    stash<HDP>(pvA, Vn[0], tid);
    if (kt0 + 1 < kt1) {
        fetch<HDP>(pkA, Kb, p.k_rs, (kt0 + 1) * 32, p.Nk, tid);
        fetch<HDP>(pvA, Vb, p.v_rs, (kt0 + 1) * 32, p.Nk, tid);
    }
    lds_barrier();
    auto step = [&](int kt, Pre<HDP>& pk, Pre<HDP>& pv) {
        const char* Kc = Kn[(kt - kt0) & 1];
        const char* Vc = Vn[(kt - kt0) & 1];
        f32x16 st;
#pragma unroll
        for (int i = 0; i < 16; ++i) st[i] = 0.f;
#pragma unroll
        for (int s = 0; s < KS; ++s) st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(nat_frag<HDP>(Kc, s, lane), qf[s], st, 0, 0, 0);
        float mx = NEG_BIG;
        const bool inside = __all(kt * 32 >= ks && kt * 32 + 32 <= ke);      // the whole tile lies inside every row's interval
        if (inside) {
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, st[r]);
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int kidx = kt * 32 + acc_row(r, hh);
                const float v = (kidx >= ks && kidx < ke) ? st[r] : NEG_BIG;
                st[r] = v;
                mx = fmaxf(mx, v);
            }
        }
        mx = xhalf_max(mx);
        const float mnew = fmaxf(m, mx);
        if (__any(mnew != m)) {                                     // (alpha = 1 on every lane otherwise: the products are exact)
            const float alpha = __builtin_amdgcn_exp2f(m - mnew);   // (both NEG_BIG: 1, with l = 0 and O = 0)
            l *= alpha;
#pragma unroll
            for (int db = 0; db < DB; ++db)
#pragma unroll
                for (int i = 0; i < 16; ++i) ot[db][i] *= alpha;
        }
        m = mnew;
        float rs = 0.f;
        if (inside) {                                               // (every score is real: mnew is one of them)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pe = __builtin_amdgcn_exp2f(st[r] - mnew);
                rs += pe;
                st[r] = pe;
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pe = (st[r] <= NEG_BIG) ? 0.f : __builtin_amdgcn_exp2f(st[r] - mnew);
                rs += pe;
                st[r] = pe;
            }
        }
        l += rs;
#pragma unroll
        for (int x = 0; x < 2; ++x) {
            const bf16x8 pf = pack8(st, x);
#pragma unroll
            for (int db = 0; db < DB; ++db) ot[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(trn_frag<HDP>(Vc, db, x, lane), pf, ot[db], 0, 0, 0);
        }
        if (kt + 1 < kt1) {
            stash<HDP>(pk, Kn[(kt - kt0 + 1) & 1], tid);            // tile kt + 1 (its stage's last readers passed the previous barrier)
            stash<HDP>(pv, Vn[(kt - kt0 + 1) & 1], tid);
            if (kt + 2 < kt1) {
                fetch<HDP>(pk, Kb, p.k_rs, (kt + 2) * 32, p.Nk, tid);
                fetch<HDP>(pv, Vb, p.v_rs, (kt + 2) * 32, p.Nk, tid);
            }
            lds_barrier();                                          // (orders LDS only: the fetches stay in flight)
        }
    };
    for (int kt = kt0; kt < kt1; ++kt) step(kt, pkA, pvA);
    const float lt = xhalf_sum(l);
    const float inv = lt > 0.f ? 1.f / lt : 0.f;
    if (q0 + ql < p.Nq) {
        const long oo = (long)b * p.o_bs + (long)qrow * p.o_rs + h * HDP;
        store_rows<HDP>(p.O + oo, p.Olo ? p.Olo + oo : nullptr, ot, inv, hh);
        if (hh == 0) p.LSE[((long)b * p.H + h) * p.Nq + qrow] = m + __builtin_amdgcn_logf(lt);       // v_log_f32 = log2
    }
}

// delta[b,h,q] = rowsum(dO o (O + O_lo)): one thread per (batch, head, query)
template <int HDP>
__global__ __launch_bounds__(256) void hd_delta_kernel(HdArgs p) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)p.B * p.H * p.Nq) return;
    const int q = (int)(i % p.Nq), h = (int)((i / p.Nq) % p.H), b = (int)(i / ((long)p.Nq * p.H));
    const bf16_t* g = p.dO + (long)b * p.do_bs + (long)q * p.do_rs + h * HDP;
    const long oo = (long)b * p.o_bs + (long)q * p.o_rs + h * HDP;
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < HDP / 8; ++c) {
        const u32x4 gv = *(const u32x4*)(g + 8 * c), ov = *(const u32x4*)(p.O + oo + 8 * c);
        u32x4 lv = {0u, 0u, 0u, 0u};
        if (p.Olo) lv = *(const u32x4*)(p.Olo + oo + 8 * c);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            s += bf16_to_f32(gv[e] & 0xffff) * (bf16_to_f32(ov[e] & 0xffff) + bf16_to_f32(lv[e] & 0xffff));
            s += bf16_to_f32(gv[e] >> 16) * (bf16_to_f32(ov[e] >> 16) + bf16_to_f32(lv[e] >> 16));
        }
    }
    p.DELTA[i] = s;
}

// ---------------------------------------------------------------------------------------------
// backward, query-major: dQ = scale * sum_k dS K,  dS = P o (dP - delta)
// ---------------------------------------------------------------------------------------------
template <int HDP, int KS>
__global__ __launch_bounds__(256, 2) void hd_dq_kernel(HdArgs p) {      // (3 waves per SIMD: 31 spilled registers, 60 % slower)
    typedef Tile<HDP> T;
    constexpr int DB = HDP / 32;
    __shared__ __attribute__((aligned(16))) char Kn[2][T::NBYTES];
    __shared__ __attribute__((aligned(16))) char Vn[2][T::NBYTES];
    __shared__ int w_lo[4], w_hi[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int pair, tile;
    pair_tile(blockIdx.x, p.B * p.H, (p.Nq + 127) >> 7, p.r_rs == 0, pair, tile);      // every tile of a pair on one XCD (common.h)
    const int h = pair % p.H, b = pair / p.H;
    const int q0 = tile * 128 + wave * 32, ql = lane & 31, hh = lane >> 5;
    const int qrow = min(q0 + ql, p.Nq - 1);
    int ks = p.ks[b * p.r_bs + qrow * p.r_rs], ke = min(p.ke[b * p.r_bs + qrow * p.r_rs], p.Nk);
    float sc = p.scale * LOG2E, gsc = p.scale;
    if (ke <= ks) { ks = 0; ke = p.Nk; sc = 0.f; gsc = 0.f; }
    int kt0, kt1;
    key_tile_range(ks, ke, (p.Nk + 31) >> 5, lane, wave, w_lo, w_hi, kt0, kt1);
    const long li = ((long)b * p.H + h) * p.Nq + qrow;
    const float lse2 = p.LSE[li], delta = p.DELTA[li];

    const bf16_t* Qp = p.Q + (long)b * p.q_bs + (long)qrow * p.q_rs + h * HDP;
    const bf16_t* Gp = p.dO + (long)b * p.do_bs + (long)qrow * p.do_rs + h * HDP;
    bf16x8 qf[KS], gf[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const bf16x8 qr = *(const bf16x8*)(Qp + 16 * s + 8 * hh);
#pragma unroll
        for (int e = 0; e < 8; ++e) qf[s][e] = (__bf16)((float)qr[e] * sc);
        gf[s] = *(const bf16x8*)(Gp + 16 * s + 8 * hh);
    }
    const bf16_t* Kb = p.K + (long)b * p.k_bs + h * HDP;
    const bf16_t* Vb = p.V + (long)b * p.v_bs + h * HDP;

    f32x16 dqt[DB];
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int i = 0; i < 16; ++i) dqt[db][i] = 0.f;
    Pre<HDP> pk, pv;                                                // two LDS stages, one barrier per tile (hd_fwd_kernel)
    fetch<HDP>(pk, Kb, p.k_rs, kt0 * 32, p.Nk, tid);
    fetch<HDP>(pv, Vb, p.v_rs, kt0 * 32, p.Nk, tid);
    stash<HDP>(pk, Kn[0], tid);
    stash<HDP>(pv, Vn[0], tid);
    if (kt0 + 1 < kt1) {
        fetch<HDP>(pk, Kb, p.k_rs, (kt0 + 1) * 32, p.Nk, tid);
        fetch<HDP>(pv, Vb, p.v_rs, (kt0 + 1) * 32, p.Nk, tid);
    }
    lds_barrier();
    for (int kt = kt0; kt < kt1; ++kt) {
        const char* Kc = Kn[(kt - kt0) & 1];
        const char* Vc = Vn[(kt - kt0) & 1];
        f32x16 st, dp;
#pragma unroll
        for (int i = 0; i < 16; ++i) { st[i] = 0.f; dp[i] = 0.f; }
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(nat_frag<HDP>(Kc, s, lane), qf[s], st, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(nat_frag<HDP>(Vc, s, lane), gf[s], dp, 0, 0, 0);
        }
        if (__all(kt * 32 >= ks && kt * 32 + 32 <= ke)) {
#pragma unroll
            for (int r = 0; r < 16; ++r) st[r] = __builtin_amdgcn_exp2f(st[r] - lse2) * (dp[r] - delta);
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int kidx = kt * 32 + acc_row(r, hh);
                const float pe = (kidx >= ks && kidx < ke) ? __builtin_amdgcn_exp2f(st[r] - lse2) : 0.f;
                st[r] = pe * (dp[r] - delta);
            }
        }
#pragma unroll
        for (int x = 0; x < 2; ++x) {
            const bf16x8 df = pack8(st, x);
#pragma unroll
            for (int db = 0; db < DB; ++db) dqt[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(trn_frag<HDP>(Kc, db, x, lane), df, dqt[db], 0, 0, 0);
        }
        if (kt + 1 < kt1) {
            stash<HDP>(pk, Kn[(kt - kt0 + 1) & 1], tid);
            stash<HDP>(pv, Vn[(kt - kt0 + 1) & 1], tid);
            if (kt + 2 < kt1) {
                fetch<HDP>(pk, Kb, p.k_rs, (kt + 2) * 32, p.Nk, tid);
                fetch<HDP>(pv, Vb, p.v_rs, (kt + 2) * 32, p.Nk, tid);
            }
            lds_barrier();
        }
    }
    if (q0 + ql < p.Nq)
        store_rows<HDP>(p.dQ + (long)b * p.dq_bs + (long)qrow * p.dq_rs + h * HDP, nullptr, dqt, gsc, hh);
}

// ---------------------------------------------------------------------------------------------
// backward, key-major: dV = P^T dO,  dK = scale * dS^T Q (workgroup = 128 keys, wave = 32 of them, key on the lane)
// ---------------------------------------------------------------------------------------------
template <int HDP, int KS>
__global__ __launch_bounds__(256, HDP <= 96 ? 2 : 1) void hd_dkv_kernel(HdArgs p) {     // (heads of 128: 2 waves per SIMD would spill 93 registers)
    typedef Tile<HDP> T;
    constexpr int DB = HDP / 32;
    __shared__ __attribute__((aligned(16))) char Qn[2][T::NBYTES];
    __shared__ __attribute__((aligned(16))) char Gn[2][T::NBYTES];
    __shared__ __attribute__((aligned(16))) float a_lse[2][32], a_delta[2][32];
    __shared__ __attribute__((aligned(16))) int a_ks[2][32], a_ke[2][32];
    __shared__ int w_lo[4], w_hi[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int pair, tile;
    pair_tile(blockIdx.x, p.B * p.H, (p.Nk + 127) >> 7, p.r_rs == 0, pair, tile);
    const int h = pair % p.H, b = pair / p.H;
    const int kl = lane & 31, hh = lane >> 5;
    const int kidx = tile * 128 + wave * 32 + kl;
    const int krow = min(kidx, p.Nk - 1);
    const float c_sc = p.scale * LOG2E;
    // Query tiles [qt0, qt1) with at least one row that attends a key of this workgroup (an empty interval attends every key):
    // every other tile contributes exact zeros.  One pass over the rows' intervals (Nq / 256 loads per thread).
    int qt0, qt1;
    {
        const int k0 = tile * 128, k1 = min(k0 + 128, p.Nk);
        int qlo = 0x7fffffff, qhi = -1;
        for (int q = tid; q < p.Nq; q += 256) {
            const int a = p.ks[b * p.r_bs + q * p.r_rs], e = min(p.ke[b * p.r_bs + q * p.r_rs], p.Nk);
            if (e <= a || (a < k1 && e > k0)) { qlo = min(qlo, q); qhi = max(qhi, q); }
        }
        qlo = wave_min_i(qlo); qhi = wave_max_i(qhi);
        if (lane == 0) { w_lo[wave] = qlo; w_hi[wave] = qhi; }
        __syncthreads();
        qlo = min(min(w_lo[0], w_lo[1]), min(w_lo[2], w_lo[3]));
        qhi = max(max(w_hi[0], w_hi[1]), max(w_hi[2], w_hi[3]));
        qt0 = qhi < 0 ? 0 : qlo >> 5;
        qt1 = qhi < 0 ? 0 : (qhi >> 5) + 1;
    }
    const bf16_t* Kp = p.K + (long)b * p.k_bs + (long)krow * p.k_rs + h * HDP;
    const bf16_t* Vp = p.V + (long)b * p.v_bs + (long)krow * p.v_rs + h * HDP;
    bf16x8 kf[KS], vf[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const bf16x8 kr = *(const bf16x8*)(Kp + 16 * s + 8 * hh);
#pragma unroll
        for (int e = 0; e < 8; ++e) kf[s][e] = (__bf16)((float)kr[e] * c_sc);
        vf[s] = *(const bf16x8*)(Vp + 16 * s + 8 * hh);
    }
    const bf16_t* Qb = p.Q + (long)b * p.q_bs + h * HDP;
    const bf16_t* Gb = p.dO + (long)b * p.do_bs + h * HDP;
    const long lb = ((long)b * p.H + h) * p.Nq;

    f32x16 dkt[DB], dvt[DB];
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int i = 0; i < 16; ++i) { dkt[db][i] = 0.f; dvt[db][i] = 0.f; }
    // the next query tile on its way: rows of Q and dO, and (threads 0-31) the tile's row constants
    Pre<HDP> pq, pg;
    float n_lse = 0.f, n_delta = 0.f;
    int n_ks = 0, n_ke = 0;
    auto fetch_tile = [&](int qt) {
        fetch<HDP>(pq, Qb, p.q_rs, qt * 32, p.Nq, tid);
        fetch<HDP>(pg, Gb, p.do_rs, qt * 32, p.Nq, tid);
        if (tid < 32) {
            const int q = qt * 32 + tid;
            const int qc = min(q, p.Nq - 1);
            n_lse = p.LSE[lb + qc];
            n_delta = p.DELTA[lb + qc];
            n_ks = p.ks[b * p.r_bs + qc * p.r_rs];
            n_ke = min(p.ke[b * p.r_bs + qc * p.r_rs], p.Nk);
            if (q >= p.Nq) { n_ks = 0x7fffffff; n_ke = -1; }             // a row past Nq sees no key (and is not "flat")
        }
    };
    auto stash_tile = [&](int st) {
        stash<HDP>(pq, Qn[st], tid);
        stash<HDP>(pg, Gn[st], tid);
        if (tid < 32) { a_lse[st][tid] = n_lse; a_delta[st][tid] = n_delta; a_ks[st][tid] = n_ks; a_ke[st][tid] = n_ke; }
    };
    if (qt0 < qt1) {                                                    // two LDS stages, one barrier per tile (hd_fwd_kernel)
        fetch_tile(qt0);
        stash_tile(0);
        if (qt0 + 1 < qt1) fetch_tile(qt0 + 1);
        lds_barrier();
    }
    for (int qt = qt0; qt < qt1; ++qt) {
        const int cur = (qt - qt0) & 1;
        const char* Qc = Qn[cur];
        const char* Gc = Gn[cur];
        f32x16 st, dp;
#pragma unroll
        for (int i = 0; i < 16; ++i) { st[i] = 0.f; dp[i] = 0.f; }
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(nat_frag<HDP>(Qc, s, lane), kf[s], st, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(nat_frag<HDP>(Gc, s, lane), vf[s], dp, 0, 0, 0);
        }
        // row constants of this lane's 16 query rows: acc_row(4g + e, hh) = 8g + 4hh + e, four consecutive rows per g
        typedef int i32x4 __attribute__((ext_vector_type(4)));
        const int kw0 = tile * 128 + wave * 32;
        bool full;
        {   // every row of the tile sees all 32 keys of this wave (no empty interval, no row past Nq, no key past Nk)?
            const int rks = a_ks[cur][kl], rke = a_ke[cur][kl];
            full = __all(rke > rks && rks <= kw0 && rke >= kw0 + 32) && kw0 + 32 <= p.Nk;
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 L = *(const f32x4*)&a_lse[cur][8 * g + 4 * hh], Dl = *(const f32x4*)&a_delta[cur][8 * g + 4 * hh];
            if (full) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int r = 4 * g + e;
                    const float pe = __builtin_amdgcn_exp2f(st[r] - L[e]);
                    st[r] = pe;
                    dp[r] = pe * (dp[r] - Dl[e]);
                }
            } else {
                const i32x4 KS4 = *(const i32x4*)&a_ks[cur][8 * g + 4 * hh], KE4 = *(const i32x4*)&a_ke[cur][8 * g + 4 * hh];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int r = 4 * g + e;
                    int rks = KS4[e], rke = KE4[e];
                    const bool past = rke < 0;
                    const bool flat = !past && rke <= rks;                  // empty interval: p = 1 / Nk, dS = 0
                    if (flat) { rks = 0; rke = p.Nk; }
                    const bool ok = !past && kidx >= rks && kidx < rke && kidx < p.Nk;
                    const float pe = ok ? __builtin_amdgcn_exp2f((flat ? 0.f : st[r]) - L[e]) : 0.f;
                    st[r] = pe;
                    dp[r] = flat ? 0.f : pe * (dp[r] - Dl[e]);
                }
            }
        }
#pragma unroll
        for (int x = 0; x < 2; ++x) {
            const bf16x8 pf = pack8(st, x), df = pack8(dp, x);
#pragma unroll
            for (int db = 0; db < DB; ++db) {
                dvt[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(trn_frag<HDP>(Gc, db, x, lane), pf, dvt[db], 0, 0, 0);
                dkt[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(trn_frag<HDP>(Qc, db, x, lane), df, dkt[db], 0, 0, 0);
            }
        }
        if (qt + 1 < qt1) {
            stash_tile(cur ^ 1);
            if (qt + 2 < qt1) fetch_tile(qt + 2);
            lds_barrier();
        }
    }
    if (kidx < p.Nk) {
        store_rows<HDP>(p.dK + (long)b * p.dk_bs + (long)krow * p.dk_rs + h * HDP, nullptr, dkt, p.scale, hh);
        store_rows<HDP>(p.dV + (long)b * p.dv_bs + (long)krow * p.dv_rs + h * HDP, nullptr, dvt, 1.f, hh);
    }
}

bool check(const HdArgs& a, int hdp) {
    return (hdp == 96 || hdp == 128) && a.B > 0 && a.H > 0 && a.Nq > 0 && a.Nk > 0 && a.q_rs % 8 == 0 && a.k_rs % 8 == 0 &&
           a.v_rs % 8 == 0 && a.q_bs % 8 == 0 && a.k_bs % 8 == 0 && a.v_bs % 8 == 0 && a.o_rs % 8 == 0 && a.o_bs % 8 == 0;
}

// hd_pad argument of the entry points: low 16 bits = elements between two stored heads (96 / 128); bits 16.. (optional, 0 =
// not given) = the head's real dimension: contraction steps over columns that are all padding (exact zeros) are left out -
// the registered ego-L's 68 of 96 need 5 of the 6 steps.  Same bits either way.
struct HdShape { int hdp, ks; };
HdShape hd_shape(int hd_pad) {
    const int hdp = hd_pad & 0xffff, hd = hd_pad >> 16;
    int ks = hdp / 16;
    if (hd > 0 && hd <= hdp && (hd + 15) / 16 <= 5) ks = 5;          // instantiated: all steps, or 5 (head dims 65 .. 80)
    return {hdp, ks};
}

}  // namespace

#define HD_DISPATCH(KERNEL, grid)                                                                      \
    do {                                                                                               \
        if (sh.hdp == 96 && sh.ks == 5) { EGO_LAUNCH((KERNEL<96, 5>), grid, dim3(256), 0, stream, a); } \
        else if (sh.hdp == 96) { EGO_LAUNCH((KERNEL<96, 6>), grid, dim3(256), 0, stream, a); }          \
        else if (sh.ks == 5) { EGO_LAUNCH((KERNEL<128, 5>), grid, dim3(256), 0, stream, a); }           \
        else { EGO_LAUNCH((KERNEL<128, 8>), grid, dim3(256), 0, stream, a); }                           \
    } while (0)

extern "C" int ego_attn_fwd_hd(const void* Q, long q_bs, long q_rs, const void* K, long k_bs, long k_rs, const void* V,
                               long v_bs, long v_rs, void* O, long o_bs, long o_rs, void* O_lo, float* LSE, const int* ks,
                               const int* ke, long r_bs, long r_rs, int B, int H, int Nq, int Nk, int hd_pad, float scale,
                               hipStream_t stream) {
    HdArgs a{};
    a.Q = (const bf16_t*)Q; a.K = (const bf16_t*)K; a.V = (const bf16_t*)V;
    a.q_bs = q_bs; a.q_rs = q_rs; a.k_bs = k_bs; a.k_rs = k_rs; a.v_bs = v_bs; a.v_rs = v_rs;
    a.O = (bf16_t*)O; a.o_bs = o_bs; a.o_rs = o_rs; a.Olo = (bf16_t*)O_lo; a.LSE = LSE; a.ks = ks; a.ke = ke; a.r_bs = r_bs; a.r_rs = r_rs;
    a.B = B; a.H = H; a.Nq = Nq; a.Nk = Nk; a.scale = scale;
    if (B == 0 || Nq == 0) return EGO_OK;
    const HdShape sh = hd_shape(hd_pad);
    // 16-byte row fragments in and out, like ego_attn_fwd_d64
    if (!check(a, sh.hdp) || ((((uintptr_t)Q) | ((uintptr_t)K) | ((uintptr_t)V) | ((uintptr_t)O) | ((uintptr_t)O_lo)) & 15)) return EGO_ERR_ARG;
    const dim3 grid(B * H * ((Nq + 127) / 128));
    HD_DISPATCH(hd_fwd_kernel, grid);
    LAUNCH_CHECK();
    return EGO_OK;
}

extern "C" int ego_attn_bwd_hd(const void* Q, long q_bs, long q_rs, const void* K, long k_bs, long k_rs, const void* V,
                               long v_bs, long v_rs, const void* O, long o_bs, long o_rs, const void* O_lo, const void* dO,
                               long do_bs, long do_rs, const float* LSE, float* DELTA, void* dQ, long dq_bs, long dq_rs,
                               void* dK, long dk_bs, long dk_rs, void* dV, long dv_bs, long dv_rs, const int* ks,
                               const int* ke, long r_bs, long r_rs, int B, int H, int Nq, int Nk, int hd_pad, float scale,
                               hipStream_t stream) {
    HdArgs a{};
    a.Q = (const bf16_t*)Q; a.K = (const bf16_t*)K; a.V = (const bf16_t*)V;
    a.q_bs = q_bs; a.q_rs = q_rs; a.k_bs = k_bs; a.k_rs = k_rs; a.v_bs = v_bs; a.v_rs = v_rs;
    a.LSE = (float*)LSE; a.ks = ks; a.ke = ke; a.r_bs = r_bs; a.r_rs = r_rs;
    a.B = B; a.H = H; a.Nq = Nq; a.Nk = Nk; a.scale = scale;
    a.O = (bf16_t*)O; a.o_bs = o_bs; a.o_rs = o_rs; a.Olo = (bf16_t*)O_lo;
    a.dO = (const bf16_t*)dO; a.do_bs = do_bs; a.do_rs = do_rs; a.DELTA = DELTA;
    a.dQ = (bf16_t*)dQ; a.dq_bs = dq_bs; a.dq_rs = dq_rs;
    a.dK = (bf16_t*)dK; a.dk_bs = dk_bs; a.dk_rs = dk_rs;
    a.dV = (bf16_t*)dV; a.dv_bs = dv_bs; a.dv_rs = dv_rs;
    if (B == 0 || Nq == 0) return EGO_OK;
    const HdShape sh = hd_shape(hd_pad);
    if (!check(a, sh.hdp) || do_rs % 8 || do_bs % 8 || dq_rs % 8 || dk_rs % 8 || dv_rs % 8 || dq_bs % 8 || dk_bs % 8 || dv_bs % 8 ||
        ((((uintptr_t)Q) | ((uintptr_t)K) | ((uintptr_t)V) | ((uintptr_t)O) | ((uintptr_t)O_lo) | ((uintptr_t)dO) | ((uintptr_t)dQ) |
          ((uintptr_t)dK) | ((uintptr_t)dV)) & 15)) return EGO_ERR_ARG;      // 16-byte row fragments, gradient rows included
    const long rows = (long)B * H * Nq;
    const dim3 gq(B * H * ((Nq + 127) / 128)), gk(B * H * ((Nk + 127) / 128)), gd((unsigned)((rows + 255) / 256));
    if (sh.hdp == 96) { EGO_LAUNCH(hd_delta_kernel<96>, gd, dim3(256), 0, stream, a); }
    else { EGO_LAUNCH(hd_delta_kernel<128>, gd, dim3(256), 0, stream, a); }
    HD_DISPATCH(hd_dq_kernel, gq);
    HD_DISPATCH(hd_dkv_kernel, gk);
    LAUNCH_CHECK();
    return EGO_OK;
}
