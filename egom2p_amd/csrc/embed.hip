// Front end of the EgoM2P hot path on gfx950: mask compaction (stable partition by prefix sum),
// fused token/positional/modality embedding gather, loss-row permutation, and the embedding backward.
//
// Replaces (reference egom2p/models/egom2p_model.py):
//   cat_encoder_tensors :251-283, forward_mask_encoder :344-396  (argsort(mask + arange*1e-6) -> gather x4)
//   cat_decoder_tensors :285-342, forward_mask_decoder :398-444, adapt_decoder_attention_mask :446-481
//   encoder/decoder embedding modules (encoder_embeddings.py:181-210,272-301; decoder_embeddings.py:337-370,455-487)
// Only the N (M) kept rows are ever embedded; the (B,T,D) tensors of the reference never exist.
// Integer outputs are bit-exact with the reference; float rows are exact fp32 sums in the same order
// (token + (pos + mod)).
#include "common.h"
#include "egom2p_hip.h"

namespace {

struct CompactArgs {
    const unsigned char* mask[EGO_MAX_MODS];   // [B, n_pos[m]]  (True = ignore)
    const long long* ids[EGO_MAX_MODS];        // [B, n_pos[m]]
    const int* dam[EGO_MAX_MODS];              // decoder_attention_mask [B, n_pos[m]] or null
    int n_pos[EGO_MAX_MODS];
    int mod_id[EGO_MAX_MODS];
    int n_mods, T, n_keep, is_decoder;
    int n_reg;                // register rows in front of every sample's kept rows (encoder only): outputs are [B, n_reg + n_keep]
    long long* ids_keep;      // [B, n_keep]
    unsigned char* pad;       // [B, n_keep]
    short* mod_mask;          // [B, n_keep]  (-1 on pads)
    int* slot;                // [B, n_keep]  modality slot in this ordering (-1 on pads)
    int* local;               // [B, n_keep]  position inside the modality
    int* tok;                 // [B, n_keep]  token id at the kept position (0 on pads)
    int* ks; int* ke;         // [B, n_keep]  allowed key interval per row
    int* n_valid;             // [B]
    int* seg;                 // [B, n_mods, 2]  (start, count) of each slot's kept unmasked rows
    int* err;                 // [1] set if the decoder mask is not one interval per row
    int* seg_bad;             // [B] (optional) set if a kept unmasked row's interval is not exactly its slot's segment
};

// CP_SPLIT workgroups per sample, each owning a run of 256-position groups.  A workgroup first sweeps the WHOLE
// sample's mask / decoder_attention_mask once with coalesced loads (10,300 bytes + 41 KB: the totals and the prefix in
// front of its own run cost less than a hand-off between workgroups would), then walks its run 256 positions at a
// time: per wave one ballot gives the rank among the unmasked positions, two 6-step shuffle scans give the running
// decoder_attention_mask sums among unmasked / masked positions, the four waves are chained through LDS.
constexpr int CP_SPLIT = 8;

__device__ __forceinline__ int wave_sum_int(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ int wave_incl_scan(int v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(v, o, 64);
        if (lane >= o) v += t;
    }
    return v;
}

__global__ __launch_bounds__(256) void compact_kernel(CompactArgs a) {
    constexpr int NC = 5 + EGO_MAX_MODS;           // pre {cnt, dam unmasked, dam masked}, total {cnt, dam unmasked}, per-modality cnt
    __shared__ int s_part[4][NC];
    __shared__ int s_wave[2][4][3];
    const int b = blockIdx.x / CP_SPLIT, part = blockIdx.x % CP_SPLIT, tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int groups = (a.T + 255) >> 8, gper = (groups + CP_SPLIT - 1) / CP_SPLIT;
    const int c0 = min(a.T, part * gper * 256), c1 = min(a.T, c0 + gper * 256);

    int moff[EGO_MAX_MODS + 1];
    moff[0] = 0;
#pragma unroll
    for (int m = 0; m < EGO_MAX_MODS; ++m) moff[m + 1] = moff[m] + (m < a.n_mods ? a.n_pos[m] : 0);
    auto locate = [&](int pos, int& m, int& loc) {
        m = 0;
        int off = 0;
#pragma unroll
        for (int k = 1; k < EGO_MAX_MODS; ++k) if (k < a.n_mods && pos >= moff[k]) { m = k; off = moff[k]; }
        loc = pos - off;
    };
    auto fetch = [&](int pos, int& m, int& loc, bool& masked, int& d) {
        locate(pos, m, loc);
        const long src = (long)b * a.n_pos[m] + loc;
        masked = a.mask[m][src] != 0;
        d = (a.is_decoder && a.dam[m]) ? a.dam[m][src] : 0;
    };

    // ---- pass 1: whole-sample counts and the prefix in front of this workgroup's run
    int acc[NC];
#pragma unroll
    for (int i = 0; i < NC; ++i) acc[i] = 0;
    for (int pos = tid; pos < a.T; pos += 256) {
        int m, loc, d; bool masked;
        fetch(pos, m, loc, masked, d);
        const bool pre = pos < c0;
        if (!masked) {
            acc[3] += 1; acc[4] += d;
            if (pre) { acc[0] += 1; acc[1] += d; }
#pragma unroll
            for (int k = 0; k < EGO_MAX_MODS; ++k) acc[5 + k] += (m == k) ? 1 : 0;
        } else if (pre) {
            acc[2] += d;
        }
    }
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        const int v = wave_sum_int(acc[i]);
        if (lane == 0) s_part[wave][i] = v;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NC; ++i) acc[i] = s_part[0][i] + s_part[1][i] + s_part[2][i] + s_part[3][i];
    const int total_valid = acc[3], total_dam = acc[4];
    const int nv = min(total_valid, a.n_keep);
    int segstart[EGO_MAX_MODS + 1];
    segstart[0] = 0;
#pragma unroll
    for (int m = 0; m < EGO_MAX_MODS; ++m) segstart[m + 1] = segstart[m] + acc[5 + m];
    const int pitch = a.n_reg + a.n_keep;           // entries per sample of every per-row output
    if (part == 0 && tid == 0) {
        a.n_valid[b] = a.n_reg + nv;
        for (int m = 0; m < a.n_mods; ++m) {
            const int s0 = min(segstart[m], a.n_keep), s1 = min(segstart[m + 1], a.n_keep);
            a.seg[((long)b * a.n_mods + m) * 2] = a.n_reg + s0;
            a.seg[((long)b * a.n_mods + m) * 2 + 1] = s1 - s0;
        }
    }
    if (part == 0 && tid < a.n_reg) {
        // register rows (egom2p_model.py:381-387): never padding, no modality (mod_mask -1), slot -2 tells the embedding kernel
        // to take row `local` of the register tokens; every key of the sample (registers included) is visible to them
        const long o = (long)b * pitch + tid;
        a.ids_keep[o] = -1;
        a.pad[o] = 0;
        a.mod_mask[o] = (short)-1;
        a.slot[o] = -2;
        a.local[o] = tid;
        a.tok[o] = 0;
        a.ks[o] = 0;
        a.ke[o] = a.n_reg + nv;
    }

    // ---- pass 2: scatter this workgroup's run
    int run_cnt = acc[0], run_du = acc[1], run_dm = acc[2];
    int par = 0;
    for (int base = c0; base < c1; base += 256, par ^= 1) {
        const int pos = base + tid;
        const bool valid = pos < c1;
        int m = 0, loc = 0, d = 0; bool masked = true;
        if (valid) fetch(pos, m, loc, masked, d);
        const bool unm = valid && !masked;
        const unsigned long long ball = __ballot(unm);
        const int rank = __popcll(ball & ((1ull << lane) - 1ull));
        const int du = wave_incl_scan(unm ? d : 0, lane), dm = wave_incl_scan((valid && masked) ? d : 0, lane);
        if (lane == 63) { s_wave[par][wave][0] = __popcll(ball); s_wave[par][wave][1] = du; s_wave[par][wave][2] = dm; }
        __syncthreads();
        int w_cnt = 0, w_du = 0, w_dm = 0, t_cnt = 0, t_du = 0, t_dm = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const int c = s_wave[par][w][0], u = s_wave[par][w][1], q = s_wave[par][w][2];
            if (w < wave) { w_cnt += c; w_du += u; w_dm += q; }
            t_cnt += c; t_du += u; t_dm += q;
        }
        if (valid) {
            const int uidx = run_cnt + w_cnt + rank;          // unmasked positions in front of this one
            int out, cs;
            if (unm) { out = uidx; cs = run_du + w_du + du; }
            else { out = total_valid + (pos - uidx); cs = total_dam + run_dm + w_dm + dm; }
            if (out < a.n_keep) {
                const long o = (long)b * pitch + a.n_reg + out;
                a.ids_keep[o] = pos;
                a.pad[o] = masked ? 1 : 0;
                a.mod_mask[o] = masked ? (short)-1 : (short)a.mod_id[m];
                a.slot[o] = masked ? -1 : m;
                a.local[o] = loc;
                a.tok[o] = masked ? 0 : (int)a.ids[m][(long)b * a.n_pos[m] + loc];
                if (a.is_decoder) {
                    // allowed keys: same modality (ids compared before pads become -1) and j < cumsum(dam)_i.
                    // Same-modality unmasked keys are the contiguous segment [s0, s1).
                    int sa = 0, sb = 0;
#pragma unroll
                    for (int k = 0; k < EGO_MAX_MODS; ++k) if (k == m) { sa = segstart[k]; sb = segstart[k + 1]; }
                    const int s0 = min(sa, a.n_keep), s1 = min(sb, a.n_keep);
                    a.ks[o] = s0;
                    a.ke[o] = min(s1, cs);
                    if (cs > nv && nv < a.n_keep) atomicOr(a.err, 1);   // a pad key would be visible: not an interval mask
                    // the data contract's marker (masking.py:262-264) makes every target row see exactly its modality's
                    // segment; anything else sends the sample down the attention kernels' per-row path
                    if (unm && a.seg_bad && cs < s1) atomicOr(a.seg_bad + b, 1);
                } else {
                    a.ks[o] = 0;
                    a.ke[o] = a.n_reg + nv;
                }
            }
        }
        run_cnt += t_cnt; run_du += t_du; run_dm += t_dm;
    }
}

// ---------------------------------------------------------------------------------------------
// embedding gather: one wave per kept row
// ---------------------------------------------------------------------------------------------
struct EmbedArgs {
    const float* table[EGO_MAX_MODS];   // token tables [V, D] or null (decoder: rows are the mask token)
    const float* pos[EGO_MAX_MODS];     // [n_pos, D]
    const float* mod[EGO_MAX_MODS];     // [D]
    const float* base_vec;              // mask_token [D] or null
    const int* slot; const int* local; const int* tok;
    float* x; float* emb;               // [rows, D]; emb may be null
    long rows; int D;
    const float* reg;                   // register tokens [n_reg, D] or null (rows with slot -2)
};

__global__ __launch_bounds__(256) void embed_kernel(EmbedArgs a) {
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= a.rows) return;
    const int s = a.slot[row];
    float* xr = a.x + row * a.D;
    float* er = a.emb ? a.emb + row * a.D : nullptr;
    const int nc = a.D >> 2;
    if (s < 0) {
        // padding row: zeros.  Register row (slot -2): x = register_tokens[local] + 0, emb = 0 (egom2p_model.py:384-385)
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        const float* rr = (s == -2 && a.reg) ? a.reg + (long)a.local[row] * a.D : nullptr;
        for (int c = lane; c < nc; c += 64) { *(f32x4*)(xr + c * 4) = rr ? *(const f32x4*)(rr + c * 4) : z; if (er) *(f32x4*)(er + c * 4) = z; }
        return;
    }
    const float *tb = nullptr, *ps = nullptr, *md = nullptr;
#pragma unroll
    for (int m = 0; m < EGO_MAX_MODS; ++m) if (m == s) { tb = a.table[m]; ps = a.pos[m]; md = a.mod[m]; }
    const float* tr = tb ? tb + (long)a.tok[row] * a.D : a.base_vec;
    const float* pr = ps + (long)a.local[row] * a.D;
    for (int c = lane; c < nc; c += 64) {
        const f32x4 e = *(const f32x4*)(pr + c * 4) + *(const f32x4*)(md + c * 4);
        const f32x4 t = *(const f32x4*)(tr + c * 4);
        *(f32x4*)(xr + c * 4) = t + e;
        if (er) *(f32x4*)(er + c * 4) = e;
    }
}

// ---------------------------------------------------------------------------------------------
// loss-row permutation: decoder rows grouped by modality (canonical order), (b, i) row-major inside,
// exactly the row order of y[decoder_mod_mask == id] (egom2p_model.py:633).  One workgroup.
//   seg   [B, n_mods, 2]  (start,count) per *decoder-order* slot
//   canon [n_mods]        decoder slot -> canonical modality index
//   perm  [B*M]           row -> grouped row (-1 for pads);  tgt_perm[grouped row] = target id
//   ranges[n_mods, 2]     (offset, count) of each canonical modality in the grouped order
// ---------------------------------------------------------------------------------------------
// Every workgroup rebuilds the small (batch x modality) offset table in LDS (B * n_mods counts: cheaper than a hand-off)
// and then maps its share of the rows.
__global__ __launch_bounds__(256) void loss_perm_kernel(const int* __restrict__ seg, const int* __restrict__ canon,
                                                        const int* __restrict__ slot, const int* __restrict__ tok,
                                                        int B, int M, int n_mods, int* __restrict__ perm,
                                                        int* __restrict__ tgt_perm, int* __restrict__ ranges,
                                                        int* __restrict__ base /* [B, n_mods] (also an output: row offsets) */) {
    extern __shared__ int lp_sm[];                  // start[B*n_mods] count[B*n_mods] base[B*n_mods] tot[EGO_MAX_MODS]
    const int nbm = B * n_mods;
    int* s_start = lp_sm; int* s_cnt = lp_sm + nbm; int* s_base = lp_sm + 2 * nbm; int* s_tot = lp_sm + 3 * nbm;
    for (int i = threadIdx.x; i < nbm; i += 256) { s_start[i] = seg[2 * i]; s_cnt[i] = seg[2 * i + 1]; }
    __syncthreads();
    if (threadIdx.x < n_mods) {                     // total kept rows of decoder slot s over the batch
        int t = 0;
        for (int b = 0; b < B; ++b) t += s_cnt[b * n_mods + threadIdx.x];
        s_tot[threadIdx.x] = t;
    }
    __syncthreads();
    if (threadIdx.x < n_mods) {                     // canonical modality c: rows of all earlier modalities come first
        const int c = threadIdx.x;
        int sl = 0, run = 0;
        for (int s = 0; s < n_mods; ++s) if (canon[s] == c) sl = s;
        for (int c2 = 0; c2 < c; ++c2)
            for (int s = 0; s < n_mods; ++s) if (canon[s] == c2) run += s_tot[s];
        if (blockIdx.x == 0) { ranges[2 * c] = run; ranges[2 * c + 1] = s_tot[sl]; }
        for (int b = 0; b < B; ++b) {
            s_base[b * n_mods + sl] = run;
            if (blockIdx.x == 0) base[b * n_mods + sl] = run;
            run += s_cnt[b * n_mods + sl];
        }
    }
    __syncthreads();
    const long total = (long)B * M;
    for (long r = (long)blockIdx.x * 256 + threadIdx.x; r < total; r += (long)gridDim.x * 256) {
        const int s = slot[r];
        if (s < 0) { perm[r] = -1; continue; }
        const int b = (int)(r / M), i = (int)(r % M);
        const int g = s_base[b * n_mods + s] + (i - s_start[b * n_mods + s]);
        perm[r] = g;
        tgt_perm[g] = tok[r];
    }
}

// ---------------------------------------------------------------------------------------------
// embedding backward: table[slot][tok] += dx row ; dmod[slot] += (dx + d2) row ; dbase += dx row.
//
// No float atomics - every sum is taken in an order that depends on the data only, so the gradients are bitwise
// reproducible (and the table scatter no longer runs at the ~1.3 TB/s atomic rate of the chip):
//   * embed_sums_kernel: modality / base column sums.  A wave walks 32 consecutive rows and keeps the sum of the CURRENT
//     slot in registers (kept rows come in runs of one modality); on a slot change it adds the run into its own LDS
//     rows.  The 4 waves' rows are combined in wave order into ONE partial row per workgroup, plain stores; the sum over
//     workgroups is colsum_kernel's ordered reduction.
//   * embed_tables_kernel: the table scatter as a gather.  Workgroup j owns the table rows with token id = j (mod ET_WGS):
//     it scans the (slot, token) keys of all rows, collects its own in ROW ORDER (ballot + prefix positions), ranks them by
//     (key, position) and sums every key's dx rows in ascending row order into the table row (one owner per table row:
//     plain read-modify-write).  Keys with 64 or more rows are summed by the 4 waves in contiguous quarters and joined in
//     wave order.
// ---------------------------------------------------------------------------------------------
struct EmbedBwdArgs {
    float* dtable[EGO_MAX_MODS];   // [V, D] grads or null
    float* dmod[EGO_MAX_MODS];     // [D]
    float* dbase;                  // [D] mask-token grad or null
    const float* dx; const float* d2;   // [rows, D]; d2 may be null
    const int* slot; const int* tok;
    long rows; int D, n_mods;
    unsigned char* touched[EGO_MAX_MODS];   // optional [V]: set to 1 for every table row that receives a gradient
    float* part;                   // [workgroups][(n_mods + 1) * D] partial column sums
};

constexpr int ES_ROWS = 128;       // rows per workgroup of embed_sums_kernel (32 per wave)

template <int MAXC>
__global__ __launch_bounds__(256) void embed_sums_kernel(EmbedBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lsum[];   // [4 waves][(n_mods + 1)][D]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nsl = a.n_mods + 1, W = nsl * a.D, nc = a.D >> 2;
    float* mine = lsum + wave * W;
    for (int i = lane; i < W; i += 64) mine[i] = 0.f;
    f32x4 acc[MAXC], accb[MAXC];
#pragma unroll
    for (int i = 0; i < MAXC; ++i) { acc[i] = f32x4{0.f, 0.f, 0.f, 0.f}; accb[i] = acc[i]; }
    int cur = -1;
    auto flush = [&]() {
        if (cur < 0) return;
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            const int c = lane + 64 * i;
            if (c < nc) {
                f32x4* d = (f32x4*)(mine + cur * a.D + c * 4);
                *d += acc[i];
                acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
    };
    const int waves = blockDim.x >> 6;                 // 4; 1 when four row sets of (n_mods + 1) x D floats do not fit the LDS
    const long r0 = ((long)blockIdx.x * waves + wave) * 32;
    for (int rr = 0; rr < 32; ++rr) {
        const long row = r0 + rr;
        if (row >= a.rows) break;
        const int s = a.slot[row];
        if (s < 0) continue;
        if (s != cur) { flush(); cur = s; }
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            const int c = lane + 64 * i;
            if (c < nc) {
                f32x4 g = *(const f32x4*)(a.dx + row * a.D + c * 4);
                if (a.dbase) accb[i] += g;
                if (a.d2) g += *(const f32x4*)(a.d2 + row * a.D + c * 4);
                acc[i] += g;
            }
        }
    }
    flush();
    if (a.dbase) {
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            const int c = lane + 64 * i;
            if (c < nc) *(f32x4*)(mine + a.n_mods * a.D + c * 4) = accb[i];
        }
    }
    __syncthreads();
    if (waves == 4) {
        for (int i = threadIdx.x; i < W; i += 256)
            a.part[(long)blockIdx.x * W + i] = ((lsum[i] + lsum[W + i]) + lsum[2 * W + i]) + lsum[3 * W + i];
    } else {
        for (int i = threadIdx.x; i < W; i += 64) a.part[(long)blockIdx.x * W + i] = lsum[i];
    }
}

constexpr int ET_WGS = 512;        // owners of the table rows (token id mod ET_WGS)
constexpr int ET_CAP = 2048;       // (key, row) pairs a workgroup holds between two flushes
constexpr int ET_HEAVY = 64;       // keys with at least this many rows are summed by the whole workgroup

template <int MAXC>
__global__ __launch_bounds__(256) void embed_tables_kernel(EmbedBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) char et_sm[];
    int* lkey = (int*)et_sm;                  // [ET_CAP] collected in row order
    int* lrow = lkey + ET_CAP;
    int* skey = lrow + ET_CAP;                // [ET_CAP] sorted by (key, position)
    int* srow = skey + ET_CAP;
    int* heads = srow + ET_CAP;               // [ET_CAP] start positions of the keys' runs in skey / srow
    int* wcnt = heads + ET_CAP;               // [8]: per-wave match counts, number of heads
    float* comb = (float*)(wcnt + 8);         // [4][D] partial sums of a heavy key
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = blockIdx.x, nc = a.D >> 2;
    int n = 0;                                 // pairs collected (uniform)

    auto sum_rows = [&](const int* rows_, int cnt, f32x4 (&acc)[MAXC]) {     // ascending positions, one wave
#pragma unroll
        for (int i = 0; i < MAXC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        int q = 0;
        for (; q + 4 <= cnt; q += 4) {
            f32x4 v[4][MAXC];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float* src = a.dx + (long)rows_[q + k] * a.D;
#pragma unroll
                for (int i = 0; i < MAXC; ++i) { const int c = lane + 64 * i; if (c < nc) v[k][i] = *(const f32x4*)(src + c * 4); }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int i = 0; i < MAXC; ++i) { const int c = lane + 64 * i; if (c < nc) acc[i] += v[k][i]; }
        }
        for (; q < cnt; ++q) {
            const float* src = a.dx + (long)rows_[q] * a.D;
#pragma unroll
            for (int i = 0; i < MAXC; ++i) { const int c = lane + 64 * i; if (c < nc) acc[i] += *(const f32x4*)(src + c * 4); }
        }
    };
    auto table_row = [&](int key, unsigned char*& fl) -> float* {
        const int s = key >> 16, t = key & 0xffff;
        float* tb = nullptr;
        fl = nullptr;
#pragma unroll
        for (int m = 0; m < EGO_MAX_MODS; ++m) if (m == s) { tb = a.dtable[m]; fl = a.touched[m]; }
        if (fl) fl += t;
        return tb + (long)t * a.D;
    };

    auto flush = [&]() {
        __syncthreads();
        // rank by (key, position): stable, so every key's rows stay in ascending row order
        for (int i = tid; i < n; i += 256) {
            const int k = lkey[i];
            int rank = 0;
            for (int e = 0; e < n; ++e) { const int ke = lkey[e]; rank += (ke < k) || (ke == k && e < i); }
            skey[rank] = k; srow[rank] = lrow[i];
        }
        if (tid == 0) wcnt[4] = 0;
        __syncthreads();
        // run heads in ascending order: every 256-block of positions appends its heads behind the previous blocks'
        for (int base = 0; base < n; base += 256) {
            const int i = base + tid;
            const bool head = i < n && (i == 0 || skey[i] != skey[i - 1]);
            const unsigned long long m = __ballot(head);
            if (lane == 0) wcnt[wave] = __builtin_popcountll(m);
            __syncthreads();
            int off = wcnt[4];
            for (int w = 0; w < wave; ++w) off += wcnt[w];
            if (head) heads[off + __builtin_popcountll(m & ((1ull << lane) - 1ull))] = i;
            __syncthreads();
            if (tid == 0) wcnt[4] += wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
            __syncthreads();
        }
        const int nh = wcnt[4];
        // light keys: wave w takes runs w, w + 4, ...
        for (int h = wave; h < nh; h += 4) {
            const int p0 = heads[h], p1 = (h + 1 < nh) ? heads[h + 1] : n;
            if (p1 - p0 >= ET_HEAVY) continue;
            f32x4 acc[MAXC];
            sum_rows(srow + p0, p1 - p0, acc);
            unsigned char* fl;
            float* dst = table_row(skey[p0], fl);
#pragma unroll
            for (int i = 0; i < MAXC; ++i) { const int c = lane + 64 * i; if (c < nc) *(f32x4*)(dst + c * 4) += acc[i]; }
            if (fl && lane == 0) *fl = 1;
        }
        // heavy keys: the four waves sum contiguous quarters, wave 0 joins them in wave order
        for (int h = 0; h < nh; ++h) {
            const int p0 = heads[h], p1 = (h + 1 < nh) ? heads[h + 1] : n, cnt = p1 - p0;
            if (cnt < ET_HEAVY) continue;
            __syncthreads();
            const int q = (cnt + 3) / 4, b0 = min(cnt, wave * q), b1 = min(cnt, b0 + q);
            f32x4 acc[MAXC];
            sum_rows(srow + p0 + b0, b1 - b0, acc);
#pragma unroll
            for (int i = 0; i < MAXC; ++i) { const int c = lane + 64 * i; if (c < nc) *(f32x4*)(comb + wave * a.D + c * 4) = acc[i]; }
            __syncthreads();
            if (wave == 0) {
                unsigned char* fl;
                float* dst = table_row(skey[p0], fl);
#pragma unroll
                for (int i = 0; i < MAXC; ++i) {
                    const int c = lane + 64 * i;
                    if (c < nc) {
                        const f32x4 t = ((*(const f32x4*)(comb + c * 4) + *(const f32x4*)(comb + a.D + c * 4)) +
                                         *(const f32x4*)(comb + 2 * a.D + c * 4)) + *(const f32x4*)(comb + 3 * a.D + c * 4);
                        *(f32x4*)(dst + c * 4) += t;
                    }
                }
                if (fl && lane == 0) *fl = 1;
            }
        }
        __syncthreads();
        n = 0;
    };

    // this lane's key for row r, or -1 (not a table row of this workgroup)
    auto key_of = [&](long r) -> int {
        if (r >= a.rows) return -1;
        const int sl = a.slot[r];
        if (sl < 0) return -1;
        bool has = false;
#pragma unroll
        for (int m = 0; m < EGO_MAX_MODS; ++m) if (m == sl) has = a.dtable[m] != nullptr;
        const int t = a.tok[r];
        return (has && (t & (ET_WGS - 1)) == j) ? ((sl << 16) | t) : -1;
    };
    // Fast path (the usual case: a workgroup owns ~rows / ET_WGS pairs): wave w scans the w-th quarter of the rows on its own
    // - no workgroup barrier per step - first counting its pairs, then, with the four counts known, writing them at its
    // offset: the list is in row order.  A lane takes 4 consecutive rows per step (16-byte loads of slot / token ids), two
    // steps are in flight at a time - the scan is L2-latency bound otherwise (measured: 2 x 512 dependent steps = 290 us).
    // Falls back to the stepwise scan with flushes when the pairs do not fit.
    const long quarter = ((a.rows + 3) / 4 + 511) / 512 * 512;
    const long q0 = wave * quarter, q1 = min(a.rows, q0 + quarter);
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    const bool vec_ok = ((((uintptr_t)a.slot) | ((uintptr_t)a.tok)) & 15) == 0;
    auto keys4 = [&](long r0, int (&k)[4]) {                      // keys of rows r0 .. r0 + 3
        if (vec_ok && r0 + 4 <= a.rows) {
            const i32x4 sl = *(const i32x4*)(a.slot + r0), tk = *(const i32x4*)(a.tok + r0);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                bool has = false;
#pragma unroll
                for (int m = 0; m < EGO_MAX_MODS; ++m) if (m == sl[e]) has = a.dtable[m] != nullptr;
                k[e] = (sl[e] >= 0 && has && (tk[e] & (ET_WGS - 1)) == j) ? ((sl[e] << 16) | tk[e]) : -1;
            }
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) k[e] = key_of(r0 + e);
        }
    };
    int mine = 0;
    for (long base = q0; base < q1; base += 512) {
        int k0[4], k1[4];
        keys4(base + lane * 4, k0);
        keys4(base + 256 + lane * 4, k1);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            mine += __builtin_popcountll(__ballot(k0[e] >= 0 && base + lane * 4 + e < q1));
            mine += __builtin_popcountll(__ballot(k1[e] >= 0 && base + 256 + lane * 4 + e < q1));
        }
    }
    if (lane == 0) wcnt[wave] = mine;
    __syncthreads();
    const int total = wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
    if (total <= ET_CAP) {
        int off = 0;
        for (int w = 0; w < wave; ++w) off += wcnt[w];
        __syncthreads();
        const unsigned long long lt = (1ull << lane) - 1ull;
        for (long base = q0; base < q1; base += 512) {
            int k[2][4];
            keys4(base + lane * 4, k[0]);
            keys4(base + 256 + lane * 4, k[1]);
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                const long r0 = base + hf * 256 + lane * 4;
                unsigned long long m[4];
                int before = 0, all = 0;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (r0 + e >= q1) k[hf][e] = -1;
                    m[e] = __ballot(k[hf][e] >= 0);
                    before += __builtin_popcountll(m[e] & lt);              // pairs of lower lanes (rows below r0)
                    all += __builtin_popcountll(m[e]);
                }
                int pos = off + before;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (k[hf][e] >= 0) { lkey[pos] = k[hf][e]; lrow[pos] = (int)(r0 + e); ++pos; }
                off += all;
            }
        }
        n = total;
        if (n > 0) flush();
        return;
    }
    __syncthreads();
    for (long base = 0; base < a.rows; base += 256) {
        const long r = base + tid;
        const int key = key_of(r);
        const unsigned long long m = __ballot(key >= 0);
        if (lane == 0) wcnt[wave] = __builtin_popcountll(m);
        __syncthreads();
        int off = n;
        for (int w = 0; w < wave; ++w) off += wcnt[w];
        if (key >= 0) {
            const int pos = off + __builtin_popcountll(m & ((1ull << lane) - 1ull));
            lkey[pos] = key; lrow[pos] = (int)r;
        }
        n += wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
        __syncthreads();
        if (n > ET_CAP - 256) flush();
    }
    if (n > 0) flush();
}

// ---------------------------------------------------------------------------------------------
// Row lists for the sparse exchange of embedding-table gradients: with few clips per
// optimiser step only a few thousand of a table's 64,000 rows carry a gradient, and ranks exchange (row id, row) lists
// instead of all-reducing the dense table.
// ---------------------------------------------------------------------------------------------
// touched[V] -> ascending row list (at most cap entries), count; the flags are cleared for the next step.  One workgroup.
__global__ __launch_bounds__(1024) void rows_compact_kernel(unsigned char* __restrict__ touched, int V, int cap,
                                                            int* __restrict__ rows, int* __restrict__ count) {
    __shared__ int s_w[16];
    __shared__ int s_run;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_run = 0;
    __syncthreads();
    for (int base = 0; base < V; base += 1024) {
        const int i = base + tid;
        const bool on = i < V && touched[i] != 0;
        if (on) touched[i] = 0;
        const unsigned long long ball = __ballot(on);
        if (lane == 0) s_w[wave] = __popcll(ball);
        __syncthreads();
        int pre = s_run;
        for (int w = 0; w < wave; ++w) pre += s_w[w];
        const int pos = pre + __popcll(ball & ((1ull << lane) - 1ull));
        if (on && pos < cap) rows[pos] = i;
        __syncthreads();
        if (tid == 0) { int t = 0; for (int w = 0; w < 16; ++w) t += s_w[w]; s_run += t; }
        __syncthreads();
    }
    if (tid == 0) *count = min(s_run, cap) | (s_run > cap ? 0x40000000 : 0);      // bit 30: the list did not fit (caller's cap too small)
    for (int i = min(s_run, cap) + tid; i < cap; i += 1024) rows[i] = -1;
}

// out[i, :] = T[rows[i], :] for i < count, 0 for count <= i < cap      (one wave per row, 16 bytes per lane)
__global__ __launch_bounds__(256) void rows_gather_kernel(const float* __restrict__ T, const int* __restrict__ rows,
                                                          const int* __restrict__ count, int cap, int D, float* __restrict__ out) {
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= cap) return;
    const int n = *count & 0x3fffffff;
    const int r = i < n ? rows[i] : -1;
    for (int c = lane * 4; c < D; c += 256) {
        const f32x4 v = r >= 0 ? *(const f32x4*)(T + (long)r * D + c) : f32x4{0.f, 0.f, 0.f, 0.f};
        *(f32x4*)(out + (long)i * D + c) = v;
    }
}

// T[rows[i], :] = (add ? T[rows[i], :] : 0) + src[i, :] for i < count; rows of one list are distinct, so plain stores
__global__ __launch_bounds__(256) void rows_scatter_kernel(float* __restrict__ T, const int* __restrict__ rows, const int* __restrict__ count,
                                                           int cap, int D, const float* __restrict__ src, int add) {
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= cap) return;
    const int n = min(*count & 0x3fffffff, cap);
    if (i >= n) return;
    const int r = rows[i];
    if (r < 0) return;
    for (int c = lane * 4; c < D; c += 256) {
        f32x4 v = src ? *(const f32x4*)(src + (long)i * D + c) : f32x4{0.f, 0.f, 0.f, 0.f};     // no source: clear the rows
        if (add) v += *(const f32x4*)(T + (long)r * D + c);
        *(f32x4*)(T + (long)r * D + c) = v;
    }
}

}  // namespace

extern "C" int ego_rows_compact(void* touched, int V, int cap, int* rows, int* count, hipStream_t stream) {
    if (V <= 0 || cap <= 0) return EGO_ERR_ARG;
    EGO_LAUNCH(rows_compact_kernel, dim3(1), dim3(1024), 0, stream, (unsigned char*)touched, V, cap, rows, count);
    LAUNCH_CHECK();
    return EGO_OK;
}

extern "C" int ego_rows_gather(const float* table, const int* rows, const int* count, int cap, int D, float* out, hipStream_t stream) {
    if (cap <= 0 || D % 4) return EGO_ERR_ARG;
    EGO_LAUNCH(rows_gather_kernel, dim3((cap + 3) / 4), dim3(256), 0, stream, table, rows, count, cap, D, out);
    LAUNCH_CHECK();
    return EGO_OK;
}

extern "C" int ego_rows_scatter(float* table, const int* rows, const int* count, int cap, int D, const float* src, int add,
                                hipStream_t stream) {
    if (cap <= 0 || D % 4) return EGO_ERR_ARG;
    EGO_LAUNCH(rows_scatter_kernel, dim3((cap + 3) / 4), dim3(256), 0, stream, table, rows, count, cap, D, src, add);
    LAUNCH_CHECK();
    return EGO_OK;
}

extern "C" int ego_compact(const ego_compact_desc* d, int B, hipStream_t stream) {
    if (!d || d->n_mods <= 0 || d->n_mods > EGO_MAX_MODS || B <= 0) return EGO_ERR_ARG;
    CompactArgs a{};
    int T = 0;
    for (int m = 0; m < d->n_mods; ++m) {
        a.mask[m] = (const unsigned char*)d->mask[m];
        a.ids[m] = (const long long*)d->ids[m];
        a.dam[m] = d->dam[m];
        a.n_pos[m] = d->n_pos[m];
        a.mod_id[m] = d->mod_id[m];
        T += d->n_pos[m];
    }
    // register tokens: encoder side only, at most one workgroup's threads; with them a pass may keep no input row at all
    if (d->n_reg < 0 || d->n_reg > 256 || (d->n_reg && d->is_decoder)) return EGO_ERR_ARG;
    if (d->n_keep < (d->n_reg ? 0 : 1) || d->n_keep > T) return EGO_ERR_ARG;
    a.n_mods = d->n_mods; a.T = T; a.n_keep = d->n_keep; a.is_decoder = d->is_decoder; a.n_reg = d->n_reg;
    a.ids_keep = (long long*)d->ids_keep; a.pad = (unsigned char*)d->pad; a.mod_mask = (short*)d->mod_mask;
    a.slot = d->slot; a.local = d->local; a.tok = d->tok; a.ks = d->ks; a.ke = d->ke;
    a.n_valid = d->n_valid; a.seg = d->seg; a.err = d->err; a.seg_bad = d->is_decoder ? d->seg_bad : nullptr;
    if (a.seg_bad && hipMemsetAsync(a.seg_bad, 0, (size_t)B * sizeof(int), stream) != hipSuccess) return EGO_ERR_LAUNCH;
    EGO_LAUNCH(compact_kernel, dim3(B * CP_SPLIT), dim3(256), 0, stream, a);
    LAUNCH_CHECK();
    return EGO_OK;
}

extern "C" int ego_embed_fwd(const ego_embed_desc* d, hipStream_t stream) {
    if (!d || d->rows <= 0 || d->D % 4) return EGO_ERR_ARG;
    EmbedArgs a{};
    for (int m = 0; m < EGO_MAX_MODS; ++m) { a.table[m] = d->table[m]; a.pos[m] = d->pos[m]; a.mod[m] = d->mod[m]; }
    a.base_vec = d->base_vec; a.slot = d->slot; a.local = d->local; a.tok = d->tok;
    a.x = d->x; a.emb = d->emb; a.rows = d->rows; a.D = d->D; a.reg = d->reg;
    EGO_LAUNCH(embed_kernel, dim3((unsigned)((d->rows + 3) / 4)), dim3(256), 0, stream, a);
    LAUNCH_CHECK();
    return EGO_OK;
}

namespace {
// register-token gradient: dreg[r][c] += sum_b dx[(b * rps + r) * D + c], b ascending (one thread per (r, 4 columns))
__global__ __launch_bounds__(256) void reg_grad_kernel(const float* __restrict__ dx, int B, long rps, int D, float* __restrict__ dreg) {
    const int r = blockIdx.y, c = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (c >= D) return;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int b = 0; b < B; ++b) acc += *(const f32x4*)(dx + ((long)b * rps + r) * D + c);
    *(f32x4*)(dreg + (long)r * D + c) += acc;
}
}  // namespace

extern "C" int ego_reg_grad(const float* dx, int B, long rows_per_sample, int n_reg, int D, float* dreg, hipStream_t stream) {
    if (n_reg <= 0) return EGO_OK;
    if (!dx || !dreg || B <= 0 || D <= 0 || D % 4 || n_reg > rows_per_sample || ((uintptr_t)dx | (uintptr_t)dreg) % 16) return EGO_ERR_ARG;
    EGO_LAUNCH(reg_grad_kernel, dim3((unsigned)((D / 4 + 255) / 256), (unsigned)n_reg), dim3(256), 0, stream, dx, B, rows_per_sample, D, dreg);
    LAUNCH_CHECK();
    return EGO_OK;
}

extern "C" int ego_loss_perm(const int* seg, const int* canon, const int* slot, const int* tok, int B, int M,
                             int n_mods, int* perm, int* tgt_perm, int* ranges, int* base, hipStream_t stream) {
    if (B <= 0 || M <= 0 || n_mods <= 0 || n_mods > EGO_MAX_MODS) return EGO_ERR_ARG;
    const size_t lds = ((size_t)3 * B * n_mods + EGO_MAX_MODS) * sizeof(int);
    if (lds > 48 * 1024) return EGO_ERR_ARG;
    const long total = (long)B * M;
    const int blocks = (int)((total + 1023) / 1024 < 256 ? (total + 1023) / 1024 : 256);
    EGO_LAUNCH(loss_perm_kernel, dim3(blocks < 1 ? 1 : blocks), dim3(256), lds, stream, seg, canon, slot, tok, B, M, n_mods, perm, tgt_perm,
               ranges, base);
    LAUNCH_CHECK();
    return EGO_OK;
}

namespace {
constexpr size_t ES_LDS_MAX = 160 * 1024;      // gfx950: 160 KiB of LDS per workgroup
// waves per workgroup of embed_sums_kernel: four (one partial row set each, joined in LDS) while the four sets fit, else one
// wave per workgroup - four times the partial rows for the ordered column sum, same result up to where the partials are cut
int es_waves(int D, int n_mods) { return (size_t)4 * (n_mods + 1) * D * sizeof(float) <= ES_LDS_MAX ? 4 : 1; }
}  // namespace

extern "C" long ego_embed_bwd_work_floats(long rows, int D, int n_mods) {
    const int rows_per_wg = 32 * es_waves(D, n_mods);
    return colsum_work_floats((rows + rows_per_wg - 1) / rows_per_wg, (n_mods + 1) * D);
}

extern "C" int ego_embed_bwd(const ego_embed_bwd_desc* d, hipStream_t stream) {
    if (!d || d->rows <= 0 || d->D % 4 || d->D > 2048 || d->n_mods <= 0 || d->n_mods > EGO_MAX_MODS || d->rows > 0x7fffffffL) return EGO_ERR_ARG;
    if (!d->work || d->work_floats < ego_embed_bwd_work_floats(d->rows, d->D, d->n_mods)) return EGO_ERR_ARG;
    EmbedBwdArgs a{};
    bool tables = false;
    for (int m = 0; m < EGO_MAX_MODS; ++m) { a.dtable[m] = d->dtable[m]; a.dmod[m] = d->dmod[m]; tables |= d->dtable[m] != nullptr; }
    a.dbase = d->dbase; a.dx = d->dx; a.d2 = d->d2; a.slot = d->slot; a.tok = d->tok;
    a.rows = d->rows; a.D = d->D; a.n_mods = d->n_mods;
    for (int m = 0; m < EGO_MAX_MODS; ++m) a.touched[m] = d->touched[m];
    a.part = d->work;
    const int W = (d->n_mods + 1) * d->D;
    const int waves = es_waves(d->D, d->n_mods);
    const size_t lds = (size_t)waves * W * sizeof(float);
    if (lds > ES_LDS_MAX) return EGO_ERR_ARG;               // (n_mods + 1) * D > 40960: beyond EGO_MAX_MODS x the largest D accepted above
    const long nwg = (d->rows + 32 * waves - 1) / (32 * waves);
#define ES_GO(C) do { if (lds > 64 * 1024 && hipFuncSetAttribute((const void*)embed_sums_kernel<C>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) \
                          return EGO_ERR_LAUNCH;                                                                                                \
                      EGO_LAUNCH(embed_sums_kernel<C>, dim3((unsigned)nwg), dim3(64 * waves), lds, stream, a); } while (0)
    if (d->D <= 768) ES_GO(3); else if (d->D <= 1024) ES_GO(4); else if (d->D <= 1536) ES_GO(6); else ES_GO(8);
#undef ES_GO
    LAUNCH_CHECK();
    ColsumDst dst{};
    for (int m = 0; m < d->n_mods; ++m) dst.p[m] = d->dmod[m];
    dst.p[d->n_mods] = d->dbase;
    dst.seg = d->D;
    colsum_launch(d->work, nwg, W, dst, stream);
    LAUNCH_CHECK();
    if (tables) {
        // table ids index 16-bit key fields: vocabularies up to 65536 (the reference's largest is 64000); a larger id would
        // alias another row's key: `vocab` (optional, 0 = not given) lets the caller have that checked here
        for (int m = 0; m < d->n_mods; ++m) if (d->dtable[m] && d->vocab[m] > 65536) return EGO_ERR_ARG;
        const size_t lds2 = (size_t)(5 * ET_CAP + 8) * sizeof(int) + (size_t)4 * d->D * sizeof(float);
#define ET_GO(C) do { if (hipFuncSetAttribute((const void*)embed_tables_kernel<C>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2) != hipSuccess) \
                          return EGO_ERR_LAUNCH;                                                                                        \
                      EGO_LAUNCH(embed_tables_kernel<C>, dim3(ET_WGS), dim3(256), lds2, stream, a); } while (0)
        if (d->D <= 768) ET_GO(3); else if (d->D <= 1024) ET_GO(4); else if (d->D <= 1536) ET_GO(6); else ET_GO(8);
#undef ET_GO
        LAUNCH_CHECK();
    }
    return EGO_OK;
}
