// bf16 MFMA GEMMs for the EgoM2P hot path (gfx950).
//
//   gemm_nt : C[M,N] = A[M,K] . B[N,K]^T        forward linears (B = nn.Linear weight [out,in]) and
//                                               dgrad (B = pre-transposed weight W^T [in,out])
//   gemm_tn : C[Ni,Nj] += P[M,Ni]^T . Q[M,Nj]   wgrad (P = dY, Q = layer input), contraction over rows
//
// Both: 128x128 output tile per 256-thread workgroup (4 waves, 2x2, 64x64 per wave as 4x4
// v_mfma_f32_16x16x32_bf16 tiles), 64-deep contraction steps, double-buffered LDS with register
// staging (global loads for step t+1 issued before the MFMAs of step t, written to the other
// LDS buffer after them: one barrier per step).  LDS images are XOR-swizzled so that the
// ds_read_b128 row reads (nt) and the ds_read_b64_tr_b16 transposed reads (tn) are bank-conflict
// free.  The accumulator is produced transposed (operands swapped in the MFMA) so each lane owns
// 4 consecutive output columns -> 8/16-byte stores.
//
// Replaces: torch.nn.functional.linear under autocast(bf16) at every call site of
// egom2p/models/egom2p_utils.py:141-169,180-203,215-242 and decoder_embeddings.py:372-383,489-500,
// and their autograd backward (mm dgrad / wgrad).
#include "common.h"
#include "egom2p_hip.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int STAGE_BYTES = 2 * BM * BK * 2;   // A tile + B tile (bf16) = 32 KiB
constexpr int GEMM_LDS = 2 * STAGE_BYTES;      // 64 KiB

struct NTArgs {
    const bf16_t* A; long lda;
    const bf16_t* B; long ldb;
    void* C; long ldc;
    const float* R; long ldr;
    const float* bias;
    const int* m_range;
    int M, N, K, epi;
};

// ---------------------------------------------------------------------------------------------
// NT kernel
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(NTArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    int M = p.M;
    long moff = 0;
    if (p.m_range) { moff = p.m_range[0]; M = min(M, p.m_range[1]); }
    const int tiles_n = (p.N + BN - 1) / BN;
    const int t = xcd_remap(blockIdx.x, gridDim.x);
    const int row0 = (t / tiles_n) * BM, col0 = (t % tiles_n) * BN;
    if (row0 >= M) return;

    const bf16_t* A = p.A + moff * p.lda;
    const bf16_t* B = p.B;

    // staging map: chunk q = tid + 256*i -> tile row q>>3, 16-byte chunk q&7
    u32x4 ra[4], rb[4];
    auto load_tile = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int q = tid + 256 * i, r = q >> 3, c = q & 7;
            const int ar = row0 + r, br = col0 + r;
            const u32x4 z = {0u, 0u, 0u, 0u};
            ra[i] = (ar < M) ? *(const u32x4*)(A + (long)ar * p.lda + k0 + c * 8) : z;
            rb[i] = (br < p.N) ? *(const u32x4*)(B + (long)br * p.ldb + k0 + c * 8) : z;
        }
    };
    auto store_tile = [&](int s) {
        char* sa = smem + s * STAGE_BYTES;
        char* sb = sa + BM * BK * 2;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int q = tid + 256 * i, r = q >> 3, c = q & 7;
            const int off = r * 128 + ((c ^ (r & 7)) << 4);
            *(u32x4*)(sa + off) = ra[i];
            *(u32x4*)(sb + off) = rb[i];
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nt = p.K / BK;
    load_tile(0);
    store_tile(0);
    __syncthreads();
    for (int kt = 0; kt < nt; ++kt) {
        const int s = kt & 1;
        if (kt + 1 < nt) load_tile((kt + 1) * BK);
        const char* sa = smem + s * STAGE_BYTES;
        const char* sb = sa + BM * BK * 2;
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
        if (kt + 1 < nt) store_tile(s ^ 1);
        __syncthreads();
    }

    // epilogue: lane owns row m, columns n..n+3
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = row0 + wm * 64 + i * 16 + (lane & 15);
        if (m >= M) continue;
        const long mrow = moff + m;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = col0 + wn * 64 + j * 16 + (lane >> 4) * 4;
            if (n >= p.N) continue;
            f32x4 v = acc[i][j];
            if (p.epi == EGO_EPI_BF16) {
                u32x2 o = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
                *(u32x2*)((bf16_t*)p.C + mrow * p.ldc + n) = o;
            } else if (p.epi == EGO_EPI_F32) {
                *(f32x4*)((float*)p.C + mrow * p.ldc + n) = v;
            } else if (p.epi == EGO_EPI_RESID) {
                const f32x4 r = *(const f32x4*)(p.R + mrow * p.ldr + n);
                f32x4 o = {r[0] + round_bf16(v[0]), r[1] + round_bf16(v[1]), r[2] + round_bf16(v[2]), r[3] + round_bf16(v[3])};
                *(f32x4*)((float*)p.C + mrow * p.ldc + n) = o;
            } else {  // EGO_EPI_BIAS_RESID
                const f32x4 r = *(const f32x4*)(p.R + mrow * p.ldr + n);
                const f32x4 b = *(const f32x4*)(p.bias + n);
                f32x4 o = {r[0] + round_bf16(v[0] + round_bf16(b[0])), r[1] + round_bf16(v[1] + round_bf16(b[1])),
                           r[2] + round_bf16(v[2] + round_bf16(b[2])), r[3] + round_bf16(v[3] + round_bf16(b[3]))};
                *(f32x4*)((float*)p.C + mrow * p.ldc + n) = o;
            }
        }
    }
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

__global__ __launch_bounds__(256, 2) void gemm_tn_kernel(TNArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
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

    // staging: chunk q = tid + 256*i -> tile row q>>4 (0..63), 16-byte chunk q&15
    u32x4 rp[4], rq[4];
    auto load_tile = [&](int m0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int q = tid + 256 * i, r = q >> 4, c = q & 15;
            const int m = m0 + r;
            const u32x4 z = {0u, 0u, 0u, 0u};
            rp[i] = (m < M) ? *(const u32x4*)(P + (long)m * p.ldp + c * 8) : z;
            rq[i] = (m < M) ? *(const u32x4*)(Q + (long)m * p.ldq + c * 8) : z;
        }
    };
    auto store_tile = [&](int s) {
        char* sp = smem + s * STAGE_BYTES;
        char* sq = sp + BM * BK * 2;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int q = tid + 256 * i, r = q >> 4, c = q & 15;
            const int off = r * 256 + ((((c >> 1) ^ tn_f(r)) << 5) | ((c & 1) << 4));
            *(u32x4*)(sp + off) = rp[i];
            *(u32x4*)(sq + off) = rq[i];
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    if (st0 < st1) {
        load_tile(st0 * BK);
        store_tile(0);
        __syncthreads();
        for (int st = st0; st < st1; ++st) {
            const int s = (st - st0) & 1;
            if (st + 1 < st1) load_tile((st + 1) * BK);
            const char* sp = smem + s * STAGE_BYTES;
            const char* sq = sp + BM * BK * 2;
            const int g = lane >> 4, tq = (lane & 15) >> 2, tp = lane & 3;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bf16x8 pf[4], qf[4];
                const int r0 = ks * 32 + 8 * g + tq;       // rows for elements 0..3; +4 for 4..7
                const int r1 = r0 + 4;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int ci = wi * 4 + i;              // 32-byte chunk (16 columns) of the P image
                    const int cj = wj * 4 + i;
                    s16x4 a0 = lds_read_tr16(sp + r0 * 256 + ((ci ^ tn_f(r0)) << 5) + tp * 8);
                    s16x4 a1 = lds_read_tr16(sp + r1 * 256 + ((ci ^ tn_f(r1)) << 5) + tp * 8);
                    s16x4 b0 = lds_read_tr16(sq + r0 * 256 + ((cj ^ tn_f(r0)) << 5) + tp * 8);
                    s16x4 b1 = lds_read_tr16(sq + r1 * 256 + ((cj ^ tn_f(r1)) << 5) + tp * 8);
                    typedef short s16x8 __attribute__((ext_vector_type(8)));
                    s16x8 a = __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7);
                    s16x8 b = __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7);
                    pf[i] = __builtin_bit_cast(bf16x8, a);
                    qf[i] = __builtin_bit_cast(bf16x8, b);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        // D[row = j (4 regs)][col = i (lane&15)]
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf[j], pf[i], acc[i][j], 0, 0, 0);
            }
            if (st + 1 < st1) store_tile(s ^ 1);
            __syncthreads();
        }
    }

#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int gi = i0 + wi * 64 + i * 16 + (lane & 15);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int gj = j0 + wj * 64 + j * 16 + (lane >> 4) * 4;
            const f32x4 v = acc[i][j];
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

bool g_attr_done = false;
void ensure_attrs() {
    if (g_attr_done) return;
    hipFuncSetAttribute((const void*)gemm_nt_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS);
    hipFuncSetAttribute((const void*)gemm_tn_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS);
    g_attr_done = true;
}

}  // namespace

extern "C" int ego_gemm_nt_bf16(const void* A, long lda, const void* B, long ldb, void* C, long ldc,
                                const float* R, long ldr, const float* bias, const int* m_range,
                                int M, int N, int K, int epi, hipStream_t stream) {
    if (M <= 0 || N <= 0) return EGO_OK;
    if (K <= 0 || K % BK || N % 8 || lda % 8 || ldb % 8 || ldc % 4 || epi < 0 || epi > EGO_EPI_BIAS_RESID) return EGO_ERR_ARG;
    if ((epi == EGO_EPI_RESID || epi == EGO_EPI_BIAS_RESID) && (!R || ldr % 4)) return EGO_ERR_ARG;
    if (epi == EGO_EPI_BIAS_RESID && !bias) return EGO_ERR_ARG;
    ensure_attrs();
    NTArgs a{(const bf16_t*)A, lda, (const bf16_t*)B, ldb, C, ldc, R, ldr, bias, m_range, M, N, K, epi};
    const int tiles = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
    EGO_LAUNCH(gemm_nt_kernel, dim3(tiles), dim3(256), GEMM_LDS, stream, a);
    LAUNCH_CHECK();
    return EGO_OK;
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
    const int tiles = (Ni / BM) * (Nj / BN);
    EGO_LAUNCH(gemm_tn_kernel, dim3(tiles * splits), dim3(256), GEMM_LDS, stream, a);
    LAUNCH_CHECK();
    if (splits > 1) {
        const long total = (long)Ni * (Nj / 4);
        const int blocks = (int)min((long)2048, (total + 255) / 256);
        EGO_LAUNCH(tn_reduce_kernel, dim3(blocks), dim3(256), 0, stream, slab, C0, C1, ldc, split_row, rows0, rows1, Ni, Nj, splits);
        LAUNCH_CHECK();
    }
    return EGO_OK;
}
