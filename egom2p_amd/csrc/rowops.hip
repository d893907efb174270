// HBM-bound row kernels of the EgoM2P hot path (gfx950): bias-free LayerNorm fwd/bwd, SwiGLU
// gate fwd/bwd, weight casts (fp32 master -> bf16 W and W^T), cross-entropy fwd/bwd over bf16
// logits, bias gradient.  One wave (64 lanes) per row, 16-byte accesses, fp32 math.
//
// Reference sites: LayerNorm egom2p/models/egom2p_utils.py:118-133 (F.layer_norm, eps 1e-6, zero bias
// buffer); GatedMlp :167-169; F.cross_entropy(reduction='mean') egom2p/models/egom2p_model.py:633-644.
#include "common.h"
#include "egom2p_hip.h"

namespace {

// term + prev with the product rounded FIRST (no fma contraction: hipcc contracts a * b + c by default, and the fused
// multi-layer backward must sum its layers exactly like the chained single-layer launches do through memory)
__device__ __forceinline__ float mul_then_add(float a, float b, float c) {
#pragma clang fp contract(off)
    const float t = a * b;
    return t + c;
}
__device__ __forceinline__ float mul_rounded(float a, float b) {
#pragma clang fp contract(off)
    return a * b;
}

constexpr int LN_MAXC_MAX = 8;   // float4 chunks per lane -> D <= 2048 (kernels are instantiated for 3, 4, 6, 8)

// ---------------------------------------------------------------------------------------------
// LayerNorm forward: y(bf16) = (x - mean) * rstd * w ; optional output row permutation
// ---------------------------------------------------------------------------------------------
template <int LN_MAXC, bool TAIL>   // TAIL: D % 4 != 0 - the last chunk holds pad columns, masked element-wise
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                     bf16_t* __restrict__ y, float* __restrict__ mean_out,
                                                     float* __restrict__ rstd_out, const int* __restrict__ out_row,
                                                     int rows, int D, int ld, float eps, unsigned char* __restrict__ q8, long ldq,
                                                     float* __restrict__ qscale) {
    // (no fma contraction anywhere in the LayerNorm kernels: the single-layer and the multi-layer kernels must agree bit for bit
    // whatever hipcc packs or fuses in one instantiation and not in another)
#pragma clang fp contract(off)
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // (uniform: row bases stay scalar)
    if (row >= rows) return;
    // D = the normalised width, ld >= D = the row pitch of x and y (a model dimension stored padded: the columns [D, ld) of
    // x are zero and do not take part; those of y are written as zeros)
    const int nc = (D + 3) >> 2, ncl = ld >> 2;
    const float* xr = x + (long)row * ld;
    f32x4 v[LN_MAXC];
    float s = 0.f;
    // every chunk's load is issued before the first use (a chunk past the row re-reads the row's last chunk and is not used): with
    // the load inside `if (c < nc)` hipcc kept each chunk's load and its sum in one exec-masked region and waited for every chunk in
    // turn - at D = 1152 (4.5 chunks per lane) the forward ran 2.7 TB/s against 5.3 at D = 768 (tools/ln_bench.py, round 5)
#pragma unroll
    for (int i = 0; i < LN_MAXC; ++i) v[i] = *(const f32x4*)(xr + min(lane + 64 * i, nc - 1) * 4);
#pragma unroll
    for (int i = 0; i < LN_MAXC; ++i) {
        const int c = lane + 64 * i;
        if (c < nc) s += v[i][0] + v[i][1] + v[i][2] + v[i][3];
    }
    const float mean = wave_sum(s) / D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXC; ++i) {
        const int c = lane + 64 * i;
        if (c < nc) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float d = v[i][e] - mean; q += (!TAIL || c * 4 + e < D) ? d * d : 0.f; }
        }
    }
    const float rstd = rsqrtf(wave_sum(q) / D + eps);
    if (lane == 0) { mean_out[row] = mean; rstd_out[row] = rstd; }
    const int orow = out_row ? out_row[row] : row;
    if (orow < 0) return;
    bf16_t* yr = y + (long)orow * ld;
    float amax = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXC; ++i) {
        const int c = lane + 64 * i;
        if (c < nc) {
            const f32x4 ww = *(const f32x4*)(w + c * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[i][e] = (!TAIL || c * 4 + e < D) ? round_bf16((v[i][e] - mean) * rstd * ww[e]) : 0.f;
                amax = fmaxf(amax, fabsf(v[i][e]));
            }
            u32x2 o = {pack_bf16x2(v[i][0], v[i][1]), pack_bf16x2(v[i][2], v[i][3])};
            *(u32x2*)(yr + c * 4) = o;
        } else if (c < ncl) {
            *(u32x2*)(yr + c * 4) = u32x2{0u, 0u};
        }
    }
    // optional e4m3 copy of the row for an fp8 GEMM (bit for bit what ego_quant_fp8_rows makes of the bf16 row)
    if (q8) {
        amax = wave_max(amax);
        const float sc = amax > 0.f ? amax * (1.f / 448.f) : 1.f, inv = 1.f / sc;
        if (lane == 0) qscale[orow] = sc;
        unsigned char* qr = q8 + (long)orow * ldq;
#pragma unroll
        for (int i = 0; i < LN_MAXC; ++i) {
            const int c = lane + 64 * i;
            if (c < nc) {
                int wd = __builtin_amdgcn_cvt_pk_fp8_f32(v[i][0] * inv, v[i][1] * inv, 0, false);
                wd = __builtin_amdgcn_cvt_pk_fp8_f32(v[i][2] * inv, v[i][3] * inv, wd, true);
                *(int*)(qr + c * 4) = wd;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// LayerNorm backward.  dx_out = (dx_in ? dx_in : 0) + rstd * (g - mean(g) - xhat * mean(g*xhat)),
// g = dy * w; dw += sum_rows dy * xhat.  dy rows may be permuted (dy_row map, -1 = zero gradient).
// Writes fp32 dx_out and an optional bf16 copy (the next GEMM's operand).
// ---------------------------------------------------------------------------------------------
constexpr int LNB_ROWS = 32;   // rows per workgroup (8 per wave)

// (second launch-bound argument = waves per SIMD the register allocation must allow: the kernel lives on the rows it has in flight)
template <int LN_MAXC, bool TAIL>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const bf16_t* __restrict__ dy, const int* __restrict__ dy_row,
                                                     const float* __restrict__ x, const float* __restrict__ mean_in,
                                                     const float* __restrict__ rstd_in, const float* __restrict__ w,
                                                     const float* dx_in, float* dx_out, bf16_t* dx_bf16,
                                                     float* __restrict__ dw_part, int rows, int D, int ld) {
    // (no fma contraction anywhere in the LayerNorm kernels: the single-layer and the multi-layer kernels must agree bit for bit
    // whatever hipcc packs or fuses in one instantiation and not in another)
#pragma clang fp contract(off)
    __shared__ float red[4][LN_MAXC * 64 * 4];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // (uniform: row offsets / statistics stay scalar)
    const int nc = (D + 3) >> 2, ncl = ld >> 2;      // D = normalised width, ld = row pitch (columns [D, ld): zero gradient)
    f32x4 ww[LN_MAXC], dwa[LN_MAXC];
#pragma unroll
    for (int i = 0; i < LN_MAXC; ++i) {
        const int c = lane + 64 * i;
        dwa[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        ww[i] = (c < nc) ? *(const f32x4*)(w + c * 4) : dwa[i];
    }
    // (one row at a time per wave, no unrolling across rows: with the row loop unrolled hipcc kept two rows' registers alive - 122 /
    //  198 / 248 VGPRs at 3 / 4 / 6 chunks per lane, i.e. 4 / 2 / 2 waves per SIMD - and the kernel, which lives on the number of
    //  rows in flight, fell from 5.0 TB/s at D = 768 to 3.3 at D = 1152 (tools/ln_bench.py, round 5).  g = dy w and xhat are
    //  formed twice - before and after the two row reductions, by the same expressions, so bit for bit the same values - instead
    //  of being held across them.)
#pragma unroll 1
    for (int rr = 0; rr < LNB_ROWS / 4; ++rr) {
        const int row = blockIdx.x * LNB_ROWS + rr * 4 + wave;
        if (row >= rows) break;
        const int grow = dy_row ? dy_row[row] : row;
        const float mean = mean_in[row], rstd = rstd_in[row];
        const float* xr = x + (long)row * ld;
        // all loads of the row first (branch-free: a chunk past the row re-reads the last chunk, a row without a gradient reads row 0
        // and is masked below): with the loads inside `if (c < nc)` every chunk waited for its own round trip (see ln_fwd_kernel)
        f32x4 xv[LN_MAXC], pin[LN_MAXC];
        u32x2 raws[LN_MAXC];
        const bf16_t* dyr = dy + (long)max(grow, 0) * ld;
#pragma unroll
        for (int i = 0; i < LN_MAXC; ++i) {
            const int cc = min(lane + 64 * i, nc - 1);
            xv[i] = *(const f32x4*)(xr + cc * 4);
            raws[i] = *(const u32x2*)(dyr + cc * 4);
        }
        if (dx_in) {          // the residual-stream gradient this row is added to: in flight under the two wave reductions
#pragma unroll
            for (int i = 0; i < LN_MAXC; ++i) pin[i] = *(const f32x4*)(dx_in + (long)row * ld + min(lane + 64 * i, nc - 1) * 4);
        }
        auto dy4 = [&](int i) {
            const u32x2 raw = raws[i];
            return grow >= 0 ? f32x4{bf16_to_f32(raw[0] & 0xffff), bf16_to_f32(raw[0] >> 16), bf16_to_f32(raw[1] & 0xffff), bf16_to_f32(raw[1] >> 16)}
                             : f32x4{0.f, 0.f, 0.f, 0.f};
        };
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < LN_MAXC; ++i) {
            const int c = lane + 64 * i;
            if (c < nc) {
                const f32x4 d = dy4(i);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float xh = (xv[i][e] - mean) * rstd;
                    const float ge = d[e] * ww[i][e];
                    dwa[i][e] += d[e] * xh;
                    s1 += ge;
                    s2 += ge * xh;
                }
            }
        }
        const float c1 = wave_sum(s1) / D, c2 = wave_sum(s2) / D;
        // (opaque to the optimiser: without it hipcc recognises the second pass's expressions and keeps every g / xhat of the first
        //  pass in registers after all - 102 VGPRs at 3 chunks instead of 70)
#pragma unroll
        for (int i = 0; i < LN_MAXC; ++i) asm volatile("" : "+v"(xv[i]), "+v"(raws[i]));
#pragma unroll
        for (int i = 0; i < LN_MAXC; ++i) {
            const int c = lane + 64 * i;
            if (c < nc) {
                const f32x4 d = dy4(i);
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float xh = (xv[i][e] - mean) * rstd;
                    const float ge = d[e] * ww[i][e];
                    o[e] = (!TAIL || c * 4 + e < D) ? mul_rounded(rstd, ge - c1 - xh * c2) : 0.f;
                }
                if (dx_in) {      // (term rounded, then added - no fma contraction: ln_bwd_multi_kernel sums its layers the same way, bit for bit)
                    const f32x4 p_ = pin[i];
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = mul_then_add(o[e], 1.0f, p_[e]);
                }
                *(f32x4*)(dx_out + (long)row * ld + c * 4) = o;
                if (dx_bf16) {
                    u32x2 b = {pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3])};
                    *(u32x2*)(dx_bf16 + (long)row * ld + c * 4) = b;
                }
            } else if (c < ncl) {
                *(f32x4*)(dx_out + (long)row * ld + c * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
                if (dx_bf16) *(u32x2*)(dx_bf16 + (long)row * ld + c * 4) = u32x2{0u, 0u};
            }
        }
    }
    // combine the 4 waves' dw partials, one atomic per column per workgroup
#pragma unroll
    for (int i = 0; i < LN_MAXC; ++i) {
        const int c = lane + 64 * i;
        if (c < nc) *(f32x4*)(&red[wave][c * 4]) = dwa[i];
    }
    __syncthreads();
    // one partial row per workgroup, plain stores (no float atomics: the sum over workgroups is taken in a fixed order by
    // colsum_kernel, so the weight gradient is bitwise reproducible - and the 768 same-address atomics per workgroup are gone)
    for (int col = threadIdx.x; col < D; col += 256)
        dw_part[(long)blockIdx.x * D + col] = red[0][col] + red[1][col] + red[2][col] + red[3][col];
}

// ---------------------------------------------------------------------------------------------
// One input, L LayerNorms (round 4).  The decoder's `context_norm` of every layer normalises the SAME tensor - the context
// (egom2p_utils.py:387-391: context_norm(context) per DecoderBlock, egom2p_model.py:520-521) - with its own weight: L = 12
// launches that each re-read 4 B / element forward, and 12 backward launches that each re-read x and read-modify-write the
// fp32 context gradient (14 B / element each).  Forward: x and its statistics once, L outputs.  Backward: x once, the L
// upstream gradients once each, ONE write of dx = sum_l rstd (g_l - mean(g_l) - xhat mean(g_l xhat)) - summed in the order
// l = L-1 .. 0 with the very expressions of ln_bwd_kernel, so dx is bit for bit what the L chained launches produce - and L
// weight-gradient partial rows per workgroup (ordered column sums as everywhere).
// ---------------------------------------------------------------------------------------------
constexpr int LN_MULTI_MAX = COLSUM_MAX_DST;
struct LnMultiFwd { const float* x; const float* w[LN_MULTI_MAX]; bf16_t* y[LN_MULTI_MAX]; float* mean; float* rstd; int L, rows, D, ld; float eps; };

template <int LN_MAXC, bool TAIL>
__global__ __launch_bounds__(256) void ln_fwd_multi_kernel(LnMultiFwd a) {
    // (no fma contraction anywhere in the LayerNorm kernels: the single-layer and the multi-layer kernels must agree bit for bit
    // whatever hipcc packs or fuses in one instantiation and not in another)
#pragma clang fp contract(off)
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (row >= a.rows) return;
    const int D = a.D, nc = (D + 3) >> 2, ncl = a.ld >> 2;
    const float* xr = a.x + (long)row * a.ld;
    f32x4 v[LN_MAXC];
    float s = 0.f;
    // every chunk's load is issued before the first use (a chunk past the row re-reads the row's last chunk and is not used): with
    // the load inside `if (c < nc)` hipcc kept each chunk's load and its sum in one exec-masked region and waited for every chunk in
    // turn - at D = 1152 (4.5 chunks per lane) the forward ran 2.7 TB/s against 5.3 at D = 768 (tools/ln_bench.py, round 5)
#pragma unroll
    for (int i = 0; i < LN_MAXC; ++i) v[i] = *(const f32x4*)(xr + min(lane + 64 * i, nc - 1) * 4);
#pragma unroll
    for (int i = 0; i < LN_MAXC; ++i) {
        const int c = lane + 64 * i;
        if (c < nc) s += v[i][0] + v[i][1] + v[i][2] + v[i][3];
    }
    const float mean = wave_sum(s) / D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXC; ++i) {
        const int c = lane + 64 * i;
        if (c < nc) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float d = v[i][e] - mean; q += (!TAIL || c * 4 + e < D) ? d * d : 0.f; }
        }
    }
    const float rstd = rsqrtf(wave_sum(q) / D + a.eps);
    if (lane == 0) { a.mean[row] = mean; a.rstd[row] = rstd; }
    for (int l = 0; l < a.L; ++l) {
        const float* w = a.w[l];
        bf16_t* yr = a.y[l] + (long)row * a.ld;
#pragma unroll
        for (int i = 0; i < LN_MAXC; ++i) {
            const int c = lane + 64 * i;
            if (c < nc) {
                const f32x4 ww = *(const f32x4*)(w + c * 4);
                float o[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (!TAIL || c * 4 + e < D) ? round_bf16((v[i][e] - mean) * rstd * ww[e]) : 0.f;
                *(u32x2*)(yr + c * 4) = u32x2{pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3])};
            } else if (c < ncl) {
                *(u32x2*)(yr + c * 4) = u32x2{0u, 0u};
            }
        }
    }
}

// rows per wave of the fused backward (their xhat and dx stay in registers across the layers): two, so that three waves fit a
// SIMD WITH the next layer's gradients and weights already in flight (a 4-row version at two waves per SIMD and no prefetch
// across layers ran at 1.7 TB/s: 8 waves per CU, every layer a fresh load -> reduce -> barrier chain)
constexpr int lnm_rw(int) { return 2; }
struct LnMultiBwd { const bf16_t* dy[LN_MULTI_MAX]; const float* w[LN_MULTI_MAX]; const float* x; const float* mean; const float* rstd;
                    const float* dx_in; float* dx_out; bf16_t* dx_bf16; float* dw_part; int L, rows, D, ld; };

template <int LN_MAXC, bool TAIL>
__global__ __launch_bounds__(256, (LN_MAXC <= 3 ? 3 : 2)) void ln_bwd_multi_kernel(LnMultiBwd a) {
    // (no fma contraction anywhere in the LayerNorm kernels: the single-layer and the multi-layer kernels must agree bit for bit
    // whatever hipcc packs or fuses in one instantiation and not in another)
#pragma clang fp contract(off)
    constexpr int LNM_RW = lnm_rw(LN_MAXC), LNM_ROWS = 4 * LNM_RW;
    __shared__ float red[4][LN_MAXC * 64 * 4];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // (uniform: row offsets / statistics stay scalar)
    const int D = a.D, ld = a.ld, nc = (D + 3) >> 2, ncl = ld >> 2;
    const int row0 = blockIdx.x * LNM_ROWS + wave * LNM_RW;
    // rows past the end are computed on the last row's data with rstd = 0 and d = 0 (no contribution, never stored): the
    // loops below are straight-line code
    f32x4 xh[LNM_RW][LN_MAXC], acc[LNM_RW][LN_MAXC];
    float rs[LNM_RW];
    long roff[LNM_RW];
#pragma unroll
    for (int r = 0; r < LNM_RW; ++r) {
        const int row = min(row0 + r, a.rows - 1);
        const bool live = row0 + r < a.rows;
        const float mean = a.mean[row];
        rs[r] = live ? a.rstd[row] : 0.f;
        roff[r] = (long)row * ld;
#pragma unroll
        for (int i = 0; i < LN_MAXC; ++i) {
            const int c = lane + 64 * i;
            acc[r][i] = f32x4{0.f, 0.f, 0.f, 0.f};
            xh[r][i] = acc[r][i];
            if (c < nc) {
                const f32x4 xv = *(const f32x4*)(a.x + roff[r] + c * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) xh[r][i][e] = (xv[e] - mean) * rs[r];
            }
        }
    }
    // gradients / weights of a layer are fetched while the previous one is computed
    u32x2 raws[LNM_RW][LN_MAXC], raws_n[LNM_RW][LN_MAXC];
    f32x4 ww[LN_MAXC], ww_n[LN_MAXC];
    auto fetch = [&](int l, u32x2 (&rw)[LNM_RW][LN_MAXC], f32x4 (&wv)[LN_MAXC]) {
        const float* w = a.w[l];
        const bf16_t* dy = a.dy[l];
#pragma unroll
        for (int i = 0; i < LN_MAXC; ++i) {
            const int c = lane + 64 * i;
            wv[i] = (c < nc) ? *(const f32x4*)(w + c * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int r = 0; r < LNM_RW; ++r) rw[r][i] = (c < nc) ? *(const u32x2*)(dy + roff[r] + c * 4) : u32x2{0u, 0u};
        }
    };
    fetch(a.L - 1, raws, ww);
    for (int l = a.L - 1; l >= 0; --l) {           // the order of the decoder's backward: last layer first
        if (l > 0) fetch(l - 1, raws_n, ww_n);
        f32x4 dwa[LN_MAXC];
#pragma unroll
        for (int i = 0; i < LN_MAXC; ++i) dwa[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < LNM_RW; ++r) {
            const float live = (row0 + r < a.rows) ? 1.f : 0.f;
            f32x4 g[LN_MAXC];
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int i = 0; i < LN_MAXC; ++i) {
                const int c = lane + 64 * i;
                g[i] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (c < nc) {
                    const u32x2 raw = raws[r][i];
                    const f32x4 d = f32x4{bf16_to_f32(raw[0] & 0xffff), bf16_to_f32(raw[0] >> 16), bf16_to_f32(raw[1] & 0xffff), bf16_to_f32(raw[1] >> 16)} * live;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        g[i][e] = d[e] * ww[i][e];
                        dwa[i][e] += d[e] * xh[r][i][e];
                        s1 += g[i][e];
                        s2 += g[i][e] * xh[r][i][e];
                    }
                }
            }
            const float c1 = wave_sum(s1) / D, c2 = wave_sum(s2) / D;
#pragma unroll
            for (int i = 0; i < LN_MAXC; ++i) {
                const int c = lane + 64 * i;
                if (c < nc) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        // (dx = term_{L-1}, then dx = term_l + dx, like the chained launches: 0 + term is the term)
                        const float inner = g[i][e] - c1 - xh[r][i][e] * c2;
                        acc[r][i][e] = (!TAIL || c * 4 + e < D) ? mul_then_add(rs[r], inner, acc[r][i][e]) : acc[r][i][e];   // term rounded first
                    }
                }
            }
        }
#pragma unroll
        for (int i = 0; i < LN_MAXC; ++i) {
            ww[i] = ww_n[i];
#pragma unroll
            for (int r = 0; r < LNM_RW; ++r) raws[r][i] = raws_n[r][i];
        }
        // this layer's weight-gradient partial row of the workgroup
        __syncthreads();
#pragma unroll
        for (int i = 0; i < LN_MAXC; ++i) {
            const int c = lane + 64 * i;
            if (c < nc) *(f32x4*)(&red[wave][c * 4]) = dwa[i];
        }
        __syncthreads();
        for (int col = threadIdx.x; col < D; col += 256)
            a.dw_part[((long)blockIdx.x * a.L + l) * D + col] = red[0][col] + red[1][col] + red[2][col] + red[3][col];
    }
#pragma unroll
    for (int r = 0; r < LNM_RW; ++r) {
        if (row0 + r >= a.rows) break;
#pragma unroll
        for (int i = 0; i < LN_MAXC; ++i) {
            const int c = lane + 64 * i;
            if (c < nc) {
                f32x4 o = acc[r][i];
                if (a.dx_in) {
                    const f32x4 p_ = *(const f32x4*)(a.dx_in + roff[r] + c * 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = mul_then_add(o[e], 1.0f, p_[e]);
                }
                *(f32x4*)(a.dx_out + roff[r] + c * 4) = o;
                if (a.dx_bf16) *(u32x2*)(a.dx_bf16 + roff[r] + c * 4) = u32x2{pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3])};
            } else if (c < ncl) {
                *(f32x4*)(a.dx_out + roff[r] + c * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
                if (a.dx_bf16) *(u32x2*)(a.dx_bf16 + roff[r] + c * 4) = u32x2{0u, 0u};
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Ordered column sums of per-workgroup partial rows: out[c] (+)= sum_r parts[r][c], rows summed in index order (fixed
// order -> bitwise reproducible).  Two launches: chunks of COLSUM_CHUNK rows -> one row per chunk, then those -> out.
// `dst` routes column c of the LAST stage to dst.p[c / seg] + c % seg (several gradient vectors behind one slab).
// ---------------------------------------------------------------------------------------------
constexpr int COLSUM_CHUNK = 64;

__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ parts, long n, int W, float* __restrict__ mid,
                                                     ColsumDst dst, int last) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= W) return;
    const long r0 = (long)blockIdx.y * COLSUM_CHUNK, r1 = min(n, r0 + COLSUM_CHUNK);
    float s = 0.f;
    long r = r0;
    for (; r + 8 <= r1; r += 8) {
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = parts[(r + k) * W + c];
#pragma unroll
        for (int k = 0; k < 8; ++k) s += v[k];
    }
    for (; r < r1; ++r) s += parts[r * W + c];
    if (!last) { mid[(long)blockIdx.y * W + c] = s; return; }
    float* o = dst.p[c / dst.seg];
    if (o) o[c % dst.seg] += s;
}

// ---------------------------------------------------------------------------------------------
// SwiGLU gate.  ab[rows, 2F]: a = cols [0,F), b = cols [F,2F).  h = bf16(bf16(silu(a)) * b)
// ---------------------------------------------------------------------------------------------

__global__ void swiglu_fwd_kernel(const bf16_t* __restrict__ ab, bf16_t* __restrict__ hout, long rows, int F) {
    const int fc = F >> 3;                      // 8-element chunks per row
    const long total = rows * fc;
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const long r = idx / fc; const int c = (int)(idx % fc) * 8;
        const u32x4 a = *(const u32x4*)(ab + r * 2 * F + c);
        const u32x4 b = *(const u32x4*)(ab + r * 2 * F + F + c);
        u32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float a0 = bf16_to_f32(a[e] & 0xffff), a1 = bf16_to_f32(a[e] >> 16);
            const float b0 = bf16_to_f32(b[e] & 0xffff), b1 = bf16_to_f32(b[e] >> 16);
            o[e] = pack_bf16x2(round_bf16(a0 * sigmoidf_(a0)) * b0, round_bf16(a1 * sigmoidf_(a1)) * b1);
        }
        *(u32x4*)(hout + r * F + c) = o;
    }
}

// dab[rows,2F]: da = dh * b * s * (1 + a (1 - s)),  db = dh * silu(a)
__global__ void swiglu_bwd_kernel(const bf16_t* __restrict__ ab, const bf16_t* __restrict__ dh,
                                  bf16_t* __restrict__ dab, long rows, int F) {
    const int fc = F >> 3;
    const long total = rows * fc;
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const long r = idx / fc; const int c = (int)(idx % fc) * 8;
        const u32x4 a = *(const u32x4*)(ab + r * 2 * F + c);
        const u32x4 b = *(const u32x4*)(ab + r * 2 * F + F + c);
        const u32x4 g = *(const u32x4*)(dh + r * F + c);
        u32x4 oa, ob;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float av[2] = {bf16_to_f32(a[e] & 0xffff), bf16_to_f32(a[e] >> 16)};
            float bv[2] = {bf16_to_f32(b[e] & 0xffff), bf16_to_f32(b[e] >> 16)};
            float gv[2] = {bf16_to_f32(g[e] & 0xffff), bf16_to_f32(g[e] >> 16)};
            float da[2], db[2];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const float s = sigmoidf_(av[k]);
                da[k] = gv[k] * bv[k] * s * (1.f + av[k] * (1.f - s));
                db[k] = gv[k] * av[k] * s;
            }
            oa[e] = pack_bf16x2(da[0], da[1]);
            ob[e] = pack_bf16x2(db[0], db[1]);
        }
        *(u32x4*)(dab + r * 2 * F + c) = oa;
        *(u32x4*)(dab + r * 2 * F + F + c) = ob;
    }
}

// ---------------------------------------------------------------------------------------------
// weight cast: fp32 W[rows_src, cols] -> bf16 Wb[rows_dst(pad), ld_w] (rows >= rows_src zero) and
// bf16 Wt[cols, ld_t] = W^T (columns >= rows_src zero up to rows_dst).  32x32 LDS transpose tiles.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cast_w_kernel(const float* __restrict__ W, int rows_src, int cols, long ld_src,
                                                     bf16_t* __restrict__ Wb, long ld_w, bf16_t* __restrict__ Wt,
                                                     long ld_t, int rows_dst) {
    __shared__ float tile[32][33];
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;       // 32 x 8
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = r0 + ty + 8 * i, c = c0 + tx;
        float v = 0.f;
        if (r < rows_src && c < cols) v = W[(long)r * ld_src + c];
        tile[ty + 8 * i][tx] = v;
        if (Wb && r < rows_dst && c < cols) Wb[(long)r * ld_w + c] = f32_to_bf16(v);
    }
    __syncthreads();
    if (Wt) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = c0 + ty + 8 * i, r = r0 + tx;           // Wt[c][r]
            if (c < cols && r < rows_dst) Wt[(long)c * ld_t + r] = f32_to_bf16(tile[tx][ty + 8 * i]);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// cross-entropy over bf16 logits rows [row_off, row_off + n) of a [*, V] buffer (ld = V).
// fwd: nll[row] = lse - logit[target]; lse[row] kept.  bwd (in place): logits <- (softmax - onehot) * coef,
// coef = *gscale / (n_mods * n) (loss_type 'mod'), or *gscale * lw[0] / n with the modality's loss weight lw (ego_loss_weights:
// 'weighted_mod' / 'token').  One workgroup per row.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ce_fwd_kernel(const bf16_t* __restrict__ logits, long ld, int V,
                                                     const int* __restrict__ targets, const int* __restrict__ range,
                                                     float* __restrict__ lse_out, float* __restrict__ nll_out) {
    __shared__ float red[8];
    const int off = range[0], n = range[1];
    const int r = blockIdx.x;
    if (r >= n) return;
    const long row = (long)off + r;
    const bf16_t* lr = logits + row * ld;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // one pass over the row (128 KB at V = 64000): running maximum and rescaled sum per thread, 4 loads in flight
    float mx = -3.0e38f, s = 0.f;
    const int vc = V >> 3;
    auto fold = [&](const u32x4& a) {
        float x[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) { x[2 * e] = bf16_to_f32(a[e] & 0xffff); x[2 * e + 1] = bf16_to_f32(a[e] >> 16); }
        float cm = x[0];
#pragma unroll
        for (int e = 1; e < 8; ++e) cm = fmaxf(cm, x[e]);
        if (cm > mx) { s *= __expf(mx - cm); mx = cm; }
#pragma unroll
        for (int e = 0; e < 8; ++e) s += __expf(x[e] - mx);
    };
    int c = threadIdx.x;
    for (; c + 768 < vc; c += 1024) {
        const u32x4 a0 = *(const u32x4*)(lr + c * 8), a1 = *(const u32x4*)(lr + (c + 256) * 8);
        const u32x4 a2 = *(const u32x4*)(lr + (c + 512) * 8), a3 = *(const u32x4*)(lr + (c + 768) * 8);
        fold(a0); fold(a1); fold(a2); fold(a3);
    }
    for (; c < vc; c += 256) fold(*(const u32x4*)(lr + c * 8));
    // combine (max, sum) pairs: wave, then the four waves
    const float wm = wave_max(mx);
    s = wave_sum(s * __expf(mx - wm));
    if (lane == 0) { red[wave] = wm; red[4 + wave] = s; }
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    if (threadIdx.x == 0) {
        const float tot = red[4] * __expf(red[0] - mx) + red[5] * __expf(red[1] - mx) + red[6] * __expf(red[2] - mx) + red[7] * __expf(red[3] - mx);
        const float lse = mx + __logf(tot);
        lse_out[row] = lse;
        nll_out[row] = lse - bf16_to_f32(lr[targets[row]]);
    }
}

__global__ __launch_bounds__(256) void ce_bwd_kernel(bf16_t* __restrict__ logits, long ld, int V,
                                                     const int* __restrict__ targets, const int* __restrict__ range,
                                                     const float* __restrict__ lse_in, const float* __restrict__ gscale,
                                                     float inv_mods, const float* __restrict__ lw) {
    const int off = range[0], n = range[1];
    const int r = blockIdx.x;
    if (r >= n) return;
    const long row = (long)off + r;
    bf16_t* lr = logits + row * ld;
    const float lse = lse_in[row];
    const float coef = gscale[0] * (lw ? lw[0] : inv_mods) / (float)n;
    const int tgt = targets[row];
    const int vc = V >> 3;
    for (int c = threadIdx.x; c < vc; c += 256) {
        const u32x4 a = *(const u32x4*)(lr + c * 8);
        u32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int i0 = c * 8 + 2 * e;
            float p0 = __expf(bf16_to_f32(a[e] & 0xffff) - lse), p1 = __expf(bf16_to_f32(a[e] >> 16) - lse);
            if (i0 == tgt) p0 -= 1.f;
            if (i0 + 1 == tgt) p1 -= 1.f;
            o[e] = pack_bf16x2(p0 * coef, p1 * coef);
        }
        *(u32x4*)(lr + c * 8) = o;
    }
}

// Forward and backward of the cross-entropy in ONE pass over the logits (training: the upstream gradient is known when
// the loss is formed).  The row lives in registers between the two phases (NCH 16-byte chunks per thread: 128 KB at
// V = 64000 is 32 chunks for each of 256 threads), so the logits are read once instead of twice.  Same fold order, same
// expressions as ce_fwd_kernel / ce_bwd_kernel: lse, nll and d logits are bitwise those of the two-call form.
template <int NCH>
__global__ __launch_bounds__(256, 2) void ce_fwd_bwd_kernel(bf16_t* __restrict__ logits, long ld, int V,
                                                         const int* __restrict__ targets, const int* __restrict__ range,
                                                         float* __restrict__ lse_out, float* __restrict__ nll_out,
                                                         const float* __restrict__ gscale, float inv_mods,
                                                         const float* __restrict__ lw) {
    __shared__ float red[9];
    const int off = range[0], n = range[1];
    const int r = blockIdx.x;
    if (r >= n) return;
    const long row = (long)off + r;
    bf16_t* lr = logits + row * ld;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int vc = V >> 3;
    const int tgt = targets[row];
    u32x4 a[NCH];
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        const int c = threadIdx.x + 256 * j;
        if (c < vc) a[j] = *(const u32x4*)(lr + c * 8);
    }
    float mx = -3.0e38f, s = 0.f;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        if (threadIdx.x + 256 * j < vc) {
            float x[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) { x[2 * e] = bf16_to_f32(a[j][e] & 0xffff); x[2 * e + 1] = bf16_to_f32(a[j][e] >> 16); }
            float cm = x[0];
#pragma unroll
            for (int e = 1; e < 8; ++e) cm = fmaxf(cm, x[e]);
            if (cm > mx) { s *= __expf(mx - cm); mx = cm; }
#pragma unroll
            for (int e = 0; e < 8; ++e) s += __expf(x[e] - mx);
        }
    }
    const float wm = wave_max(mx);
    s = wave_sum(s * __expf(mx - wm));
    if (lane == 0) { red[wave] = wm; red[4 + wave] = s; }
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    if (threadIdx.x == 0) {
        const float tot = red[4] * __expf(red[0] - mx) + red[5] * __expf(red[1] - mx) + red[6] * __expf(red[2] - mx) + red[7] * __expf(red[3] - mx);
        const float l = mx + __logf(tot);
        red[8] = l;
        lse_out[row] = l;
        nll_out[row] = l - bf16_to_f32(lr[tgt]);
    }
    __syncthreads();                         // (also orders thread 0's read of the target logit before the stores below)
    const float lse = red[8];
    const float coef = gscale[0] * (lw ? lw[0] : inv_mods) / (float)n;
    // (the row stays PACKED across the barrier: without this hipcc keeps the first phase's 8 floats per chunk alive too)
#pragma unroll
    for (int j = 0; j < NCH; ++j) asm volatile("" : "+v"(a[j]));
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        const int c = threadIdx.x + 256 * j;
        if (c < vc) {
            u32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int i0 = c * 8 + 2 * e;
                float p0 = __expf(bf16_to_f32(a[j][e] & 0xffff) - lse), p1 = __expf(bf16_to_f32(a[j][e] >> 16) - lse);
                if (i0 == tgt) p0 -= 1.f;
                if (i0 + 1 == tgt) p1 -= 1.f;
                o[e] = pack_bf16x2(p0 * coef, p1 * coef);
            }
            *(u32x4*)(lr + c * 8) = o;
        }
    }
}

// mod_loss[m] = sum(nll[off..off+n)) / n (0 if n == 0); loss = sum_m mod_loss / n_mods.  One block.
// One workgroup of 1024 threads, 4 independent 16-byte loads in flight per thread: the 65,536 per-row losses of a
// micro-batch (256 KB) are four round trips, and the sum order is fixed (bitwise reproducible).  `err` (optional): the
// compaction's "decoder mask is not an interval" flag - a set flag poisons the loss (NaN: the train loop's non-finite
// check then stops the run like the reference's, run_training_egom2p.py:731-734) and is cleared for the next step.
// lw / ms (optional, ego_loss_weights): loss = sum_m lw[m] * mean_nll[m] and the reported mod_loss[m] = ms[m] * mean_nll[m]
// (forward_weighted_mod_loss / forward_token_loss, egom2p_model.py:583-612, 646-681).
__global__ __launch_bounds__(1024) void loss_finalize_kernel(const float* __restrict__ nll, const int* __restrict__ ranges,
                                                             int n_mods, float* __restrict__ out, int* __restrict__ err,
                                                             const float* __restrict__ lw, const float* __restrict__ ms) {
    __shared__ float red[16];
    const int tid = threadIdx.x;
    float total = 0.f;
    for (int m = 0; m < n_mods; ++m) {
        const int off = ranges[2 * m], n = ranges[2 * m + 1];
        const float* p = nll + off;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        const int head = min(n, (int)((4 - (off & 3)) & 3));       // up to the first 16-byte boundary
        if (tid < head) s0 += p[tid];
        const int n4 = (n - head) >> 2;
        const f32x4* p4 = (const f32x4*)(p + head);
        int i = tid;
        for (; i + 3 * 1024 < n4; i += 4 * 1024) {
            const f32x4 a = p4[i], b = p4[i + 1024], c = p4[i + 2048], d = p4[i + 3072];
            s0 += (a[0] + a[1]) + (a[2] + a[3]); s1 += (b[0] + b[1]) + (b[2] + b[3]);
            s2 += (c[0] + c[1]) + (c[2] + c[3]); s3 += (d[0] + d[1]) + (d[2] + d[3]);
        }
        for (; i < n4; i += 1024) { const f32x4 a = p4[i]; s0 += (a[0] + a[1]) + (a[2] + a[3]); }
        const int tail0 = head + 4 * n4;
        if (tail0 + tid < n) s1 += p[tail0 + tid];
        float s = wave_sum((s0 + s1) + (s2 + s3));
        __syncthreads();
        if ((tid & 63) == 0) red[tid >> 6] = s;
        __syncthreads();
        float r = 0.f;
#pragma unroll
        for (int w = 0; w < 16; ++w) r += red[w];
        const float ml = n > 0 ? r / (float)n : 0.f;
        if (tid == 0) out[1 + m] = ms ? ms[m] * ml : ml;
        total += lw ? lw[m] * ml : ml;
    }
    if (tid == 0) {
        float v = lw ? total : total / (float)n_mods;
        if (err && *err) {
            v = __int_as_float(0x7fc00000);
            for (int m = 0; m < n_mods; ++m) out[1 + m] = v;
            *err = 0;
        }
        out[0] = v;
    }
}

// db[col] += sum_rows g[row][col]   (bf16 g, fp32 accumulate).  A workgroup owns 256 rows; a thread sums 8 adjacent columns
// (one 16-byte load per row) of every (256 / (D / 8))-th row, the row groups are combined in LDS, one partial row per
// workgroup (no atomics: bitwise reproducible).
__global__ __launch_bounds__(256) void bias_grad_kernel(const bf16_t* __restrict__ g, long rows, int D, float* __restrict__ db_part) {
    extern __shared__ __attribute__((aligned(16))) float bg_sm[];      // [rgroups][D]
    const int nvec = D >> 3, rgroups = 256 / nvec;
    const int cv = threadIdx.x % nvec, rg = threadIdx.x / nvec;
    const long r0 = (long)blockIdx.x * 256, r1 = min(rows, r0 + 256);
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    if (rg < rgroups) {
        for (long r = r0 + rg; r < r1; r += rgroups) {
            const u32x4 v = *(const u32x4*)(g + r * D + cv * 8);
#pragma unroll
            for (int e = 0; e < 4; ++e) { acc[2 * e] += bf16_to_f32(v[e] & 0xffff); acc[2 * e + 1] += bf16_to_f32(v[e] >> 16); }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) bg_sm[rg * D + cv * 8 + e] = acc[e];
    }
    __syncthreads();
    for (int c = threadIdx.x; c < D; c += 256) {
        float s = 0.f;
        for (int q = 0; q < rgroups; ++q) s += bg_sm[q * D + c];
        db_part[(long)blockIdx.x * D + c] = s;              // summed over workgroups in order by colsum_kernel
    }
}

// dst_bf16 = src_f32 (flat)
__global__ void cast_f32_bf16_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, long n4) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const f32x4 v = *(const f32x4*)(src + i * 4);
        u32x2 o = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
        *(u32x2*)(dst + i * 4) = o;
    }
}

// Row-wise fp8 quantisation (OCP e4m3, max 448): scale[r] = amax(row) / 448 (1 for an all-zero row), q = e4m3(x / scale).
// One wave per row, 16 bytes of bf16 per lane and pass; HBM-bound: 2 B read + 1 B written per element.
__global__ __launch_bounds__(256) void quant_fp8_rows_kernel(const bf16_t* __restrict__ X, long ld, long rows, int K,
                                                              unsigned char* __restrict__ Q, long ldq, float* __restrict__ scale) {
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    const bf16_t* x = X + row * ld;
    float amax = 0.f;
    for (int c = lane * 8; c < K; c += 512) {
        const u32x4 v = *(const u32x4*)(x + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) amax = fmaxf(amax, fmaxf(fabsf(bf16_to_f32(v[e] & 0xffff)), fabsf(bf16_to_f32(v[e] >> 16))));
    }
    amax = wave_max(amax);
    const float sc = amax > 0.f ? amax * (1.f / 448.f) : 1.f, inv = 1.f / sc;
    if (lane == 0) scale[row] = sc;
    unsigned char* q = Q + row * ldq;
    for (int c = lane * 8; c < K; c += 512) {
        const u32x4 v = *(const u32x4*)(x + c);
        u32x2 o;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const float f0 = bf16_to_f32(v[2 * e] & 0xffff) * inv, f1 = bf16_to_f32(v[2 * e] >> 16) * inv;
            const float f2 = bf16_to_f32(v[2 * e + 1] & 0xffff) * inv, f3 = bf16_to_f32(v[2 * e + 1] >> 16) * inv;
            int w = __builtin_amdgcn_cvt_pk_fp8_f32(f0, f1, 0, false);
            w = __builtin_amdgcn_cvt_pk_fp8_f32(f2, f3, w, true);
            o[e] = (unsigned)w;
        }
        *(u32x2*)(q + c) = o;
    }
}

inline int grid_for(long total, int cap = 4096) { return (int)((total + 255) / 256 < cap ? (total + 255) / 256 : cap); }

}  // namespace

// parts [n][W] -> dst (accumulating).  work = parts followed by room for the intermediate rows.
__attribute__((visibility("hidden"))) int colsum_launch(float* parts, long n, int W, const ColsumDst& dst, hipStream_t stream) {
    ColsumDst none{};
    while (n > COLSUM_CHUNK) {
        const long nc = (n + COLSUM_CHUNK - 1) / COLSUM_CHUNK;
        float* mid = parts + n * W;
        EGO_LAUNCH(colsum_kernel, dim3((W + 255) / 256, (unsigned)nc), dim3(256), 0, stream, parts, n, W, mid, none, 0);
        parts = mid; n = nc;
    }
    EGO_LAUNCH(colsum_kernel, dim3((W + 255) / 256, 1), dim3(256), 0, stream, parts, n, W, (float*)nullptr, dst, 1);
    return 0;
}
__attribute__((visibility("hidden"))) long colsum_work_floats(long n, int W) {     // parts + every intermediate level
    long tot = 0;
    while (true) { tot += n * W; if (n <= COLSUM_CHUNK) break; n = (n + COLSUM_CHUNK - 1) / COLSUM_CHUNK; }
    return tot;
}


extern "C" int ego_layernorm_fwd(const float* x, const float* w, void* y, float* mean, float* rstd,
                                 const int* out_row, int rows, int D, long ld, float eps, void* q8, long ldq, float* qscale,
                                 hipStream_t stream) {
    if (rows <= 0) return EGO_OK;
    if (D <= 0 || ld % 4 || ld < D || ld > LN_MAXC_MAX * 256 || (q8 && (!qscale || ldq % 4 || ld != D))) return EGO_ERR_ARG;
#define LN_FWD_(C, T) EGO_LAUNCH((ln_fwd_kernel<C, T>), dim3((rows + 3) / 4), dim3(256), 0, stream, x, w, (bf16_t*)y, mean, rstd, out_row, rows, D, (int)ld, eps, \
                             (unsigned char*)q8, ldq, qscale)
#define LN_FWD(C) do { if (D % 4) LN_FWD_(C, true); else LN_FWD_(C, false); } while (0)
    if (ld <= 768) LN_FWD(3); else if (ld <= 1024) LN_FWD(4); else if (ld <= 1536) LN_FWD(6); else LN_FWD(8);
#undef LN_FWD_
#undef LN_FWD
    LAUNCH_CHECK();
    return EGO_OK;
}

extern "C" long ego_layernorm_bwd_work_floats(int rows, int D) { return colsum_work_floats((rows + LNB_ROWS - 1) / LNB_ROWS, D); }

extern "C" int ego_layernorm_bwd(const void* dy, const int* dy_row, const float* x, const float* mean,
                                 const float* rstd, const float* w, const float* dx_in, float* dx_out,
                                 void* dx_bf16, float* dw, float* work, long work_floats, int rows, int D, long ld,
                                 hipStream_t stream) {
    if (rows <= 0) return EGO_OK;
    if (D <= 0 || ld % 4 || ld < D || ld > LN_MAXC_MAX * 256 || !work || work_floats < ego_layernorm_bwd_work_floats(rows, D))
        return EGO_ERR_ARG;
    const int nwg = (rows + LNB_ROWS - 1) / LNB_ROWS;
#define LN_BWD_(C, T) EGO_LAUNCH((ln_bwd_kernel<C, T>), dim3(nwg), dim3(256), 0, stream, (const bf16_t*)dy, \
                       dy_row, x, mean, rstd, w, dx_in, dx_out, (bf16_t*)dx_bf16, work, rows, D, (int)ld)
#define LN_BWD(C) do { if (D % 4) LN_BWD_(C, true); else LN_BWD_(C, false); } while (0)
    if (ld <= 768) LN_BWD(3); else if (ld <= 1024) LN_BWD(4); else if (ld <= 1536) LN_BWD(6); else LN_BWD(8);
#undef LN_BWD_
#undef LN_BWD
    LAUNCH_CHECK();
    ColsumDst dst{};
    dst.p[0] = dw; dst.seg = D;
    colsum_launch(work, nwg, D, dst, stream);
    LAUNCH_CHECK();
    return EGO_OK;
}

extern "C" int ego_layernorm_fwd_multi(const float* x, int n_layers, const float* const* w, void* const* y, float* mean, float* rstd,
                                       int rows, int D, long ld, float eps, hipStream_t stream) {
    if (rows <= 0 || n_layers <= 0) return EGO_OK;
    if (n_layers > LN_MULTI_MAX || !w || !y || D <= 0 || ld % 4 || ld < D || ld > LN_MAXC_MAX * 256) return EGO_ERR_ARG;
    LnMultiFwd a{};
    a.x = x; a.mean = mean; a.rstd = rstd; a.L = n_layers; a.rows = rows; a.D = D; a.ld = (int)ld; a.eps = eps;
    for (int l = 0; l < n_layers; ++l) { if (!w[l] || !y[l]) return EGO_ERR_ARG; a.w[l] = w[l]; a.y[l] = (bf16_t*)y[l]; }
#define LNM_FWD_(C, T) EGO_LAUNCH((ln_fwd_multi_kernel<C, T>), dim3((rows + 3) / 4), dim3(256), 0, stream, a)
#define LNM_FWD(C) do { if (D % 4) LNM_FWD_(C, true); else LNM_FWD_(C, false); } while (0)
    if (ld <= 768) LNM_FWD(3); else if (ld <= 1024) LNM_FWD(4); else if (ld <= 1536) LNM_FWD(6); else LNM_FWD(8);
#undef LNM_FWD_
#undef LNM_FWD
    LAUNCH_CHECK();
    return EGO_OK;
}

namespace {
int lnm_rows(long) { return 4 * lnm_rw(0); }
}  // namespace

extern "C" long ego_layernorm_bwd_multi_work_floats(int rows, int D, int n_layers) {
    // (the partial rows are laid out by the row pitch's kernel instantiation; D <= ld, so size for the smaller workgroups of D itself)
    const int per = lnm_rows(D);
    return colsum_work_floats((rows + per - 1) / per, n_layers * D);
}

extern "C" int ego_layernorm_bwd_multi(int n_layers, const void* const* dy, const float* const* w, float* const* dw, const float* x,
                                       const float* mean, const float* rstd, const float* dx_in, float* dx_out, void* dx_bf16,
                                       float* work, long work_floats, int rows, int D, long ld, hipStream_t stream) {
    if (rows <= 0 || n_layers <= 0) return EGO_OK;
    if (n_layers > LN_MULTI_MAX || !dy || !w || !dw || D <= 0 || ld % 4 || ld < D || ld > 6 * 256 || !work ||
        work_floats < ego_layernorm_bwd_multi_work_floats(rows, D, n_layers)) return EGO_ERR_ARG;      // (ld <= 1536: four rows of xhat + dx per lane)
    LnMultiBwd a{};
    a.x = x; a.mean = mean; a.rstd = rstd; a.dx_in = dx_in; a.dx_out = dx_out; a.dx_bf16 = (bf16_t*)dx_bf16; a.dw_part = work;
    a.L = n_layers; a.rows = rows; a.D = D; a.ld = (int)ld;
    ColsumDst dst{};
    for (int l = 0; l < n_layers; ++l) {
        if (!dy[l] || !w[l] || !dw[l]) return EGO_ERR_ARG;
        a.dy[l] = (const bf16_t*)dy[l]; a.w[l] = w[l]; dst.p[l] = dw[l];
    }
    dst.seg = D;
    const int per = lnm_rows(ld);
    const int nwg = (rows + per - 1) / per;
    if (work_floats < colsum_work_floats(nwg, n_layers * D)) return EGO_ERR_ARG;
#define LNM_BWD_(C, T) EGO_LAUNCH((ln_bwd_multi_kernel<C, T>), dim3(nwg), dim3(256), 0, stream, a)
#define LNM_BWD(C) do { if (D % 4) LNM_BWD_(C, true); else LNM_BWD_(C, false); } while (0)
    if (ld <= 768) LNM_BWD(3); else if (ld <= 1024) LNM_BWD(4); else LNM_BWD(6);
#undef LNM_BWD_
#undef LNM_BWD
    LAUNCH_CHECK();
    colsum_launch(work, nwg, n_layers * D, dst, stream);
    LAUNCH_CHECK();
    return EGO_OK;
}

extern "C" int ego_swiglu_fwd(const void* ab, void* h, long rows, int F, hipStream_t stream) {
    if (rows <= 0) return EGO_OK;
    if (F % 8) return EGO_ERR_ARG;
    EGO_LAUNCH(swiglu_fwd_kernel, dim3(grid_for(rows * (F / 8))), dim3(256), 0, stream, (const bf16_t*)ab, (bf16_t*)h, rows, F);
    LAUNCH_CHECK();
    return EGO_OK;
}

extern "C" int ego_swiglu_bwd(const void* ab, const void* dh, void* dab, long rows, int F, hipStream_t stream) {
    if (rows <= 0) return EGO_OK;
    if (F % 8) return EGO_ERR_ARG;
    EGO_LAUNCH(swiglu_bwd_kernel, dim3(grid_for(rows * (F / 8))), dim3(256), 0, stream, (const bf16_t*)ab, (const bf16_t*)dh, (bf16_t*)dab, rows, F);
    LAUNCH_CHECK();
    return EGO_OK;
}

extern "C" int ego_cast_weight(const float* W, int rows, int cols, long ld_src, void* Wb, long ld_w, void* Wt, long ld_t,
                               int rows_dst, hipStream_t stream) {
    if (rows <= 0 || cols <= 0 || rows_dst < rows) return EGO_ERR_ARG;
    EGO_LAUNCH(cast_w_kernel, dim3((cols + 31) / 32, (rows_dst + 31) / 32), dim3(256), 0, stream, W, rows, cols, ld_src,
                       (bf16_t*)Wb, ld_w, (bf16_t*)Wt, ld_t, rows_dst);
    LAUNCH_CHECK();
    return EGO_OK;
}

extern "C" int ego_cast_f32_bf16(const float* src, void* dst, long n, hipStream_t stream) {
    if (n <= 0) return EGO_OK;
    if (n % 4) return EGO_ERR_ARG;
    EGO_LAUNCH(cast_f32_bf16_kernel, dim3(grid_for(n / 4)), dim3(256), 0, stream, src, (bf16_t*)dst, n / 4);
    LAUNCH_CHECK();
    return EGO_OK;
}

extern "C" int ego_ce_fwd(const void* logits, long ld, int V, const int* targets, const int* range, int max_rows,
                          float* lse, float* nll, hipStream_t stream) {
    if (max_rows <= 0) return EGO_OK;
    if (V % 8 || ld % 8) return EGO_ERR_ARG;
    EGO_LAUNCH(ce_fwd_kernel, dim3(max_rows), dim3(256), 0, stream, (const bf16_t*)logits, ld, V, targets, range, lse, nll);
    LAUNCH_CHECK();
    return EGO_OK;
}

extern "C" int ego_ce_bwd(void* logits, long ld, int V, const int* targets, const int* range, int max_rows,
                          const float* lse, const float* gscale, int n_mods, const float* loss_w, hipStream_t stream) {
    if (max_rows <= 0) return EGO_OK;
    if (V % 8 || ld % 8 || n_mods <= 0) return EGO_ERR_ARG;
    EGO_LAUNCH(ce_bwd_kernel, dim3(max_rows), dim3(256), 0, stream, (bf16_t*)logits, ld, V, targets, range, lse, gscale, 1.f / n_mods, loss_w);
    LAUNCH_CHECK();
    return EGO_OK;
}

extern "C" int ego_ce_fwd_bwd(void* logits, long ld, int V, const int* targets, const int* range, int max_rows, float* lse,
                              float* nll, const float* gscale, int n_mods, const float* loss_w, hipStream_t stream) {
    if (max_rows <= 0) return EGO_OK;
    if (V % 8 || ld % 8 || n_mods <= 0 || V > 65536) return EGO_ERR_ARG;      // a row must fit the registers of one workgroup
    if (V <= 2048) {
        EGO_LAUNCH(ce_fwd_bwd_kernel<1>, dim3(max_rows), dim3(256), 0, stream, (bf16_t*)logits, ld, V, targets, range, lse, nll, gscale, 1.f / n_mods, loss_w);
    } else {
        EGO_LAUNCH(ce_fwd_bwd_kernel<32>, dim3(max_rows), dim3(256), 0, stream, (bf16_t*)logits, ld, V, targets, range, lse, nll, gscale, 1.f / n_mods, loss_w);
    }
    LAUNCH_CHECK();
    return EGO_OK;
}

extern "C" int ego_loss_finalize(const float* nll, const int* ranges, int n_mods, float* out, int* err, const float* loss_w,
                                 const float* mod_scale, hipStream_t stream) {
    if (n_mods <= 0 || n_mods > EGO_MAX_MODS) return EGO_ERR_ARG;
    EGO_LAUNCH(loss_finalize_kernel, dim3(1), dim3(1024), 0, stream, nll, ranges, n_mods, out, err, loss_w, mod_scale);
    LAUNCH_CHECK();
    return EGO_OK;
}

namespace {
struct LossWArgs { const int* ranges; int n_mods, mode; int vocab[EGO_MAX_MODS]; float* lw; float* ms; };
// one thread: the loss weights of `weighted_mod` (mode 1) and `token` (mode 2) from the per-modality row counts of this batch
__global__ void loss_weights_kernel(LossWArgs a) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float tot = 0.f;
    for (int m = 0; m < a.n_mods; ++m) tot += (float)a.ranges[2 * m + 1] * (float)a.vocab[m];      // logits.numel() per modality
    for (int m = 0; m < a.n_mods; ++m) {
        if (a.mode == 1) {
            // loss / math.log(vocab_size) * 5.545177444479562 (= ln 256: "set 256 as default codebook size"), then the mean over modalities
            const float sc = 5.545177444479562f / logf((float)a.vocab[m]);
            a.ms[m] = sc;
            a.lw[m] = sc / (float)a.n_mods;
        } else {
            // sum(mod_loss * mod_count) / sum(mod_count), mod_count = logits.numel() = rows x vocab (0 / 0 = NaN like the reference)
            a.ms[m] = 1.f;
            a.lw[m] = (float)a.ranges[2 * m + 1] * (float)a.vocab[m] / tot;
        }
    }
}
}  // namespace

extern "C" int ego_loss_weights(const int* ranges, const int* vocab, int n_mods, int mode, float* loss_w, float* mod_scale,
                                hipStream_t stream) {
    if (n_mods <= 0 || n_mods > EGO_MAX_MODS || (mode != 1 && mode != 2) || !ranges || !vocab || !loss_w || !mod_scale) return EGO_ERR_ARG;
    LossWArgs a{ranges, n_mods, mode, {0}, loss_w, mod_scale};
    for (int m = 0; m < n_mods; ++m) { if (vocab[m] < 2) return EGO_ERR_ARG; a.vocab[m] = vocab[m]; }
    EGO_LAUNCH(loss_weights_kernel, dim3(1), dim3(64), 0, stream, a);
    LAUNCH_CHECK();
    return EGO_OK;
}

extern "C" long ego_bias_grad_work_floats(long rows, int D) { return colsum_work_floats((rows + 255) / 256, D); }

extern "C" int ego_bias_grad(const void* g, long rows, int D, float* db, float* work, long work_floats, hipStream_t stream) {
    if (rows <= 0) return EGO_OK;
    if (D % 8 || D > 2048 || (((uintptr_t)g) & 15) || !work || work_floats < ego_bias_grad_work_floats(rows, D)) return EGO_ERR_ARG;
    const int rgroups = 256 / (D / 8);
    const long nwg = (rows + 255) / 256;
    EGO_LAUNCH(bias_grad_kernel, dim3((unsigned)nwg), dim3(256), (size_t)rgroups * D * sizeof(float), stream,
               (const bf16_t*)g, rows, D, work);
    LAUNCH_CHECK();
    ColsumDst dst{};
    dst.p[0] = db; dst.seg = D;
    colsum_launch(work, nwg, D, dst, stream);
    LAUNCH_CHECK();
    return EGO_OK;
}

extern "C" int ego_quant_fp8_rows(const void* X, long ld, long rows, int K, void* Q, long ldq, float* scale, hipStream_t stream) {
    if (rows <= 0) return EGO_OK;
    if (K <= 0 || K % 8 || ld % 8 || ldq % 8) return EGO_ERR_ARG;
    EGO_LAUNCH(quant_fp8_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream, (const bf16_t*)X, ld, rows, K, (unsigned char*)Q, ldq,
               scale);
    LAUNCH_CHECK();
    return EGO_OK;
}
