/* C-ABI of libegom2p_hip.so - the MI355X (gfx950) engine under the EgoM2P Python surface.
 *
 * The reference (lgen-sudo/EgoM2P) is pure PyTorch and has no FFI; its drop-in boundary is the
 * nn.Module surface of `EgoM2P` (egom2p/models/egom2p_model.py:57-734).  These entry points are what
 * sits under that surface here: plain C functions over raw device pointers, explicit shapes/strides
 * and a hipStream_t.  They return 0 on success (EGO_ERR_ARG = rejected arguments, EGO_ERR_LAUNCH =
 * launch failed), never allocate caller-visible memory and never synchronise, so every call is capturable
 * in a hipGraph and re-entrant per stream.  The library reads no environment variables; its only process-wide
 * state is the GEMM tile-family selection set by ego_gemm_kernel_mode / ego_gemm_small_tiles (test / tuning hooks, default "by shape").
 *
 * Each declaration cites the reference code it replaces (paths relative to the reference root).
 * All "bf16" pointers are raw 16-bit bfloat16; "row-major [R, C] with ld" means element (r, c) at
 * base + r * ld + c.
 */
#ifndef EGOM2P_HIP_H
#define EGOM2P_HIP_H

#include <hip/hip_runtime_api.h>

#ifdef __cplusplus
extern "C" {
#endif

/* bumped whenever an existing entry point changes its signature or meaning (2: round 2 added arguments to
 * ego_layernorm_fwd / ego_attn_*_d64 / ego_loss_finalize and removed ego_grad_scale; 3: round 3 gave ego_layernorm_bwd,
 * ego_bias_grad and ego_embed_bwd a scratch buffer for their atomic-free reductions; 4: ego_layernorm_fwd / _bwd take the
 * row pitch `ld` beside the normalised width D; 5: ego_compact_desc grew `seg_bad`, ego_embed_bwd_desc `vocab`, ego_ce_bwd / ego_ce_fwd_bwd / ego_loss_finalize take loss weights; 6 (round 5): register tokens - ego_compact_desc grew `n_reg`,
 * ego_embed_desc `reg`, new ego_reg_grad - and ego_sample_cfg_topp takes top_k; loaders must refuse other versions) */
#define EGO_ABI_VERSION 6
#define EGO_MAX_MODS 8

/* GEMM epilogues */
#define EGO_EPI_BF16 0        /* C(bf16) = acc                                            */
#define EGO_EPI_F32 1         /* C(f32)  = acc                                            */
#define EGO_EPI_RESID 2       /* C(f32)  = R(f32) + bf16(acc)        residual add         */
#define EGO_EPI_BIAS_RESID 3  /* C(f32)  = R(f32) + bf16(acc + bias) context projection   */

int ego_abi_version(void);
/* Tile family the GEMM entries may pick (1 = by shape, 0 = 128x128 kernels only, 2 = 256x256 wherever legal); same
 * results up to fp32 summation order.  No reference counterpart: test / tuning hook (tests/test_engine_gpu.py). */
int ego_gemm_kernel_mode(int nt256, int tn256);
/* NT launches of at most `max_tiles128` 128x128 output tiles (under-filled grids: the 1707-row linears of the generation
 * path, egom2p/models/generate.py:747-766) run on the 64x64-tile small-grid kernel; 0 = never, < 0 = query only.  Returns
 * the previous threshold (default 400).  Same results (same K order); test / tuning hook like ego_gemm_kernel_mode. */
int ego_gemm_small_tiles(int max_tiles128);
/* Probe hook (timing only, results never change; no reference counterpart).  key 1: every other persistent 256 x 256 NT workgroup
 * of an XCD starts `value` x ~0.5 us late (de-phases the workgroups' epilogue store bursts: tools/dephase_probe.py); 0 = off (the
 * product setting).  Returns the previous value; value < 0 only queries; unknown key: -1. */
int ego_gemm_tune(int key, int value);
/* The same for the attention kernels: key 1 = start delay (x 512 clocks) of the dK / dV workgroups dispatched second on their CU
 * (phase probe of DESIGN.md section 4f; 0 = off, the product setting). */
int ego_attn_tune(int key, int value);

/* ---- front end ------------------------------------------------------------------------------- */

/* Stable-partition compaction of one side (encoder inputs / decoder targets) of a clip batch.
 * Replaces cat_encoder_tensors + forward_mask_encoder (egom2p/models/egom2p_model.py:251-283, 344-396)
 * and cat_decoder_tensors + forward_mask_decoder + adapt_decoder_attention_mask (:285-342, 398-481).
 * Modalities are given in concatenation order (decoder: the shuffled order of :312). */
typedef struct {
    int n_mods, n_keep, is_decoder;
    const void* mask[EGO_MAX_MODS];  /* uint8/bool [B, n_pos[m]], nonzero = ignore                       */
    const void* ids[EGO_MAX_MODS];   /* int64 [B, n_pos[m]] token ids                                     */
    const int* dam[EGO_MAX_MODS];    /* int32 [B, n_pos[m]] decoder_attention_mask (decoder only, or NULL)*/
    int n_pos[EGO_MAX_MODS];
    int mod_id[EGO_MAX_MODS];        /* sha256-derived modality ids (egom2p/utils/misc.py:39-41)          */
    void* ids_keep;                  /* out int64 [B, n_keep]                                             */
    void* pad;                       /* out uint8 [B, n_keep] 1 = padding row                             */
    void* mod_mask;                  /* out int16 [B, n_keep] modality id, -1 on padding                  */
    int* slot;                       /* out int32 [B, n_keep] modality slot, -1 on padding                */
    int* local;                      /* out int32 [B, n_keep] position inside the modality                */
    int* tok;                        /* out int32 [B, n_keep] token id (decoder: target id), 0 on padding */
    int* ks; int* ke;                /* out int32 [B, n_keep] allowed attention key interval per row      */
    int* n_valid;                    /* out int32 [B]                                                     */
    int* seg;                        /* out int32 [B, n_mods, 2] (start, count) of each slot's kept rows  */
    int* err;                        /* in/out int32 [1]: |= 1 if the decoder mask is not an interval     */
    int* seg_bad;                    /* out int32 [B] (decoder only, or NULL): nonzero if some kept unmasked row's interval
                                        is not exactly its slot's segment (start, count) - ego_attn_*_d64_seg then takes
                                        the per-row path for that sample                                                  */
    int n_reg;                       /* register tokens (encoder only; egom2p_model.py:381-387): every per-row output has
                                        n_reg + n_keep entries per sample, the first n_reg of them the register rows - pad 0,
                                        mod_mask -1, slot -2, local = register index, tok 0, ids_keep -1 (the reference's
                                        ids_keep has no such entries: compare [:, n_reg:]); n_valid and the key intervals
                                        count them: n_valid = n_reg + kept, [ks, ke) = [0, n_valid)                         */
} ego_compact_desc;
int ego_compact(const ego_compact_desc* d, int B, hipStream_t stream);

/* Fused embedding of the kept rows: x = token_row + (pos_row + mod_emb), emb = pos_row + mod_emb;
 * padding rows are zero.  Replaces the embedding modules' forward (encoder_embeddings.py:181-210,
 * 272-301; decoder_embeddings.py:337-370, 455-487) + torch.gather x2 + masked zeroing
 * (egom2p_model.py:375-391, 427-438, 718, 723).  Decoder: table[] NULL and base_vec = mask_token (:328). */
typedef struct {
    const float* table[EGO_MAX_MODS];
    const float* pos[EGO_MAX_MODS];
    const float* mod[EGO_MAX_MODS];
    const float* base_vec;
    const int* slot; const int* local; const int* tok;
    float* x; float* emb;
    long rows; int D;
    const float* reg;                /* register tokens [n_reg, D] (egom2p_model.py:170-171) or NULL: a row with slot -2 is
                                        x = reg[local], emb = 0 (`torch.cat([register_tokens, ...])`, zeros_like for emb)   */
} ego_embed_desc;
int ego_embed_fwd(const ego_embed_desc* d, hipStream_t stream);

/* Backward of the above: embedding_dense_backward scatter-add + mod_emb / mask_token reductions. */
typedef struct {
    float* dtable[EGO_MAX_MODS];
    float* dmod[EGO_MAX_MODS];
    float* dbase;
    const float* dx; const float* d2;
    const int* slot; const int* tok;
    long rows; int D, n_mods;
    unsigned char* touched[EGO_MAX_MODS];   /* optional uint8 [V] per table: set to 1 for every row that got a gradient */
    float* work; long work_floats;          /* scratch, at least ego_embed_bwd_work_floats(rows, D, n_mods) floats */
    int vocab[EGO_MAX_MODS];                /* rows of dtable[m] (0 = not given): > 65536 is refused (16-bit table keys) */
} ego_embed_bwd_desc;
/* No float atomics: column sums through per-workgroup partial rows + an ordered reduction, the table scatter as a gather by
 * the table row's owner in ascending row order - results are bitwise reproducible.  Vocabularies up to 65536. */
long ego_embed_bwd_work_floats(long rows, int D, int n_mods);
int ego_embed_bwd(const ego_embed_bwd_desc* d, hipStream_t stream);
/* Gradient of the register tokens (the backward of `repeat(self.register_tokens, '() n d -> b n d', b=B)`,
 * egom2p_model.py:382): dreg[r] += sum over b (in batch order: bitwise reproducible) of dx[b * rows_per_sample + r], r < n_reg.
 * (ego_embed_bwd skips rows with a negative slot, the register rows among them.) */
int ego_reg_grad(const float* dx, int B, long rows_per_sample, int n_reg, int D, float* dreg, hipStream_t stream);

/* Row lists for the sparse data-parallel exchange of an embedding table's gradient (replaces, for few clips per step,
 * the dense all-reduce DDP performs on the 64000 x 768 tables, run_training_egom2p.py:514; encoder_embeddings.py:200,291).
 * ego_rows_compact: touched[V] -> rows[cap] ascending (-1 padded), *count = number of rows (bit 30 set: more than cap
 * rows were touched, the list is truncated); clears the flags.  ego_rows_gather: out[i] = table[rows[i]] (zeros past
 * count).  ego_rows_scatter: table[rows[i]] = (add ? table[rows[i]] : 0) + src[i] for i < *count (src NULL: zeros). */
int ego_rows_compact(void* touched, int V, int cap, int* rows, int* count, hipStream_t stream);
int ego_rows_gather(const float* table, const int* rows, const int* count, int cap, int D, float* out, hipStream_t stream);
int ego_rows_scatter(float* table, const int* rows, const int* count, int cap, int D, const float* src, int add,
                     hipStream_t stream);

/* Row order of `y[decoder_mod_mask == id]` for every modality (egom2p_model.py:633): perm[row] = row
 * in the modality-grouped order (-1 for padding), targets gathered alongside, (offset,count) per
 * modality in `ranges`.  `base` is int32 [B, n_mods] scratch. */
int ego_loss_perm(const int* seg, const int* canon, const int* slot, const int* tok, int B, int M, int n_mods,
                  int* perm, int* tgt_perm, int* ranges, int* base, hipStream_t stream);

/* ---- transformer blocks ---------------------------------------------------------------------- */

/* Bias-free LayerNorm (egom2p/models/egom2p_utils.py:118-133): y(bf16)[out_row[r]] = LN(x[r]) * w.
 * out_row may be NULL (identity); -1 drops the row. mean/rstd are saved for the backward.  q8 / qscale (optional): the
 * row also leaves as e4m3 bytes + scale, exactly what ego_quant_fp8_rows would make of y (operand of an fp8 GEMM; ld == D).
 * D = the normalised width, ld >= D = the row pitch of x and y in elements (ld > D: a model dimension stored padded - the
 * registered ego-L's 1020 in rows of 1024; the columns [D, ld) of x must be zero, those of y are written as zeros; ld % 4 == 0,
 * any D <= ld; w / dw hold at least D rounded up to 4 floats). */
int ego_layernorm_fwd(const float* x, const float* w, void* y_bf16, float* mean, float* rstd, const int* out_row,
                      int rows, int D, long ld, float eps, void* q8, long ldq, float* qscale, hipStream_t stream);
/* dx_out = (dx_in ? dx_in : 0) + LN'(dy); dw += sum_rows dy * xhat.  dy_row: same map as out_row.  The weight gradient goes
 * through one partial row per workgroup in `work` (>= ego_layernorm_bwd_work_floats(rows, D) floats) and an ordered
 * reduction: no float atomics, bitwise reproducible.  D / ld as in the forward (dy, x, dx_* rows have pitch ld; the gradient's
 * columns [D, ld) are written as zeros). */
long ego_layernorm_bwd_work_floats(int rows, int D);
int ego_layernorm_bwd(const void* dy_bf16, const int* dy_row, const float* x, const float* mean, const float* rstd,
                      const float* w, const float* dx_in, float* dx_out, void* dx_out_bf16, float* dw, float* work,
                      long work_floats, int rows, int D, long ld, hipStream_t stream);

/* C[M,N] = A[M,K] . B[N,K]^T, bf16 inputs, fp32 MFMA accumulate.  Replaces F.linear under
 * autocast(bf16) (egom2p_utils.py:141-169, 180-203, 215-242; decoder_embeddings.py:372-383, 489-500)
 * and its dgrad (B = W^T).  m_range: optional device int[2] = {row offset, row count}; then M is an
 * upper bound for the grid and A/C/R rows start at the offset. */
int ego_gemm_nt_bf16(const void* A, long lda, const void* B, long ldb, void* C, long ldc, const float* R, long ldr,
                     const float* bias, const int* m_range, int M, int N, int K, int epi, hipStream_t stream);
/* C[Ni,Nj] += P[M,Ni]^T . Q[M,Nj] (wgrad).  Output rows [0,split_row) go to C0 (first rows0 valid),
 * rows >= split_row to C1 (first rows1 valid) - lets fused/padded weights scatter to their own grads.
 * splits > 1: deterministic split over M through `slab` (fp32 [splits, Ni, Nj]). */
int ego_gemm_tn_bf16(const void* P, long ldp, const void* Q, long ldq, float* C0, float* C1, long ldc, int split_row,
                     int rows0, int rows1, const int* m_range, int Ni, int Nj, int M, int splits, float* slab,
                     hipStream_t stream);

/* fp8 forward linears (BASELINE config 5: "bf16 + fp8 MFMA GEMMs"; the reference has no fp8 path - the contract is the
 * same F.linear, egom2p_utils.py:141-169, 180-203, 215-242, to a stated tolerance).  Operands are OCP e4m3 bytes with one
 * fp32 scale per row (activations) / per output channel (weights): C[m,n] = sa[m] * sb[n] * sum_k A8[m,k] B8[n,k], MFMA
 * v_mfma_scale_f32_16x16x128_f8f6f4 with fp32 accumulation, same epilogues as ego_gemm_nt_bf16.  K % 128 == 0, K >= 256,
 * N % 128 == 0.  ego_quant_fp8_rows makes the operands: scale[r] = amax(row r) / 448, Q = e4m3(X / scale). */
int ego_quant_fp8_rows(const void* X_bf16, long ld, long rows, int K, void* Q, long ldq, float* scale, hipStream_t stream);
int ego_gemm_nt_fp8(const void* A8, long lda, const float* sa, const void* B8, long ldb, const float* sb, void* C, long ldc,
                    const float* R, long ldr, const float* bias, int M, int N, int K, int epi, hipStream_t stream);
int ego_gemm_nt_swiglu_fwd_fp8(const void* X8, long ldx, const float* sx, const void* W13_8, long ldw, const float* sw,
                               void* ab, long ld_ab, void* h, long ld_h, int M, int F, int K, hipStream_t stream);

/* Split-K factor `splits` the wgrad above should be called with for this shape (>= 1; 1 when the row range lives on the
 * device): one round of workgroups, bounded by the slab the caller owns (slab_elems fp32).  The launcher's own rule -
 * the host never mirrors it. */
int ego_gemm_tn_plan(int Ni, int Nj, int M, long ldp, long ldq, long slab_elems, int ranged);

/* Fused attention, head_dim 64 (Attention / CrossAttention, egom2p_utils.py:185-205, 222-244).
 * Element (b, row, head h, d) of X at X + b * x_bs + row * x_rs + h * 64 + d; every tensor 16-byte aligned with batch and
 * row strides that are multiples of 8 elements (16-byte row fragments in and out), else EGO_ERR_ARG.  ks/ke: allowed key
 * interval of query row (b, q) at [b * r_bs + q * r_rs] (r_rs = 0: one interval per sample).
 * LSE: fp32 [B, H, Nq], MINUS the log2-domain log-sum-exp of the scaled scores (an opaque hand-over from the forward to
 * the backward, which uses it - and DELTA, MINUS rowsum(dO o O), written by the backward's first kernel - as the initial
 * accumulator of its score / dP MFMA chains).
 * O_lo (optional, NULL = off; laid out like O): the forward also stores bf16(o - bf16(o)) and the backward forms
 * delta = rowsum(dO o (O + O_lo)) - torch's softmax backward gets its row sums from the fp32 probabilities, the
 * flash-style rowsum(dO o O) from the bf16-rounded output loses that (3.5 % on the cross-attention query gradients at
 * N = 2048; with O_lo they are at the tensor's own bf16 noise). */
int ego_attn_fwd_d64(const void* Q, long q_bs, long q_rs, const void* K, long k_bs, long k_rs, const void* V, long v_bs,
                     long v_rs, void* O, long o_bs, long o_rs, void* O_lo, float* LSE, const int* ks, const int* ke,
                     long r_bs, long r_rs, int B, int H, int Nq, int Nk, float scale, hipStream_t stream);
int ego_attn_bwd_d64(const void* Q, long q_bs, long q_rs, const void* K, long k_bs, long k_rs, const void* V, long v_bs,
                     long v_rs, const void* O, long o_bs, long o_rs, const void* O_lo, const void* dO, long do_bs, long do_rs,
                     const float* LSE, float* DELTA, void* dQ, long dq_bs, long dq_rs, void* dK, long dk_bs, long dk_rs,
                     void* dV, long dv_bs, long dv_rs, const int* ks, const int* ke, long r_bs, long r_rs, int B, int H,
                     int Nq, int Nk, float scale, hipStream_t stream);
/* ego_attn_fwd_d64 for UNDER-FILLED grids (the generation path's 1707 decoder rows x 12 heads = 168 workgroups on 256 CUs,
 * egom2p/models/generate.py:747-766): the keys of every query tile are cut into kv_splits runs handled by separate workgroups
 * (flash-decoding), whose unnormalised outputs and (max, sum) pairs go through `ws` (at least
 * ego_attn_fwd_split_floats(B, H, Nq, kv_splits) floats, 16-byte aligned) and are joined by a second kernel.  Same result as
 * ego_attn_fwd_d64 up to the order of the fp32 sums; LSE may be NULL; no O_lo; kv_splits <= 1: the plain launch. */
long ego_attn_fwd_split_floats(int B, int H, int Nq, int kv_splits);
int ego_attn_fwd_d64_split(const void* Q, long q_bs, long q_rs, const void* K, long k_bs, long k_rs, const void* V, long v_bs,
                           long v_rs, void* O, long o_bs, long o_rs, float* LSE, const int* ks, const int* ke, long r_bs,
                           long r_rs, int B, int H, int Nq, int Nk, float scale, int kv_splits, float* ws, long ws_floats,
                           hipStream_t stream);

/* The same two for SELF-attention under a block-diagonal mask given as row groups (the decoder's modality-wise mask,
 * egom2p_model.py:446-481, as ego_compact writes it): seg int32 [B, n_seg, 2] = (first row, row count) of each group of a
 * sample, ascending and disjoint; every row of a group attends exactly the group's rows.  The kernels cut every group into
 * tiles of its own, so each workgroup sees ONE interval (the fast path of the per-sample launches: no per-row mask work, no
 * tile that straddles two groups).  Rows behind the last group, and every row of a sample with seg_bad[b] != 0 (seg_bad
 * may be NULL), use their per-row intervals ks / ke (r_rs = 1, r_bs = Nq, required) - results are those of
 * ego_attn_fwd_d64 / ego_attn_bwd_d64 on the same intervals up to fp32 summation order.  Nq == Nk.  seg = NULL: the plain
 * entry points. */
int ego_attn_fwd_d64_seg(const void* Q, long q_bs, long q_rs, const void* K, long k_bs, long k_rs, const void* V, long v_bs,
                         long v_rs, void* O, long o_bs, long o_rs, void* O_lo, float* LSE, const int* ks, const int* ke,
                         long r_bs, long r_rs, const int* seg, int n_seg, const int* seg_bad, int B, int H, int Nq, int Nk,
                         float scale, hipStream_t stream);
int ego_attn_bwd_d64_seg(const void* Q, long q_bs, long q_rs, const void* K, long k_bs, long k_rs, const void* V, long v_bs,
                         long v_rs, const void* O, long o_bs, long o_rs, const void* O_lo, const void* dO, long do_bs,
                         long do_rs, const float* LSE, float* DELTA, void* dQ, long dq_bs, long dq_rs, void* dK, long dk_bs,
                         long dk_rs, void* dV, long dv_bs, long dv_rs, const int* ks, const int* ke, long r_bs, long r_rs,
                         const int* seg, int n_seg, const int* seg_bad, int B, int H, int Nq, int Nk, float scale,
                         hipStream_t stream);


/* The same two entries for a head dimension other than 64: every head is stored padded with zero columns to
 * hd_pad = 96 or 128 elements (the registered ego-L, egom2p_model.py:1080-1092, has 15 heads of 68: 68^-0.5 is `scale`),
 * head h of a row at element offset h * hd_pad.  LDS-staged 32-row tiles (tiles outside every row's interval are skipped);
 * not the throughput path of the head-dim-64 models.  hd_pad: the low 16 bits are that pitch; bits 16 and up may carry the
 * head's real dimension (68: hd_pad = 96 | 68 << 16; 0 = not given) - contraction steps over columns that are all padding
 * are then left out, the results are the same bits.  LSE / DELTA are opaque to the caller, as above; other pitches ->
 * EGO_ERR_ARG. */
int ego_attn_fwd_hd(const void* Q, long q_bs, long q_rs, const void* K, long k_bs, long k_rs, const void* V, long v_bs,
                    long v_rs, void* O, long o_bs, long o_rs, void* O_lo, float* LSE, const int* ks, const int* ke,
                    long r_bs, long r_rs, int B, int H, int Nq, int Nk, int hd_pad, float scale, hipStream_t stream);
int ego_attn_bwd_hd(const void* Q, long q_bs, long q_rs, const void* K, long k_bs, long k_rs, const void* V, long v_bs,
                    long v_rs, const void* O, long o_bs, long o_rs, const void* O_lo, const void* dO, long do_bs, long do_rs,
                    const float* LSE, float* DELTA, void* dQ, long dq_bs, long dq_rs, void* dK, long dk_bs, long dk_rs,
                    void* dV, long dv_bs, long dv_rs, const int* ks, const int* ke, long r_bs, long r_rs, int B, int H,
                    int Nq, int Nk, int hd_pad, float scale, hipStream_t stream);

/* SwiGLU gate on the fused fc1||fc3 output ab[rows, 2F] (GatedMlp, egom2p_utils.py:167-169). */
int ego_swiglu_fwd(const void* ab, void* h, long rows, int F, hipStream_t stream);
int ego_swiglu_bwd(const void* ab, const void* dh, void* dab, long rows, int F, hipStream_t stream);
/* fc1||fc3 and the gate in one launch: ab[M,2F] = X[M,K] @ W13[2F,K]^T and h = bf16(bf16(silu(a)) * b) (GatedMlp.forward,
 * egom2p_utils.py:167-169).  A tile of the GEMM is 128 columns of the a half and the matching 128 columns of the b half,
 * so the gate is formed in the epilogue; ab is still stored (the backward reads it) but never read back.  Bitwise the
 * result of ego_gemm_nt_bf16(EPI_BF16) + ego_swiglu_fwd.  F % 128 == 0, K % 64 == 0, K >= 128 (else EGO_ERR_ARG). */
int ego_gemm_nt_swiglu_fwd(const void* X, long ldx, const void* W13, long ldw, void* ab, long ld_ab, void* h, long ld_h,
                           int M, int F, int K, hipStream_t stream);
/* The fc2 input gradient and the gate backward in one launch: dh = dY[M,K] @ W2t[F,K]^T is formed tile by tile in
 * the GEMM and consumed by its epilogue (autograd of GatedMlp.forward, egom2p_utils.py:167-169, through fc2 and the
 * SiLU gate); dab[M,2F] receives exactly what ego_gemm_nt_bf16(EPI_BF16) + ego_swiglu_bwd would write, without the
 * 2 x M x F x 2 bytes of dh going to HBM and back.  F % 256 == 0, K % 64 == 0, K >= 128 (EGO_ERR_ARG otherwise: the
 * caller then uses the two-call form). */
int ego_gemm_nt_swiglu_bwd(const void* dY, long ldy, const void* W2t, long ldw, const void* ab, void* dab, long ld_ab,
                           int M, int F, int K, hipStream_t stream);

/* ---- input contract on the device ------------------------------------------------------------ */
/* Synthetic `mod_dict` entries of one modality for B clips (egom2p/data/masking.py:236-266 builds them on CPU workers:
 * random permutation of the positions, first k_in inputs, next k_tgt targets, decoder_attention_mask = k_tgt at the
 * first target).  ids[B,n] int64 in [0,vocab), masks[B,n] uint8 (1 = ignore), dam[B,n] int32.  key_ids / key_perm:
 * one 64-bit stream key per clip (host: sha256 of "<seed>:clip<s>.<mod>.ids|perm"); bit-identical to
 * egom2p_amd/synth.py:make_clip_batch.  n <= 8192.  ids may be NULL: masks and marker only, for clips whose tokens already
 * exist (egom2p_amd/masking.py: UnifiedMasking on real token tensors). */
int ego_clip_synth(const void* key_ids, const void* key_perm, const int* k_in, const int* k_tgt, int B, int n, int vocab,
                   long* ids, void* input_mask, void* target_mask, int* dam, hipStream_t stream);

/* Token budgets per clip and modality, the reference's Dirichlet-mixture sampler (`UnifiedMasking.input_token_budget` /
 * `target_token_budget`, egom2p/data/masking.py:181-234, mixture / token-count draws :530-541) on the device: floor of
 * Dirichlet(alpha) * N, leftover tokens to the arg-max of further draws, clamp to max_tokens (targets: to what the inputs
 * left), redraw below min_tokens.  clip_keys: one 64-bit stream key per clip.  k_in / k_tgt: int32 [n_mods, B] - the
 * per-modality rows feed ego_clip_synth directly.  alphas are clamped by the caller (> 0; the reference clamps at 1e-9). */
#define EGO_MAX_MIX 8
typedef struct {
    int n_mods, n_mix;
    float in_alpha[EGO_MAX_MIX][EGO_MAX_MODS];     /* [mixture component][modality] */
    float tgt_alpha[EGO_MAX_MIX][EGO_MAX_MODS];
    float mix_weight[EGO_MAX_MIX];                 /* sampling_weights of the mixture */
    int max_tokens[EGO_MAX_MODS], min_tokens[EGO_MAX_MODS];
    int not_seq[EGO_MAX_MODS];                     /* 1: img / cam / gaze type (targets cannot reuse input positions) */
    int n_in_lo, n_in_hi, n_tgt_lo, n_tgt_hi;      /* input_tokens_range / target_tokens_range, inclusive */
    int max_tries;
} ego_budget_desc;
int ego_budget_dirichlet(const ego_budget_desc* d, const void* clip_keys, int B, int* k_in, int* k_tgt, hipStream_t stream);

/* One input, n_layers LayerNorms: the decoder's `context_norm` of every layer normalises the same context tensor with its own
 * weight (egom2p_utils.py:387-391 per DecoderBlock, egom2p_model.py:520-521).  Forward: x and its statistics are read /
 * formed once, y[l] = bf16(LN(x) * w[l]) for every layer (bitwise ego_layernorm_fwd's rows).  Backward: x once, the upstream
 * gradients dy[l] once each, dx_out = (dx_in) + sum_l LN-backward_l - summed in the order l = n_layers-1 .. 0 like the chained
 * ego_layernorm_bwd calls (bit for bit their dx) - and dw[l] += the layer's weight gradient (ordered column sums: bitwise
 * reproducible, not bitwise the chained calls' - the partial rows are cut differently).  w / y / dy / dw: HOST arrays of
 * n_layers device pointers (n_layers <= 32).  Backward: ld <= 1536. */
int ego_layernorm_fwd_multi(const float* x, int n_layers, const float* const* w, void* const* y, float* mean, float* rstd, int rows,
                            int D, long ld, float eps, hipStream_t stream);
long ego_layernorm_bwd_multi_work_floats(int rows, int D, int n_layers);
int ego_layernorm_bwd_multi(int n_layers, const void* const* dy, const float* const* w, float* const* dw, const float* x,
                            const float* mean, const float* rstd, const float* dx_in, float* dx_out, void* dx_bf16, float* work,
                            long work_floats, int rows, int D, long ld, hipStream_t stream);

/* ---- loss head ------------------------------------------------------------------------------- */

/* F.cross_entropy(reduction='mean') per modality over bf16 logits rows [range[0], range[0]+range[1])
 * (egom2p_model.py:633-644).  bwd overwrites the logits with d loss / d logits (bf16): (softmax - onehot) * gscale / (n_mods * n)
 * for loss_type 'mod'; with loss_w (a pointer to THIS modality's entry of ego_loss_weights' loss_w; NULL = 'mod')
 * (softmax - onehot) * gscale * loss_w[0] / n. */
int ego_ce_fwd(const void* logits, long ld, int V, const int* targets, const int* range, int max_rows, float* lse,
               float* nll, hipStream_t stream);
int ego_ce_bwd(void* logits, long ld, int V, const int* targets, const int* range, int max_rows, const float* lse,
               const float* gscale, int n_mods, const float* loss_w, hipStream_t stream);
/* ego_ce_fwd and ego_ce_bwd in one pass over the logits (the training step knows the upstream gradient `gscale` when it
 * forms the loss): lse / nll as ego_ce_fwd writes them, the logits overwritten as ego_ce_bwd does, bitwise the two-call
 * result; the rows are read once.  V <= 65536 (a row is held in one workgroup's registers), else EGO_ERR_ARG: the caller
 * then uses the two calls. */
int ego_ce_fwd_bwd(void* logits, long ld, int V, const int* targets, const int* range, int max_rows, float* lse, float* nll,
                   const float* gscale, int n_mods, const float* loss_w, hipStream_t stream);
/* Per-modality loss weights of the reference's other two loss types from the row counts of this batch (ranges[m] = (first
 * row, rows) on the device): mode 1 = 'weighted_mod' (forward_weighted_mod_loss, egom2p_model.py:583-612: every modality's
 * mean CE divided by ln(vocab) and multiplied by ln 256, then the mean over modalities), mode 2 = 'token'
 * (forward_token_loss, :646-681: the modalities' mean CEs weighted by logits.numel() = rows x vocab).  loss_w[m] multiplies
 * modality m's mean nll in the total, mod_scale[m] the reported per-modality loss.  vocab: HOST array of n_mods sizes. */
int ego_loss_weights(const int* ranges, const int* vocab, int n_mods, int mode, float* loss_w, float* mod_scale,
                     hipStream_t stream);
/* out[0] = mean over modalities of per-modality mean nll (empty modality = 0), out[1+m] = per modality; with loss_w /
 * mod_scale (ego_loss_weights; NULL = loss_type 'mod'): out[0] = sum_m loss_w[m] * mean_nll[m], out[1+m] = mod_scale[m] * mean_nll[m].
 * err (optional): ego_compact's flag word; if set, every output is NaN (the reference's non-finite-loss exit,
 * run_training_egom2p.py:731-734, then stops the run instead of training on a wrong attention mask) and it is cleared. */
int ego_loss_finalize(const float* nll, const int* ranges, int n_mods, float* out, int* err, const float* loss_w,
                      const float* mod_scale, hipStream_t stream);

/* ---- generation (config 4) ------------------------------------------------------------------- */

/* Per decoded row: logits = uncond + (cond - uncond) * cfg_scale (uncond NULL: logits = cond); nucleus
 * top_k (> 0: keep the tokens whose logit is not below the k-th largest one - ties with it stay, generate.py:335-345; 0: off),
 * then the nucleus filter top_p (<= 0: off) on the softmax of the surviving logits; sample from softmax(kept / temperature) with
 * the caller's uniform number uniforms[row] (temperature <= 1e-10: arg-max).  Replaces guided_roar_step_batched's CFG mix,
 * top_k_top_p_filtering, softmax and torch.multinomial (egom2p/models/generate.py:332-371, 805-808).  top_k is the COUNT
 * (the reference turns a float top_k into int(top_k * V) on the host, :337-340).
 * cond/uncond: bf16 [rows, ld >= V]; out_prob (optional): probability of the sampled token. */
int ego_sample_cfg_topp(const void* cond, const void* uncond, long ld, int V, float cfg_scale, float top_p, int top_k,
                        float temperature, const float* uniforms, int* out_tokens, float* out_prob, int rows,
                        hipStream_t stream);

/* ---- parameters / optimiser ------------------------------------------------------------------ */

/* fp32 master W[rows, cols] -> bf16 W (zero-padded to rows_dst rows) and/or bf16 W^T [cols, rows_dst]. */
int ego_cast_weight(const float* W, int rows, int cols, long ld_src, void* Wb, long ld_w, void* Wt, long ld_t,
                    int rows_dst, hipStream_t stream);
int ego_cast_f32_bf16(const float* src, void* dst, long n, hipStream_t stream);
/* db += column sums of g (the bias gradient of decoder_proj_context, egom2p_model.py:157); same partial-row scheme. */
long ego_bias_grad_work_floats(long rows, int D);
int ego_bias_grad(const void* g_bf16, long rows, int D, float* db, float* work, long work_floats, hipStream_t stream);

/* clip_grad_norm_ + AdamW over flat buffers (egom2p/utils/native_scaler.py:28-43, optim_factory.py:226). */
/* *out += sum g^2.  work: EGO_SQNORM_WORK doubles of scratch - per-workgroup partial sums added in workgroup order (bitwise
 * reproducible); NULL: one double atomic per workgroup. */
#define EGO_SQNORM_WORK 2048
int ego_grad_sqnorm(const float* g, long n, double* out, double* work, hipStream_t stream);
int ego_adamw_step(float* p, float* g, float* m, float* v, long n, float lr, float wd, float beta1, float beta2, float eps,
                   int step, float gscale, float max_norm, const double* sqnorm, int zero_grad, hipStream_t stream);

/* ---- data-parallel gradient exchange (RCCL over xGMI) ---------------------------------------- */

/* Replaces DistributedDataParallel's gradient all-reduce for the reference (run_training_egom2p.py:514-515, :723;
 * egom2p/utils/dist.py:78-100): one process per GPU, one communicator per process.  Rank 0 calls ego_dp_unique_id and hands
 * the 128 bytes to the other ranks by whatever launcher channel exists (egom2p_amd/dp.py broadcasts them through
 * torch.distributed's store); every rank then calls ego_dp_comm_create (collective).  ego_dp_allreduce_begin sums
 * buf[0, count) over the ranks IN PLACE on comm_stream, ordered behind what compute_stream holds at the time of the call, and
 * returns at once (algo 0: all-reduce, 1: reduce-scatter + all-gather in place); ego_dp_wait makes compute_stream wait for
 * the exchanges issued so far.  RCCL is resolved with dlopen at first use (no link-time dependency). */
typedef struct ego_dp_comm ego_dp_comm;
int ego_dp_unique_id(void* out128);
int ego_dp_comm_create(const void* unique_id128, int rank, int world, ego_dp_comm** out);
int ego_dp_comm_destroy(ego_dp_comm* comm);
int ego_dp_allreduce_begin(ego_dp_comm* comm, float* buf, long count, int algo, hipStream_t compute_stream,
                           hipStream_t comm_stream);
int ego_dp_wait(ego_dp_comm* comm, hipStream_t compute_stream, hipStream_t comm_stream);

#ifdef __cplusplus
}
#endif
#endif
