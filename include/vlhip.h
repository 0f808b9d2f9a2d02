/* vlhip.h -- C ABI of libvlhip.so: hand-written gfx950 (MI355X / CDNA4) kernels for the UC2 / M3P VQA
 * fine-tuning hot path of nooralahzadeh/CLG-VQA.
 *
 * This boundary replaces the reference's only native plugin boundary on this path -- the pybind11 torch
 * extensions of vendored apex (volta/apex/csrc/layer_norm_cuda.cpp:121-239 forward_affine/backward_affine,
 * volta/apex/csrc/flatten_unflatten.cpp:1-18, volta/apex/csrc/multi_tensor_scale_kernel.cu) -- and the eager
 * torch ops the reference issues around it (file:line cited per entry point below; paths are relative to
 * /root/reference/volta).
 *
 * Conventions
 *   - Plain pointers and sizes only; no torch / HIP types in signatures (`stream` is a hipStream_t passed as
 *     void*; NULL = the default stream).  All pointers are DEVICE pointers unless stated otherwise.
 *   - The caller owns every buffer (incl. workspaces); the library never allocates, frees or retains them.
 *   - Every call only enqueues work on `stream` and returns immediately (asynchronous, re-entrant).
 *   - Return value: 0 on success, negative on error (-1 bad argument / unsupported shape, -3 HIP launch
 *     error).  The message is available from vl_last_error() (thread-local).  Nothing throws or aborts.
 *   - "bf16" buffers hold raw bfloat16 bits (uint16).  A "(hi, lo) split" of an fp32 tensor x is the pair
 *     hi = bf16(x), lo = bf16(x - hi): x ~= hi + lo to 16 significant bits (Precision).
 *   - Row-major everywhere; `ld*` are leading dimensions in ELEMENTS.
 */
#ifndef VLHIP_H_
#define VLHIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

int vl_version(void);
const char* vl_last_error(void);

/* ------------------------------------------------------------------------------------------------------------
 * GEMM  C[M,N] = A[M,K] * B[N,K]^T  (+ epilogue), bf16 MFMA with fp32 accumulation.
 * Replaces: nn.Linear forward GEMMs encoders.py:229-246 (Q,K,V), :411-414 (attention out-proj), :496-501
 * (FFN1 + erf-GELU :131-137), :553-556 (FFN2), embeddings.py:660 (region projection), encoders.py:603-607
 * (pooler), :788-815 (classifier) and the autograd dX / dW products of the same layers.
 * passes = 1: bf16 operands (a_lo / b_lo ignored).  passes = 3: 3-term split product (fp32-grade).
 * K, lda, ldb multiples of 8; operand pointers 16-byte aligned; M, N arbitrary.
 * ------------------------------------------------------------------------------------------------------------ */
enum {
  VL_EPI_F32 = 0,        /* out32[m,n]  = acc + bias[n] (bias may be NULL) + resid32[m,n] (may be NULL)           */
  VL_EPI_GELU_SPLIT = 1, /* u = acc+bias; aux16 = bf16(gelu_erf'(u)); (out_hi,out_lo) = split(gelu_erf(u))          */
  VL_EPI_DGELU_BF16 = 2, /* out_hi = bf16(acc * aux16[m,n]), aux16 = the derivative saved by GELU_SPLIT (FFN1 backward) */
  VL_EPI_BF16 = 3,       /* out_hi = bf16(acc + bias)                                                             */
  VL_EPI_SPLIT = 4       /* (out_hi,out_lo) = split(acc + bias)                                                   */
};
int vl_gemm_nt(const void* a_hi, const void* a_lo, int64_t lda, const void* b_hi, const void* b_lo, int64_t ldb,
               int64_t M, int64_t N, int64_t K, int passes, int epilogue, const float* bias, const float* resid32,
               float* out32, int64_t ldc, void* out_hi, void* out_lo, void* aux16, int64_t ld16, void* stream);
/* The same with optional arguments in `extra`, a HOST array of VL_GX_FIELDS int64 (NULL = none; the library keeps no
 * tuning state):
 *  VL_GX_TILE  0 automatic (what vl_gemm_nt does) | 2 / 3 / 5 eight-wave ping-pong kernel with 256x256 / 256x192 / 224x256
 *              tiles | 4 ping-pong, width by cost model only | 6 single-barrier kernel | 7 generic 128x128 kernel |
 *              128 / 192 / 256 single-barrier kernel of that width | 8 small-M path.
 *  VL_GX_WS / VL_GX_WS_FLOATS  caller-owned fp32 workspace of the SMALL-M PATH -- products whose M is the batch (the pooled
 *              rows of the last layer, pooler, classifier: encoders.py:597-608, :788-815) would run on 4-16 of the 256 CUs
 *              with the big tiles, bound by memory latency; with >= vl_gemm_small_ws_floats(M, N, K) floats they run as
 *              64x64 tiles x K ranges (~500 workgroups, raw partial tiles into the workspace) + one launch that sums the
 *              ranges in a fixed order and applies the epilogue.  Automatic (tile 0) below 100 big tiles.
 *  VL_GX_IMG / VL_GX_IMG_COLS  (16-bit epilogues) additionally store out_hi as the K-MAJOR IMAGE the weight-gradient GEMM
 *              reads (vl_transpose_blocked's layout: img[((m >> 6) * IMG_COLS + n) * 64 + (m & 63)]): the producer writes
 *              it, no re-layout pass re-reads the row-major copy.  Rows [M, ceil64(M)) of the image are NOT written: use it
 *              when M is a multiple of 64, or zero them.
 *  VL_GX_COLSUM  (DGELU epilogue, ping-pong kernel only -- vl_gemm_nt_path() == 2) fp32 [rows, N] partial column sums of
 *              the rounded out_hi values, the bias-gradient partials the re-layout pass used to produce; on return
 *              VL_GX_COLSUM_ROWS holds the number of partial rows written (<= 8 * ceil(M / 256)): sum them with
 *              vl_colreduce_multi.
 *  VL_GX_PERSIST  > 0: the ping-pong kernel in its PERSISTENT form on that many workgroups (use the CU count, 256): a
 *              workgroup runs tiles w, w + G, ... and issues the operand DMA of its next tile before the epilogue of the
 *              finished one, so epilogue stores and first loads overlap the matrix pipe instead of arriving as one
 *              chip-wide burst per round.  Same tiles, same arithmetic: bit-identical results.  Ignored for products with
 *              <= G tiles, for the other kernels and together with VL_GX_IMG / VL_GX_COLSUM. */
enum { VL_GX_TILE = 0, VL_GX_WS = 1, VL_GX_WS_FLOATS = 2, VL_GX_IMG = 3, VL_GX_IMG_COLS = 4, VL_GX_COLSUM = 5,
       VL_GX_COLSUM_ROWS = 6, VL_GX_PERSIST = 7, VL_GX_FIELDS = 8 };
int vl_gemm_nt_ex(const void* a_hi, const void* a_lo, int64_t lda, const void* b_hi, const void* b_lo, int64_t ldb,
                  int64_t M, int64_t N, int64_t K, int passes, int epilogue, const float* bias, const float* resid32,
                  float* out32, int64_t ldc, void* out_hi, void* out_lo, void* aux16, int64_t ld16, int64_t* extra,
                  void* stream);
/* the kernel the automatic choice takes for a shape: 0 generic | 1 single-barrier | 2 ping-pong | 3 small-M */
int vl_gemm_nt_path(int64_t M, int64_t N, int64_t K, int passes, int has_ws);
int64_t vl_gemm_small_ws_floats(int64_t M, int64_t N, int64_t K);

/* Split-K form for the weight gradients dW[M,N] = A[M,K] * B[N,K]^T with K = B*S rows (bf16 single pass, fp32 out,
 * ld = N): `splits` K-ranges accumulate into fp32 slabs in `ws` (>= vl_gemm_splitk_ws_floats floats), then one
 * streaming pass sums them into out32 (deterministic, no atomics). */
int64_t vl_gemm_splitk_plan(int64_t M, int64_t N, int64_t K); /* recommended number of splits for this shape */
int64_t vl_gemm_splitk_ws_floats(int64_t M, int64_t N, int64_t splits);
int vl_gemm_nt_splitk(const void* a_hi, int64_t lda, const void* b_hi, int64_t ldb, int64_t M, int64_t N, int64_t K,
                      int64_t splits, float* ws, float* out32, void* stream);

/* Same product straight from row-major activations: out32[M,N] = A^T B with A [K,M], B [K,N] (K = batch rows), using
 * transposing LDS reads (ds_read_b64_tr_b16) instead of transposed HBM copies.  Returns -2 when the shape is outside
 * the fast path (K % 64, M >= 256, N >= 128, M/N multiples of 8): use vl_transpose_bf16 + vl_gemm_nt_splitk then. */
int vl_gemm_tn_splitk(const void* a, int64_t lda, const void* b, int64_t ldb, int64_t M, int64_t N, int64_t K,
                      int64_t splits, float* ws, float* out32, void* stream);
/* Weight gradients on K-major operands (csrc/dw.hip).  vl_transpose_blocked re-lays row-major bf16 activations
 * X [M, N] (ld) as XT[mb][n][mi] = X[64*mb + mi][n] (ceil(M/64) blocks of [N][64]; rows past M are zero; N a multiple of
 * 64): up to 8 matrices per launch; `tab` is a HOST array of n x VL_TR_FIELDS int64 {src, ld, N, dst, colsum_partial,
 * 0}; a non-zero colsum_partial ([ceil(M/64), N] floats) receives the per-block column sums of that matrix (the bias
 * gradient's partial sums, summed by vl_colsum_finalize into 1..4 equal destination segments, e.g. the query / key /
 * value bias gradients of the packed projection).  vl_blocked_elems(M, N) = elements of one blocked image.
 * vl_dw_grouped: out_p[M_p, N_p] (+)= A_p^T B_p (* mask_p) for up to 8 problems in ONE launch, A_p^T / B_p^T given as
 * blocked images (a_rows_total / b_rows_total = N of the image the pointer points into; the pointer may address a row
 * sub-range, e.g. the key rows of the packed [Q|K|V] gradient): an NT ping-pong product whose K loop runs over the
 * ceil(K/64) row blocks, accumulated in registers over the whole K range (no split-K slabs, no reduce pass).
 * `probs`: HOST array of nprob x VL_DW_FIELDS int64 {aT, a_rows_total, bT, b_rows_total, out, ldo, mask (0 = none; fp32,
 * layout of out), M, N, 0}.  Replaces autograd's grad_weight = grad_output.t() @ input of nn.Linear
 * (volta/encoders.py:229-246, 411-414, 496-501, 553-556) and, with `mask`, grad(weight_orig) = grad(weight) *
 * weight_mask of torch.nn.utils.prune under train_task_sft.py:128-132. */
#define VL_TR_FIELDS 6
#define VL_DW_FIELDS 10
int64_t vl_blocked_elems(int64_t M, int64_t N);
int vl_transpose_blocked(const int64_t* tab, int64_t n, int64_t M, int64_t max_blocks /* 0 = default (4096) */, void* stream);
int vl_colsum_finalize(const float* partial, int64_t nblk, int64_t N, float* const* outs, int64_t nout, int accumulate,
                       void* stream);
int vl_dw_grouped(const int64_t* probs, int64_t nprob, int64_t K, int accumulate, void* stream);
/* The same products straight from the ROW-MAJOR activations (no K-major images, no re-layout pass): problem fields
 * {dY [rows, lda] at the problem's first column, lda, X [rows, ldb], ldb, out, ldo, mask (0 = none), M, N,
 *  colsum partials (0 = none)}: out[M, N] (+)= dY[:, :M]^T . X[:, :N] (* mask).  MFMA fragments are gathered with
 * transposing LDS reads (ds_read_b64_tr_b16).  colsum partials: fp32 [ceil(N / 256), M] -- row t holds the column sums of
 * dY over the 64-row blocks kt = t (mod ceil(N / 256)); their sum (vl_colreduce_multi) is the bias gradient of the
 * Linear, produced by the pass that reads dY anyway.  Needs rows % 64 == 0, M, N, lda, ldb multiples of 8. */
int vl_dw_grouped_rowmajor(const int64_t* probs, int64_t nprob, int64_t rows, int accumulate, void* stream);
/* Per-operand layouts: mode bit 0 = dY row-major (fields 0 / 1 = pointer, lda), else its K-major image (image, image
 * columns); bit 1 = the same for X (fields 2 / 3).  mode 0 = vl_dw_grouped, 3 = vl_dw_grouped_rowmajor; field 9 (colsum
 * partials) works in every mode.  A transposing read moves half the bytes of a plain one per LDS instruction, so per
 * K-tile and wave the fragment reads are 24 (mode 0), 32 (mode 2), 40 (mode 1), 48 (mode 3) instructions. */
int vl_dw_grouped_mixed(const int64_t* probs, int64_t nprob, int64_t rows, int accumulate, int mode, void* stream);
/* Stream-K form of the same launch: a FIXED `budget` of workgroups (= CUs; <= 256) each walks an equal contiguous share of
 * the launch's {tile} x {64-row K-tile} iterations, so the weight gradients occupy `budget` CUs for the whole launch and
 * the kernels of the other stream keep the rest (autograd's grad_weight products, volta/encoders.py:229-246, 411-414,
 * 496-501, 553-556, run beside the dX products instead of fighting them for CUs).  A tile cut by a share boundary is
 * finished by the workgroup that starts it, from the partial tiles the others leave in `ws`
 * (vl_dw_streamk_ws_bytes(budget) bytes, caller-owned, ZERO-FILLED once; launches sharing it must be stream-ordered).
 * Deterministic for a given budget; differs from vl_dw_grouped_mixed by fp32 re-association only. */
int64_t vl_dw_streamk_ws_bytes(int64_t budget);
int vl_dw_grouped_streamk(const int64_t* probs, int64_t nprob, int64_t rows, int accumulate, int mode, int64_t budget,
                          void* ws, int64_t ws_bytes, void* stream);
/* Up to 8 column reductions in one launch: `tab` = HOST array of n x VL_CR_FIELDS int64 {src [nrows, ncols] fp32, nrows,
 * ncols, seg, out0, out1, out2, 0}: out_t[c] (+)= sum_rows src[row][t*seg + c] for the ncols / seg <= 3 segments (a
 * zero out_t skips a segment).  Deterministic.  Used per layer for the LayerNorm partials of vl_ln_bwd (dgamma, dbeta,
 * dbias) and the bias-gradient partials of vl_transpose_blocked. */
#define VL_CR_FIELDS 8
int vl_colreduce_multi(const int64_t* tab, int64_t n, int accumulate, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * V&L attention core over the single stream X = [text ; boxes]  (S = T + V <= 160, head dim 64 or 32).
 * Replaces encoders.py:255-341: four gated score blocks, two concatenated softmaxes, four dropouts, four P.V
 * products == one multi-head attention with the additive key mask [t_mask ; v_mask] (encoders.py:978-995).
 * addmask [B*S] fp32 additive key mask (0 or -10000; -inf for M3P).
 * ctx     (hi, lo) split [B*S, nh*dh] -- attention output in [b, s, h*dh+d] order (permute fused).
 * lse     [B*nh*S] fp32 row log-sum-exp of the masked, scaled scores (saved for backward).
 * ------------------------------------------------------------------------------------------------------------ */
/* On the bf16 matrix pipe, fed by the (hi, lo) split of Q/K/V that the QKV projection writes with VL_EPI_SPLIT (no fp32
 * copy of the projection goes through HBM):
 * qkv_hi / qkv_lo [B*S, 3*nh*64] bf16, columns [Q | K | V].  Forward: 3-term split products (fp32-grade, like the
 * projections around it); ctx (hi, lo) and lse as above.  Backward: single-pass bf16 on the hi halves (like every
 * other backward product): dctx16 bf16 -> dqkv16 bf16; delta = rowsum(P * dP) is computed in the kernel, so the saved
 * ctx is not read.  Dropout: counter-based, regenerated in backward from (seed).
 * nq = number of queries per sample that are live: S normally; nq < S ("only the pooled row of the last layer feeds
 * the head", encoders.py:597-608) computes / differentiates queries [0, nq) only, with ctx / dctx in the COMPACT layout
 * [B*nq, nh*64]; dqkv16 is always the full [B*S, 3*nh*64] matrix (dQ rows >= nq are written as zeros). */
int vl_attn2_fwd(const void* qkv_hi, const void* qkv_lo, const float* addmask, void* ctx_hi, void* ctx_lo, float* lse,
                 int64_t B, int64_t S, int64_t nh, int64_t dh, int64_t nq, float p_drop, uint64_t seed, void* stream);
int vl_attn2_bwd(const void* qkv_hi, const float* addmask, const void* dctx16, const float* lse, void* dqkv16,
                 int64_t B, int64_t S, int64_t nh, int64_t dh, int64_t nq, float p_drop, uint64_t seed, void* stream);

/* The op north_star names -- QKV projection + masked softmax(QK^T)V over the [text ; box] sequence (encoders.py:229-341) --
 * as one entry point per direction (SURVEY 8b vl_qkv_attention_{fwd,bwd}).  Forward = the 3-pass projection GEMM with the
 * (hi, lo) epilogue + vl_attn2_fwd, two launches sharing the split-bf16 Q|K|V (x (hi, lo) [B*S, H], wqkv (hi, lo) [3H, H],
 * bqkv [3H]; qkv_hi / qkv_lo [B*S, 3H] are outputs, kept for backward).  Backward w.r.t. the input = vl_attn2_bwd + the dX
 * GEMM dx32 = dqkv16 . wqkv_t (+ resid32; wqkv_t = transposed hi weights [H, 3H]).  Why two launches and not one kernel:
 * DESIGN.md "Fused V&L attention". */
int vl_qkv_attention_fwd(const void* x_hi, const void* x_lo, const void* wqkv_hi, const void* wqkv_lo, const float* bqkv,
                         const float* addmask, void* qkv_hi, void* qkv_lo, void* ctx_hi, void* ctx_lo, float* lse,
                         int64_t B, int64_t S, int64_t nh, int64_t dh, int64_t nq, float p_drop, uint64_t seed, void* stream);
int vl_qkv_attention_bwd(const void* qkv_hi, const float* addmask, const void* dctx16, const float* lse, const void* wqkv_t,
                         const float* resid32, void* dqkv16, float* dx32, int64_t B, int64_t S, int64_t nh, int64_t dh,
                         int64_t nq, float p_drop, uint64_t seed, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * (dropout +) residual + LayerNorm, forward and backward.
 * Replaces apex fused_layer_norm_cuda.forward_affine / backward_affine (apex/csrc/layer_norm_cuda.cpp:139-239,
 * kernels layer_norm_cuda_kernel.cu:279-322, :403-637) == BertLayerNorm (encoders.py:44-62), fused with the
 * surrounding eager ops of BertGatedSelfOutput / BertGatedOutput (encoders.py:411-425, :553-567) and of
 * UC2Embeddings (embeddings.py:653-666):
 *     z   = (dropout_pre(y) + resid + addvec[r % addvec_rows]) * row_pre[r]     (each term optional; z overwrites y)
 *     out = dropout_post(gamma * (z - mean) * rsqrt(var + eps) + beta) * row_post[r]      biased var, eps in sqrt
 * addvec is an [addvec_rows, H] table (1 row = a broadcast vector: UC2's type embedding; V or T rows = M3P's
 * position embeddings over the [image ; text] stream, m3p_transformer.py:929-933); row_pre / row_post are the [M]
 * multiplicative length masks of M3P (`tensor *= mask`, m3p_transformer.py:937, :955).
 * Output row r of the M input rows goes to row (r / group)*out_stride + out_off + (r % group) of out32/out_hi/
 * out_lo (group == M, out_stride == 0, out_off == 0 for the identity map) -- this is how the text rows and box
 * rows are interleaved into the single [B, S, H] stream.  H must be a multiple of 256, H <= 2048.
 * mean, rstd: [M] fp32 saved for backward.  out_hi/out_lo may be NULL.
 * ------------------------------------------------------------------------------------------------------------ */
int vl_ln_fwd(float* y32_z32, const float* resid32, const float* addvec, int64_t addvec_rows, const float* row_pre,
              const float* row_post, const float* gamma, const float* beta, float eps, float* out32, void* out_hi, void* out_lo, float* mean, float* rstd, int64_t M, int64_t H,
              int64_t group, int64_t out_stride, int64_t out_off, float p_pre, float p_post, uint64_t seed,
              int64_t orig_row_stride, int64_t resid_row_stride, void* stream);
/* The same with the residual RECOMPUTED instead of read: when the residual is the output of an earlier LayerNorm call of this
 * library (the transformer's post-LN chain: every residual is), resid_ln = {z32, mean, rstd, gamma, beta, row_post or 0} of that
 * call (six pointers as int64) lets this kernel re-evaluate gamma * ((z - mean) * rstd) + beta (* row_post) -- the expression that
 * call stored -- from buffers backward keeps anyway, so that call need not store its fp32 output (out32 = NULL there: 44 MB per
 * launch at c2).  Row r reads row r * resid_row_stride of z32 / mean / rstd / row_post.  That call must have run with
 * p_post = 0 and the identity row map.  resid32 and resid_ln exclude each other; resid_ln = NULL is vl_ln_fwd. */
int vl_ln_fwd_rr(float* y32_z32, const float* resid32, const int64_t* resid_ln, const float* addvec, int64_t addvec_rows,
                 const float* row_pre, const float* row_post, const float* gamma, const float* beta, float eps, float* out32,
                 void* out_hi, void* out_lo, float* mean, float* rstd, int64_t M, int64_t H, int64_t group, int64_t out_stride,
                 int64_t out_off, float p_pre, float p_post, uint64_t seed, int64_t orig_row_stride, int64_t resid_row_stride,
                 void* stream);
/* Compact-row calls (orig_row_stride > 1): only a subset of the rows of a larger [M_full, H] problem is live -- the pooled
 * row of every sample in the last layer (encoders.py:597-608 reads hidden_states[:, 0] only).  Row r of this call is row
 * r * orig_row_stride of the full problem: the dropout counter and row_pre / row_post are indexed by that ORIGINAL row, so
 * the result is bit-identical to the corresponding rows of the dense call; resid32 is read at row r * resid_row_stride.
 * Pass 1, 1 for ordinary calls.
 * Backward: dy32 is read through the same row map; dz32 [M,H] = dL/dz (what flows to the residual branch);
 * dpre16 (bf16, may be NULL) / dpre32 (fp32, may be NULL) = dL/dy = dz * keep_pre  (what flows into the producing
 * GEMM); column sums over the M rows: dgamma, dbeta, dbias (= colsum(dL/dy), the producing dense layer's bias
 * gradient; may be NULL).  partial_ws: >= vl_ln_bwd_ws_floats(M, H) floats of scratch. */
int64_t vl_ln_bwd_ws_floats(int64_t M, int64_t H);
int vl_ln_bwd(const float* dy32, const float* z32, const float* mean, const float* rstd, const float* gamma,
              const float* row_pre, const float* row_post, float* dz32, void* dpre16, float* dpre32, float* dgamma, float* dbeta, float* dbias,
              float* partial_ws, int64_t M, int64_t H, int64_t group, int64_t out_stride, int64_t out_off,
              float p_pre, float p_post, uint64_t seed, int64_t orig_row_stride, void* stream);
/* dgamma = dbeta = dbias = NULL makes vl_ln_bwd stop after the per-workgroup partials; this sums them later, on any
 * stream ordered after that call (the engine uses the weight-gradient stream: off the backward critical path). */
int vl_ln_bwd_reduce(const float* partial_ws, int64_t M, int64_t H, float* dgamma, float* dbeta, float* dbias,
                     void* stream);
/* two independent reductions in one launch (the two LayerNorms of a transformer layer) */
int vl_ln_bwd_reduce2(const float* ws_a, int64_t M_a, float* dgamma_a, float* dbeta_a, float* dbias_a,
                      const float* ws_b, int64_t M_b, float* dgamma_b, float* dbeta_b, float* dbias_b, int64_t H,
                      int accumulate, void* stream); /* accumulate != 0: add to the destinations */

/* ------------------------------------------------------------------------------------------------------------
 * The transformer trunk as ONE call per direction (csrc/stack.hip).  Replaces the Python loop of BertEncoder.forward
 * (volta/volta/encoders.py:848-892) over 24 sub-layers and autograd's replay of it: the host fills a descriptor (a HOST
 * array of int64: VL_ST_FIELDS header values, then VL_LY_FIELDS values per layer; pointers are device pointers unless
 * stated) once per (batch shape, parameter placement) and re-uses it every step.
 * Header: dims; eps / p_hid / p_att as float BITS; seed0 (site s of layer l draws from seed0*4096 + 16 l + s); addmask
 * [B*S]; row_post [B*S] or 0 (M3P `tensor *= mask`); EV_FORK = a hipEvent_t of the caller (stream fork, backward);
 * ACCUMULATE != 0 adds the gradients to their destinations; T_* = blocked K-major images (vl_blocked_elems(B*S, N)
 * bf16 each; the X-side fields T_X / T_CTX / T_X1 / T_H of the HEADER are unused since the images became per-layer)
 * and CS_* = column-sum partials ([ceil(B*S/64), N] fp32), scratch of the weight-gradient stream.
 * Layer: X32 / X_HI / X_LO = the layer's input (fp32 stream + its split), OUT* = its output (= the next layer's input);
 * prepared weights W*_HI / W*_LO [N,K], W*_T = transposed hi [K,N], biases, LayerNorm parameters; the activations saved
 * for backward (QKV_HI/LO [B*S,3H], CTX_HI/LO, LSE, Z1, MEAN1, RSTD1, X1_*, U16, H_HI/LO, Z2, MEAN2, RSTD2); backward
 * buffers (DY in, DX out, DZ2, DT2, DU16, DX1, DZ1, DT1, DCTX16, DQKV, LayerNorm partial workspaces LNWS1/2 of
 * vl_ln_bwd_ws_floats(B*S, H) floats each -- DT2 / DU16 / DT1 / DQKV / LNWS* are read by the side stream and must be
 * private to the layer); GRAD0..GRAD0+15 = gradient destinations in the order {Wq, bq, Wk, bk, Wv, bv, Wo, bo, ln1.g,
 * ln1.b, W1, b1, W2, b2, ln2.g, ln2.b}; MASK0..MASK0+5 = SFT masks of {Wq, Wk, Wv, Wo, W1, W2} (fp32, 0 = dense).
 * vl_stack_fwd runs layers [layer_begin, layer_end) on stream_main; when the layer records carry T_X / T_CTX / T_X1 /
 * T_H (training), the K-major images of the layer's X operands {layer input, attention context, LayerNorm-1 output,
 * GELU output} are written behind each layer on stream_side (HBM-bound work under the MFMA-bound forward GEMMs), so
 * that backward only re-lays the four dY operands.  vl_stack_bwd runs layers [layer_lo, layer_hi) in
 * descending order: the critical path on stream_main, the optimizer-only work (K-major re-layout, column sums, grouped
 * weight-gradient GEMM) forked per layer onto stream_side (NULL = everything on stream_main); the caller joins them.
 * ------------------------------------------------------------------------------------------------------------ */
#define VL_ST_MAGIC_VALUE 0x564c5354414b34ll
enum {
  VL_ST_MAGIC = 0,
  VL_ST_B = 1,
  VL_ST_S = 2,
  VL_ST_H = 3,
  VL_ST_I = 4,
  VL_ST_NH = 5,
  VL_ST_NLAYERS = 6,
  VL_ST_EPS = 7,
  VL_ST_P_HID = 8,
  VL_ST_P_ATT = 9,
  VL_ST_SEED0 = 10,
  VL_ST_ADDMASK = 11,
  VL_ST_ROW_POST = 12,
  VL_ST_EV_FORK = 13,
  VL_ST_ACCUMULATE = 14,
  VL_ST_T_DQKV = 15,
  VL_ST_T_DT1 = 16,
  VL_ST_T_DU = 17,
  VL_ST_T_DT2 = 18,
  VL_ST_T_X = 19,
  VL_ST_T_CTX = 20,
  VL_ST_T_X1 = 21,
  VL_ST_T_H = 22,
  VL_ST_CS_QKV = 23,
  VL_ST_CS_U = 24,
  VL_ST_PROF = 25, /* HOST pointer to a VlProf block (0 = no timing), see below */
  VL_ST_POOLED_ONLY = 26, /* != 0: only row 0 of every sample of the LAST layer's output is live (see below) */
  VL_ST_ROWS0 = 27, /* int64 [B] device array {0, S, 2S, ...}: the live rows (pooled-row mode) */
  VL_ST_TR_BLOCKS_FWD = 28, /* workgroup caps of the K-major re-layout launches (0 = default) */
  VL_ST_TR_BLOCKS_BWD = 29,
  VL_ST_SMALL_WS = 30, /* fp32 workspace of the small-M GEMM path (vl_gemm_nt_ex) for the B-row products of the pooled-row mode; 0 = none */
  VL_ST_SMALL_WS_FLOATS = 31,
  VL_ST_TR_BWD_LAYERS = 32, /* K-major X images: those of the bottom n layers are written in backward (side stream, ahead of the
                               layer's own re-layout), the others at the end of forward (under the task head); 0 = all in forward */
  VL_ST_FUSE_IMAGES = 33, /* bit 0: the FFN1 epilogue writes the K-major image of the GELU output (VL_LY_T_H); bit 1: the GELU'
                             epilogue of FFN1's backward writes du's image + column sums (VL_LY_T_DU / VL_LY_CS_DU) */
  VL_ST_DW_ROWMAJOR = 34, /* operand layouts of the weight-gradient GEMM (mode of vl_dw_grouped_mixed; bit 0: dY row-major,
                             bit 1: X row-major -- no K-major image / re-layout pass for that side), used whenever B*S (and B
                             in the pooled-row mode) is a multiple of 64; 0 = both sides through the re-layout pass */
  VL_ST_DX_TILE = 35, /* tile selection (VL_GX_TILE) of the single-pass products of backward: low byte for N <= 1024, next byte
                         for the wider ones; 0 = automatic */
  VL_ST_DW_BUDGET = 36, /* > 0: the weight-gradient GEMMs run in the stream-K form on this many workgroups (vl_dw_grouped_streamk) */
  VL_ST_DW_SK_WS = 37,  /* its workspace (device pointer, zero-filled once) and size in bytes */
  VL_ST_DW_SK_WS_BYTES = 38,
  VL_ST_GEMM_PERSIST = 39, /* VL_GX_PERSIST of the stack's GEMMs: low 16 bits for the 3-pass products of forward, next 16 for the
                              single-pass products of backward (0 = one workgroup per tile) */
  VL_ST_DW_TAIL_BUDGET = 40, /* > 0: the weight-gradient GEMM of the LAST layer vl_stack_bwd runs (layer 0: nothing is left on the
                                main stream to share the chip with) takes the stream-K form on this many workgroups (needs
                                VL_ST_DW_SK_WS sized for it) */
  VL_ST_FIELDS = 48
};
enum {
  VL_LY_X32 = 0,
  VL_LY_X_HI = 1,
  VL_LY_X_LO = 2,
  VL_LY_WQKV_HI = 3,
  VL_LY_WQKV_LO = 4,
  VL_LY_WQKV_T = 5,
  VL_LY_BQKV = 6,
  VL_LY_WO_HI = 7,
  VL_LY_WO_LO = 8,
  VL_LY_WO_T = 9,
  VL_LY_BO = 10,
  VL_LY_W1_HI = 11,
  VL_LY_W1_LO = 12,
  VL_LY_W1_T = 13,
  VL_LY_B1 = 14,
  VL_LY_W2_HI = 15,
  VL_LY_W2_LO = 16,
  VL_LY_W2_T = 17,
  VL_LY_B2 = 18,
  VL_LY_LN1_G = 19,
  VL_LY_LN1_B = 20,
  VL_LY_LN2_G = 21,
  VL_LY_LN2_B = 22,
  VL_LY_QKV_HI = 23,
  VL_LY_QKV_LO = 24,
  VL_LY_CTX_HI = 25,
  VL_LY_CTX_LO = 26,
  VL_LY_LSE = 27,
  VL_LY_Z1 = 28,
  VL_LY_MEAN1 = 29,
  VL_LY_RSTD1 = 30,
  VL_LY_X1_32 = 31,
  VL_LY_X1_HI = 32,
  VL_LY_X1_LO = 33,
  VL_LY_U16 = 34,
  VL_LY_H_HI = 35,
  VL_LY_H_LO = 36,
  VL_LY_Z2 = 37,
  VL_LY_MEAN2 = 38,
  VL_LY_RSTD2 = 39,
  VL_LY_OUT32 = 40,
  VL_LY_OUT_HI = 41,
  VL_LY_OUT_LO = 42,
  VL_LY_DY = 43,
  VL_LY_DX = 44,
  VL_LY_DZ2 = 45,
  VL_LY_DT2 = 46,
  VL_LY_DU16 = 47,
  VL_LY_DX1 = 48,
  VL_LY_DZ1 = 49,
  VL_LY_DT1 = 50,
  VL_LY_DCTX16 = 51,
  VL_LY_DQKV = 52,
  VL_LY_LNWS1 = 53,
  VL_LY_LNWS2 = 54,
  VL_LY_GRAD0 = 55, /* 16 values */
  VL_LY_MASK0 = 71, /* 6 values */
  VL_LY_T_X = 77, /* per-layer K-major images of the X operands, written during FORWARD (0 = inference) */
  VL_LY_T_CTX = 78,
  VL_LY_T_X1 = 79,
  VL_LY_T_H = 80,
  VL_LY_T_DU = 81, /* per-layer K-major image of du16 + its column-sum partials [8 * ceil(M / 256), I]: written by the GELU'
                      epilogue of the FFN1-backward GEMM on the MAIN stream while the side stream may still read the layer
                      above's (0 = use the shared VL_ST_T_DU / VL_ST_CS_U through the re-layout pass) */
  VL_LY_CS_DU = 82,
  VL_LY_EV_READY = 83, /* a hipEvent_t of the caller or 0: vl_stack_fwd makes its stream wait for it before the layer's first
                          kernel (the layer's parameters / prepared weights are being written on another stream: the optimizer
                          update of the previous step running under this forward) */
  VL_LY_FIELDS = 96
};
/* Pooled-row mode (VL_ST_POOLED_ONLY): the head reads hidden_states[:, 0] only (BertTextPooler, encoders.py:597-608; M3P
 * BertPooler, m3p_transformer.py:548-560), so in the last layer every row but one per sample is dead after the K/V
 * projection.  The last layer then runs attention for query 0 only and everything after it on the B live rows in COMPACT
 * [B, .] buffers (the first B rows of the layer's buffers): OUT32 is [B, H], and in backward DY is the [B, H] gradient of
 * those rows.  Dropout counters and row masks are those of the original rows, so the live rows are bit-identical to the
 * dense run; DX is the full [B*S, H] gradient.
 * Optional launch timing (benchmarks: roofline numbers measured live, on the stream the kernel runs on).  VL_ST_PROF
 * points to a HOST int64 block owned by the caller: [0] stride (every stride-th GEMM launch is bracketed), [1] capacity
 * (event pairs), [2] launch counter, [3] pairs used, then per pair 4 values {event0, event1 (hipEvent_t handles created
 * by the caller with timing enabled), tag, flops}: the library records the events around the launch and fills tag
 * (= passes * 16 + epilogue for a GEMM; VL_PROF_TAG_QKV_ATTN for the fused attention op) and the algorithmic flops; the
 * caller reads the elapsed times. */
#define VL_PROF_HEADER 4
#define VL_PROF_PAIR 4
#define VL_PROF_TAG_QKV_ATTN 1000 /* the fused V&L attention op (projection + core), forward */
int64_t vl_stack_desc_len(int64_t n_layers);
int vl_stack_fwd(const int64_t* desc, int64_t layer_begin, int64_t layer_end, void* stream_main, void* stream_side);
int vl_stack_bwd(const int64_t* desc, int64_t layer_hi, int64_t layer_lo, void* stream_main, void* stream_side);

/* ------------------------------------------------------------------------------------------------------------
 * Sparse fine-tuning mask kernels.
 * vl_mask_mul: out = a (*) m -- torch.nn.utils.prune forward pre-hook weight = weight_orig * weight_mask and its
 * autograd grad(weight_orig) = grad(weight) * mask (train_task_sft.py:128-132; torch prune.py:20-31).
 * vl_weight_prep: one pass per optimizer step over a Linear weight W[N,K] (optionally (*) mask): writes the
 * (hi, lo) split for forward GEMMs and the transposed hi copy Wt[K,N] for the dX GEMM.  Any of the outputs may
 * be NULL.
 * ------------------------------------------------------------------------------------------------------------ */
int vl_mask_mul(const float* a, const float* m, float* out, int64_t n, void* stream);
int vl_weight_prep(const float* w32, const float* mask32, void* w_hi, void* w_lo, void* wt_hi, int64_t N,
                   int64_t K, int64_t ldw, int64_t ldt, void* stream);

/* One round of global magnitude pruning (IMP): new_mask = mask with the k smallest |w| among entries with mask == 1
 * set to 0 (w, mask, new_mask: n fp32, the flat concatenation of the prunable weights in named_modules() order;
 * mask holds {0,1}).  Replaces prune.global_unstructured(L1Unstructured, amount) as called at
 * train_task_prunning.py:80-84 (k = round(amount * n_remaining), torch prune.py:514-534).  Exact radix select on the
 * fp32 bits; threshold ties are pruned lowest-flat-index-first.  ws >= vl_imp_ws_bytes(n) bytes.  No host sync. */
int64_t vl_imp_ws_bytes(int64_t n);
int vl_imp_select(const float* w, const float* mask, float* new_mask, int64_t n, int64_t k, void* ws, void* stream);

/* Batched form: ONE launch for every Linear weight of the model.  table_dev: device array of ndesc x 10 int64
 * [w32, mask32 (0 = none), w_hi, w_lo, wt_hi (0 = none), N, K, ldw, ldt, first_tile] with first_tile = running sum of
 * ceil(N/64)*ceil(K/64); total_tiles = the grand total. */
int vl_weight_prep_multi(const int64_t* table_dev, int64_t ndesc, int64_t total_tiles, void* stream);

/* Elementwise / layout helpers. */
int vl_memset_zero(void* p, int64_t bytes, void* stream); /* zero-fill on the caller's stream */
int vl_split_f32(const float* x32, void* hi, void* lo, int64_t n, void* stream); /* lo may be NULL (plain cast) */
int vl_transpose_bf16(const void* in, void* out, int64_t M, int64_t N, int64_t ld_in, int64_t ld_out, void* stream);
/* out32[n] = sum_m x16[m,n]; ws >= vl_colsum_ws_floats(M,N) floats. */
int64_t vl_colsum_ws_floats(int64_t M, int64_t N);
int vl_colsum_bf16(const void* x16, int64_t M, int64_t N, int64_t ld, float* ws, float* out32, void* stream);
int vl_addmask(const int64_t* text_mask, const int64_t* img_mask, float* addmask, int64_t B, int64_t T, int64_t V,
               void* stream); /* (1 - m) * -10000 over [text ; boxes], encoders.py:978-995 */

/* ------------------------------------------------------------------------------------------------------------
 * UC2Embeddings pieces (embeddings.py:636-669).
 * text: z[b*T+t,:] = word[ids] + pos[cumsum(ids!=pad)*(ids!=pad)+pad] + type[seg]   (RoBERTa position ids,
 *       embeddings.py:157-170); backward scatter-adds dz into the dense tables with float atomics.
 *       dword may be NULL (the word-table scatter is then done elsewhere, e.g. after a sparse multi-GPU exchange).
 * loc : y[r,:] = loc[r,0:L] . Wl[:,0:L]^T + bl  (L = num_locs <= 8) and its backward.
 * ------------------------------------------------------------------------------------------------------------ */
int vl_embed_text_fwd(const int64_t* ids, const int64_t* seg, const float* word, const float* pos,
                      const float* type, float* z32, int64_t B, int64_t T, int64_t H, int64_t pad_id, void* stream);
int vl_embed_text_bwd(const int64_t* ids, const int64_t* seg, const float* dz32, float* dword, float* dpos,
                      float* dtype, int64_t B, int64_t T, int64_t H, int64_t pad_id, uint8_t* row_flags, void* stream);
/* plain row gather out[r,:] = table[ids[r],:] and its scatter-add (atomics; rows with ids[r] == pad_id skipped; pass
 * pad_id = -1 for none) -- M3P's text embedding `self.embeddings(x)` (m3p_transformer.py:908). */
int vl_embed_gather_fwd(const int64_t* ids, const float* table, float* out32, int64_t R, int64_t H, void* stream);
int vl_embed_scatter_add(const int64_t* ids, const float* dz32, float* dtable, int64_t R, int64_t H, int64_t pad_id,
                         uint8_t* row_flags, void* stream);
/* row_flags (optional, [table rows] bytes): set to 1 for every row that receives a gradient; vl_adamw uses it to
 * skip the optimizer state of embedding rows that were never touched (exactly: their update is p *= 1 - lr*wd). */
int vl_loc_linear_fwd(const float* loc, const float* w, const float* b, float* y32, int64_t R, int64_t L, int64_t H,
                      void* stream);
/* dw [H,L], db [H] are ADDED to: zero them first.  ws (may be NULL): vl_loc_bwd_ws_floats(R, H) floats -- with it the row
 * blocks are summed in a fixed order by a second launch (bit-reproducible); without it float atomics add them in arrival
 * order. */
int64_t vl_loc_bwd_ws_floats(int64_t R, int64_t H);
int vl_loc_linear_bwd(const float* loc, const float* dy32, float* dw, float* db, int64_t R, int64_t L, int64_t H,
                      float* ws, void* stream);
/* DETERMINISTIC scatter-add of gradient rows into up to 4 embedding tables that all take their rows from dz32 [R, H]
 * (csrc/scatter.hip): the backward of nn.Embedding(sparse=False) (word / position / token-type tables of UC2Embeddings,
 * volta/volta/embeddings.py:617-655; M3P's self.embeddings(x), m3p_transformer.py:908) with a FIXED summation order -- sort
 * of (row id, source index) pairs, run sums in source order, one owner per table row, no atomics -- where torch's backward
 * and vl_embed_text_bwd / vl_embed_scatter_add add in arrival order.  `tab`: HOST array of n x VL_SC_FIELDS int64 {ids
 * (device int64 [R]), kind, table (fp32 [rows, H], ADDED to), skip, row_flags (device uint8 per table row, or 0), T}:
 * kind 0 -- row = ids[r], rows with ids[r] == skip receive nothing (skip = -1: none); kind 1 -- row = the RoBERTa position
 * id of token r of `ids` viewed as [R / T, T] with pad id `skip` (embeddings.py:157-170).  R <= 16384, H a multiple of 64.
 * ws: vl_scatter_det_ws_bytes(n, R, H) bytes, 256-byte aligned, caller-owned. */
#define VL_SC_FIELDS 6
int64_t vl_scatter_det_ws_bytes(int64_t n, int64_t R, int64_t H);
int vl_scatter_add_det(const int64_t* tab, int64_t n, const float* dz32, int64_t R, int64_t H, void* ws, int64_t ws_bytes,
                       void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Element-wise stages of the pooler / classifier head on [M = batch, N] fp32 tensors (contiguous, ld = N).
 * Replaces: BertTextPooler's ReLU (volta/volta/encoders.py:597-608), M3P BertPooler's tanh
 * (volta/volta/m3p/m3p_transformer.py:548-560), nn.Dropout on the pooled vector (encoders.py:1238-1239) and
 * SimpleClassifier's GeLU (encoders.py:788-815, :131-137), each with autograd's replay.
 *   vl_act_fwd: y = dropout_p(act(z)) -> out32 [M,N] (may be NULL) and / or its (hi, lo) bf16 split (out_lo may be NULL),
 *               leading dimension ld16 >= N, columns [N, ld16) written as zeros (the operand form of the next GEMM).
 *   vl_act_bwd: dz = dy * dropout_mask_p * act'(z) -> dz32 (may be NULL) and / or bf16 dz16 (ld16, zero pad columns):
 *               the operand of the next dX GEMM and of the weight-gradient path; act = VL_ACT_NONE is the plain
 *               cast + pad (z32 may then be NULL).
 * Dropout masks are counter-based (seed, element index m*N + n), regenerated in backward.
 * ------------------------------------------------------------------------------------------------------------ */
enum { VL_ACT_NONE = 0, VL_ACT_RELU = 1, VL_ACT_TANH = 2, VL_ACT_GELU = 3 };
int vl_act_fwd(const float* z32, int64_t M, int64_t N, int act, float p_drop, uint64_t seed, float* out32, void* out_hi,
               void* out_lo, int64_t ld16, void* stream);
int vl_act_bwd(const float* dy32, const float* z32, int64_t M, int64_t N, int act, float p_drop, uint64_t seed, float* dz32,
               void* dz16, int64_t ld16, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * GQA loss with semantic prior, its score and d(loss)/d(logits) in one launch (csrc/loss.hip).
 * Replaces the ~45 eager kernels of task_utils.py:413-428 + compute_score_with_logits (:706-711):
 *   p = softmax(logits); (p10, idx) = topk(p, 10); prior = mean_b sum_k p10 * distances[b, idx];
 *   loss = CE(logits, argmax(target.long())) * C + semantic_lambda * prior * C;  score = sum_b target[b, argmax logits_b] / B.
 * logits / target / distances / dlogits: [B, C] fp32 (C <= 4096); loss_score: 2 floats {loss, score}; ws: >=
 * vl_gqa_loss_ws_bytes(B) bytes of scratch.
 * dlogits = d(loss)/d(logits) for upstream gradient 1 (top-k indices are constants for autograd, as in torch).
 * ------------------------------------------------------------------------------------------------------------ */
int64_t vl_gqa_loss_ws_bytes(int64_t B);
int vl_gqa_loss(const float* logits, const float* target, const float* distances, int64_t B, int64_t C,
                float semantic_lambda, float* loss_score, float* dlogits, void* ws, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Fused optimizer step over a flat parameter arena (next-row f1 of SURVEY.md section 8):
 * AdamW as pytorch_transformers.optimization.AdamW (call site train_task.py:264-268): bias-corrected step size,
 * decoupled weight decay applied after the Adam update; grads pre-scaled by *grad_scale_dev (device scalar, e.g. the
 * clip coefficient of train_task.py:330) when non-NULL, else by `grad_scale`.  seg_end[i] = exclusive end offset of
 * segment i in the arena, seg_lr / seg_wd its base lr and weight decay (215 one-tensor param groups,
 * train_task.py:249-260); `lr_mult` = the LR schedule's multiplier for this step (WarmupLinearSchedule,
 * train_task.py:274).  `step` = 1-based optimizer step.  Also zeroes the gradient when zero_grad != 0.
 * sumsq_dev (may be NULL): device scalar sum(g^2) (vl_sumsq): the kernel then computes the clip_grad_norm_ coefficient
 * itself, grad scale = min(1, max_norm / (sqrt(sumsq) * post + 1e-6)) * post (post = 1/world_size of the all-reduce),
 * overriding grad_scale_dev / grad_scale; sumsq_next (may be NULL) = ANOTHER device scalar that is set to 0 (the next
 * step's accumulator: two alternate, so no separate fill launch is needed).
 * row_flags (may be NULL): per-row "has ever received a gradient" bytes for the table occupying arena elements
 * [flag_begin, flag_begin + flag_rows*flag_row_len): rows with flag 0 have g = m = v = 0, so only p *= (1 - lr*wd)
 * is applied (8 B/param of traffic instead of 32; bit-identical to the dense update).
 * A NEGATIVE seg_lr marks a segment whose parameter received no gradient: it is skipped entirely (no moment decay, no
 * weight decay, no memory traffic), like `if p.grad is None: continue` in pytorch_transformers.AdamW.
 * seg_step (may be NULL): DEVICE array of nseg 1-based per-segment step counts for the bias correction -- the reference
 * optimizer keeps state['step'] per parameter and advances it only on steps where that parameter has a gradient; NULL =
 * every segment is at `step`.
 * ------------------------------------------------------------------------------------------------------------ */
int vl_adamw(float* param, float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, const int64_t* seg_end,
             const float* seg_lr, const float* seg_wd, int64_t nseg, float beta1, float beta2, float eps,
             int64_t step, const int64_t* seg_step, int correct_bias, float lr_mult, const float* grad_scale_dev, float grad_scale,
             const float* sumsq_dev, float max_norm, float post, float* sumsq_next,
             int zero_grad, const uint8_t* row_flags, int64_t flag_begin, int64_t flag_rows, int64_t flag_row_len,
             void* stream);
/* out[0] += sum(x^2) over n floats (zero out[0] first).  ws (may be NULL): vl_sumsq_ws_floats() floats -- with it the
 * workgroups' partial sums are added in a fixed order by a second launch (bit-reproducible clip coefficient); without it
 * they are added by float atomics in arrival order. */
int64_t vl_sumsq_ws_floats(void);
int vl_sumsq(const float* x, int64_t n, float* out, float* ws, void* stream);
/* same, not reading the rows of the flagged table range (vl_adamw's row_flags arguments) whose flag is 0: they never
 * received a gradient and hold exact zeros */
int vl_sumsq_flagged(const float* x, int64_t n, float* out, const uint8_t* row_flags, int64_t flag_begin,
                     int64_t flag_rows, int64_t flag_row_len, float* ws, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VLHIP_H_ */
