// vl_stack_fwd / vl_stack_bwd: the layer sequencing of the transformer trunk in native code.
//
// Reference: BertEncoder.forward (volta/volta/encoders.py:848-892) walks 24 sub-layers from Python, and autograd
// replays them backwards; every op is a separate eager dispatch.  Here one C call enqueues the whole stack -- per
// layer { QKV projection -> attention -> out-projection -> dropout+residual+LayerNorm -> FFN1 (+GELU) -> FFN2 ->
// dropout+residual+LayerNorm [-> * row mask] } -- from a descriptor the host builds ONCE per (batch shape, parameter
// placement): no interpreter, allocator or binding cost per kernel (the Python driver needed ~10 ctypes calls and ~12
// allocations per layer and direction, 6-13 ms of host time per step).
//
// Backward: the critical path (LayerNorm / dX GEMMs / attention) runs on `stream_main`; everything that only feeds the
// optimizer -- the K-major re-layout of the operands, the bias / LayerNorm column sums and the grouped weight-gradient
// GEMM (csrc/dw.hip) -- is forked per layer onto `stream_side`, where it fills the CUs the critical path leaves idle.
// The caller joins the streams (after the embedding backward).  Gradients are written straight to the 16 destinations
// the descriptor names (the optimizer's flat arena), optionally accumulating.
//
// The library keeps no state: every buffer, both streams and the fork event belong to the caller.
#include <string.h>

#include "common.h"
#include "../../include/vlhip.h"

namespace {

inline float f_of(int64_t bits) {
  const uint32_t u = (uint32_t)bits;
  float f;
  memcpy(&f, &u, 4);
  return f;
}
template <class T> inline T* ptr(int64_t v) { return reinterpret_cast<T*>(static_cast<uintptr_t>(v)); }
inline uint64_t seed_of(int64_t seed0, int site) { return ((uint64_t)seed0 * 4096ull + (uint64_t)site); }

#define VL_TRY(call) do { if (int rc_ = (call)) return rc_; } while (0)

// caller-owned event pairs (descriptor field VL_ST_PROF): every stride-th timed launch is bracketed
int64_t* prof_begin(int64_t* prof, void* stream) {
  if (!prof) return nullptr;
  const int64_t n = prof[2]++;
  if (!(prof[0] > 0 && n % prof[0] == 0 && prof[3] < prof[1])) return nullptr;
  int64_t* pair = prof + VL_PROF_HEADER + VL_PROF_PAIR * prof[3]++;
  (void)hipEventRecord(reinterpret_cast<hipEvent_t>(static_cast<uintptr_t>(pair[0])), (hipStream_t)stream);
  return pair;
}
void prof_end(int64_t* pair, int64_t tag, int64_t flops, void* stream) {
  if (!pair) return;
  (void)hipEventRecord(reinterpret_cast<hipEvent_t>(static_cast<uintptr_t>(pair[1])), (hipStream_t)stream);
  pair[2] = tag;
  pair[3] = flops;
}

// vl_gemm_nt, optionally bracketed by a caller-owned event pair (descriptor field VL_ST_PROF); B-row products of the
// pooled-row mode take the small-M path through the descriptor's workspace (VL_ST_SMALL_WS)
struct GemmCtx { int64_t* prof; float* ws; int64_t ws_floats; int64_t tile1; int64_t persist; };
struct GemmImage { void* img; int64_t cols; float* colsum; int64_t* colsum_rows; };
int gemm(const GemmCtx& g, const void* a_hi, const void* a_lo, int64_t lda, const void* b_hi, const void* b_lo, int64_t ldb,
         int64_t M, int64_t N, int64_t K, int passes, int epi, const float* bias, const float* resid, float* out32,
         int64_t ldc, void* out_hi, void* out_lo, void* aux16, int64_t ld16, void* stream, const GemmImage* im = nullptr) {
  int64_t* pair = prof_begin(g.prof, stream);
  int64_t extra[VL_GX_FIELDS] = {0};
  extra[VL_GX_WS] = (int64_t)(uintptr_t)g.ws; extra[VL_GX_WS_FLOATS] = g.ws_floats;
  // (VL_ST_DX_TILE, A/B knob: low byte = tile of the narrow single-pass products, next byte = of the wide one)
  if (passes == 1 && g.tile1 && M >= 2048) extra[VL_GX_TILE] = N <= 1024 ? (g.tile1 & 255) : ((g.tile1 >> 8) & 255);
  extra[VL_GX_PERSIST] = passes == 3 ? (g.persist & 0xffff) : ((g.persist >> 16) & 0xffff);
  if (im) {
    extra[VL_GX_IMG] = (int64_t)(uintptr_t)im->img; extra[VL_GX_IMG_COLS] = im->cols;
    extra[VL_GX_COLSUM] = (int64_t)(uintptr_t)im->colsum;
    if (im->colsum) extra[VL_GX_WS] = extra[VL_GX_WS_FLOATS] = 0;
  }
  const int rc = vl_gemm_nt_ex(a_hi, a_lo, lda, b_hi, b_lo, ldb, M, N, K, passes, epi, bias, resid, out32, ldc, out_hi, out_lo,
                               aux16, ld16, extra, stream);
  if (im && im->colsum_rows) *im->colsum_rows = extra[VL_GX_COLSUM_ROWS];
  prof_end(pair, passes * 16 + epi, 2 * M * N * K, stream);
  return rc;
}

// The two [M, I] operands of the weight-gradient GEMMs -- GELU output h (X side of dW2) and du (dY side of dW1, + the
// column sums that are b1's gradient) -- are written as K-major images by the epilogues of the GEMMs that produce them
// when those run on the ping-pong kernel over whole 64-row blocks: half of the re-layout traffic never happens.
// operand layouts of the weight-gradient GEMM (vl_dw_grouped_mixed): bit 0 = dY row-major, bit 1 = X row-major (no K-major
// image, no re-layout pass for that side); whole 64-row blocks only, else 0 = both through the re-layout pass
int dw_mode(const int64_t* d) {
  const int64_t B = d[VL_ST_B], S = d[VL_ST_S], H = d[VL_ST_H], I = d[VL_ST_I];
  const bool ok = (B * S) % 64 == 0 && (d[VL_ST_POOLED_ONLY] == 0 || B % 64 == 0) && H % 8 == 0 && I % 8 == 0 &&
                  d[VL_ST_CS_QKV] && d[VL_ST_CS_U];
  return ok ? (int)(d[VL_ST_DW_ROWMAJOR] & 3) : 0;
}
bool fused_shape(const int64_t* d, int64_t l) {
  const int64_t B = d[VL_ST_B], S = d[VL_ST_S], H = d[VL_ST_H], I = d[VL_ST_I], L = d[VL_ST_NLAYERS], M = B * S;
  const bool pooled = d[VL_ST_POOLED_ONLY] != 0 && l == L - 1;
  return !pooled && M % 64 == 0 && vl_gemm_nt_path(M, I, H, 1, d[VL_ST_SMALL_WS] != 0) == 2;
}
bool fused_h(const int64_t* d, int64_t l) {  // VL_ST_FUSE_IMAGES bit 0: h by the FFN1 (erf-GELU) epilogue
  const int64_t* y = d + VL_ST_FIELDS + l * VL_LY_FIELDS;
  return (d[VL_ST_FUSE_IMAGES] & 1) && y[VL_LY_T_H] && fused_shape(d, l);
}
bool fused_du(const int64_t* d, int64_t l) {  // bit 1: du + its column sums by the GELU' epilogue
  const int64_t* y = d + VL_ST_FIELDS + l * VL_LY_FIELDS;
  return (d[VL_ST_FUSE_IMAGES] & 2) && y[VL_LY_T_DU] && y[VL_LY_CS_DU] && fused_shape(d, l);
}

int check_header(const char* fn, const int64_t* d) {
  VL_CHECK_ARG(d, "%s: null descriptor", fn);
  VL_CHECK_ARG(d[VL_ST_MAGIC] == VL_ST_MAGIC_VALUE, "%s: descriptor magic mismatch (built for another library version?)", fn);
  VL_CHECK_ARG(d[VL_ST_B] > 0 && d[VL_ST_S] > 0 && d[VL_ST_H] > 0 && d[VL_ST_I] > 0 && d[VL_ST_NH] > 0 && d[VL_ST_NLAYERS] > 0,
               "%s: bad dimensions in the descriptor", fn);
  VL_CHECK_ARG(d[VL_ST_H] == d[VL_ST_NH] * 64 || d[VL_ST_H] == d[VL_ST_NH] * 32, "%s: head dim must be 64 or 32", fn);
  return 0;
}

// K-major images of layer l's X operands {layer input, attention context, LayerNorm-1 output, GELU output} for the
// weight-gradient GEMMs (dw.hip)
int x_images(const int64_t* d, int64_t l, int64_t max_blocks, hipStream_t ss) {
  const int64_t B = d[VL_ST_B], S = d[VL_ST_S], H = d[VL_ST_H], I = d[VL_ST_I], L = d[VL_ST_NLAYERS], M = B * S;
  const int64_t* y = d + VL_ST_FIELDS + l * VL_LY_FIELDS;
  const bool pooled = d[VL_ST_POOLED_ONLY] != 0 && l == L - 1;
  const int64_t tr[4 * VL_TR_FIELDS] = {
      y[VL_LY_X_HI], H, H, y[VL_LY_T_X], 0, 0,
      y[VL_LY_CTX_HI], H, H, y[VL_LY_T_CTX], 0, 0,
      y[VL_LY_X1_HI], H, H, y[VL_LY_T_X1], 0, 0,
      y[VL_LY_H_HI], I, I, y[VL_LY_T_H], 0, 0};
  if (pooled) {  // the layer input has M rows, the other three only the B live ones
    VL_TRY(vl_transpose_blocked(tr, 1, M, max_blocks, ss));
    return vl_transpose_blocked(tr + VL_TR_FIELDS, 3, B, max_blocks, ss);
  }
  return vl_transpose_blocked(tr, fused_h(d, l) ? 3 : 4, M, max_blocks, ss);  // (h: by the FFN1 epilogue)
}

}  // namespace

// The op north_star names -- QKV projection + softmax(QK^T)V over the mixed (token, box) sequence -- as one entry point.
// Two launches share the split-bf16 Q|K|V (no fp32 copy of the projection goes through HBM): the 3-pass projection GEMM
// with the (hi, lo) epilogue, then the attention kernel.  A single-kernel fusion was analysed and not built: the
// projection must land in HBM anyway (backward reads it), so fusion only saves the attention kernel's re-read of it
// (132 MB per layer, largely Infinity-Cache hits right after the write, ~10-20 us), while a 224 x 192 per-(4 samples,
// head) GEMM tile re-reads X twelve times through L2 and has 20 % fewer FLOP per LDS byte than the 256 x 256 tile
// ("Fused V&L attention").
extern "C" int vl_qkv_attention_fwd(const void* x_hi, const void* x_lo, const void* wqkv_hi, const void* wqkv_lo,
                                    const float* bqkv, const float* addmask, void* qkv_hi, void* qkv_lo, void* ctx_hi,
                                    void* ctx_lo, float* lse, int64_t B, int64_t S, int64_t nh, int64_t dh, int64_t nq,
                                    float p_drop, uint64_t seed, void* stream) {
  const int64_t H = nh * dh, M = B * S;
  VL_TRY(vl_gemm_nt(x_hi, x_lo, H, wqkv_hi, wqkv_lo, H, M, 3 * H, H, 3, VL_EPI_SPLIT, bqkv, nullptr, nullptr, 0, qkv_hi, qkv_lo,
                    nullptr, 3 * H, stream));
  return vl_attn2_fwd(qkv_hi, qkv_lo, addmask, ctx_hi, ctx_lo, lse, B, S, nh, dh, nq, p_drop, seed, stream);
}

// Backward of the same op w.r.t. its input: dctx16 -> dqkv16 (attention backward) -> dx32 = dqkv . W_qkv (+ resid32).
// (The weight / bias gradients of the projection are products over the batch rows: vl_transpose_blocked + vl_dw_grouped.)
extern "C" int vl_qkv_attention_bwd(const void* qkv_hi, const float* addmask, const void* dctx16, const float* lse,
                                    const void* wqkv_t, const float* resid32, void* dqkv16, float* dx32, int64_t B,
                                    int64_t S, int64_t nh, int64_t dh, int64_t nq, float p_drop, uint64_t seed, void* stream) {
  const int64_t H = nh * dh, M = B * S;
  VL_TRY(vl_attn2_bwd(qkv_hi, addmask, dctx16, lse, dqkv16, B, S, nh, dh, nq, p_drop, seed, stream));
  return vl_gemm_nt(dqkv16, nullptr, 3 * H, wqkv_t, nullptr, 3 * H, M, H, 3 * H, 1, VL_EPI_F32, nullptr, resid32, dx32, H, nullptr,
                    nullptr, nullptr, 0, stream);
}

extern "C" int64_t vl_stack_desc_len(int64_t n_layers) { return VL_ST_FIELDS + n_layers * VL_LY_FIELDS; }

extern "C" int vl_stack_fwd(const int64_t* d, int64_t layer_begin, int64_t layer_end, void* stream, void* stream_side) {
  VL_TRY(check_header("vl_stack_fwd", d));
  const int64_t B = d[VL_ST_B], S = d[VL_ST_S], H = d[VL_ST_H], I = d[VL_ST_I], nh = d[VL_ST_NH], L = d[VL_ST_NLAYERS];
  VL_CHECK_ARG(layer_begin >= 0 && layer_begin <= layer_end && layer_end <= L, "vl_stack_fwd: bad layer range");
  const int64_t M = B * S;
  const float eps = f_of(d[VL_ST_EPS]), p_hid = f_of(d[VL_ST_P_HID]), p_att = f_of(d[VL_ST_P_ATT]);
  const float* addmask = ptr<const float>(d[VL_ST_ADDMASK]);
  const float* row_post = ptr<const float>(d[VL_ST_ROW_POST]);
  int64_t* prof = ptr<int64_t>(d[VL_ST_PROF]);
  const GemmCtx gc{prof, ptr<float>(d[VL_ST_SMALL_WS]), d[VL_ST_SMALL_WS_FLOATS], d[VL_ST_DX_TILE], d[VL_ST_GEMM_PERSIST]};
  hipStream_t ss = stream_side ? (hipStream_t)stream_side : (hipStream_t)stream;
  hipEvent_t fork = ptr<ihipEvent_t>(d[VL_ST_EV_FORK]);
  for (int64_t l = layer_begin; l < layer_end; ++l) {
    const int64_t* y = d + VL_ST_FIELDS + l * VL_LY_FIELDS;
    const int s3 = (int)(16 * l + 3);
    if (y[VL_LY_EV_READY]) {  // this layer's parameters are being updated on another stream: wait for that update
      hipError_t e = hipStreamWaitEvent((hipStream_t)stream, ptr<ihipEvent_t>(y[VL_LY_EV_READY]), 0);
      if (e != hipSuccess) return vl_set_error(-3, "vl_stack_fwd: wait for layer %lld's parameters: %s", (long long)l, hipGetErrorString(e));
    }
    // only the pooled row of every sample leaves the LAST layer (BertTextPooler reads hidden_states[:, 0],
    // encoders.py:597-608; M3P's BertPooler likewise): everything after the K/V projection of that layer runs on the
    // B live rows (R), in compact [B, .] buffers, with the dropout counters / row masks of the original rows (stride S)
    const bool pooled = d[VL_ST_POOLED_ONLY] != 0 && l == L - 1;
    const int64_t R = pooled ? B : M, nq = pooled ? 1 : S, os = pooled ? S : 1;
    // Q | K | V = X W^T + b as the (hi, lo) split, then attention: the fused V&L attention op (timed as ONE op, tag
    // VL_PROF_TAG_QKV_ATTN: projection + core FLOPs against the interval around both launches)
    {
      int64_t* pair = prof_begin(prof, stream);
      VL_TRY(vl_qkv_attention_fwd(ptr<void>(y[VL_LY_X_HI]), ptr<void>(y[VL_LY_X_LO]), ptr<void>(y[VL_LY_WQKV_HI]),
                                  ptr<void>(y[VL_LY_WQKV_LO]), ptr<const float>(y[VL_LY_BQKV]), addmask, ptr<void>(y[VL_LY_QKV_HI]),
                                  ptr<void>(y[VL_LY_QKV_LO]), ptr<void>(y[VL_LY_CTX_HI]), ptr<void>(y[VL_LY_CTX_LO]),
                                  ptr<float>(y[VL_LY_LSE]), B, S, nh, H / nh, nq, p_att, seed_of(d[VL_ST_SEED0], s3), stream));
      prof_end(pair, VL_PROF_TAG_QKV_ATTN, 2 * M * 3 * H * H + 4 * B * nq * S * H, stream);
    }
    VL_TRY(gemm(gc, ptr<void>(y[VL_LY_CTX_HI]), ptr<void>(y[VL_LY_CTX_LO]), H, ptr<void>(y[VL_LY_WO_HI]), ptr<void>(y[VL_LY_WO_LO]), H,
                R, H, H, 3, VL_EPI_F32, ptr<const float>(y[VL_LY_BO]), nullptr, ptr<float>(y[VL_LY_Z1]), H, nullptr, nullptr,
                nullptr, 0, stream));
    // The residual stream is never stored in fp32 between the LayerNorms of one call: every residual is the output of the
    // LayerNorm before it, whose z / mean / rstd backward keeps anyway -- the consumer re-evaluates gamma * ((z - mean) * rstd) +
    // beta (vl_ln_fwd_rr) instead of reading a copy the producer would have to write (44 MB per LayerNorm at c2).  The first
    // layer of a call reads the caller's x32, the last one writes out32 for the caller.
#ifdef VL_STACK_STORED_RESIDUAL  // (A/B builds: the fp32 residual stream stored by every LayerNorm and read by the next, as before)
    constexpr bool RR = false;
#else
    constexpr bool RR = true;
#endif
    const bool rr1 = RR && l > layer_begin;  // the residual of LayerNorm 1 = output of the previous layer's LayerNorm 2
    const int64_t* yp = y - VL_LY_FIELDS;
    const int64_t rl1[6] = {rr1 ? yp[VL_LY_Z2] : 0, rr1 ? yp[VL_LY_MEAN2] : 0, rr1 ? yp[VL_LY_RSTD2] : 0, rr1 ? yp[VL_LY_LN2_G] : 0,
                            rr1 ? yp[VL_LY_LN2_B] : 0, d[VL_ST_ROW_POST]};
    VL_TRY(vl_ln_fwd_rr(ptr<float>(y[VL_LY_Z1]), rr1 ? nullptr : ptr<const float>(y[VL_LY_X32]), rr1 ? rl1 : nullptr, nullptr, 1,
                        nullptr, nullptr, ptr<const float>(y[VL_LY_LN1_G]), ptr<const float>(y[VL_LY_LN1_B]), eps,
                        RR ? nullptr : ptr<float>(y[VL_LY_X1_32]), ptr<void>(y[VL_LY_X1_HI]), ptr<void>(y[VL_LY_X1_LO]), ptr<float>(y[VL_LY_MEAN1]), ptr<float>(y[VL_LY_RSTD1]),
                        R, H, R, 0, 0, p_hid, 0.f, seed_of(d[VL_ST_SEED0], s3 + 1), os, os, stream));
    const GemmImage him{ptr<void>(y[VL_LY_T_H]), I, nullptr, nullptr};
    VL_TRY(gemm(gc, ptr<void>(y[VL_LY_X1_HI]), ptr<void>(y[VL_LY_X1_LO]), H, ptr<void>(y[VL_LY_W1_HI]), ptr<void>(y[VL_LY_W1_LO]), H,
                R, I, H, 3, VL_EPI_GELU_SPLIT, ptr<const float>(y[VL_LY_B1]), nullptr, nullptr, 0, ptr<void>(y[VL_LY_H_HI]),
                ptr<void>(y[VL_LY_H_LO]), ptr<void>(y[VL_LY_U16]), I, stream, fused_h(d, l) ? &him : nullptr));
    VL_TRY(gemm(gc, ptr<void>(y[VL_LY_H_HI]), ptr<void>(y[VL_LY_H_LO]), I, ptr<void>(y[VL_LY_W2_HI]), ptr<void>(y[VL_LY_W2_LO]), I,
                R, H, I, 3, VL_EPI_F32, ptr<const float>(y[VL_LY_B2]), nullptr, ptr<float>(y[VL_LY_Z2]), H, nullptr, nullptr,
                nullptr, 0, stream));
    const int64_t rl2[6] = {y[VL_LY_Z1], y[VL_LY_MEAN1], y[VL_LY_RSTD1], y[VL_LY_LN1_G], y[VL_LY_LN1_B], 0};
    VL_TRY(vl_ln_fwd_rr(ptr<float>(y[VL_LY_Z2]), RR ? nullptr : ptr<const float>(y[VL_LY_X1_32]), RR ? rl2 : nullptr, nullptr, 1,
                        nullptr, row_post, ptr<const float>(y[VL_LY_LN2_G]), ptr<const float>(y[VL_LY_LN2_B]), eps,
                        (!RR || l == layer_end - 1) ? ptr<float>(y[VL_LY_OUT32]) : nullptr,
                        ptr<void>(y[VL_LY_OUT_HI]), ptr<void>(y[VL_LY_OUT_LO]), ptr<float>(y[VL_LY_MEAN2]), ptr<float>(y[VL_LY_RSTD2]),
                        R, H, R, 0, 0, p_hid, 0.f, seed_of(d[VL_ST_SEED0], s3 + 2), os, 1, stream));
  }
  // training: the K-major images of the layers' X operands {layer input, attention context, LayerNorm-1 output, GELU
  // output} for the weight-gradient GEMMs of backward.  Those of the top layers (all but the bottom VL_ST_TR_BWD_LAYERS)
  // are written on the side stream once the LAST layer of the stack has been enqueued, i.e. under the task head / loss
  // (tiny latency-bound kernels, the chip is otherwise idle); the others in backward, on the side stream ahead of the
  // layer's own re-layout.  History: written layer by layer beside the forward GEMMs the re-layout cost the QKV
  // projection +40 % (net zero); all 12 layers under the head was best while the head was ~75 eager launches (0.6 ms);
  // with the head as one native node (~0.3 ms) only the top layer's images still fit there (16.98 vs 17.28 ms / step).
  if (layer_end == L && d[VL_ST_FIELDS + VL_LY_T_X] && !(dw_mode(d) & 2)) {
    if (ss != (hipStream_t)stream) {
      VL_CHECK_ARG(fork, "vl_stack_fwd: a side stream needs the fork event of the descriptor");
      hipError_t e = hipEventRecord(fork, (hipStream_t)stream);
      if (e == hipSuccess) e = hipStreamWaitEvent(ss, fork, 0);
      if (e != hipSuccess) return vl_set_error(-3, "vl_stack_fwd: stream fork: %s", hipGetErrorString(e));
    }
    for (int64_t l = L - 1; l >= d[VL_ST_TR_BWD_LAYERS] && l >= 0; --l)  // the order backward consumes them in
      VL_TRY(x_images(d, l, d[VL_ST_TR_BLOCKS_FWD], ss));
  }
  return 0;
}

// layers [layer_lo, layer_hi) in DESCENDING order.  dy of the top layer = y[VL_LY_DY] of layer layer_hi - 1; each layer
// writes dL/d(its input) to y[VL_LY_DX] (the host makes DX of layer l the DY of layer l - 1).
extern "C" int vl_stack_bwd(const int64_t* d, int64_t layer_hi, int64_t layer_lo, void* stream_main, void* stream_side) {
  VL_TRY(check_header("vl_stack_bwd", d));
  const int64_t B = d[VL_ST_B], S = d[VL_ST_S], H = d[VL_ST_H], I = d[VL_ST_I], nh = d[VL_ST_NH], L = d[VL_ST_NLAYERS];
  VL_CHECK_ARG(layer_lo >= 0 && layer_lo <= layer_hi && layer_hi <= L, "vl_stack_bwd: bad layer range");
  const int64_t M = B * S;
  const float p_hid = f_of(d[VL_ST_P_HID]), p_att = f_of(d[VL_ST_P_ATT]);
  const float* addmask = ptr<const float>(d[VL_ST_ADDMASK]);
  const float* row_post = ptr<const float>(d[VL_ST_ROW_POST]);
  const int accumulate = (int)d[VL_ST_ACCUMULATE];
  int64_t* prof = ptr<int64_t>(d[VL_ST_PROF]);
  const GemmCtx gc{prof, ptr<float>(d[VL_ST_SMALL_WS]), d[VL_ST_SMALL_WS_FLOATS], d[VL_ST_DX_TILE], d[VL_ST_GEMM_PERSIST]};
  hipStream_t sm = (hipStream_t)stream_main;
  hipStream_t ss = stream_side ? (hipStream_t)stream_side : sm;
  hipEvent_t fork = ptr<ihipEvent_t>(d[VL_ST_EV_FORK]);
  VL_CHECK_ARG(ss == sm || fork, "vl_stack_bwd: a side stream needs the fork event of the descriptor");
  for (int64_t l = layer_hi - 1; l >= layer_lo; --l) {
    const int64_t* y = d + VL_ST_FIELDS + l * VL_LY_FIELDS;
    const int s3 = (int)(16 * l + 3);
    const bool pooled = d[VL_ST_POOLED_ONLY] != 0 && l == L - 1;  // see vl_stack_fwd: DY is the compact [B, H] gradient
    const int64_t R = pooled ? B : M, nq = pooled ? 1 : S, os = pooled ? S : 1;
    VL_CHECK_ARG(!pooled || d[VL_ST_ROWS0], "vl_stack_bwd: the pooled-row mode needs VL_ST_ROWS0");
    // ---- critical path ------------------------------------------------------------------------------------------
    VL_TRY(vl_ln_bwd(ptr<const float>(y[VL_LY_DY]), ptr<const float>(y[VL_LY_Z2]), ptr<const float>(y[VL_LY_MEAN2]),
                     ptr<const float>(y[VL_LY_RSTD2]), ptr<const float>(y[VL_LY_LN2_G]), nullptr, row_post, ptr<float>(y[VL_LY_DZ2]),
                     ptr<void>(y[VL_LY_DT2]), nullptr, nullptr, nullptr, nullptr, ptr<float>(y[VL_LY_LNWS2]), R, H, R, 0, 0, p_hid, 0.f,
                     seed_of(d[VL_ST_SEED0], s3 + 2), os, sm));
    const bool fused = fused_du(d, l);
    int64_t cs_du_rows = 0;
    const GemmImage duim{ptr<void>(y[VL_LY_T_DU]), I, ptr<float>(y[VL_LY_CS_DU]), &cs_du_rows};
    VL_TRY(gemm(gc, ptr<void>(y[VL_LY_DT2]), nullptr, H, ptr<void>(y[VL_LY_W2_T]), nullptr, H, R, I, H, 1, VL_EPI_DGELU_BF16,
                nullptr, nullptr, nullptr, 0, ptr<void>(y[VL_LY_DU16]), nullptr, ptr<void>(y[VL_LY_U16]), I, sm,
                fused ? &duim : nullptr));
    VL_TRY(gemm(gc, ptr<void>(y[VL_LY_DU16]), nullptr, I, ptr<void>(y[VL_LY_W1_T]), nullptr, I, R, H, I, 1, VL_EPI_F32, nullptr,
                ptr<const float>(y[VL_LY_DZ2]), ptr<float>(y[VL_LY_DX1]), H, nullptr, nullptr, nullptr, 0, sm));
    VL_TRY(vl_ln_bwd(ptr<const float>(y[VL_LY_DX1]), ptr<const float>(y[VL_LY_Z1]), ptr<const float>(y[VL_LY_MEAN1]),
                     ptr<const float>(y[VL_LY_RSTD1]), ptr<const float>(y[VL_LY_LN1_G]), nullptr, nullptr, ptr<float>(y[VL_LY_DZ1]),
                     ptr<void>(y[VL_LY_DT1]), nullptr, nullptr, nullptr, nullptr, ptr<float>(y[VL_LY_LNWS1]), R, H, R, 0, 0, p_hid, 0.f,
                     seed_of(d[VL_ST_SEED0], s3 + 1), os, sm));
    VL_TRY(gemm(gc, ptr<void>(y[VL_LY_DT1]), nullptr, H, ptr<void>(y[VL_LY_WO_T]), nullptr, H, R, H, H, 1, VL_EPI_BF16, nullptr,
                nullptr, nullptr, 0, ptr<void>(y[VL_LY_DCTX16]), nullptr, nullptr, H, sm));
    VL_TRY(vl_attn2_bwd(ptr<void>(y[VL_LY_QKV_HI]), addmask, ptr<void>(y[VL_LY_DCTX16]), ptr<const float>(y[VL_LY_LSE]),
                        ptr<void>(y[VL_LY_DQKV]), B, S, nh, H / nh, nq, p_att, seed_of(d[VL_ST_SEED0], s3), sm));
    if (ss != sm) {  // everything the side stream reads of this layer has been enqueued on the main stream
      hipError_t e = hipEventRecord(fork, sm);
      if (e == hipSuccess) e = hipStreamWaitEvent(ss, fork, 0);
      if (e != hipSuccess) return vl_set_error(-3, "vl_stack_bwd: stream fork: %s", hipGetErrorString(e));
    }
    // dL/dX = dQKV W_qkv + dz1 (the residual branch); in the pooled-row mode dz1 only has the B live rows: they are
    // added to rows b * S afterwards (one fp32 add per element either way: bit-identical to the dense epilogue)
    VL_TRY(gemm(gc, ptr<void>(y[VL_LY_DQKV]), nullptr, 3 * H, ptr<void>(y[VL_LY_WQKV_T]), nullptr, 3 * H, M, H, 3 * H, 1, VL_EPI_F32,
                nullptr, pooled ? nullptr : ptr<const float>(y[VL_LY_DZ1]), ptr<float>(y[VL_LY_DX]), H, nullptr, nullptr, nullptr, 0,
                sm));
    if (pooled)
      VL_TRY(vl_embed_scatter_add(ptr<const int64_t>(d[VL_ST_ROWS0]), ptr<const float>(y[VL_LY_DZ1]), ptr<float>(y[VL_LY_DX]), B, H,
                                  -1, nullptr, sm));
    // ---- optimizer-only work (side stream): weight-gradient GEMM + column sums ------------------------------------
    const int64_t nws = vl_ln_bwd_ws_floats(R, H) / (3 * H);
    if (const int mode = dw_mode(d)) {
      // per side: row-major operand (transposing LDS reads in the GEMM) or its K-major image (one re-layout launch); the
      // bias column sums of dqkv / du ride in the GEMM (partials [3][.] per problem in the CS buffers)
      const bool dy_img = !(mode & 1), x_img = !(mode & 2);
      if (x_img && l < d[VL_ST_TR_BWD_LAYERS]) VL_TRY(x_images(d, l, d[VL_ST_TR_BLOCKS_BWD], ss));
      if (dy_img) {
        const int64_t tr[4 * VL_TR_FIELDS] = {
            y[VL_LY_DQKV], 3 * H, 3 * H, d[VL_ST_T_DQKV], 0, 0,
            y[VL_LY_DT1], H, H, d[VL_ST_T_DT1], 0, 0,
            y[VL_LY_DT2], H, H, d[VL_ST_T_DT2], 0, 0,
            y[VL_LY_DU16], I, I, d[VL_ST_T_DU], 0, 0};
        if (pooled) {
          VL_TRY(vl_transpose_blocked(tr, 1, M, d[VL_ST_TR_BLOCKS_BWD], ss));
          VL_TRY(vl_transpose_blocked(tr + VL_TR_FIELDS, 3, R, d[VL_ST_TR_BLOCKS_BWD], ss));
        } else {
          VL_TRY(vl_transpose_blocked(tr, 4, M, d[VL_ST_TR_BLOCKS_BWD], ss));
        }
      }
      const int64_t tq = (H + 255) / 256;  // partial rows of a problem whose X operand has H columns
      float* csq = ptr<float>(d[VL_ST_CS_QKV]);
      const int64_t cq = (int64_t)(uintptr_t)csq, ck = (int64_t)(uintptr_t)(csq + tq * H), cv = (int64_t)(uintptr_t)(csq + 2 * tq * H);
      // dY operands {q, k, v, o, w1, w2} and X operands, each as (pointer, leading dimension | image columns)
      const int64_t aq = dy_img ? d[VL_ST_T_DQKV] : y[VL_LY_DQKV], astep = dy_img ? 2 * 64 * H : 2 * H;
      const int64_t a_o = dy_img ? d[VL_ST_T_DT1] : y[VL_LY_DT1], a_1 = dy_img ? d[VL_ST_T_DU] : y[VL_LY_DU16];
      const int64_t a_2 = dy_img ? d[VL_ST_T_DT2] : y[VL_LY_DT2];
      const int64_t bx = x_img ? y[VL_LY_T_X] : y[VL_LY_X_HI], bc = x_img ? y[VL_LY_T_CTX] : y[VL_LY_CTX_HI];
      const int64_t b1 = x_img ? y[VL_LY_T_X1] : y[VL_LY_X1_HI], bh = x_img ? y[VL_LY_T_H] : y[VL_LY_H_HI];
      const int64_t pr[6 * VL_DW_FIELDS] = {
          aq, 3 * H, bx, H, y[VL_LY_GRAD0 + 0], H, y[VL_LY_MASK0 + 0], H, H, cq,
          aq + astep, 3 * H, bx, H, y[VL_LY_GRAD0 + 2], H, y[VL_LY_MASK0 + 1], H, H, ck,
          aq + 2 * astep, 3 * H, bx, H, y[VL_LY_GRAD0 + 4], H, y[VL_LY_MASK0 + 2], H, H, cv,
          a_o, H, bc, H, y[VL_LY_GRAD0 + 6], H, y[VL_LY_MASK0 + 3], H, H, 0,
          a_1, I, b1, H, y[VL_LY_GRAD0 + 10], H, y[VL_LY_MASK0 + 4], I, H, d[VL_ST_CS_U],
          a_2, H, bh, I, y[VL_LY_GRAD0 + 12], I, y[VL_LY_MASK0 + 5], H, I, 0};
      // (VL_ST_DW_BUDGET > 0: the stream-K form on a fixed number of workgroups -- the dX products of the main stream
      // keep the other CUs; launches of the side stream share the workspace in stream order)
      // (the last layer of backward has the chip to itself once the main stream has run dry: VL_ST_DW_TAIL_BUDGET)
      const int64_t budget = (l == 0 && d[VL_ST_DW_TAIL_BUDGET] > 0) ? d[VL_ST_DW_TAIL_BUDGET] : d[VL_ST_DW_BUDGET];
      void* skws = ptr<void>(d[VL_ST_DW_SK_WS]);
      auto dw = [&](const int64_t* probs, int64_t n, int64_t rows) {
        if (budget > 0 && rows >= 1024)
          return vl_dw_grouped_streamk(probs, n, rows, accumulate, mode, budget, skws, d[VL_ST_DW_SK_WS_BYTES], ss);
        return vl_dw_grouped_mixed(probs, n, rows, accumulate, mode, ss);
      };
      if (pooled) {  // Q/K/V gradients reduce over all M rows, the other three over the B live rows
        VL_TRY(dw(pr, 3, M));
        VL_TRY(dw(pr + 3 * VL_DW_FIELDS, 3, R));
      } else {
        VL_TRY(dw(pr, 6, M));
      }
      const int64_t cr[6 * VL_CR_FIELDS] = {
          y[VL_LY_LNWS2], nws, 3 * H, H, y[VL_LY_GRAD0 + 14], y[VL_LY_GRAD0 + 15], y[VL_LY_GRAD0 + 13], 0,
          y[VL_LY_LNWS1], nws, 3 * H, H, y[VL_LY_GRAD0 + 8], y[VL_LY_GRAD0 + 9], y[VL_LY_GRAD0 + 7], 0,
          cq, tq, H, H, y[VL_LY_GRAD0 + 1], 0, 0, 0,
          ck, tq, H, H, y[VL_LY_GRAD0 + 3], 0, 0, 0,
          cv, tq, H, H, y[VL_LY_GRAD0 + 5], 0, 0, 0,
          d[VL_ST_CS_U], tq, I, I, y[VL_LY_GRAD0 + 11], 0, 0, 0};
      VL_TRY(vl_colreduce_multi(cr, 6, accumulate, ss));
      continue;
    }
    // ---- optimizer-only work: K-major re-layout, column sums, grouped weight-gradient GEMM ---------------------------
    const int64_t mblk = (M + 63) / 64, rblk = (R + 63) / 64;
    const int64_t tr[4 * VL_TR_FIELDS] = {
        y[VL_LY_DQKV], 3 * H, 3 * H, d[VL_ST_T_DQKV], d[VL_ST_CS_QKV], 0,
        y[VL_LY_DT1], H, H, d[VL_ST_T_DT1], 0, 0,
        y[VL_LY_DT2], H, H, d[VL_ST_T_DT2], 0, 0,
        y[VL_LY_DU16], I, I, d[VL_ST_T_DU], d[VL_ST_CS_U], 0};  // (last: written by the GELU' epilogue when `fused`)
    VL_CHECK_ARG(y[VL_LY_T_X] && y[VL_LY_T_CTX] && y[VL_LY_T_X1] && y[VL_LY_T_H],
                 "vl_stack_bwd: the layer record lacks the K-major X images (forward ran without them)");
    if (l < d[VL_ST_TR_BWD_LAYERS])  // not written at the end of forward
      VL_TRY(x_images(d, l, d[VL_ST_TR_BLOCKS_BWD], ss));
    if (pooled) {  // dqkv has M rows, the other three only the B live ones
      VL_TRY(vl_transpose_blocked(tr, 1, M, d[VL_ST_TR_BLOCKS_BWD], ss));
      VL_TRY(vl_transpose_blocked(tr + VL_TR_FIELDS, 3, R, d[VL_ST_TR_BLOCKS_BWD], ss));
    } else {
      VL_TRY(vl_transpose_blocked(tr, fused ? 3 : 4, M, d[VL_ST_TR_BLOCKS_BWD], ss));
    }
    // one launch: LayerNorm partials -> (dgamma, dbeta, bias gradient of the producing Linear) x 2, column-sum partials
    // of dqkv -> (bq, bk, bv) and of du -> b1
    const int64_t cr[4 * VL_CR_FIELDS] = {
        y[VL_LY_LNWS2], nws, 3 * H, H, y[VL_LY_GRAD0 + 14], y[VL_LY_GRAD0 + 15], y[VL_LY_GRAD0 + 13], 0,
        y[VL_LY_LNWS1], nws, 3 * H, H, y[VL_LY_GRAD0 + 8], y[VL_LY_GRAD0 + 9], y[VL_LY_GRAD0 + 7], 0,
        d[VL_ST_CS_QKV], mblk, 3 * H, H, y[VL_LY_GRAD0 + 1], y[VL_LY_GRAD0 + 3], y[VL_LY_GRAD0 + 5], 0,
        fused ? y[VL_LY_CS_DU] : d[VL_ST_CS_U], fused ? cs_du_rows : rblk, I, I, y[VL_LY_GRAD0 + 11], 0, 0, 0};
    VL_TRY(vl_colreduce_multi(cr, 4, accumulate, ss));
    const int64_t pr[6 * VL_DW_FIELDS] = {
        d[VL_ST_T_DQKV], 3 * H, y[VL_LY_T_X], H, y[VL_LY_GRAD0 + 0], H, y[VL_LY_MASK0 + 0], H, H, 0,
        d[VL_ST_T_DQKV] + 2 * 64 * H, 3 * H, y[VL_LY_T_X], H, y[VL_LY_GRAD0 + 2], H, y[VL_LY_MASK0 + 1], H, H, 0,
        d[VL_ST_T_DQKV] + 2 * 64 * 2 * H, 3 * H, y[VL_LY_T_X], H, y[VL_LY_GRAD0 + 4], H, y[VL_LY_MASK0 + 2], H, H, 0,
        d[VL_ST_T_DT1], H, y[VL_LY_T_CTX], H, y[VL_LY_GRAD0 + 6], H, y[VL_LY_MASK0 + 3], H, H, 0,
        fused ? y[VL_LY_T_DU] : d[VL_ST_T_DU], I, y[VL_LY_T_X1], H, y[VL_LY_GRAD0 + 10], H, y[VL_LY_MASK0 + 4], I, H, 0,
        d[VL_ST_T_DT2], H, y[VL_LY_T_H], I, y[VL_LY_GRAD0 + 12], I, y[VL_LY_MASK0 + 5], H, I, 0};
    if (pooled) {  // Q/K/V gradients reduce over all M rows, the other three over the B live rows
      VL_TRY(vl_dw_grouped(pr, 3, M, accumulate, ss));
      VL_TRY(vl_dw_grouped(pr + 3 * VL_DW_FIELDS, 3, R, accumulate, ss));
    } else {
      VL_TRY(vl_dw_grouped(pr, 6, M, accumulate, ss));
    }
  }
  return 0;
}
