// vl_act_fwd / vl_act_bwd: the element-wise stages of the pooler / classifier head on [batch, features] tensors.
//
// Reference: BertTextPooler (volta/volta/encoders.py:597-608: dense -> ReLU), M3P BertPooler
// (volta/volta/m3p/m3p_transformer.py:548-560: dense -> tanh), nn.Dropout on the pooled vector (encoders.py:1238-1239),
// SimpleClassifier's GeLU (encoders.py:788-815, :131-137).  The reference runs each as 1-6 eager kernels plus autograd's
// replay; at [256, 768..1842] every one of them is pure launch latency, so one launch here does activation + dropout +
// the (hi, lo) bf16 split the next GEMM reads (forward), or dropout-mask * activation' + the bf16 cast the next dX / dW
// GEMM reads, zero-padded to its leading dimension (backward).  Dropout masks come from the counter RNG of common.h
// (regenerated in backward, never stored).
#include "common.h"
#include "../../include/vlhip.h"

namespace {

struct ActArgs {
  const float* z; const float* dy; long M; int N; int ld16; int act; float p, inv_keep; uint64_t seed;
  float* out32; bf16_raw* out_hi; bf16_raw* out_lo;
};

__device__ __forceinline__ float act_value(int act, float z) {
  switch (act) {
    case VL_ACT_RELU: return z > 0.f ? z : 0.f;
    case VL_ACT_TANH: return tanhf(z);
    case VL_ACT_GELU: return gelu_erf(z);
  }
  return z;
}
__device__ __forceinline__ float act_grad(int act, float z) {
  switch (act) {
    case VL_ACT_RELU: return z > 0.f ? 1.f : 0.f;
    case VL_ACT_TANH: { const float t = tanhf(z); return 1.f - t * t; }
    case VL_ACT_GELU: return gelu_erf_grad(z);
  }
  return 1.f;
}

// one thread per element of the [M, max(N, ld16)] iteration space; columns >= N only exist in the 16-bit outputs (zeros)
template <bool BWD>
__global__ __launch_bounds__(256) void act_kernel(ActArgs a) {
  const int W = a.ld16 > a.N ? a.ld16 : a.N;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= a.M * W) return;
  const long m = idx / W;
  const int n = (int)(idx - m * W);
  if (n >= a.N) {
    if (a.out_hi) a.out_hi[m * a.ld16 + n] = 0;
    if (a.out_lo) a.out_lo[m * a.ld16 + n] = 0;
    return;
  }
  const long e = m * a.N + n;
  const float keep = vl_dropout_scale(a.seed, (uint64_t)e, a.p, a.inv_keep);
  float v;
  if (BWD) {
    v = a.dy[e] * keep;
    if (a.act != VL_ACT_NONE) v *= act_grad(a.act, a.z[e]);
  } else {
    v = act_value(a.act, a.z[e]) * keep;
  }
  if (a.out32) a.out32[e] = v;
  if (a.out_hi) {
    bf16_raw hi, lo;
    split_bf16(v, hi, lo);
    a.out_hi[m * a.ld16 + n] = hi;
    if (a.out_lo) a.out_lo[m * a.ld16 + n] = lo;
  }
}

int check(const char* fn, int64_t M, int64_t N, int act, float p, const void* hi, int64_t ld16) {
  VL_CHECK_ARG(M > 0 && N > 0 && M * (N > ld16 ? N : ld16) < (1ll << 40), "%s: bad shape", fn);
  VL_CHECK_ARG(act >= VL_ACT_NONE && act <= VL_ACT_GELU, "%s: unknown activation %d", fn, act);
  VL_CHECK_ARG(p >= 0.f && p < 1.f, "%s: dropout probability must be in [0, 1)", fn);
  VL_CHECK_ARG(!hi || ld16 >= N, "%s: 16-bit outputs need ld16 >= N", fn);
  return 0;
}

}  // namespace

extern "C" int vl_act_fwd(const float* z32, int64_t M, int64_t N, int act, float p_drop, uint64_t seed, float* out32,
                          void* out_hi, void* out_lo, int64_t ld16, void* stream) {
  if (int rc = check("vl_act_fwd", M, N, act, p_drop, out_hi, ld16)) return rc;
  VL_CHECK_ARG(z32 && (out32 || out_hi) && (out_hi || !out_lo), "vl_act_fwd: null pointer");
  ActArgs a{};
  a.z = z32; a.M = M; a.N = (int)N; a.ld16 = out_hi ? (int)ld16 : 0; a.act = act; a.p = p_drop; a.inv_keep = 1.f / (1.f - p_drop);
  a.seed = seed; a.out32 = out32; a.out_hi = (bf16_raw*)out_hi; a.out_lo = (bf16_raw*)out_lo;
  const long n = M * (a.ld16 > a.N ? a.ld16 : a.N);
  hipLaunchKernelGGL(act_kernel<false>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
  VL_CHECK_LAUNCH("vl_act_fwd");
  return 0;
}

extern "C" int vl_act_bwd(const float* dy32, const float* z32, int64_t M, int64_t N, int act, float p_drop, uint64_t seed,
                          float* dz32, void* dz16, int64_t ld16, void* stream) {
  if (int rc = check("vl_act_bwd", M, N, act, p_drop, dz16, ld16)) return rc;
  VL_CHECK_ARG(dy32 && (dz32 || dz16) && (z32 || act == VL_ACT_NONE), "vl_act_bwd: null pointer");
  ActArgs a{};
  a.dy = dy32; a.z = z32; a.M = M; a.N = (int)N; a.ld16 = dz16 ? (int)ld16 : 0; a.act = act; a.p = p_drop;
  a.inv_keep = 1.f / (1.f - p_drop); a.seed = seed; a.out32 = dz32; a.out_hi = (bf16_raw*)dz16;
  const long n = M * (a.ld16 > a.N ? a.ld16 : a.N);
  hipLaunchKernelGGL(act_kernel<true>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
  VL_CHECK_LAUNCH("vl_act_bwd");
  return 0;
}
