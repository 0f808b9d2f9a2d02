// vl_gqa_loss: the GQA fine-tuning loss with semantic prior, its score and its gradient in ONE launch.
//
// Reference: volta/volta/task_utils.py:413-428 (ForwardModelsTrain, "VL-classifier-GQA") and :706-711
// (compute_score_with_logits):
//     p = softmax(logits);  (p10, idx) = topk(p, 10);  prior = mean_b sum_k p10 * distances[b, idx]
//     loss = CrossEntropy(logits, argmax(target.long())) * C  +  semantic_lambda * prior * C          (C = target.size(1))
//     score = sum_b target[b, argmax logits[b]] / B
// The eager reference issues ~45 tiny kernels for this on a [B, 1842] matrix (softmax, top-k, gathers, reductions,
// their autograd); at 14 k samples/s that is 0.7 ms of launch latency per step.  Here one workgroup per row computes the
// row's loss terms and d(loss)/d(logits) directly (top-k indices are constants for autograd, exactly as in torch):
//     dL/dz_j = (C/B) (p_j - [j = label])  +  (lambda C / B) p_j (d_j [j in top10] - s_b),     s_b = sum_{k in top10} p_k d_k
// and a one-workgroup second launch sums the B row terms in a fixed order (deterministic; a kernel boundary instead of a
// cross-workgroup ticket: the row terms come from workgroups on other XCDs / L2s).
#include "common.h"
#include "../../include/vlhip.h"

namespace {

constexpr int TOPK = 10, MAXPT = 16;  // up to 256 * 16 = 4096 labels

struct LossArgs {
  const float* logits; const float* target; const float* dist; float* dlogits; float* row_terms; float* out;
  int B, C; float lambda;
};

__device__ __forceinline__ void block_argmax(float& v, int& i, float* sv, int* si) {
  // lexicographic (value desc, index asc): the first occurrence of the maximum, like torch.max / torch.argmax
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(v, o, 64);
    const int oi = __shfl_xor(i, o, 64);
    if (ov > v || (ov == v && oi < i)) { v = ov; i = oi; }
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __syncthreads();
  if (lane == 0) { sv[wave] = v; si[wave] = i; }
  __syncthreads();
  v = sv[0]; i = si[0];
#pragma unroll
  for (int w = 1; w < 4; ++w)
    if (sv[w] > v || (sv[w] == v && si[w] < i)) { v = sv[w]; i = si[w]; }
}
__device__ __forceinline__ float block_sum(float v, float* sv) {
  v = wave_sum(v);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __syncthreads();
  if (lane == 0) sv[wave] = v;
  __syncthreads();
  return (sv[0] + sv[1]) + (sv[2] + sv[3]);
}

__global__ __launch_bounds__(256) void gqa_loss_kernel(LossArgs a) {
  __shared__ float sv[4];
  __shared__ int si[4];
  __shared__ int s_top[TOPK];
  const int b = blockIdx.x, tid = threadIdx.x, C = a.C;
  const float* z = a.logits + (long)b * C;
  const float* t = a.target + (long)b * C;
  const float* d = a.dist + (long)b * C;
  float x[MAXPT], p[MAXPT];
  // logits, their arg max (score), and the label = argmax(target.long()) (first maximum; .long() truncates like the reference)
  float zmax = -INFINITY; int zarg = 0x7fffffff;
  float tmax = -INFINITY; int targ = 0x7fffffff;
#pragma unroll
  for (int k = 0; k < MAXPT; ++k) {
    const int j = tid + 256 * k;
    x[k] = j < C ? z[j] : -INFINITY;
    if (x[k] > zmax) { zmax = x[k]; zarg = j; }
    if (j < C) {
      const float tv = truncf(t[j]);
      if (tv > tmax) { tmax = tv; targ = j; }
    }
  }
  block_argmax(zmax, zarg, sv, si);
  block_argmax(tmax, targ, sv, si);
  // a row with a NaN logit (every `>` is false) or with -inf everywhere leaves the sentinel: fall back to index 0 -- the
  // loss of such a row is NaN through `se`, as in the reference, and no index ever leaves the row
  if ((unsigned)zarg >= (unsigned)C) zarg = 0;
  if ((unsigned)targ >= (unsigned)C) targ = 0;
  const int label = targ;
  float se = 0.f;
#pragma unroll
  for (int k = 0; k < MAXPT; ++k) {
    p[k] = (tid + 256 * k) < C ? __expf(x[k] - zmax) : 0.f;
    se += p[k];
  }
  se = block_sum(se, sv);
  const float inv = 1.0f / se;
#pragma unroll
  for (int k = 0; k < MAXPT; ++k) p[k] *= inv;
  // top-10 of p (selection by repeated arg max; selected entries are masked in a private copy)
  float q[MAXPT];
#pragma unroll
  for (int k = 0; k < MAXPT; ++k) q[k] = (tid + 256 * k) < C ? p[k] : -1.f;
  float s_b = 0.f;
  for (int r = 0; r < TOPK && r < C; ++r) {
    float v = -1.f; int i = 0x7fffffff;
#pragma unroll
    for (int k = 0; k < MAXPT; ++k)
      if (q[k] > v) { v = q[k]; i = tid + 256 * k; }
    block_argmax(v, i, sv, si);
    if ((unsigned)i >= (unsigned)C) i = 0;  // NaN probabilities: nothing compares greater (see above)
    if (tid == 0) s_top[r] = i;
#pragma unroll
    for (int k = 0; k < MAXPT; ++k)
      if (tid + 256 * k == i) q[k] = -1.f;
    s_b += v * d[i];  // (every thread accumulates the same value)
  }
  __syncthreads();
  const float ce = (zmax + __logf(se)) - z[label];
  const float cB = (float)C / (float)a.B, lam = a.lambda * cB;
#pragma unroll
  for (int k = 0; k < MAXPT; ++k) {
    const int j = tid + 256 * k;
    if (j < C) {
      bool in_top = false;
      for (int r = 0; r < TOPK && r < C; ++r) in_top |= (s_top[r] == j);
      const float g = cB * (p[k] - (j == label ? 1.f : 0.f)) + lam * p[k] * ((in_top ? d[j] : 0.f) - s_b);
      a.dlogits[(long)b * C + j] = g;
    }
  }
  if (tid == 0) {
    a.row_terms[3 * b] = ce;
    a.row_terms[3 * b + 1] = s_b;
    a.row_terms[3 * b + 2] = t[zarg];
  }
}

__global__ __launch_bounds__(256) void gqa_loss_finish_kernel(LossArgs a) {
  __shared__ float sv[4];
  const int tid = threadIdx.x;
  float c = 0.f, s = 0.f, sc = 0.f;
  for (int r = tid; r < a.B; r += 256) {
    c += a.row_terms[3 * r];
    s += a.row_terms[3 * r + 1];
    sc += a.row_terms[3 * r + 2];
  }
  c = block_sum(c, sv); s = block_sum(s, sv); sc = block_sum(sc, sv);
  if (tid == 0) {
    const float fB = (float)a.B, fC = (float)a.C;
    a.out[0] = (c / fB) * fC + (a.lambda * (s / fB)) * fC;
    a.out[1] = sc / fB;
  }
}

}  // namespace

extern "C" int64_t vl_gqa_loss_ws_bytes(int64_t B) { return 12 * B; }

extern "C" int vl_gqa_loss(const float* logits, const float* target, const float* distances, int64_t B, int64_t C,
                           float semantic_lambda, float* loss_score, float* dlogits, void* ws, void* stream) {
  VL_CHECK_ARG(logits && target && distances && loss_score && dlogits && ws && B >= 1 && C >= 1 && C <= 256 * MAXPT,
               "vl_gqa_loss: bad arguments (C must be <= %d)", 256 * MAXPT);
  LossArgs a{};
  a.logits = logits; a.target = target; a.dist = distances; a.dlogits = dlogits; a.out = loss_score;
  a.row_terms = (float*)ws;
  a.B = (int)B; a.C = (int)C; a.lambda = semantic_lambda;
  hipLaunchKernelGGL(gqa_loss_kernel, dim3((unsigned)B), dim3(256), 0, (hipStream_t)stream, a);
  hipLaunchKernelGGL(gqa_loss_finish_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, a);
  VL_CHECK_LAUNCH("vl_gqa_loss");
  return 0;
}
