// vl_imp_select: one round of iterative magnitude pruning on the device -- the k smallest |w| among the still
// unmasked entries of the (concatenated) prunable weights get mask 0.
//
// Reference: volta/train_task_prunning.py:45-91 -> torch.nn.utils.prune.global_unstructured(L1Unstructured, amount)
// (torch nn/utils/prune.py:1095-1151 concatenation, :315-409 PruningContainer slice mask==1, :514-534
// topk(|t|, k, largest=False); k = round(amount * n_remaining)).  Integer/index work: the result must be the same
// SET of indices.  |w| >= 0, so IEEE-754 bit patterns order like unsigned integers: an exact 3-pass radix select
// (11 + 11 + 10 bits) over the fp32 bits finds the k-th smallest value T; everything below T is pruned, and of the
// entries equal to T the lowest flat indices are pruned until k is reached.  (torch.topk's choice among threshold
// ties is implementation-defined -- see DESIGN.md; without a tie at T the sets are identical.)
// No host synchronisation: k-th bucket bookkeeping stays in a small device workspace.
#include "common.h"
#include "../../include/vlhip.h"

namespace {

constexpr int CHUNK = 4096;      // elements per workgroup in the ordered passes (16 per thread, contiguous)
constexpr int WS_HIST = 0;       // [2048] histogram
constexpr int WS_STATE = 2048;   // [0]=prefix value so far, [1]=remaining k, [2]=T, [3]=need (ties to prune)
constexpr int WS_BLOCKS = 2048 + 8;

__device__ __forceinline__ unsigned key_of(float w) { return __float_as_uint(fabsf(w)); }

// pass p: 0 -> bits 31..21 (2048 bins), 1 -> bits 20..10 (2048 bins), 2 -> bits 9..0 (1024 bins)
template <int PASS>
__global__ __launch_bounds__(256) void hist_kernel(const float* __restrict__ w, const float* __restrict__ mask, long n,
                                                   unsigned* __restrict__ ws) {
  __shared__ unsigned h[2048];
  for (int i = threadIdx.x; i < 2048; i += 256) h[i] = 0;
  __syncthreads();
  const unsigned prefix = ws[WS_STATE + 0];
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    if (mask[i] != 1.0f) continue;
    const unsigned key = key_of(w[i] * mask[i]);
    if (PASS == 0) atomicAdd(&h[key >> 21], 1u);
    else if (PASS == 1) { if ((key >> 21) == prefix) atomicAdd(&h[(key >> 10) & 0x7FF], 1u); }
    else { if ((key >> 10) == prefix) atomicAdd(&h[key & 0x3FF], 1u); }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2048; i += 256)
    if (h[i]) atomicAdd(&ws[WS_HIST + i], h[i]);
}

// single workgroup: find the bin holding the k-th element, update (prefix, k), clear the histogram
template <int PASS>
__global__ __launch_bounds__(256) void pick_kernel(unsigned* ws, unsigned k_in) {
  __shared__ unsigned cum[2048];
  const int nb = PASS == 2 ? 1024 : 2048;
  for (int i = threadIdx.x; i < 2048; i += 256) cum[i] = i < nb ? ws[WS_HIST + i] : 0;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned k = PASS == 0 ? k_in : ws[WS_STATE + 1];  // 1-based rank still to find
    unsigned acc = 0;
    int b = 0;
    for (; b < nb; ++b) {
      if (acc + cum[b] >= k) break;
      acc += cum[b];
    }
    const unsigned prefix = ws[WS_STATE + 0];
    const unsigned np = PASS == 0 ? (unsigned)b : (PASS == 1 ? (prefix << 11) | (unsigned)b : (prefix << 10) | (unsigned)b);
    ws[WS_STATE + 0] = np;
    ws[WS_STATE + 1] = k - acc;  // rank inside the chosen bin (>= 1)
    if (PASS == 2) { ws[WS_STATE + 2] = np; ws[WS_STATE + 3] = k - acc; }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2048; i += 256) ws[WS_HIST + i] = 0;
}

// ordered passes over contiguous CHUNKs: count ties per chunk, then write the new mask
__global__ __launch_bounds__(256) void tie_count_kernel(const float* __restrict__ w, const float* __restrict__ mask,
                                                        long n, unsigned* __restrict__ ws) {
  __shared__ unsigned cnt;
  if (threadIdx.x == 0) cnt = 0;
  __syncthreads();
  const unsigned T = ws[WS_STATE + 2];
  const long base = (long)blockIdx.x * CHUNK + threadIdx.x * 16;
  unsigned c = 0;
  for (int j = 0; j < 16; ++j) {
    const long i = base + j;
    if (i < n && mask[i] == 1.0f && key_of(w[i]) == T) ++c;
  }
  if (c) atomicAdd(&cnt, c);
  __syncthreads();
  if (threadIdx.x == 0) ws[WS_BLOCKS + blockIdx.x] = cnt;
}

__global__ __launch_bounds__(1024) void tie_scan_kernel(unsigned* ws, int nblocks) {  // exclusive scan, one workgroup
  __shared__ unsigned part[1024];
  const int per = (nblocks + 1023) / 1024;
  const int b0 = threadIdx.x * per;
  unsigned s = 0;
  for (int j = 0; j < per; ++j) if (b0 + j < nblocks) s += ws[WS_BLOCKS + b0 + j];
  part[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned acc = 0;
    for (int i = 0; i < 1024; ++i) { const unsigned t = part[i]; part[i] = acc; acc += t; }
  }
  __syncthreads();
  unsigned acc = part[threadIdx.x];
  for (int j = 0; j < per; ++j)
    if (b0 + j < nblocks) { const unsigned t = ws[WS_BLOCKS + b0 + j]; ws[WS_BLOCKS + b0 + j] = acc; acc += t; }
}

__global__ __launch_bounds__(256) void write_mask_kernel(const float* __restrict__ w, const float* __restrict__ mask,
                                                         float* __restrict__ new_mask, long n,
                                                         const unsigned* __restrict__ ws) {
  __shared__ unsigned tpre[256];
  const unsigned T = ws[WS_STATE + 2], need = ws[WS_STATE + 3];
  const long base = (long)blockIdx.x * CHUNK + threadIdx.x * 16;
  unsigned c = 0;
  for (int j = 0; j < 16; ++j) {
    const long i = base + j;
    if (i < n && mask[i] == 1.0f && key_of(w[i]) == T) ++c;
  }
  tpre[threadIdx.x] = c;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned acc = ws[WS_BLOCKS + blockIdx.x];
    for (int i = 0; i < 256; ++i) { const unsigned t = tpre[i]; tpre[i] = acc; acc += t; }
  }
  __syncthreads();
  unsigned rank = tpre[threadIdx.x];  // global rank (in flat index order) of this thread's first tie
  for (int j = 0; j < 16; ++j) {
    const long i = base + j;
    if (i >= n) break;
    float m = mask[i];
    if (m == 1.0f) {
      const unsigned key = key_of(w[i]);
      if (key < T) m = 0.0f;
      else if (key == T) { if (rank < need) m = 0.0f; ++rank; }
    }
    new_mask[i] = m;
  }
}

}  // namespace

extern "C" int64_t vl_imp_ws_bytes(int64_t n) { return 4 * (WS_BLOCKS + (n + CHUNK - 1) / CHUNK + 8); }

extern "C" int vl_imp_select(const float* w, const float* mask, float* new_mask, int64_t n, int64_t k, void* ws,
                             void* stream) {
  VL_CHECK_ARG(w && mask && new_mask && ws && n > 0 && k >= 0 && k <= n && n < (1LL << 40) && k < (1LL << 32),
               "vl_imp_select: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  unsigned* u = (unsigned*)ws;
  const int nblocks = (int)((n + CHUNK - 1) / CHUNK);
  VL_CHECK_ARG(nblocks <= 1024 * 1024, "vl_imp_select: too many elements");
  hipError_t e = hipMemsetAsync(ws, 0, (size_t)vl_imp_ws_bytes(n), s);
  if (e != hipSuccess) return vl_set_error(-3, "vl_imp_select: memset: %s", hipGetErrorString(e));
  if (k == 0) {
    e = hipMemcpyAsync(new_mask, mask, (size_t)n * 4, hipMemcpyDeviceToDevice, s);
    if (e != hipSuccess) return vl_set_error(-3, "vl_imp_select: copy: %s", hipGetErrorString(e));
    return 0;
  }
  const unsigned g = 2048;
  hipLaunchKernelGGL(hist_kernel<0>, dim3(g), dim3(256), 0, s, w, mask, (long)n, u);
  hipLaunchKernelGGL(pick_kernel<0>, dim3(1), dim3(256), 0, s, u, (unsigned)k);
  hipLaunchKernelGGL(hist_kernel<1>, dim3(g), dim3(256), 0, s, w, mask, (long)n, u);
  hipLaunchKernelGGL(pick_kernel<1>, dim3(1), dim3(256), 0, s, u, 0u);
  hipLaunchKernelGGL(hist_kernel<2>, dim3(g), dim3(256), 0, s, w, mask, (long)n, u);
  hipLaunchKernelGGL(pick_kernel<2>, dim3(1), dim3(256), 0, s, u, 0u);
  hipLaunchKernelGGL(tie_count_kernel, dim3(nblocks), dim3(256), 0, s, w, mask, (long)n, u);
  hipLaunchKernelGGL(tie_scan_kernel, dim3(1), dim3(1024), 0, s, u, nblocks);
  hipLaunchKernelGGL(write_mask_kernel, dim3(nblocks), dim3(256), 0, s, w, mask, new_mask, (long)n, u);
  VL_CHECK_LAUNCH("vl_imp_select");
  return 0;
}
