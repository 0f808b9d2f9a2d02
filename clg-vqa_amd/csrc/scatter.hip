// vl_scatter_add_det: DETERMINISTIC scatter-add of gradient rows into embedding tables -- the backward of
// nn.Embedding(sparse=False) for the text side of UC2Embeddings (word / position / token-type tables,
// volta/volta/embeddings.py:617-655) and of M3P's `self.embeddings(x)` (m3p_transformer.py:908).
//
// torch's own backward (and the atomic kernels of elementwise.hip) add the rows that hit one table row in whatever order the
// hardware retires the atomics: run-to-run the sums differ in the last bits.  Here the order is fixed:
//   1. (row id, source index) pairs are sorted -- a rank sort: every workgroup holds all keys of its table in LDS (<= 16 384
//      source rows: B*T of every configuration of the path) and each thread counts the keys below its own -- so equal row
//      ids are adjacent, in source order;
//   2. one wave per 32 sorted positions walks them in order and sums every run of equal row ids in registers: a run that lies
//      inside the block is added to its table row by this wave alone (plain read-modify-write, no atomics); a run that
//      crosses a block boundary leaves a partial row (at most a head and a tail per block);
//   3. one wave per block adds up the partials of the run that STARTS in it (tail + the heads of the following blocks, in
//      order) and adds that to the table row.
// Every table row has exactly one owner and one summation order: bit-reproducible, independent of the launch timing.
#include "common.h"
#include "../../include/vlhip.h"

namespace {

constexpr int SC_MAXT = 4;        // tables per call
constexpr int SC_MAXV = 32;       // H <= 64 * SC_MAXV
constexpr unsigned SC_NONE = 0xFFFFFFFFu;

struct ScTable { const int64_t* ids; int kind; float* table; int64_t skip; unsigned char* flags; int T; };
struct ScArgs {
  ScTable t[SC_MAXT]; int n; int R; int Rp; int H; int nblocks;
  const float* dz; unsigned long long* sorted; float* part; int* meta;  // meta[table][block][3] = {head row, head continues, tail row}
};

// (key, source index) pairs of table tb -> s[0, R): the table row in the high word (SC_NONE: receives nothing)
__device__ __forceinline__ void sc_stage_keys(const ScTable& tb, unsigned long long* s, int R, int tid) {
  if (tb.kind == 1) {
    // RoBERTa position ids: (number of non-pad ids in [0, t] of the sample, 0 for a pad) + pad id.  The ids go to LDS with
    // coalesced loads first; then one thread walks one sample there (a running count instead of a re-count per token, LDS
    // reads instead of T global loads one after the other)
    const int T = tb.T;
    for (int i = tid; i < R; i += 256) s[i] = (unsigned long long)tb.ids[i];
    __syncthreads();
    for (int b = tid; b < R / T; b += 256) {
      int cnt = 0;
      for (int t = 0; t < T; ++t) {
        const int i = b * T + t;
        const bool nz = (int64_t)s[i] != tb.skip;
        cnt += nz;
        s[i] = ((unsigned long long)(unsigned)((nz ? cnt : 0) + (int)tb.skip) << 32) | (unsigned)i;
      }
    }
  } else {
    for (int i = tid; i < R; i += 256) {
      const int64_t id = tb.ids[i];
      s[i] = ((unsigned long long)(id != tb.skip ? (unsigned)id : SC_NONE) << 32) | (unsigned)i;
    }
  }
}

// Rank sort: every workgroup stages all R keys of its table in LDS and ranks 64 of them: four threads per key, thread p counting
// the keys j = p (mod 4) below its own (the four lanes of a key read 32 contiguous bytes, the 16 keys of a wave the same ones: a
// broadcast), the four counts added across the lanes; the pair goes to sorted[rank].  O(R^2) compares spread over R / 64
// workgroups (R = 5120: 80 workgroups per table x 1280 compares per thread; 20 workgroups x 5120 took 70 us, a one-workgroup
// bitonic sort of the same keys ~250 us).
__global__ __launch_bounds__(256) void sc_sort_kernel(ScArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  unsigned long long* s = reinterpret_cast<unsigned long long*>(smem_raw);
  const ScTable& tb = a.t[blockIdx.y];
  const int tid = threadIdx.x;
  sc_stage_keys(tb, s, a.R, tid);
  __syncthreads();
  const int i = blockIdx.x * 64 + (tid >> 2), part = tid & 3;
  unsigned long long* out = a.sorted + (long)blockIdx.y * a.Rp;
  const bool real = i < a.R;
  const unsigned long long mine = real ? s[i] : ~0ull;
  int rank = 0;
  if (real) {
    int j = part;
    for (; j + 12 < a.R; j += 16)
      rank += (int)(s[j] < mine) + (int)(s[j + 4] < mine) + (int)(s[j + 8] < mine) + (int)(s[j + 12] < mine);
    for (; j < a.R; j += 4) rank += (int)(s[j] < mine);
  }
  rank += __shfl_xor(rank, 1, 64);
  rank += __shfl_xor(rank, 2, 64);
  if (part != 0 || i >= a.Rp) return;
  if (!real) { out[i] = ~0ull; return; }  // the real pairs take ranks [0, R): the tail is padding
  const unsigned row = (unsigned)(mine >> 32);
  if (row != SC_NONE && tb.flags) tb.flags[row] = 1;  // this table row now carries optimizer state
  out[rank] = mine;
}

// one wave per block of SC_BLK sorted positions.  The rows of a block are fetched G at a time (all G rows' loads in flight,
// then added in order): a row-by-row walk exposes one memory latency per row (measured: 1 ms per step at c2).
constexpr int SC_BLK = 32;
template <int NVT>
__global__ __launch_bounds__(256) void sc_block_kernel(ScArgs a) {
  constexpr int G = NVT <= 12 ? 8 : 2;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int blk = blockIdx.x * 4 + wave, ti = blockIdx.y;
  if (blk >= a.nblocks) return;
  const ScTable& tb = a.t[ti];
  const unsigned long long* keys = a.sorted + (long)ti * a.Rp;
  const int p0 = blk * SC_BLK, nv = a.H / 64;
  const unsigned long long mine = lane < SC_BLK ? keys[p0 + lane] : ~0ull;  // (Rp is a multiple of 64)
  const unsigned prev_row = p0 > 0 ? (unsigned)(keys[p0 - 1] >> 32) : SC_NONE;
  const unsigned next_row = p0 + SC_BLK < a.Rp ? (unsigned)(keys[p0 + SC_BLK] >> 32) : SC_NONE;
  int* meta = a.meta + ((long)ti * a.nblocks + blk) * 3;
  float* part = a.part + ((long)ti * a.nblocks + blk) * 2 * a.H;
  float acc[NVT];
#pragma unroll
  for (int j = 0; j < NVT; ++j) acc[j] = 0.f;
  int head_row = -1, head_through = 0, tail_row = -1;
  int run_start = 0;
  for (int g0 = 0; g0 < SC_BLK; g0 += G) {
    float v[G][NVT], d[G][NVT];
    unsigned rows[G + 1];
    unsigned idx[G];
    int code[G];  // 0: the run goes on | 1: a run that lies inside the block ends here (this wave owns the table row) |
                  // 2: the head run (started in an earlier block) ends or leaves here | 3: the tail run leaves the block
#pragma unroll
    for (int u = 0; u < G; ++u) {
      const unsigned long long kp = __shfl(mine, g0 + u, 64);
      rows[u] = (unsigned)(kp >> 32);
      idx[u] = (unsigned)kp;
    }
    rows[G] = g0 + G < SC_BLK ? (unsigned)(__shfl(mine, g0 + G, 64) >> 32) : next_row;
    if (rows[0] == SC_NONE) break;  // padding / skipped ids sort to the end
    // pass 1, integer logic only: where runs end and who owns them -- so that the table rows this wave will add to can be
    // fetched together with the source rows (a load -> add -> store chain per position exposes a memory latency per row)
#pragma unroll
    for (int u = 0; u < G; ++u) {
      code[u] = 0;
      if (rows[u] == SC_NONE) continue;
      const int p = g0 + u;
      if (rows[u + 1] != rows[u] || p == SC_BLK - 1) {
        const bool started_here = run_start > 0 || prev_row != rows[u];
        const bool ends_here = rows[u + 1] != rows[u];
        code[u] = started_here ? (ends_here ? 1 : 3) : 2;
        if (code[u] == 3) tail_row = (int)rows[u];
        if (code[u] == 2) { head_row = (int)rows[u]; head_through = ends_here ? 0 : 1; }
        run_start = p + 1;
      }
    }
#pragma unroll
    for (int u = 0; u < G; ++u) {
      const float* src = a.dz + (long)idx[u] * a.H + lane;
      const float* own = tb.table + (long)rows[u] * a.H + lane;
#pragma unroll
      for (int j = 0; j < NVT; ++j) {
        v[u][j] = (rows[u] != SC_NONE && j < nv) ? src[64 * j] : 0.f;
        d[u][j] = (code[u] == 1 && j < nv) ? own[64 * j] : 0.f;
      }
    }
    // pass 2: the sums, in source order
#pragma unroll
    for (int u = 0; u < G; ++u) {
      if (rows[u] == SC_NONE) break;
#pragma unroll
      for (int j = 0; j < NVT; ++j) acc[j] += v[u][j];
      if (code[u] == 0) continue;
      if (code[u] == 1) {
        float* dst = tb.table + (long)rows[u] * a.H + lane;
#pragma unroll
        for (int j = 0; j < NVT; ++j)
          if (j < nv) dst[64 * j] = d[u][j] + acc[j];
      } else {
        float* dst = part + (code[u] == 3 ? a.H : 0) + lane;
#pragma unroll
        for (int j = 0; j < NVT; ++j)
          if (j < nv) dst[64 * j] = acc[j];
      }
#pragma unroll
      for (int j = 0; j < NVT; ++j) acc[j] = 0.f;
    }
  }
  if (lane == 0) { meta[0] = head_row; meta[1] = head_through; meta[2] = tail_row; }
}

// one wave per block whose tail run continues into the following blocks: chain length first, then the partial rows G at a
// time
template <int NVT>
__global__ __launch_bounds__(256) void sc_boundary_kernel(ScArgs a) {
  constexpr int G = NVT <= 12 ? 8 : 2;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int blk = blockIdx.x * 4 + wave, ti = blockIdx.y;
  if (blk >= a.nblocks) return;
  const int* meta = a.meta + (long)ti * a.nblocks * 3;
  const int row = meta[blk * 3 + 2];
  if (row < 0) return;
  const int nv = a.H / 64;
  const float* part = a.part + (long)ti * a.nblocks * 2 * a.H;
  int len = 0;  // blocks blk + 1 .. blk + len hold the heads of this run (64 blocks are examined per step, one per lane)
  for (int base = blk + 1; base < a.nblocks; base += 64) {
    const int k = base + lane;
    const bool is_head = k < a.nblocks && meta[k * 3] == row;
    const bool goes_on = is_head && meta[k * 3 + 1] != 0;
    const unsigned long long headm = __ballot(is_head), stopm = ~__ballot(goes_on);
    if (stopm) {  // the chain ends inside this group: at the first block that is not a continuing head
      const int first = __ffsll((long long)stopm) - 1;
      len += first + (int)((headm >> first) & 1ull);
      break;
    }
    len += 64;
  }
  float acc[NVT];
  {
    const float* src = part + ((long)blk * 2 + 1) * a.H + lane;
#pragma unroll
    for (int j = 0; j < NVT; ++j) acc[j] = j < nv ? src[64 * j] : 0.f;
  }
  for (int k0 = 0; k0 < len; k0 += G) {
    float v[G][NVT];
#pragma unroll
    for (int u = 0; u < G; ++u) {
      const float* src = part + (long)(blk + 1 + k0 + u) * 2 * a.H + lane;
#pragma unroll
      for (int j = 0; j < NVT; ++j) v[u][j] = (k0 + u < len && j < nv) ? src[64 * j] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < G; ++u)
#pragma unroll
      for (int j = 0; j < NVT; ++j) acc[j] += v[u][j];  // (+ 0.0f for the slots past the chain: exact)
  }
  float* dst = a.t[ti].table + (long)row * a.H + lane;
#pragma unroll
  for (int j = 0; j < NVT; ++j)
    if (j < nv) dst[64 * j] += acc[j];
}

inline int64_t pow2_at_least(int64_t n) {
  int64_t p = 64;
  while (p < n) p <<= 1;
  return p;
}
inline int64_t align256(int64_t n) { return (n + 255) / 256 * 256; }

}  // namespace

extern "C" int64_t vl_scatter_det_ws_bytes(int64_t n, int64_t R, int64_t H) {
  if (n <= 0 || R <= 0 || H <= 0) return 0;
  const int64_t Rp = pow2_at_least(R), nb = Rp / SC_BLK;
  return align256(n * Rp * 8) + align256(n * nb * 2 * H * 4) + align256(n * nb * 3 * 4);
}

// tab: HOST array of n (<= 4) x VL_SC_FIELDS int64 {ids, kind, table, skip, row_flags, T}; all tables take their rows from dz32
extern "C" int vl_scatter_add_det(const int64_t* tab, int64_t n, const float* dz32, int64_t R, int64_t H, void* ws,
                                  int64_t ws_bytes, void* stream) {
  VL_CHECK_ARG(tab && dz32 && ws && n >= 1 && n <= SC_MAXT && R >= 1 && H >= 64, "vl_scatter_add_det: bad arguments");
  VL_CHECK_ARG(H % 64 == 0 && H <= 64 * SC_MAXV, "vl_scatter_add_det: H must be a multiple of 64, at most %d", 64 * SC_MAXV);
  VL_CHECK_ARG(R <= 16384, "vl_scatter_add_det: at most 16384 source rows (the sort keeps all keys in a workgroup's LDS); got %lld",
               (long long)R);
  VL_CHECK_ARG(ws_bytes >= vl_scatter_det_ws_bytes(n, R, H) && (reinterpret_cast<uintptr_t>(ws) & 255) == 0,
               "vl_scatter_add_det: workspace too small or not 256-byte aligned");
  ScArgs a{};
  a.n = (int)n; a.R = (int)R; a.Rp = (int)pow2_at_least(R); a.H = (int)H; a.nblocks = a.Rp / SC_BLK; a.dz = dz32;
  char* w = reinterpret_cast<char*>(ws);
  a.sorted = reinterpret_cast<unsigned long long*>(w);
  w += align256(n * a.Rp * 8);
  a.part = reinterpret_cast<float*>(w);
  w += align256(n * (int64_t)a.nblocks * 2 * H * 4);
  a.meta = reinterpret_cast<int*>(w);
  for (int i = 0; i < n; ++i) {
    const int64_t* t = tab + i * VL_SC_FIELDS;
    ScTable& s = a.t[i];
    s.ids = reinterpret_cast<const int64_t*>(static_cast<uintptr_t>(t[0])); s.kind = (int)t[1];
    s.table = reinterpret_cast<float*>(static_cast<uintptr_t>(t[2])); s.skip = t[3];
    s.flags = reinterpret_cast<unsigned char*>(static_cast<uintptr_t>(t[4])); s.T = (int)t[5];
    VL_CHECK_ARG(s.ids && s.table && (s.kind == 0 || (s.kind == 1 && s.T >= 1 && R % s.T == 0)),
                 "vl_scatter_add_det: table %d: bad entry", i);
  }
  const size_t lds = (size_t)a.R * 8;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&sc_sort_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       16384 * 8);
    if (e != hipSuccess) return vl_set_error(-3, "vl_scatter_add_det: hipFuncSetAttribute: %s", hipGetErrorString(e));
    attr_set = true;
  }
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(sc_sort_kernel, dim3((unsigned)((a.Rp + 63) / 64), (unsigned)n), dim3(256), lds, s, a);
  const dim3 grid((unsigned)((a.nblocks + 3) / 4), (unsigned)n);
  const int nv = (int)(H / 64);
#define SC_LAUNCH(NVT)                                                          \
  do {                                                                          \
    hipLaunchKernelGGL(sc_block_kernel<NVT>, grid, dim3(256), 0, s, a);         \
    hipLaunchKernelGGL(sc_boundary_kernel<NVT>, grid, dim3(256), 0, s, a);      \
  } while (0)
  if (nv <= 2) SC_LAUNCH(2);
  else if (nv <= 4) SC_LAUNCH(4);
  else if (nv <= 12) SC_LAUNCH(12);
  else SC_LAUNCH(32);
#undef SC_LAUNCH
  VL_CHECK_LAUNCH("vl_scatter_add_det");
  return 0;
}
