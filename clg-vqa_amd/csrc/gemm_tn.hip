// Grouped weight-gradient GEMM: for each problem p,  out_p[M_p, N_p] = A_p^T B_p  (* mask_p)
// with A_p = dY [K, M_p], B_p = X [K, N_p] row-major bf16 activations exactly as they sit in HBM (K = the B*S batch
// rows, shared by all problems of a launch) and fp32 output -- the four dW products of one transformer layer
// (reference: autograd of nn.Linear, volta/encoders.py:229-246, 411-414, 496-501, 553-556) in ONE launch.
//
// Why grouped: one layer has 9 + 27 + 36 + 36 = 108 output tiles of 256 x 256; split-K per product (the earlier
// design) needed 7-28 K-ranges per tile to fill the chip, i.e. 260 MB of fp32 slabs written and re-read per layer and
// 8-K-tile main loops.  Launched together the 108 tiles are 108 long-running workgroups (224 K-tiles each, no slabs,
// no reduce pass) on the side stream, and the dX / LayerNorm / attention kernels of the main stream take the other CUs.
//
// Kernel: the 8-wave ping-pong structure of gemm3_kernel (gemm.hip: counted vmcnt, staggered barriers, one
// workgroup per CU, 128 KiB LDS) with a k-major LDS image: a half-tile is [64 k][128 cols] bf16 (256-B rows), staged
// by LDS-DMA in 4-row units and read with ds_read_b64_tr_b16 (two 4(k) x 16(col) transposed blocks per MFMA
// fragment).  32-byte column blocks are XOR-swizzled with h(k) = (k & 3) | ((k >> 3) & 1) << 2 on the DMA source
// address (undone by the reads): the 8 k-rows a 32-lane half touches land in 8 different 32-B bank groups.
#include "common.h"
#include "../../include/vlhip.h"

namespace {

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;
typedef __attribute__((ext_vector_type(4))) short s16x4;

__device__ __forceinline__ bf16x8 tr_frag(const unsigned char* p0, const unsigned char* p1) {
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p0);
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p1);
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8, v);
}

constexpr int MAX_PROBS = 8;
struct TnProblem {
  const bf16_raw* a; const bf16_raw* b; float* out; const float* mask;
  long lda, ldb, ldo;
  int M, N, tiles_n, tile0;  // tile0: first tile id of this problem in the launch's tile list
};
struct TnArgs {
  TnProblem p[MAX_PROBS];
  int nprob, tiles, splits, k_len, K;
};

template <int N> __device__ __forceinline__ void wait_vmcnt() {
  static_assert(N == 0 || N == 4 || N == 8, "unsupported count");
  if (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  if (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
}

__global__ __launch_bounds__(512) void gemm_tn_grouped_kernel(TnArgs g) {
  constexpr int HALF = 16384, STAGE = 65536, ROWB = 256;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const bool late = wave >= 4;

  // workgroups resident on one XCD (bid % 8) take a contiguous chunk of the (split, problem, tile_m, tile_n) list:
  // tiles that share dY columns (same tile_m) or a K-range sit behind the same L2
  const int nwg = g.tiles * g.splits;
  const int bid = blockIdx.x;
  const int xcd = bid & 7, q8 = nwg >> 3, r8 = nwg & 7;
  const int swz = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int split = swz / g.tiles, tile = swz - split * g.tiles;
  int pi = 0;
#pragma unroll
  for (int i = 1; i < MAX_PROBS; ++i)
    if (i < g.nprob && tile >= g.p[i].tile0) pi = i;
  const TnProblem& P = g.p[pi];
  const int lt = tile - P.tile0;
  const int tm = lt / P.tiles_n, tn = lt - tm * P.tiles_n;
  const int row0 = tm * 256, col0 = tn * 256;
  const int kbeg = split * g.k_len;
  const int kend = min(g.K, kbeg + g.k_len);
  const int nk = (kend - kbeg) >> 6;

  // DMA sources: half-tile x in {A0, A1, B0, B1}; unit u = wave + 8*j = k-rows 4u..4u+3; lane -> (k-row, 16-B chunk)
  const bf16_raw* src[4][2];
#pragma unroll
  for (int x = 0; x < 4; ++x)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const bool isB = x >= 2;
      const int k = 4 * (wave + 8 * j) + (lane >> 4);
      const int c = lane & 15;
      const int lb = (c >> 1) ^ ((k & 3) | (((k >> 3) & 1) << 2));  // logical 32-B block landing at physical c >> 1
      int gcol = (isB ? col0 : row0) + (x & 1) * 128 + lb * 16 + (c & 1) * 8;
      const int lim = (isB ? P.N : P.M) - 8;
      gcol = gcol < lim ? gcol : lim;  // columns past the edge re-read valid data; their products are never stored
      src[x][j] = (isB ? P.b : P.a) + (long)(kbeg + k) * (isB ? P.ldb : P.lda) + gcol;
    }
  const long stepA = 64L * P.lda, stepB = 64L * P.ldb;
#define TN_ISSUE(x, kt)                                                                                          \
  do {                                                                                                           \
    _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_)                                                             \
        __builtin_amdgcn_global_load_lds((glb_ptr_t)(src[x][j_] + (long)(kt) * ((x) >= 2 ? stepB : stepA)),     \
                                         (lds_ptr_t)(smem + ((kt) & 1) * STAGE + (x) * HALF + (wave + 8 * j_) * 1024), \
                                         16, 0, 0);                                                              \
  } while (0)
#define TN_WAIT(issued)                                                                                          \
  do {                                                                                                           \
    if (issued) wait_vmcnt<8>();                                                                                 \
    else wait_vmcnt<0>();                                                                                        \
  } while (0)

  f32x4 acc[2][2][4][2];
#pragma unroll
  for (int a_ = 0; a_ < 2; ++a_)
#pragma unroll
    for (int b_ = 0; b_ < 2; ++b_)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[a_][b_][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // transposed-read lane geometry: group gq = lane>>4 owns k = 8gq..8gq+7 of a 32-deep MFMA step; inside the group
  // lane 4q+pp addresses k-row q, columns 4pp..4pp+3 of the 4 x 16 block
  const int gq = lane >> 4, qq = (lane & 15) >> 2, pp = lane & 3;
  const int hq = qq | ((gq & 1) << 2);
  const int kbase = (8 * gq + qq) * ROWB + pp * 8;
  int a_t[4], b_t[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) a_t[i] = kbase + (((wr * 4 + i) ^ hq) << 5);
#pragma unroll
  for (int j = 0; j < 2; ++j) b_t[j] = kbase + (((wc * 2 + j) ^ hq) << 5);

  bf16x8 fa[4][2], fb0[2][2], fb1[2][2];
#define TN_READ_A(st, qm)                                                                                        \
  _Pragma("unroll") for (int i = 0; i < 4; ++i) _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) {              \
    const unsigned char* ap_ = (st) + (qm) * HALF + kk * 32 * ROWB + a_t[i];                                     \
    fa[i][kk] = tr_frag(ap_, ap_ + 4 * ROWB);                                                                    \
  }
#define TN_READ_B(st, qn, fb)                                                                                    \
  _Pragma("unroll") for (int j = 0; j < 2; ++j) _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) {              \
    const unsigned char* bp_ = (st) + (2 + (qn)) * HALF + kk * 32 * ROWB + b_t[j];                               \
    fb[j][kk] = tr_frag(bp_, bp_ + 4 * ROWB);                                                                    \
  }
#define TN_MFMA(qm, qn, fb)                                                                                      \
  do {                                                                                                           \
    __builtin_amdgcn_s_barrier();                                                                                \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                           \
    __builtin_amdgcn_sched_barrier(0);                                                                           \
    __builtin_amdgcn_s_setprio(1);                                                                               \
    _Pragma("unroll") for (int kk = 0; kk < 2; ++kk)                                                             \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                                \
    _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                                \
        acc[qm][qn][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j][kk], fa[i][kk], acc[qm][qn][i][j], 0, 0, 0); \
    __builtin_amdgcn_s_setprio(0);                                                                               \
    __builtin_amdgcn_sched_barrier(0);                                                                           \
    __builtin_amdgcn_s_barrier();                                                                                \
  } while (0)

  // prologue: K-tile 0 complete, A0 / B0 of K-tile 1 in flight (schedule and hazards: see gemm3_kernel)
  TN_ISSUE(0, 0); TN_ISSUE(2, 0); TN_ISSUE(3, 0); TN_ISSUE(1, 0);
  if (nk > 1) {
    TN_ISSUE(0, 1); TN_ISSUE(2, 1);
    wait_vmcnt<4>();
  } else {
    wait_vmcnt<0>();
  }
  __builtin_amdgcn_s_barrier();
  if (late) __builtin_amdgcn_s_barrier();

  for (int kt = 0; kt < nk; ++kt) {
    const unsigned char* st = smem + (kt & 1) * STAGE;
    const bool n1 = kt + 1 < nk, n2 = kt + 2 < nk;
    TN_READ_B(st, 0, fb0);
    __builtin_amdgcn_sched_barrier(0);
    TN_READ_A(st, 0);
    if (n1) TN_ISSUE(3, kt + 1);
    TN_WAIT(n1);
    TN_MFMA(0, 0, fb0);
    TN_READ_B(st, 1, fb1);
    if (n1) TN_ISSUE(1, kt + 1);
    TN_WAIT(n1);
    TN_MFMA(0, 1, fb1);
    TN_READ_A(st, 1);
    if (n2) TN_ISSUE(0, kt + 2);
    TN_WAIT(n2);
    TN_MFMA(1, 1, fb1);
    if (n2) TN_ISSUE(2, kt + 2);
    TN_WAIT(n2);
    TN_MFMA(1, 0, fb0);
  }
  if (!late) __builtin_amdgcn_s_barrier();
#undef TN_ISSUE
#undef TN_WAIT
#undef TN_READ_A
#undef TN_READ_B
#undef TN_MFMA

  // epilogue.  D^T layout: lane&15 -> m inside the 16-row tile, 4*(lane>>4) + reg -> n inside the 16-col tile.
  // splits == 2: both K-halves add into the zero-initialised output; 0 + a + b is order independent in fp32, so the
  // result is deterministic (and a {0,1} mask distributes over the sum exactly).
  const bool vec = ((P.ldo & 3) == 0) && ((reinterpret_cast<uintptr_t>(P.out) & 15) == 0) &&
                   (!P.mask || (reinterpret_cast<uintptr_t>(P.mask) & 15) == 0);
#pragma unroll
  for (int qm = 0; qm < 2; ++qm)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = row0 + qm * 128 + wr * 64 + i * 16 + (lane & 15);
      if (m >= P.M) continue;
#pragma unroll
      for (int qn = 0; qn < 2; ++qn)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int n0 = col0 + qn * 128 + wc * 32 + j * 16 + 4 * (lane >> 4);
          if (n0 >= P.N) continue;
          f32x4 v = acc[qm][qn][i][j];
          const long o = (long)m * P.ldo + n0;
          if (vec && n0 + 3 < P.N) {
            if (P.mask) {
              const float4 mk = *reinterpret_cast<const float4*>(P.mask + o);
              v[0] *= mk.x; v[1] *= mk.y; v[2] *= mk.z; v[3] *= mk.w;
            }
            if (g.splits == 1) {
              *reinterpret_cast<float4*>(P.out + o) = make_float4(v[0], v[1], v[2], v[3]);
            } else {
#pragma unroll
              for (int r = 0; r < 4; ++r) atomicAdd(P.out + o + r, v[r]);
            }
          } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              if (n0 + r >= P.N) break;
              const float x = P.mask ? v[r] * P.mask[o + r] : v[r];
              if (g.splits == 1) P.out[o + r] = x;
              else atomicAdd(P.out + o + r, x);
            }
          }
        }
    }
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

}  // namespace

// probs: nprob x VL_TN_FIELDS int64 values {a, lda, b, ldb, out, ldo, mask, M, N, 0} (host memory).
extern "C" int vl_gemm_tn_grouped(const int64_t* probs, int64_t nprob, int64_t K, int64_t splits, void* stream) {
  VL_CHECK_ARG(probs && nprob >= 1 && nprob <= MAX_PROBS, "vl_gemm_tn_grouped: 1..%d problems per launch", MAX_PROBS);
  VL_CHECK_ARG(K > 0 && K < (1 << 30) && (K % 64) == 0, "vl_gemm_tn_grouped: K must be a positive multiple of 64 (got %lld)",
               (long long)K);
  TnArgs g{};
  int tiles = 0;
  for (int i = 0; i < nprob; ++i) {
    const int64_t* f = probs + i * VL_TN_FIELDS;
    TnProblem& p = g.p[i];
    p.a = (const bf16_raw*)f[0]; p.lda = f[1]; p.b = (const bf16_raw*)f[2]; p.ldb = f[3];
    p.out = (float*)f[4]; p.ldo = f[5]; p.mask = (const float*)f[6];
    VL_CHECK_ARG(f[7] >= 16 && f[8] >= 16 && f[7] < (1 << 30) && f[8] < (1 << 30) && (f[7] % 8) == 0 && (f[8] % 8) == 0,
                 "vl_gemm_tn_grouped: problem %d: M, N must be multiples of 8 and >= 16 (M=%lld N=%lld)", i,
                 (long long)f[7], (long long)f[8]);
    p.M = (int)f[7]; p.N = (int)f[8];
    VL_CHECK_ARG(p.a && p.b && p.out, "vl_gemm_tn_grouped: problem %d: null pointer", i);
    VL_CHECK_ARG(p.lda >= p.M && p.ldb >= p.N && p.ldo >= p.N && (p.lda % 8) == 0 && (p.ldb % 8) == 0 &&
                 aligned16(p.a) && aligned16(p.b),
                 "vl_gemm_tn_grouped: problem %d: leading dimensions must be multiples of 8 and >= the row length, "
                 "operands 16-byte aligned", i);
    p.tiles_n = (p.N + 255) / 256;
    p.tile0 = tiles;
    tiles += ((p.M + 255) / 256) * p.tiles_n;
  }
  if (splits <= 0) splits = (tiles < 80 && (K % 128) == 0) ? 2 : 1;  // 2 K-halves when one round would leave most CUs idle
  VL_CHECK_ARG(splits == 1 || (splits == 2 && (K % 128) == 0),
               "vl_gemm_tn_grouped: splits must be 1, or 2 with K %% 128 == 0 (more addends would make the atomic sum order dependent)");
  g.nprob = (int)nprob; g.tiles = tiles; g.splits = (int)splits; g.K = (int)K; g.k_len = (int)(K / splits);
  hipStream_t s = (hipStream_t)stream;
  if (splits == 2) {
    for (int i = 0; i < nprob; ++i) {
      const TnProblem& p = g.p[i];
      hipError_t e = hipMemset2DAsync(p.out, (size_t)p.ldo * 4, 0, (size_t)p.N * 4, (size_t)p.M, s);
      if (e != hipSuccess) return vl_set_error(-3, "vl_gemm_tn_grouped: memset: %s", hipGetErrorString(e));
    }
  }
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tn_grouped_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    if (e != hipSuccess) return vl_set_error(-3, "vl_gemm_tn_grouped: hipFuncSetAttribute: %s", hipGetErrorString(e));
    attr_set = true;
  }
  hipLaunchKernelGGL(gemm_tn_grouped_kernel, dim3(tiles * (int)splits), dim3(512), 131072, s, g);
  VL_CHECK_LAUNCH("vl_gemm_tn_grouped");
  return 0;
}
