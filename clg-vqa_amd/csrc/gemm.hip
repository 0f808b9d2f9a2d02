// vl_gemm_nt: C[M,N] = A[M,K] * B[N,K]^T on the bf16 MFMA pipe (v_mfma_f32_16x16x32_bf16, fp32 accumulate).
//
// This one kernel carries ~98.8 % of the step's FLOPs: the reference issues them as
// eager fp32 nn.Linear GEMMs (volta/volta/encoders.py:229-246 Q/K/V, :411-414 out-proj, :496-501 FFN1,
// :553-556 FFN2) and autograd's dX / dW products.
//
// Precision modes (DESIGN.md "Precision"):
//   passes = 1   plain bf16 operands                       -> used for the backward products
//   passes = 3   operands given as (hi, lo) bf16 pairs with x ~= hi + lo (16 significant bits);
//                acc += Alo*Bhi + Ahi*Blo + Ahi*Bhi        -> fp32-grade forward (logits within 1e-3)
//
// Tiling: 128x128 output tile per 256-thread workgroup (4 waves as 2x2, 64x64 per wave = 4x4 MFMA
// tiles), BK = 64.  Operands are K-contiguous; tiles are staged global -> registers -> LDS with an XOR
// swizzle of the 16-byte chunk index ((row>>1)&7) so that every ds_read_b128 fragment read is
// bank-conflict free on the 64-bank LDS; the next tile's global loads are issued before the MFMA block
// of the current one.  blockIdx is remapped so that the workgroups resident on one XCD (private L2)
// walk neighbouring tiles of the same A row-panel.
#include "common.h"
#include "../../include/vlhip.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;  // 16 KiB per operand tile

struct GemmArgs {
  const bf16_raw* a_hi; const bf16_raw* a_lo; const bf16_raw* b_hi; const bf16_raw* b_lo;
  long lda, ldb;
  int M, N, K;
  const float* bias; const float* resid; float* out32; long ldc;
  bf16_raw* out_hi; bf16_raw* out_lo; bf16_raw* aux16; long ld16;
  int tiles_m, tiles_n;
  int k_len; long slab_stride;  // split-K: blockIdx.y owns k in [y*k_len, (y+1)*k_len) and writes slab y of out32
};

__device__ __forceinline__ int lds_off(int row, int kc) { return row * 128 + ((kc ^ ((row >> 1) & 7)) << 4); }

__device__ __forceinline__ uint4 load_chunk(const bf16_raw* base, long ld, int row, int nrows, int k, int K) {
  if (row < nrows && k < K) return *reinterpret_cast<const uint4*>(base + (long)row * ld + k);
  return make_uint4(0u, 0u, 0u, 0u);
}

template <int NSPLIT, int EPI>
__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* sA_hi = smem;
  unsigned char* sB_hi = smem + TILE_BYTES;
  unsigned char* sA_lo = smem + 2 * TILE_BYTES;
  unsigned char* sB_lo = smem + 3 * TILE_BYTES;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  // XCD-aware (bijective) remap of the linear block id -> (tile_m, tile_n)
  const int nwg = p.tiles_m * p.tiles_n;
  const int bid = blockIdx.x;
  const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
  const int swz = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  const int tm = swz / p.tiles_n, tn = swz - tm * p.tiles_n;
  const int row0 = tm * BM, col0 = tn * BN;

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int kbeg = blockIdx.y * p.k_len;
  const int kend = min(p.K, kbeg + p.k_len);
  uint4 ra_hi[4], rb_hi[4], ra_lo[4], rb_lo[4];
  auto load_tile = [&](int k0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = tid + i * 256;
      const int row = c >> 3, k = k0 + ((c & 7) << 3);
      ra_hi[i] = load_chunk(p.a_hi, p.lda, row0 + row, p.M, k, kend);
      rb_hi[i] = load_chunk(p.b_hi, p.ldb, col0 + row, p.N, k, kend);
      if (NSPLIT == 3) {
        ra_lo[i] = load_chunk(p.a_lo, p.lda, row0 + row, p.M, k, kend);
        rb_lo[i] = load_chunk(p.b_lo, p.ldb, col0 + row, p.N, k, kend);
      }
    }
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = tid + i * 256;
      const int off = lds_off(c >> 3, c & 7);
      *reinterpret_cast<uint4*>(sA_hi + off) = ra_hi[i];
      *reinterpret_cast<uint4*>(sB_hi + off) = rb_hi[i];
      if (NSPLIT == 3) {
        *reinterpret_cast<uint4*>(sA_lo + off) = ra_lo[i];
        *reinterpret_cast<uint4*>(sB_lo + off) = rb_lo[i];
      }
    }
  };

  const int nk = (kend - kbeg + BK - 1) / BK;
  load_tile(kbeg);
  const int frow = lane & 15, fk = lane >> 4;
  for (int kt = 0; kt < nk; ++kt) {
    __syncthreads();  // every wave is done reading the previous tile
    store_tile();
    __syncthreads();
    if (kt + 1 < nk) load_tile(kbeg + (kt + 1) * BK);  // in flight under the MFMA block below
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8 a_h[4], b_h[4], a_l[4], b_l[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int ra = wm * 64 + i * 16 + frow, rb = wn * 64 + i * 16 + frow;
        const int oa = lds_off(ra, kk * 4 + fk), ob = lds_off(rb, kk * 4 + fk);
        a_h[i] = *reinterpret_cast<const bf16x8*>(sA_hi + oa);
        b_h[i] = *reinterpret_cast<const bf16x8*>(sB_hi + ob);
        if (NSPLIT == 3) {
          a_l[i] = *reinterpret_cast<const bf16x8*>(sA_lo + oa);
          b_l[i] = *reinterpret_cast<const bf16x8*>(sB_lo + ob);
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (NSPLIT == 3) {  // small cross terms first, dominant term last
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_l[i], b_h[j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_h[i], b_l[j], acc[i][j], 0, 0, 0);
          }
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_h[i], b_h[j], acc[i][j], 0, 0, 0);
        }
    }
  }

  // epilogue: C/D layout of 16x16 MFMA: col = lane&15, row = 4*(lane>>4) + reg
  float* out32 = p.out32 + (long)blockIdx.y * p.slab_stride;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = col0 + wn * 64 + j * 16 + (lane & 15);
      if (n >= p.N) continue;
      const float bv = p.bias ? p.bias[n] : 0.f;
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int m = row0 + wm * 64 + i * 16 + 4 * (lane >> 4) + rr;
        if (m >= p.M) continue;
        float v = acc[i][j][rr] + bv;
        if (EPI == VL_EPI_F32) {
          if (p.resid) v += p.resid[(long)m * p.ldc + n];
          out32[(long)m * p.ldc + n] = v;
        } else if (EPI == VL_EPI_GELU_SPLIT) {
          p.aux16[(long)m * p.ld16 + n] = f32_to_bf16(v);
          bf16_raw hi, lo;
          split_bf16(gelu_erf(v), hi, lo);
          p.out_hi[(long)m * p.ld16 + n] = hi;
          p.out_lo[(long)m * p.ld16 + n] = lo;
        } else if (EPI == VL_EPI_DGELU_BF16) {
          const float u = bf16_to_f32(p.aux16[(long)m * p.ld16 + n]);
          p.out_hi[(long)m * p.ld16 + n] = f32_to_bf16(v * gelu_erf_grad(u));
        } else if (EPI == VL_EPI_BF16) {
          p.out_hi[(long)m * p.ld16 + n] = f32_to_bf16(v);
        } else {  // VL_EPI_SPLIT
          bf16_raw hi, lo;
          split_bf16(v, hi, lo);
          p.out_hi[(long)m * p.ld16 + n] = hi;
          p.out_lo[(long)m * p.ld16 + n] = lo;
        }
      }
    }
}

// out[i] = sum_s ws[s][i]  (float4 granularity)
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ ws, int splits, long n4,
                                                            float* __restrict__ out) {
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    float4 acc = reinterpret_cast<const float4*>(ws)[i];
    for (int s = 1; s < splits; ++s) {
      const float4 v = reinterpret_cast<const float4*>(ws)[(long)s * n4 + i];
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    reinterpret_cast<float4*>(out)[i] = acc;
  }
}

template <int NSPLIT, int EPI>
int launch(const GemmArgs& a, hipStream_t stream, int splits = 1) {
  const size_t lds = (NSPLIT == 3 ? 4 : 2) * TILE_BYTES;
  static bool attr_set = false;  // per instantiation; idempotent, so a race only repeats the call
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_kernel<NSPLIT, EPI>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return vl_set_error(-3, "vl_gemm_nt: hipFuncSetAttribute: %s", hipGetErrorString(e));
    attr_set = true;
  }
  hipLaunchKernelGGL((gemm_nt_kernel<NSPLIT, EPI>), dim3(a.tiles_m * a.tiles_n, splits), dim3(256), lds, stream, a);
  VL_CHECK_LAUNCH("vl_gemm_nt");
  return 0;
}

template <int NSPLIT>
int dispatch_epi(int epi, const GemmArgs& a, hipStream_t s) {
  switch (epi) {
    case VL_EPI_F32: return launch<NSPLIT, VL_EPI_F32>(a, s);
    case VL_EPI_GELU_SPLIT: return launch<NSPLIT, VL_EPI_GELU_SPLIT>(a, s);
    case VL_EPI_DGELU_BF16: return launch<NSPLIT, VL_EPI_DGELU_BF16>(a, s);
    case VL_EPI_BF16: return launch<NSPLIT, VL_EPI_BF16>(a, s);
    case VL_EPI_SPLIT: return launch<NSPLIT, VL_EPI_SPLIT>(a, s);
  }
  return vl_set_error(-1, "vl_gemm_nt: unknown epilogue %d", epi);
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

}  // namespace

extern "C" int vl_gemm_nt(const void* a_hi, const void* a_lo, int64_t lda, const void* b_hi, const void* b_lo,
                          int64_t ldb, int64_t M, int64_t N, int64_t K, int passes, int epilogue,
                          const float* bias, const float* resid32, float* out32, int64_t ldc, void* out_hi,
                          void* out_lo, void* aux16, int64_t ld16, void* stream) {
  VL_CHECK_ARG(passes == 1 || passes == 3, "vl_gemm_nt: passes must be 1 or 3 (got %d)", passes);
  VL_CHECK_ARG(M > 0 && N > 0 && K > 0 && M < (1 << 30) && N < (1 << 30) && K < (1 << 30),
               "vl_gemm_nt: bad dims M=%lld N=%lld K=%lld", (long long)M, (long long)N, (long long)K);
  VL_CHECK_ARG(a_hi && b_hi, "vl_gemm_nt: null operand");
  VL_CHECK_ARG(passes == 1 || (a_lo && b_lo), "vl_gemm_nt: passes=3 needs the lo halves of both operands");
  VL_CHECK_ARG((K & 7) == 0 && (lda & 7) == 0 && (ldb & 7) == 0 && lda >= K && ldb >= K,
               "vl_gemm_nt: K, lda, ldb must be multiples of 8 with ld >= K (K=%lld lda=%lld ldb=%lld)",
               (long long)K, (long long)lda, (long long)ldb);
  VL_CHECK_ARG(aligned16(a_hi) && aligned16(b_hi) && aligned16(a_lo) && aligned16(b_lo),
               "vl_gemm_nt: operand pointers must be 16-byte aligned");
  if (epilogue == VL_EPI_F32) {
    VL_CHECK_ARG(out32 && ldc >= N, "vl_gemm_nt: F32 epilogue needs out32 with ldc >= N");
  } else {
    VL_CHECK_ARG(out_hi && ld16 >= N, "vl_gemm_nt: 16-bit epilogue needs out_hi with ld16 >= N");
    VL_CHECK_ARG(epilogue != VL_EPI_GELU_SPLIT || (out_lo && aux16 && bias), "vl_gemm_nt: GELU_SPLIT needs out_lo, aux16, bias");
    VL_CHECK_ARG(epilogue != VL_EPI_SPLIT || out_lo, "vl_gemm_nt: SPLIT needs out_lo");
    VL_CHECK_ARG(epilogue != VL_EPI_DGELU_BF16 || aux16, "vl_gemm_nt: DGELU needs aux16");
    VL_CHECK_ARG(!resid32, "vl_gemm_nt: resid32 only with the F32 epilogue");
  }
  GemmArgs a;
  a.a_hi = (const bf16_raw*)a_hi; a.a_lo = (const bf16_raw*)a_lo;
  a.b_hi = (const bf16_raw*)b_hi; a.b_lo = (const bf16_raw*)b_lo;
  a.lda = lda; a.ldb = ldb; a.M = (int)M; a.N = (int)N; a.K = (int)K;
  a.bias = bias; a.resid = resid32; a.out32 = out32; a.ldc = ldc;
  a.out_hi = (bf16_raw*)out_hi; a.out_lo = (bf16_raw*)out_lo; a.aux16 = (bf16_raw*)aux16; a.ld16 = ld16;
  a.tiles_m = (int)((M + BM - 1) / BM); a.tiles_n = (int)((N + BN - 1) / BN);
  a.k_len = (int)K; a.slab_stride = 0;
  hipStream_t s = (hipStream_t)stream;
  return passes == 3 ? dispatch_epi<3>(epilogue, a, s) : dispatch_epi<1>(epilogue, a, s);
}

// Split-K variant for the weight-gradient products dW[N_out,K_in] = dY^T X, whose reduction dimension is the
// B*S = 14336 rows of the batch while the output is only 36..144 tiles: `splits` workgroups per output tile each
// reduce a K-range into their own fp32 slab (plain stores), then one streaming pass sums the slabs (deterministic;
// float atomics would run at ~1.3 TB/s chip-wide and are slower here).
extern "C" int64_t vl_gemm_splitk_ws_floats(int64_t M, int64_t N, int64_t splits) { return M * N * splits; }

extern "C" int vl_gemm_nt_splitk(const void* a_hi, int64_t lda, const void* b_hi, int64_t ldb, int64_t M, int64_t N,
                                 int64_t K, int64_t splits, float* ws, float* out32, void* stream) {
  VL_CHECK_ARG(M > 0 && N > 0 && K > 0 && M < (1 << 30) && N < (1 << 30) && K < (1 << 30) && splits >= 1 && splits <= 64,
               "vl_gemm_nt_splitk: bad dims / splits");
  VL_CHECK_ARG(a_hi && b_hi && out32 && (splits == 1 || ws), "vl_gemm_nt_splitk: null pointer");
  VL_CHECK_ARG((K & 7) == 0 && (lda & 7) == 0 && (ldb & 7) == 0 && lda >= K && ldb >= K && aligned16(a_hi) && aligned16(b_hi),
               "vl_gemm_nt_splitk: K, lda, ldb must be multiples of 8, pointers 16-byte aligned");
  VL_CHECK_ARG(splits == 1 || ((M * N) % 4 == 0 && aligned16(ws) && aligned16(out32)),
               "vl_gemm_nt_splitk: M*N must be a multiple of 4 and ws/out32 16-byte aligned");
  GemmArgs a{};
  a.a_hi = (const bf16_raw*)a_hi; a.b_hi = (const bf16_raw*)b_hi;
  a.lda = lda; a.ldb = ldb; a.M = (int)M; a.N = (int)N; a.K = (int)K;
  a.ldc = N;
  a.tiles_m = (int)((M + BM - 1) / BM); a.tiles_n = (int)((N + BN - 1) / BN);
  int64_t k_len = (K + splits - 1) / splits;
  k_len = (k_len + BK - 1) / BK * BK;
  const int eff = (int)((K + k_len - 1) / k_len);
  a.k_len = (int)k_len;
  hipStream_t s = (hipStream_t)stream;
  if (eff == 1) {
    a.out32 = out32; a.slab_stride = 0;
    return launch<1, VL_EPI_F32>(a, s, 1);
  }
  a.out32 = ws; a.slab_stride = M * N;
  if (int rc = launch<1, VL_EPI_F32>(a, s, eff)) return rc;
  const long n4 = (long)(M * N / 4);
  long g = (n4 + 255) / 256;
  if (g > 2048) g = 2048;
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)g), dim3(256), 0, s, ws, eff, n4, out32);
  VL_CHECK_LAUNCH("vl_gemm_nt_splitk(reduce)");
  return 0;
}
