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
// Two kernels share the epilogues:
//  * gemm2_kernel (fast path, K % 64 == 0 / % 32 for 3-pass, M >= 256): 256 x BN output tile (BN = 256 / 192 / 128
//    picked per shape so the tile count fills the 256 CUs), 512 threads = 8 waves as 2(M) x 4(N), operand tiles
//    DMA'd global -> LDS with global_load_lds_dwordx4 (no VGPR staging, no ds_write), two LDS stages so the next
//    K-step's DMA flies under the current step's MFMAs, one barrier per K-step.  The LDS image is lane-linear per
//    DMA instruction (1 KiB = 8 rows x 128 B); the bank-conflict-free XOR swizzle is applied to the per-lane SOURCE
//    address and undone by the ds_read_b128 fragment reads.  For the 3-pass mode a 128-byte LDS row holds
//    [hi k0..31 | lo k0..31], so both modes share one image / one set of addresses.
//  * gemm_nt_kernel (generic path, any K multiple of 8): described next.
//
// Tiling (generic path): 128x128 output tile per 256-thread workgroup (4 waves as 2x2, 64x64 per wave = 4x4 MFMA
// tiles), BK = 64.  Operands are K-contiguous; tiles are staged global -> registers -> LDS with an XOR
// swizzle of the 16-byte chunk index ((row>>1)&7) so that every ds_read_b128 fragment read is
// bank-conflict free on the 64-bank LDS; the next tile's global loads are issued before the MFMA block
// of the current one.  blockIdx is remapped so that the workgroups resident on one XCD (private L2)
// walk neighbouring tiles of the same A row-panel.
#include <type_traits>

// Only the explicit fmaf()s fuse: which a * b + c the optimizer contracts differs between the unrolled instances of an
// epilogue and between tile configurations, and the epilogues of all configurations are tested bit-identical.
#ifndef VL_GEMM_CONTRACT_FAST  // (A/B builds only)
#pragma clang fp contract(off)
#endif
#include "common.h"
#include "../../include/vlhip.h"

// (timing experiment, WRONG RESULTS: -DVL_EXP_NO_STORE keeps the epilogue's arithmetic and loads but never executes its stores)
#ifdef VL_EXP_NO_STORE
#define VL_EXP_ST if (p.K == -1)
#else
#define VL_EXP_ST
#endif

namespace {

// Kernel / tile selection is a per-call argument (`tile` of vl_gemm_nt_ex; the library holds no mutable state):
//   0 automatic | 2 / 3 / 5 ping-pong kernel with 256x256 / 256x192 / 224x256 tiles | 4 ping-pong, cost model only
//   6 single-barrier kernel (automatic width) | 7 generic 128x128 kernel | 128 / 192 / 256 single-barrier kernel of that width
//   8 small-M path (64x64 tiles + split K through the caller's workspace); automatic for shapes whose big tiles would
//     leave 3/4 of the chip idle, when the caller passes a workspace

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;  // 16 KiB per operand tile

struct GemmArgs {
  const bf16_raw* a_hi; const bf16_raw* a_lo; const bf16_raw* b_hi; const bf16_raw* b_lo;
  long lda, ldb;
  int M, N, K;
  const float* bias; const float* resid; float* out32; long ldc;
  bf16_raw* out_hi; bf16_raw* out_lo; bf16_raw* aux16; long ld16;
  int tiles_m, tiles_n;
  int vec8;                     // 1: additionally 16-byte bf16 epilogue accesses are legal (ld16 % 8 == 0, 16-B pointers)
  int vec;                      // 1: leading dimensions / pointers allow the 16-byte (fp32) / 8-byte (bf16) epilogue
  int splits;                   // TN kernel: number of K-ranges (1-D grid over splits x tiles)
  int k_len; long slab_stride;  // split-K: blockIdx.y owns k in [y*k_len, (y+1)*k_len) and writes slab y of out32
  int tile;                     // host-side kernel / tile selection (see above); not read by the kernels
  float* ws; long ws_floats;    // host-side: caller-owned workspace of the small-M path (may be NULL)
  int persist;                  // host-side: > 0 = ping-pong kernel in its persistent form on this many workgroups
  // 16-bit epilogues: additionally store out_hi in the K-major blocked layout of the weight-gradient GEMM
  // (img[((m >> 6) * img_n + n) * 64 + (m & 63)], dw.hip) -- the producer writes the image, no re-layout pass reads the
  // row-major copy back; cs (ping-pong kernel only): fp32 column sums of the rounded out_hi values per (tile, M half,
  // wave row) -> cs[(tm * 2 + qm) * WR + wr][n], the bias-gradient partials the re-layout pass used to produce
  bf16_raw* img; long img_n; float* cs;
};

__device__ __forceinline__ int lds_off(int row, int kc) { return row * 128 + ((kc ^ ((row >> 1) & 7)) << 4); }

__device__ __forceinline__ uint4 load_chunk(const bf16_raw* base, long ld, int row, int nrows, int k, int K) {
  if (row < nrows && k < K) return *reinterpret_cast<const uint4*>(base + (long)row * ld + k);
  return make_uint4(0u, 0u, 0u, 0u);
}

// K-major image store of one row's consecutive columns: column n + t of row m lands 128 B after column n + t - 1; the 16
// lanes that share lane >> 4 hold 16 consecutive rows of the same columns, i.e. one 32-byte segment per store and column
template <int NV>
__device__ __forceinline__ void image_store(const GemmArgs& p, int m, int n0, const bf16_raw* v) {
  bf16_raw* dst = p.img + ((long)(m >> 6) * p.img_n + n0) * 64 + (m & 63);
#pragma unroll
  for (int t = 0; t < NV; ++t)
    if (n0 + t < p.N) dst[t * 64] = v[t];
}

// Epilogue for 4 consecutive output columns n0..n0+3 of row m (the MFMA is issued as D^T = B.A^T so that a lane's
// 4 accumulator registers are 4 consecutive n: 16-byte fp32 / 8-byte bf16 accesses instead of scalar ones).
// want_cs: csum (4 accumulators of the caller) += the rounded out_hi values (column sums for the bias gradient); the
// accumulators are passed by reference with constant indices only, so they stay in registers
// lcol (may be NULL) / lpitch: LDS staging of the K-major image (ping-pong kernel): element t goes to lcol[t * lpitch];
// without it the image is written by 2-byte global stores (correct on every kernel, slow: 32-byte segments)
template <int EPI>
// has_bq / bq0 (bq1): the bias of these columns, preloaded by the caller (the ping-pong kernel fetches a wave's bias values once
// per M half instead of once per 16-row tile: a dependent load per store group otherwise)
__device__ __forceinline__ void epilogue_store4(const GemmArgs& p, float* out32, int m, int n0, f32x4 v, bool want_cs,
                                                float (&csum)[4], bf16_raw* lcol = nullptr, int lpitch = 0,
                                                bool has_bq = false, float4 bq0 = float4{0.f, 0.f, 0.f, 0.f},
                                                bool has_rq = false, float4 rq = float4{0.f, 0.f, 0.f, 0.f}) {
  const bool vec = p.vec && (n0 + 3 < p.N);
  if (vec) {
    if (p.bias) {
      const float4 b = has_bq ? bq0 : *reinterpret_cast<const float4*>(p.bias + n0);
      v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
    }
    if (EPI == VL_EPI_F32) {
      const long o = (long)m * p.ldc + n0;
      if (p.resid) {
        // (has_rq: preloaded by the caller -- the residual may alias nothing, but the compiler cannot know: without the
        // preload every group's load waits behind the previous group's store)
        const float4 r = has_rq ? rq : *reinterpret_cast<const float4*>(p.resid + o);
        v[0] += r.x; v[1] += r.y; v[2] += r.z; v[3] += r.w;
      }
      VL_EXP_ST *reinterpret_cast<float4*>(out32 + o) = make_float4(v[0], v[1], v[2], v[3]);
    } else {
      const long o = (long)m * p.ld16 + n0;
      ushort4 hi, lo;
      if (EPI == VL_EPI_GELU_SPLIT) {
        ushort4 u;
        float y[4], d[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) gelu_erf_both(v[t], y[t], d[t]);
        u.x = f32_to_bf16(d[0]); u.y = f32_to_bf16(d[1]); u.z = f32_to_bf16(d[2]); u.w = f32_to_bf16(d[3]);
        VL_EXP_ST *reinterpret_cast<ushort4*>(p.aux16 + o) = u;
        split_bf16(y[0], hi.x, lo.x); split_bf16(y[1], hi.y, lo.y);
        split_bf16(y[2], hi.z, lo.z); split_bf16(y[3], hi.w, lo.w);
        VL_EXP_ST *reinterpret_cast<ushort4*>(p.out_hi + o) = hi;
        VL_EXP_ST *reinterpret_cast<ushort4*>(p.out_lo + o) = lo;
      } else if (EPI == VL_EPI_DGELU_BF16) {
        const ushort4 u = *reinterpret_cast<const ushort4*>(p.aux16 + o);
        hi.x = f32_to_bf16(v[0] * bf16_to_f32(u.x));
        hi.y = f32_to_bf16(v[1] * bf16_to_f32(u.y));
        hi.z = f32_to_bf16(v[2] * bf16_to_f32(u.z));
        hi.w = f32_to_bf16(v[3] * bf16_to_f32(u.w));
        VL_EXP_ST *reinterpret_cast<ushort4*>(p.out_hi + o) = hi;
      } else if (EPI == VL_EPI_BF16) {
        hi.x = f32_to_bf16(v[0]); hi.y = f32_to_bf16(v[1]); hi.z = f32_to_bf16(v[2]); hi.w = f32_to_bf16(v[3]);
        VL_EXP_ST *reinterpret_cast<ushort4*>(p.out_hi + o) = hi;
      } else {  // VL_EPI_SPLIT
        split_bf16(v[0], hi.x, lo.x); split_bf16(v[1], hi.y, lo.y);
        split_bf16(v[2], hi.z, lo.z); split_bf16(v[3], hi.w, lo.w);
        VL_EXP_ST *reinterpret_cast<ushort4*>(p.out_hi + o) = hi;
        VL_EXP_ST *reinterpret_cast<ushort4*>(p.out_lo + o) = lo;
      }
      if (p.img || want_cs) {
        const bf16_raw h4[4] = {hi.x, hi.y, hi.z, hi.w};
        if (lcol) {
#pragma unroll
          for (int t = 0; t < 4; ++t) lcol[t * lpitch] = h4[t];
        } else if (p.img) {
          image_store<4>(p, m, n0, h4);
        }
        if (want_cs) {
#pragma unroll
          for (int t = 0; t < 4; ++t) csum[t] += bf16_to_f32(h4[t]);
        }
      }
    }
    return;
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {  // ragged edge / unaligned leading dimension: scalar path
    const int n = n0 + r;
    if (n >= p.N) break;
    float x = v[r] + (p.bias ? p.bias[n] : 0.f);
    bf16_raw hi = 0, lo = 0;
    if (EPI == VL_EPI_F32) {
      if (p.resid) x += p.resid[(long)m * p.ldc + n];
      out32[(long)m * p.ldc + n] = x;
    } else if (EPI == VL_EPI_GELU_SPLIT) {
      float y, d;
      gelu_erf_both(x, y, d);
      p.aux16[(long)m * p.ld16 + n] = f32_to_bf16(d);
      split_bf16(y, hi, lo);
      p.out_hi[(long)m * p.ld16 + n] = hi;
      p.out_lo[(long)m * p.ld16 + n] = lo;
    } else if (EPI == VL_EPI_DGELU_BF16) {
      const float u = bf16_to_f32(p.aux16[(long)m * p.ld16 + n]);
      hi = f32_to_bf16(x * u);
      p.out_hi[(long)m * p.ld16 + n] = hi;
    } else if (EPI == VL_EPI_BF16) {
      hi = f32_to_bf16(x);
      p.out_hi[(long)m * p.ld16 + n] = hi;
    } else {
      split_bf16(x, hi, lo);
      p.out_hi[(long)m * p.ld16 + n] = hi;
      p.out_lo[(long)m * p.ld16 + n] = lo;
    }
    if (EPI != VL_EPI_F32) {
      if (lcol) lcol[r * lpitch] = hi;
      else if (p.img) p.img[((long)(m >> 6) * p.img_n + n) * 64 + (m & 63)] = hi;
      if (want_cs) csum[r] += bf16_to_f32(hi);
    }
  }
}
template <int EPI>
__device__ __forceinline__ void epilogue_store4(const GemmArgs& p, float* out32, int m, int n0, f32x4 v) {
  float unused[4] = {0.f, 0.f, 0.f, 0.f};
  epilogue_store4<EPI>(p, out32, m, n0, v, false, unused);
}

// 8 consecutive output columns n0..n0+7 of row m (16-bit epilogues of the fast path: two paired MFMA tiles hold the two
// halves, see the B-row permutation in gemm2_kernel): one 16-byte access per bf16 array instead of two 8-byte ones.
__device__ __forceinline__ uint4 pack8(const ushort4& a, const ushort4& b) {
  return make_uint4((unsigned)a.x | ((unsigned)a.y << 16), (unsigned)a.z | ((unsigned)a.w << 16),
                    (unsigned)b.x | ((unsigned)b.y << 16), (unsigned)b.z | ((unsigned)b.w << 16));
}
template <int EPI>
__device__ __forceinline__ void epilogue_store8(const GemmArgs& p, float* out32, int m, int n0, f32x4 v0, f32x4 v1,
                                                bool want_cs, float (&c0)[4], float (&c1)[4], bf16_raw* lcol = nullptr,
                                                int lpitch = 0, bool has_bq = false, float4 bq0 = float4{0.f, 0.f, 0.f, 0.f},
                                                float4 bq1 = float4{0.f, 0.f, 0.f, 0.f}, bool has_uq = false,
                                                uint4 uq = uint4{0u, 0u, 0u, 0u}) {
  if (!(p.vec8 && n0 + 7 < p.N)) {
    epilogue_store4<EPI>(p, out32, m, n0, v0, want_cs, c0, lcol, lpitch, has_bq, bq0);
    if (n0 + 4 < p.N)
      epilogue_store4<EPI>(p, out32, m, n0 + 4, v1, want_cs, c1, lcol ? lcol + 4 * lpitch : nullptr, lpitch, has_bq, bq1);
    return;
  }
  float x[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
  if (p.bias) {
    const float4 b0 = has_bq ? bq0 : *reinterpret_cast<const float4*>(p.bias + n0);
    const float4 b1 = has_bq ? bq1 : *reinterpret_cast<const float4*>(p.bias + n0 + 4);
    x[0] += b0.x; x[1] += b0.y; x[2] += b0.z; x[3] += b0.w; x[4] += b1.x; x[5] += b1.y; x[6] += b1.z; x[7] += b1.w;
  }
  const long o = (long)m * p.ld16 + n0;
  ushort4 h0, h1, l0, l1;
  if (EPI == VL_EPI_GELU_SPLIT) {
    ushort4 u0, u1;
    float y[8], d[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) gelu_erf_both(x[t], y[t], d[t]);
    u0.x = f32_to_bf16(d[0]); u0.y = f32_to_bf16(d[1]); u0.z = f32_to_bf16(d[2]); u0.w = f32_to_bf16(d[3]);
    u1.x = f32_to_bf16(d[4]); u1.y = f32_to_bf16(d[5]); u1.z = f32_to_bf16(d[6]); u1.w = f32_to_bf16(d[7]);
    VL_EXP_ST *reinterpret_cast<uint4*>(p.aux16 + o) = pack8(u0, u1);
    split_bf16(y[0], h0.x, l0.x); split_bf16(y[1], h0.y, l0.y);
    split_bf16(y[2], h0.z, l0.z); split_bf16(y[3], h0.w, l0.w);
    split_bf16(y[4], h1.x, l1.x); split_bf16(y[5], h1.y, l1.y);
    split_bf16(y[6], h1.z, l1.z); split_bf16(y[7], h1.w, l1.w);
    VL_EXP_ST *reinterpret_cast<uint4*>(p.out_hi + o) = pack8(h0, h1);
    VL_EXP_ST *reinterpret_cast<uint4*>(p.out_lo + o) = pack8(l0, l1);
  } else if (EPI == VL_EPI_DGELU_BF16) {
    const uint4 uu = has_uq ? uq : *reinterpret_cast<const uint4*>(p.aux16 + o);  // (preloaded: see epilogue_store4's residual)
    const unsigned w[4] = {uu.x, uu.y, uu.z, uu.w};
    bf16_raw r[8];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      r[2 * t] = f32_to_bf16(x[2 * t] * __uint_as_float(w[t] << 16));
      r[2 * t + 1] = f32_to_bf16(x[2 * t + 1] * __uint_as_float(w[t] & 0xFFFF0000u));
    }
    h0.x = r[0]; h0.y = r[1]; h0.z = r[2]; h0.w = r[3]; h1.x = r[4]; h1.y = r[5]; h1.z = r[6]; h1.w = r[7];
    VL_EXP_ST *reinterpret_cast<uint4*>(p.out_hi + o) = pack8(h0, h1);
  } else if (EPI == VL_EPI_BF16) {
    h0.x = f32_to_bf16(x[0]); h0.y = f32_to_bf16(x[1]); h0.z = f32_to_bf16(x[2]); h0.w = f32_to_bf16(x[3]);
    h1.x = f32_to_bf16(x[4]); h1.y = f32_to_bf16(x[5]); h1.z = f32_to_bf16(x[6]); h1.w = f32_to_bf16(x[7]);
    VL_EXP_ST *reinterpret_cast<uint4*>(p.out_hi + o) = pack8(h0, h1);
  } else {  // VL_EPI_SPLIT
    split_bf16(x[0], h0.x, l0.x); split_bf16(x[1], h0.y, l0.y); split_bf16(x[2], h0.z, l0.z); split_bf16(x[3], h0.w, l0.w);
    split_bf16(x[4], h1.x, l1.x); split_bf16(x[5], h1.y, l1.y); split_bf16(x[6], h1.z, l1.z); split_bf16(x[7], h1.w, l1.w);
    VL_EXP_ST *reinterpret_cast<uint4*>(p.out_hi + o) = pack8(h0, h1);
    VL_EXP_ST *reinterpret_cast<uint4*>(p.out_lo + o) = pack8(l0, l1);
  }
  if (p.img || want_cs) {
    const bf16_raw h8[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
    if (lcol) {
#pragma unroll
      for (int t = 0; t < 8; ++t) lcol[t * lpitch] = h8[t];
    } else if (p.img) {
      image_store<8>(p, m, n0, h8);
    }
    if (want_cs) {
#pragma unroll
      for (int t = 0; t < 4; ++t) { c0[t] += bf16_to_f32(h8[t]); c1[t] += bf16_to_f32(h8[4 + t]); }
    }
  }
}
template <int EPI>
__device__ __forceinline__ void epilogue_store8(const GemmArgs& p, float* out32, int m, int n0, f32x4 v0, f32x4 v1) {
  float u0[4] = {0.f, 0.f, 0.f, 0.f}, u1[4] = {0.f, 0.f, 0.f, 0.f};
  epilogue_store8<EPI>(p, out32, m, n0, v0, v1, false, u0, u1);
}

// ---------------------------------------------------------------------------------------------------------------
// fast path: 256 x BN tile, LDS-DMA staging, double-buffered
// ---------------------------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

template <int NSPLIT, int EPI, int BN, int WM>
__global__ __launch_bounds__(WM * 256, WM) void gemm2_kernel(GemmArgs p) {
  constexpr int BM2 = 256, NT = BN / 64;                  // 16-wide n-tiles per wave (wave tile 256/WM x BN/4)
  constexpr int MT = 16 / WM;                              // 16-high m-tiles per wave
  constexpr int NWAVES = WM * 4;
  constexpr int A_UNITS = BM2 / 8, UNITS = (BM2 + BN) / 8;  // 1-KiB DMA units (8 rows x 128 B) per stage
  constexpr int UPW = (UNITS + NWAVES - 1) / NWAVES;       // units per wave
  constexpr int STAGE = UNITS * 1024;
  constexpr int KSTEP = NSPLIT == 3 ? 32 : 64;             // k elements consumed per stage
  // 16-bit epilogues: n-tiles are processed in PAIRS whose 16 MFMA columns interleave in groups of 4, so that a lane's
  // accumulators of tiles (2t, 2t+1) are 8 consecutive output columns (one 16-byte bf16 store).  This is a pure
  // permutation of which B row feeds which MFMA column; the B tile then uses its own bank swizzle fB (the permuted
  // fragment reads touch rows {0-3, 8-11, 16-19, 24-27} (+4) instead of 16 consecutive ones).
  constexpr bool PAIR = (EPI != VL_EPI_F32);
  constexpr int NTP = PAIR ? (NT & ~1) : 0;                // tiles [0, NTP) are paired
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;

  const int nwg = p.tiles_m * p.tiles_n;
  const int bid = blockIdx.x;
  const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
  const int swz = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  const int tm = swz / p.tiles_n, tn = swz - tm * p.tiles_n;
  const int row0 = tm * BM2, col0 = tn * BN;
  const int kbeg = blockIdx.y * p.k_len;
  const int kend = min(p.K, kbeg + p.k_len);
  const int nk = (kend - kbeg) / KSTEP;

  // per-lane DMA source pointers: unit u = wave + 8*j covers tile rows [8u', 8u'+8) of A (u < A_UNITS) or B
  const bf16_raw* src[UPW];
#pragma unroll
  for (int j = 0; j < UPW; ++j) {
    const int u = wave + NWAVES * j;
    const bool isB = u >= A_UNITS;
    const int trow = (isB ? u - A_UNITS : u) * 8 + (lane >> 3);
    if (u >= UNITS) { src[j] = p.a_hi; continue; }
    int grow = (isB ? col0 : row0) + trow;
    const int lim = (isB ? p.N : p.M) - 1;
    grow = grow < lim ? grow : lim;  // rows past the edge re-read the last row; their products are never stored
    const int fsw = (PAIR && isB) ? (((0x78 >> (2 * ((trow >> 3) & 3))) & 3) | (((trow >> 1) & 1) << 2))
                                  : ((trow >> 1) & 7);
    const int lc = (lane & 7) ^ fsw;  // logical 16-B chunk that lands at physical position lane&7
    const bf16_raw* base;
    int koff;
    if (NSPLIT == 3) {
      base = isB ? ((lc & 4) ? p.b_lo : p.b_hi) : ((lc & 4) ? p.a_lo : p.a_hi);
      koff = (lc & 3) * 8;
    } else {
      base = isB ? p.b_hi : p.a_hi;
      koff = lc * 8;
    }
    src[j] = base + (long)grow * (isB ? p.ldb : p.lda) + kbeg + koff;
  }
  auto issue = [&](int kt, int stage) {
#pragma unroll
    for (int j = 0; j < UPW; ++j)
      if (wave + NWAVES * j < UNITS)
        __builtin_amdgcn_global_load_lds((glb_ptr_t)(src[j] + kt * KSTEP),
                                         (lds_ptr_t)(smem + stage * STAGE + (wave + NWAVES * j) * 1024), 16, 0, 0);
  };

  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int frow = lane & 15, fk = lane >> 4;
  // fragment byte offsets inside a stage (row-dependent swizzle folded in); chunk index is XORed per read
  int a_off[MT], a_sw[MT], b_off[NT], b_sw[NT];
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int row = wm * (16 * MT) + i * 16 + frow;
    a_off[i] = row * 128;
    a_sw[i] = (row >> 1) & 7;
  }
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int row = wn * (BN / 4) + (j < NTP ? 32 * (j >> 1) + 8 * (frow >> 2) + 4 * (j & 1) + (frow & 3) : j * 16 + frow);
    b_off[j] = A_UNITS * 1024 + row * 128;
    b_sw[j] = PAIR ? (((0x78 >> (2 * ((row >> 3) & 3))) & 3) | (((row >> 1) & 1) << 2)) : ((row >> 1) & 7);
  }

  issue(0, 0);
  __syncthreads();  // waits vmcnt(0) for the DMA, then the workgroup barrier
  for (int kt = 0; kt < nk; ++kt) {
    const unsigned char* st = smem + (kt & 1) * STAGE;
    if (kt + 1 < nk) issue(kt + 1, (kt + 1) & 1);
    if (NSPLIT == 1) {
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        bf16x8 b[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j)
          b[j] = *reinterpret_cast<const bf16x8*>(st + b_off[j] + (((kk * 4 + fk) ^ b_sw[j]) << 4));
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          const bf16x8 a = *reinterpret_cast<const bf16x8*>(st + a_off[i] + (((kk * 4 + fk) ^ a_sw[i]) << 4));
#pragma unroll
          for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a, acc[i][j], 0, 0, 0);
        }
      }
    } else {
      bf16x8 bh[NT], bl[NT];
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        bh[j] = *reinterpret_cast<const bf16x8*>(st + b_off[j] + ((fk ^ b_sw[j]) << 4));
        bl[j] = *reinterpret_cast<const bf16x8*>(st + b_off[j] + (((4 + fk) ^ b_sw[j]) << 4));
      }
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const bf16x8 ah = *reinterpret_cast<const bf16x8*>(st + a_off[i] + ((fk ^ a_sw[i]) << 4));
        const bf16x8 al = *reinterpret_cast<const bf16x8*>(st + a_off[i] + (((4 + fk) ^ a_sw[i]) << 4));
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[j], al, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl[j], ah, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[j], ah, acc[i][j], 0, 0, 0);
        }
      }
    }
    __syncthreads();  // next stage landed (vmcnt(0)) and everyone is done reading this one
  }

  // epilogue.  D^T layout: lane&15 -> m inside the 16-row tile, 4*(lane>>4) + reg -> n inside the 16-col tile
  float* out32 = p.out32 + (long)blockIdx.y * p.slab_stride;
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int m = row0 + wm * (16 * MT) + i * 16 + (lane & 15);
    if (m >= p.M) continue;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      if (j < NTP) {
        if (j & 1) continue;
        const int n0 = col0 + wn * (BN / 4) + 32 * (j >> 1) + 8 * (lane >> 4);
        if (n0 < p.N) epilogue_store8<EPI>(p, out32, m, n0, acc[i][j], acc[i][j + 1 < NT ? j + 1 : j]);
      } else {
        const int n0 = col0 + wn * (BN / 4) + j * 16 + 4 * (lane >> 4);
        if (n0 < p.N) epilogue_store4<EPI>(p, out32, m, n0, acc[i][j]);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// "ping-pong" path (all large products with K % 64 == 0 / K % 32 == 0): 8 waves, one workgroup per CU, two waves per
// SIMD (wave w and w + 4).  Two tile configurations
//      CFG 0: 256 x 256, waves 2 (M) x 4 (N), wave tile 128 x 64 = four 64 x 32 quadrants
//      CFG 1: 256 x 192, waves 4 (M) x 2 (N), wave tile  64 x 96 = four 32 x 48 quadrants   (N = 768 -> 224 tiles)
// quadrant (qm, qn) of wave (wr, wc) = rows qm*128 + wr*MI*16 .., cols qn*BH + wc*NJ*16 ..: it needs exactly one
// A half-tile (128 rows x 128 B) and one B half-tile (BH rows x 128 B) of the K-tile.  A 128-B LDS row holds 64 k
// values (1 pass) or 32 k values as [hi | lo] (3 passes), so both precisions share the staging and the reads.
// LDS: 2 K-tiles x {A0, A1, B0, B1}.  A K-tile is consumed in 4 phases, one quadrant each:
//      [ds_read the sub-tiles the quadrant needs | LDS-DMA one half-tile of a later K-tile | s_waitcnt vmcnt(N)]
//      s_barrier   [lgkmcnt(0); the quadrant's MFMAs at raised priority]   s_barrier
// and waves 4-7 run one barrier behind waves 0-3, so on every SIMD one wave is in its MFMA section while the other
// issues its LDS reads / DMA: the matrix pipe never waits for a load section.  Hazards, by barrier count:
//  * RAW: a half-tile issued in phase q is retired by every wave's counted vmcnt in phase q+4 (N = the DMA
//    instructions of the 4 younger half-tiles) and first read in phase >= q+5, i.e. after a barrier that follows every
//    wave's wait even with the one-barrier stagger.
//  * WAR: a half-tile is re-staged >= 2 phases after the phase of its last ds_read (A0, B0: read in phase 0 --
//    B0 stays in registers for phase 3 -- re-staged in phases 2, 3; B1: phase 1 -> next phase 0; A1: phase 2 ->
//    next phase 1).
// Quadrant order (0,0) (0,1) (1,1) (1,0): consecutive phases share the A or the B sub-tile.
// Bank swizzle f(row) (XOR on the 16-B chunk index, applied on the DMA source address and undone by the reads) is
// conflict-free both for 16 consecutive rows and for the permuted rows {0-3, 8-11, 16-19, 24-27} + 4*(j&1) that the
// paired n-tiles of the 16-bit epilogues read (see gemm2_kernel).
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int swz3(int row) {
  return ((row >> 1) & 1) | (((row >> 3) & 1) << 1) | ((((row >> 4) ^ (row >> 2)) & 1) << 2);
}
template <int N> __device__ __forceinline__ void wait_vmcnt() {
  static_assert(N >= 0 && N <= 12, "unsupported count");
  if (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (N == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
  if (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  if (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
  if (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  if (N == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
  if (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  if (N == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
  if (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  if (N == 9) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
  if (N == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
  if (N == 11) asm volatile("s_waitcnt vmcnt(11)" ::: "memory");
  if (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
}

// RING (VL_GEMM_RING): LDS slots per operand, each one half-tile.  4 = two K-tiles resident (128 KB at 256 x 256): a half-tile
// is issued 4 - 5 phases (one K-tile) before its first read.  5 = all 160 KB of the CU's LDS: every half-tile is issued TWO
// PHASES EARLIER -- 7 - 8 phases before its read -- because the timing experiments (profiles/r03_ab_log.txt section 7) put 11 - 21 %
// of every product into waits for operand DMA: one K-tile of prefetch distance is shorter than the memory latency of the
// loaded chip.  Slot of half h of K-tile kt: (2 kt + h) mod RING, per operand.  Schedule for RING = 5, K-tile kt:
//      phase 0: read B0, A0 (kt) | issue A0 (kt + 2)      phase 1: read B1 (kt) | issue B0 (kt + 2)
//      phase 2: read A1 (kt)     | issue B1 (kt + 2)      phase 3:              | issue A1 (kt + 2)
//  * WAR: A0 (kt + 2) takes the slot of A1 (kt - 1) (last read: phase 2 of kt - 1), B0 (kt + 2) that of B1 (kt - 1) (phase 1 of
//    kt - 1), B1 (kt + 2) that of B0 (kt) (phase 0), A1 (kt + 2) that of A0 (kt) (phase 0): always >= 2 phases after the read,
//    the same margin as the 4-slot ring.
//  * RAW: after its own issue every phase waits until at most the SIX youngest half-tiles are outstanding, so a half-tile
//    issued in phase q has landed by the wait of phase q + 6 and is first read in phase q + 7 or q + 8, behind the barriers of
//    phase q + 6 (also with the one-barrier stagger of waves 4 - 7).
// Measured (same box, profiles/r03_ab_log.txt section 9): no difference -- QKV 140.2 / 141.2 vs 150.4 / 138.6 us, FFN2 215 / 208 vs
// 205 / 210, step 16.07 / 16.13 vs 16.10 / 16.09 ms.  The cost the "no DMA" experiment exposes is therefore NOT prefetch distance
// (the data is there in time); the 4-slot ring stays the default (128 KB), the 5-slot path stays tested behind the macro.
#ifndef VL_GEMM_RING
#define VL_GEMM_RING 4
#endif

// PERSIST: the grid is a fixed number of workgroups G (<= the CU count); workgroup w runs tiles w, w + G, w + 2G, ... (the
// XCD-contiguous tile order is kept: tile t and t + G land on the same XCD when G % 8 == 0).  Between two tiles the DMA
// prologue of the NEXT tile (K-tile 0 + A0 / B0 of K-tile 1) is issued right behind the last MFMA section, BEFORE the
// epilogue of the finished tile: its HBM latency hides under the epilogue's arithmetic, and the epilogue's stores drain
// under the next tile's first MFMA sections instead of in front of an idle matrix pipe (with one tile per workgroup all
// 256 CUs reach their epilogues together: a burst of 64 MB of stores, then a burst of first loads, per round).  Counted
// vmcnt waits stay valid with the younger stores in flight: loads return in order among themselves, so "at most N
// outstanding" still implies that everything older than the N youngest LOADS has landed (stores only make the wait
// stricter).
template <int NSPLIT, int EPI, int CFG, bool PERSIST>
__global__ __launch_bounds__(512) void gemm3_kernel(GemmArgs p) {
  constexpr int WR = CFG == 1 ? 4 : 2, WC = 8 / WR;        // wave grid
  constexpr int MI = CFG == 1 ? 2 : 4, NJ = CFG == 1 ? 3 : 2;  // 16 x 16 MFMA tiles per quadrant (first M half)
  constexpr int MI1 = CFG == 2 ? 3 : MI;                    // ... of the second M half (CFG 2: 224-row tile)
  constexpr int AH = WR * MI * 16, AH1 = WR * MI1 * 16, BH = WC * NJ * 16;  // rows per half-tile
  constexpr int BMT = AH + AH1, BNT = 2 * BH;               // tile height / width
  constexpr bool BSHORT = BH / 8 < 16;                      // waves 4-7 stage one 1-KiB unit of a B half-tile, not two
  constexpr bool ASHORT = AH1 / 8 < 16;                     // same for the second A half-tile
  constexpr int OFF_A1 = AH * 128, OFF_B0 = (AH + AH1) * 128, OFF_B1 = OFF_B0 + BH * 128, STAGE = OFF_B1 + BH * 128;
  constexpr int KSTEP = NSPLIT == 3 ? 32 : 64;
  constexpr bool PAIR = (EPI != VL_EPI_F32);
  constexpr int NJP = PAIR ? (NJ & ~1) : 0;                 // n-tiles [0, NJP) of a quadrant are paired
  constexpr int RING = VL_GEMM_RING;                        // LDS slots per operand (see above)
  constexpr int SLOT_A = AH * 128, SLOT_B = BH * 128;       // RING == 5: A ring at 0, B ring behind it
  constexpr int RING_B0 = RING * SLOT_A;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave / WC, wc = wave % WC;
  const bool late = wave >= 4;  // the wave group that runs one barrier behind

  const int nwg = p.tiles_m * p.tiles_n;
  const int nk = p.K / KSTEP;
  const int q8 = nwg >> 3, r8 = nwg & 7;
  int row0, col0, tm;
  // DMA sources: half-tile x in {A0, A1, B0, B1}; unit u = wave + 8*j (8 rows x 128 B); lane -> (row, physical chunk)
  const bf16_raw* src[4][2];
  auto set_tile = [&](int t) {  // tile index in dispatch order -> (row0, col0), per-lane DMA source pointers
    const int xcd = t & 7;
    const int swz = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (t >> 3);
    tm = swz / p.tiles_n;
    const int tn = swz - tm * p.tiles_n;
    row0 = tm * BMT; col0 = tn * BNT;
#pragma unroll
    for (int x = 0; x < 4; ++x)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const bool isB = x >= 2;
        const int r = 8 * (wave + 8 * j) + (lane >> 3);
        const int lc = (lane & 7) ^ swz3(r);  // logical chunk that lands at physical position lane & 7
        int g = (isB ? col0 + (x & 1) * BH : row0 + (x & 1) * AH) + r;
        const int lim = (isB ? p.N : p.M) - 1;
        g = g < lim ? g : lim;  // rows past the edge re-read a valid row; their products are never stored
        const bf16_raw* base;
        int koff;
        if (NSPLIT == 3) {
          base = isB ? ((lc & 4) ? p.b_lo : p.b_hi) : ((lc & 4) ? p.a_lo : p.a_hi);
          koff = (lc & 3) * 8;
        } else {
          base = isB ? p.b_hi : p.a_hi;
          koff = lc * 8;
        }
        src[x][j] = base + (long)g * (isB ? p.ldb : p.lda) + koff;
      }
  };
  constexpr int XOFF[4] = {0, OFF_A1, OFF_B0, OFF_B1};
#define G3_ISSUE(x, kt)                                                                                          \
  do {                                                                                                           \
    __builtin_amdgcn_global_load_lds((glb_ptr_t)(src[x][0] + (long)(kt) * KSTEP),                                \
                                     (lds_ptr_t)(smem + ((kt) & 1) * STAGE + XOFF[x] + wave * 1024), 16, 0, 0);  \
    if (!(late && ((BSHORT && (x) >= 2) || (ASHORT && (x) == 1))))                                               \
      __builtin_amdgcn_global_load_lds((glb_ptr_t)(src[x][1] + (long)(kt) * KSTEP),                              \
                                       (lds_ptr_t)(smem + ((kt) & 1) * STAGE + XOFF[x] + (wave + 8) * 1024), 16, 0, 0); \
  } while (0)
  // wait until everything older than the youngest (nA A-half-tiles + nB B-half-tiles) has landed
#define G3_WAIT_YOUNGER(nA, nB)                                                                                  \
  do {                                                                                                           \
    if (late) wait_vmcnt<2 * (nA) - ((nA) == 2 && ASHORT ? 1 : 0) + (BSHORT ? 1 : 2) * (nB)>();                  \
    else wait_vmcnt<2 * (nA) + 2 * (nB)>();                                                                      \
  } while (0)
#define G3_WAIT(issued)                                                                                          \
  do {                                                                                                           \
    if (issued) G3_WAIT_YOUNGER(2, 2);                                                                           \
    else wait_vmcnt<0>();                                                                                        \
  } while (0)
  // RING == 5: half-tile x of K-tile kt into ring slot `slot` (a scalar) of its operand
#define G5_ISSUE(x, kt, slot)                                                                                    \
  do {                                                                                                           \
    unsigned char* dst_ = smem + ((x) >= 2 ? RING_B0 + (slot) * SLOT_B : (slot) * SLOT_A);                       \
    __builtin_amdgcn_global_load_lds((glb_ptr_t)(src[x][0] + (long)(kt) * KSTEP), (lds_ptr_t)(dst_ + wave * 1024), 16, 0, 0); \
    if (!(late && ((BSHORT && (x) >= 2) || (ASHORT && (x) == 1))))                                               \
      __builtin_amdgcn_global_load_lds((glb_ptr_t)(src[x][1] + (long)(kt) * KSTEP), (lds_ptr_t)(dst_ + (wave + 8) * 1024), 16, 0, 0); \
  } while (0)
  // ... wait until at most the youngest (nA0 first-half A + nA1 second-half A + nB B) half-tiles are outstanding
#define G5_WAIT_YOUNGER(nA0, nA1, nB)                                                                            \
  do {                                                                                                           \
    if (late) wait_vmcnt<2 * (nA0) + (ASHORT ? 1 : 2) * (nA1) + (BSHORT ? 1 : 2) * (nB)>();                      \
    else wait_vmcnt<2 * ((nA0) + (nA1) + (nB))>();                                                               \
  } while (0)
#define G5_WAIT(issued, nA0, nA1, nB)                                                                            \
  do {                                                                                                           \
    if (issued) G5_WAIT_YOUNGER(nA0, nA1, nB);                                                                   \
    else wait_vmcnt<0>();                                                                                        \
  } while (0)

  f32x4 acc[2][2][MI][NJ];  // [1][*][i >= MI1] unused

  // fragment byte offsets inside a half-tile: [tile][first / second 64-B half of the row], swizzle folded in
  const int frow = lane & 15, fk = lane >> 4;
  constexpr int QA = MI1 == MI ? 1 : 2;  // the row offsets of the two M halves differ only when their heights do
  int a_o[QA][MI][2], b_o[NJ][2];
#pragma unroll
  for (int qa = 0; qa < QA; ++qa)
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int row = wr * ((qa == 0 ? MI : MI1) * 16) + i * 16 + frow;
      a_o[qa][i][0] = row * 128 + ((fk ^ swz3(row)) << 4);
      a_o[qa][i][1] = row * 128 + (((4 + fk) ^ swz3(row)) << 4);
    }
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int row = wc * (NJ * 16) + (j < NJP ? 32 * (j >> 1) + 8 * (frow >> 2) + 4 * (j & 1) + (frow & 3) : j * 16 + frow);
    b_o[j][0] = row * 128 + ((fk ^ swz3(row)) << 4);
    b_o[j][1] = row * 128 + (((4 + fk) ^ swz3(row)) << 4);
  }

  bf16x8 fa[MI][2], fb0[NJ][2], fb1[NJ][2];
#ifdef VL_EXP_NO_READS
#define G3_RD_GUARD if (kt == 0)
#else
#define G3_RD_GUARD
#endif
#define G3_READ_A(st, qm)                                                                                        \
  G3_RD_GUARD _Pragma("unroll") for (int i = 0; i < ((qm) == 0 ? MI : MI1); ++i) {                                          \
    fa[i][0] = *reinterpret_cast<const bf16x8*>((st) + (qm) * OFF_A1 + a_o[(qm) % QA][i][0]);                    \
    fa[i][1] = *reinterpret_cast<const bf16x8*>((st) + (qm) * OFF_A1 + a_o[(qm) % QA][i][1]);                    \
  }
#define G3_READ_B(st, qn, fb)                                                                                    \
  G3_RD_GUARD _Pragma("unroll") for (int j = 0; j < NJ; ++j) {                                                              \
    fb[j][0] = *reinterpret_cast<const bf16x8*>((st) + OFF_B0 + (qn) * (BH * 128) + b_o[j][0]);                  \
    fb[j][1] = *reinterpret_cast<const bf16x8*>((st) + OFF_B0 + (qn) * (BH * 128) + b_o[j][1]);                  \
  }
#define G5_READ_A(pa, qm)                                                                                        \
  G3_RD_GUARD _Pragma("unroll") for (int i = 0; i < ((qm) == 0 ? MI : MI1); ++i) {                              \
    fa[i][0] = *reinterpret_cast<const bf16x8*>((pa) + a_o[(qm) % QA][i][0]);                                    \
    fa[i][1] = *reinterpret_cast<const bf16x8*>((pa) + a_o[(qm) % QA][i][1]);                                    \
  }
#define G5_READ_B(pb, fb)                                                                                        \
  G3_RD_GUARD _Pragma("unroll") for (int j = 0; j < NJ; ++j) {                                                  \
    fb[j][0] = *reinterpret_cast<const bf16x8*>((pb) + b_o[j][0]);                                               \
    fb[j][1] = *reinterpret_cast<const bf16x8*>((pb) + b_o[j][1]);                                               \
  }
  // 1 pass: [0] / [1] are the two 32-deep k halves of the row; 3 passes: [0] = hi, [1] = lo of one 32-deep k step and
  // the products are issued product-major so that dependent accumulations are MI*NJ MFMAs apart
  // (timing experiments, WRONG RESULTS, only meaningful together with VL_EXP_NO_DMA + VL_EXP_NO_READS: -DVL_EXP_NO_BARRIER drops the
  // two barriers around every MFMA section, -DVL_EXP_NO_PRIO the priority flips)
#ifdef VL_EXP_NO_BARRIER
#define G3_BAR() do { } while (0)
#else
#define G3_BAR() __builtin_amdgcn_s_barrier()
#endif
#ifdef VL_EXP_NO_PRIO
#define G3_PRIO(x) do { } while (0)
#else
#define G3_PRIO(x) __builtin_amdgcn_s_setprio(x)
#endif
#define G3_MFMA(qm, qn, fb)                                                                                      \
  do {                                                                                                           \
    G3_BAR();                                                                                                    \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                           \
    __builtin_amdgcn_sched_barrier(0);                                                                           \
    G3_PRIO(1);                                                                                                  \
    if (NSPLIT == 1) {                                                                                           \
      _Pragma("unroll") for (int kk = 0; kk < 2; ++kk)                                                           \
      _Pragma("unroll") for (int i = 0; i < ((qm) == 0 ? MI : MI1); ++i)                                         \
      _Pragma("unroll") for (int j = 0; j < NJ; ++j)                                                             \
          acc[qm][qn][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j][kk], fa[i][kk], acc[qm][qn][i][j], 0, 0, 0); \
    } else {                                                                                                     \
      _Pragma("unroll") for (int i = 0; i < ((qm) == 0 ? MI : MI1); ++i)                                         \
      _Pragma("unroll") for (int j = 0; j < NJ; ++j)                                                             \
          acc[qm][qn][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j][0], fa[i][1], acc[qm][qn][i][j], 0, 0, 0); \
      _Pragma("unroll") for (int i = 0; i < ((qm) == 0 ? MI : MI1); ++i)                                         \
      _Pragma("unroll") for (int j = 0; j < NJ; ++j)                                                             \
          acc[qm][qn][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j][1], fa[i][0], acc[qm][qn][i][j], 0, 0, 0); \
      _Pragma("unroll") for (int i = 0; i < ((qm) == 0 ? MI : MI1); ++i)                                         \
      _Pragma("unroll") for (int j = 0; j < NJ; ++j)                                                             \
          acc[qm][qn][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j][0], fa[i][0], acc[qm][qn][i][j], 0, 0, 0); \
    }                                                                                                            \
    G3_PRIO(0);                                                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                                           \
    G3_BAR();                                                                                                    \
  } while (0)

  // prologue: K-tile 0 complete, A0 / B0 of K-tile 1 in flight
  // (RING == 5: K-tile 0 complete, ALL of K-tile 1 in flight, issued in the loop's order A0, B0, B1, A1)
#define G3_PROLOGUE_ISSUE()                                                                                      \
  do {                                                                                                           \
    if (RING == 5) {                                                                                             \
      G5_ISSUE(0, 0, 0); G5_ISSUE(2, 0, 0); G5_ISSUE(3, 0, 1); G5_ISSUE(1, 0, 1);                                \
      if (nk > 1) { G5_ISSUE(0, 1, 2); G5_ISSUE(2, 1, 2); G5_ISSUE(3, 1, 3); G5_ISSUE(1, 1, 3); }                \
    } else {                                                                                                     \
      G3_ISSUE(0, 0); G3_ISSUE(2, 0); G3_ISSUE(3, 0); G3_ISSUE(1, 0);                                            \
      if (nk > 1) { G3_ISSUE(0, 1); G3_ISSUE(2, 1); }                                                            \
    }                                                                                                            \
  } while (0)
#define G3_PROLOGUE_WAIT()                                                                                       \
  do {                                                                                                           \
    if (nk > 1) { if (RING == 5) G5_WAIT_YOUNGER(1, 1, 2); else G3_WAIT_YOUNGER(1, 1); }                          \
    else wait_vmcnt<0>();                                                                                        \
    __builtin_amdgcn_s_barrier();                                                                                \
    if (late) __builtin_amdgcn_s_barrier(); /* stagger */                                                        \
  } while (0)
  set_tile(blockIdx.x);
  G3_PROLOGUE_ISSUE();
  G3_PROLOGUE_WAIT();

 int t = blockIdx.x;
 do {  // tiles of this workgroup (exactly one unless PERSIST)
#pragma unroll
  for (int a_ = 0; a_ < 2; ++a_)
#pragma unroll
    for (int b_ = 0; b_ < 2; ++b_)
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[a_][b_][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  // (timing experiments, WRONG RESULTS: -DVL_EXP_NO_DMA: no operand DMA inside the K loop; -DVL_EXP_NO_READS: no LDS fragment
  // reads inside it; what is left of the loop is MFMA sections + barriers)
#if defined(VL_EXP_NO_DMA) || defined(VL_EXP_NO_READS)
#pragma message("gemm3_kernel: timing-experiment build -- results are wrong")
#endif
  if constexpr (RING == 5) {
    int ia = 0, ib = 0;  // ring slots of A0 (kt) / B0 (kt): (2 kt) mod 5
    for (int kt = 0; kt < nk; ++kt) {
#ifdef VL_EXP_NO_DMA
      const bool n2 = false;
#else
      const bool n2 = kt + 2 < nk;
#endif
      const int ia1 = ia + 1 >= 5 ? ia - 4 : ia + 1, ia4 = ia + 4 >= 5 ? ia - 1 : ia + 4;  // slots of A1 (kt), A0 (kt + 2)
      const int ib1 = ib + 1 >= 5 ? ib - 4 : ib + 1, ib4 = ib + 4 >= 5 ? ib - 1 : ib + 4;  //          B1 (kt), B0 (kt + 2)
      // phase 0: quadrant (0,0)
      G5_READ_B(smem + RING_B0 + ib * SLOT_B, fb0);
      __builtin_amdgcn_sched_barrier(0);
      G5_READ_A(smem + ia * SLOT_A, 0);
      if (n2) G5_ISSUE(0, kt + 2, ia4);
      G5_WAIT(n2, 2, 2, 2);
      G3_MFMA(0, 0, fb0);
      // phase 1: quadrant (0,1)
      G5_READ_B(smem + RING_B0 + ib1 * SLOT_B, fb1);
      if (n2) G5_ISSUE(2, kt + 2, ib4);
      G5_WAIT(n2, 2, 1, 3);
      G3_MFMA(0, 1, fb1);
      // phase 2: quadrant (1,1)
      G5_READ_A(smem + ia1 * SLOT_A, 1);
      if (n2) G5_ISSUE(3, kt + 2, ib);  // (B1 (kt + 2): the slot B0 (kt) was read from in phase 0)
      G5_WAIT(n2, 1, 1, 4);
      G3_MFMA(1, 1, fb1);
      // phase 3: quadrant (1,0)
      if (n2) G5_ISSUE(1, kt + 2, ia);  // (A1 (kt + 2): the slot of A0 (kt))
      G5_WAIT(n2, 1, 2, 3);
      G3_MFMA(1, 0, fb0);
      ia = ia + 2 >= 5 ? ia - 3 : ia + 2;
      ib = ib + 2 >= 5 ? ib - 3 : ib + 2;
    }
  } else {
  for (int kt = 0; kt < nk; ++kt) {
    const unsigned char* st = smem + (kt & 1) * STAGE;
#ifdef VL_EXP_NO_DMA
    const bool n1 = false, n2 = false;
#else
    const bool n1 = kt + 1 < nk, n2 = kt + 2 < nk;
#endif
    // phase 0: quadrant (0,0)
    G3_READ_B(st, 0, fb0);
    __builtin_amdgcn_sched_barrier(0);
    G3_READ_A(st, 0);
    if (n1) G3_ISSUE(3, kt + 1);
    G3_WAIT(n1);
    G3_MFMA(0, 0, fb0);
    // phase 1: quadrant (0,1)
    G3_READ_B(st, 1, fb1);
    if (n1) G3_ISSUE(1, kt + 1);
    G3_WAIT(n1);
    G3_MFMA(0, 1, fb1);
    // phase 2: quadrant (1,1)
    G3_READ_A(st, 1);
    if (n2) G3_ISSUE(0, kt + 2);
    G3_WAIT(n2);
    G3_MFMA(1, 1, fb1);
    // phase 3: quadrant (1,0)
    if (n2) G3_ISSUE(2, kt + 2);
    G3_WAIT(n2);
    G3_MFMA(1, 0, fb0);
  }
  }
  if (!late) __builtin_amdgcn_s_barrier();  // matches the stagger barrier
  // the finished tile's coordinates for the epilogue; then (PERSIST) the next tile's DMA prologue goes out first
  const int e_row0 = row0, e_col0 = col0, e_tm = tm;
  const bool has_next = PERSIST && t + (int)gridDim.x < nwg;
  if (has_next) {
    set_tile(t + (int)gridDim.x);
    G3_PROLOGUE_ISSUE();
  }

  // epilogue.  D^T layout: lane&15 -> m inside the 16-row tile, 4*(lane>>4) + reg -> n inside the 16-col tile
  // (PERSIST: the lane index is laundered, so the per-lane address arithmetic of the epilogue is computed here instead of
  // being hoisted out of the tile loop and kept alive across the K loop, which has no register to spare)
  int lane_e = lane;
  if (PERSIST) asm volatile("" : "+v"(lane_e));
  float* out32 = p.out32;
  // K-major image of out_hi: a wave stages its [MI*16 rows] x [2 x NJ*16 columns] part of an M half column-major in a
  // private LDS region (the K loop is over, LDS is free; 2-byte ds_writes, pitch rows*2 + 8 B: conflict-free), then
  // writes it as 16-byte chunks of 8 consecutive rows -- 8 lanes cover the 128 contiguous bytes of one column of a
  // 64-row block.  (Written straight from the accumulator layout it would be 2-byte global stores in 32-byte segments:
  // measured +75 us per FFN1-sized GEMM, more than the re-layout pass it replaces.)
  constexpr int IMG_PITCH = MI * 16 * 2 + 8;        // bytes per staged column
  constexpr int IMG_NCOL = 2 * NJ * 16;             // both N halves of the wave
  constexpr int IMG_LP = IMG_PITCH / 2;             // pitch in elements
  unsigned char* limg = smem + wave * (IMG_NCOL * IMG_PITCH);
  const bool use_limg = EPI != VL_EPI_F32 && p.img != nullptr;
  // (column sums are wired for the GELU' epilogue only -- the one product whose bias gradient needs them; the erf-GELU
  // epilogue on the 256 x 256 tile has no registers to spare for 16 more accumulators)
  const bool want_cs = EPI == VL_EPI_DGELU_BF16 && p.cs != nullptr;
  // (one call per M half with a compile-time index: as an outer loop the optimizer refuses to unroll it once the body
  // holds the erf-GELU epilogue and the flush loop, and the accumulators go to scratch)
  auto half = [&](auto QM) {
    constexpr int qm = decltype(QM)::value;
    float csum[2][NJ][4];  // column sums of the rounded out_hi values over this wave's rows of the M half
#pragma unroll
    for (int qn = 0; qn < 2; ++qn)
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int t = 0; t < 4; ++t) csum[qn][j][t] = 0.f;
    // this lane's bias values: they depend on the column group only -- fetched once here, not once per 16-row tile
    float4 bpre[2][NJ + 1];
    const bool use_bpre = p.bias != nullptr && p.vec;
#pragma unroll
    for (int qn = 0; qn < 2; ++qn)
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int cb = e_col0 + qn * BH + wc * (NJ * 16);
        const int n0 = j < NJP ? cb + 32 * (j >> 1) + 8 * (lane_e >> 4) + 4 * (j & 1) : cb + j * 16 + 4 * (lane_e >> 4);
        bpre[qn][j] = (use_bpre && n0 + 3 < p.N) ? *reinterpret_cast<const float4*>(p.bias + n0) : make_float4(0.f, 0.f, 0.f, 0.f);
        if (j == NJ - 1) bpre[qn][NJ] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    // the residual (fp32 epilogue) / the saved GELU' (its backward epilogue) of this half: all loads go out before the first
    // store, instead of one load -> use -> store chain per group
    constexpr bool PRE_R = EPI == VL_EPI_F32, PRE_U = EPI == VL_EPI_DGELU_BF16;
    float4 rpre[PRE_R ? MI : 1][2][PRE_R ? NJ : 1];
    uint4 upre[PRE_U ? MI : 1][2][PRE_U ? (NJ + 1) / 2 : 1];
    const bool use_rpre = PRE_R && p.resid != nullptr && p.vec;
    const bool use_upre = PRE_U && p.vec8 && NJP == NJ;
    if (PRE_R || PRE_U) {
#pragma unroll
      for (int i = 0; i < (qm == 0 ? MI : MI1); ++i) {
        const int m = e_row0 + qm * AH + wr * ((qm == 0 ? MI : MI1) * 16) + i * 16 + (lane_e & 15);
#pragma unroll
        for (int qn = 0; qn < 2; ++qn)
#pragma unroll
          for (int j = 0; j < NJ; ++j) {
            const int cb = e_col0 + qn * BH + wc * (NJ * 16);
            if (PRE_R) {
              const int n0 = cb + j * 16 + 4 * (lane_e >> 4);
              rpre[PRE_R ? i : 0][qn][PRE_R ? j : 0] = (use_rpre && m < p.M && n0 + 3 < p.N)
                  ? *reinterpret_cast<const float4*>(p.resid + (long)m * p.ldc + n0) : make_float4(0.f, 0.f, 0.f, 0.f);
            } else if ((j & 1) == 0) {
              const int n0 = cb + 32 * (j >> 1) + 8 * (lane_e >> 4);
              upre[PRE_U ? i : 0][qn][PRE_U ? j / 2 : 0] = (use_upre && m < p.M && n0 + 7 < p.N)
                  ? *reinterpret_cast<const uint4*>(p.aux16 + (long)m * p.ld16 + n0) : make_uint4(0u, 0u, 0u, 0u);
            }
          }
      }
    }
#pragma unroll
    for (int i = 0; i < (qm == 0 ? MI : MI1); ++i) {
      const int m = e_row0 + qm * AH + wr * ((qm == 0 ? MI : MI1) * 16) + i * 16 + (lane_e & 15);
      const bool row_ok = m < p.M;
      bf16_raw* lrow = use_limg ? reinterpret_cast<bf16_raw*>(limg) + i * 16 + (lane_e & 15) : nullptr;
#pragma unroll
      for (int qn = 0; qn < 2; ++qn)
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const int cb = e_col0 + qn * BH + wc * (NJ * 16);
          if (j < NJP) {
            if ((j & 1) == 0) {
              const int nl = 32 * (j >> 1) + 8 * (lane_e >> 4);  // column inside the wave's NJ*16-wide span
              const int n0 = cb + nl;
              if (row_ok && n0 < p.N)
                epilogue_store8<EPI>(p, out32, m, n0, acc[qm][qn][i][j], acc[qm][qn][i][j + 1 < NJ ? j + 1 : j], want_cs,
                                     csum[qn][j], csum[qn][j + 1 < NJ ? j + 1 : j],
                                     lrow ? lrow + (qn * NJ * 16 + nl) * IMG_LP : nullptr, IMG_LP, use_bpre, bpre[qn][j],
                                     bpre[qn][j + 1], use_upre, upre[PRE_U ? i : 0][qn][PRE_U ? j / 2 : 0]);
            }
          } else {
            const int nl = j * 16 + 4 * (lane_e >> 4);
            const int n0 = cb + nl;
            if (row_ok && n0 < p.N)
              epilogue_store4<EPI>(p, out32, m, n0, acc[qm][qn][i][j], want_cs, csum[qn][j],
                                   lrow ? lrow + (qn * NJ * 16 + nl) * IMG_LP : nullptr, IMG_LP, use_bpre, bpre[qn][j],
                                   use_rpre, rpre[PRE_R ? i : 0][qn][PRE_R ? j : 0]);
          }
        }
    }
    if (EPI != VL_EPI_F32 && use_limg) {  // flush the staged part: 16-byte chunks of 8 consecutive rows of one column
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      constexpr int ROWS = (qm == 0 ? MI : MI1) * 16, CPC = ROWS / 8;
      const int mbase = e_row0 + qm * AH + wr * ROWS;
      for (int idx = lane_e; idx < IMG_NCOL * CPC; idx += 64) {
        const int nl = idx / CPC, c = idx - nl * CPC;
        const int qn = nl / (NJ * 16);
        const int n = e_col0 + qn * BH + wc * (NJ * 16) + (nl - qn * (NJ * 16));
        const int m8 = mbase + c * 8;
        if (n < p.N && m8 < p.M) {
          const unsigned char* src = limg + nl * IMG_PITCH + c * 16;
          const uint2 lo8 = *reinterpret_cast<const uint2*>(src), hi8 = *reinterpret_cast<const uint2*>(src + 8);
          *reinterpret_cast<uint4*>(p.img + ((long)(m8 >> 6) * p.img_n + n) * 64 + (m8 & 63)) =
              make_uint4(lo8.x, lo8.y, hi8.x, hi8.y);
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    if (EPI == VL_EPI_DGELU_BF16 && want_cs) {  // one partial row per (tile row, M half, wave row): sum over the 16 row lanes, then store
      float* dst = p.cs + (long)((e_tm * 2 + qm) * WR + wr) * p.N;
#pragma unroll
      for (int qn = 0; qn < 2; ++qn)
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          float v[4];
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            float x = csum[qn][j][t];
            x += __shfl_xor(x, 1); x += __shfl_xor(x, 2); x += __shfl_xor(x, 4); x += __shfl_xor(x, 8);
            v[t] = x;
          }
          if ((lane_e & 15) == 0) {
            const int cb = e_col0 + qn * BH + wc * (NJ * 16);
            // paired tiles (j, j+1): csum[j] = columns +0..3, csum[j+1] = columns +4..7 of the lane_e's 8-column group
            const int n0 = j < NJP ? cb + 32 * (j >> 1) + 8 * (lane_e >> 4) + 4 * (j & 1) : cb + j * 16 + 4 * (lane_e >> 4);
#pragma unroll
            for (int t = 0; t < 4; ++t)
              if (n0 + t < p.N) dst[n0 + t] = v[t];
          }
        }
    }
  };
  half(std::integral_constant<int, 0>{});
  half(std::integral_constant<int, 1>{});
  if (has_next) G3_PROLOGUE_WAIT();
  t += (int)gridDim.x;
 } while (PERSIST && t < nwg);
#undef G3_ISSUE
#undef G3_WAIT_YOUNGER
#undef G3_WAIT
#undef G3_READ_A
#undef G3_READ_B
#undef G3_MFMA
#undef G3_PROLOGUE_ISSUE
#undef G3_PROLOGUE_WAIT
#undef G5_ISSUE
#undef G5_WAIT_YOUNGER
#undef G5_WAIT
#undef G5_READ_A
#undef G5_READ_B
}

// ---------------------------------------------------------------------------------------------------------------
// TN fast path for the weight gradients: C[M,N] = A^T B with A [K][M], B [K][N] both "k-strided" (row = one of the
// B*S batch rows), i.e. dW = dY^T X straight from the row-major activations -- no transposed copies.  Same 256 x BN
// tile / 8 waves / LDS-DMA double buffer as gemm2_kernel; the LDS image is [k][cols] and MFMA fragments (8 k-values
// of one column per lane) are gathered with ds_read_b64_tr_b16 (two 4(k) x 16(col) transposed block reads per
// fragment).  32-byte column blocks are XOR-swizzled with h(k) = (k&3) | ((k>>3)&1)<<2 (applied on the DMA source
// address, undone by the reads) so that the 8 k-rows a 32-lane half touches hit 8 different 32-byte bank groups.
// ---------------------------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(4))) short s16x4;
__device__ __forceinline__ bf16x8 tr_frag(const unsigned char* p0, const unsigned char* p1) {
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p0);
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p1);
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8, v);
}

template <int BN>
__global__ __launch_bounds__(1024, 4) void gemm2_tn_kernel(GemmArgs p) {
  constexpr int BM2 = 256, NT = BN / 64, MT = 4, NWAVES = 16;
  constexpr int A_ROWB = BM2 * 2, B_ROWB = BN * 2;          // bytes per k-row of the A / B tile
  constexpr int A_UNITS = 64 * A_ROWB / 1024, B_UNITS = 64 * B_ROWB / 1024, UNITS = A_UNITS + B_UNITS;
  constexpr int UPW = UNITS / NWAVES, STAGE = UNITS * 1024;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  // 1-D grid over (split, tile): the workgroups resident on one XCD (bid % 8) get a contiguous range of the
  // split-major work list, i.e. (almost) one K-range of the whole output -> each K-range of dY and X is pulled from
  // HBM by one XCD's L2 only, instead of by every XCD that happens to host one of its tiles.
  const int tiles = p.tiles_m * p.tiles_n;
  const int nwg = tiles * p.splits;
  const int bid = blockIdx.x;
  const int xcd = bid & 7, q8 = nwg >> 3, r8 = nwg & 7;
  const int swz = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int split = swz / tiles, tile = swz - split * tiles;
  const int tm = tile / p.tiles_n, tn = tile - tm * p.tiles_n;
  const int row0 = tm * BM2, col0 = tn * BN;
  const int kbeg = split * p.k_len;
  const int kend = min(p.K, kbeg + p.k_len);
  const int nk = (kend - kbeg) / 64;

  const bf16_raw* src[UPW];
#pragma unroll
  for (int j = 0; j < UPW; ++j) {
    const int u = wave + NWAVES * j;
    const bool isB = u >= A_UNITS;
    const int rowb = isB ? B_ROWB : A_ROWB;
    const int cpr = rowb / 16;                         // 16-byte chunks per k-row (32 or 16)
    const int k = (isB ? u - A_UNITS : u) * (1024 / rowb) + lane / cpr;
    const int c = lane % cpr;
    const int lb = (c >> 1) ^ ((k & 3) | (((k >> 3) & 1) << 2));  // logical 32-B block landing at physical c>>1
    int gcol = (isB ? col0 : row0) + lb * 16 + (c & 1) * 8;
    const int lim = (isB ? p.N : p.M) - 8;
    gcol = gcol < lim ? gcol : lim;  // columns past the edge re-read valid data; their products are never stored
    src[j] = (isB ? p.b_hi : p.a_hi) + (long)(kbeg + k) * (isB ? p.ldb : p.lda) + gcol;
  }
  const long stepA = 64L * p.lda, stepB = 64L * p.ldb;
  auto issue = [&](int kt, int stage) {
#pragma unroll
    for (int j = 0; j < UPW; ++j) {
      const bool isB = (wave + NWAVES * j) >= A_UNITS;
      __builtin_amdgcn_global_load_lds((glb_ptr_t)(src[j] + kt * (isB ? stepB : stepA)),
                                       (lds_ptr_t)(smem + stage * STAGE + (wave + NWAVES * j) * 1024), 16, 0, 0);
    }
  };

  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // transposed-read lane geometry: group g = lane>>4 owns k = 8g..8g+7 of the 32-deep MFMA step; inside the group
  // lane 4q+pp addresses row q, columns 4pp..4pp+3 of the 4 x 16 block
  const int g = lane >> 4, qq = (lane & 15) >> 2, pp = lane & 3;
  const int hq = qq | ((g & 1) << 2);
  const int a_k = (8 * g + qq) * A_ROWB + pp * 8;
  const int b_k = A_UNITS * 1024 + (8 * g + qq) * B_ROWB + pp * 8;

  issue(0, 0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const unsigned char* st = smem + (kt & 1) * STAGE;
    if (kt + 1 < nk) issue(kt + 1, (kt + 1) & 1);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8 b[NT];
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const unsigned char* bp = st + b_k + kk * 32 * B_ROWB + (((wn * NT + j) ^ hq) << 5);
        b[j] = tr_frag(bp, bp + 4 * B_ROWB);
      }
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const unsigned char* ap = st + a_k + kk * 32 * A_ROWB + (((wm * MT + i) ^ hq) << 5);
        const bf16x8 a = tr_frag(ap, ap + 4 * A_ROWB);
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a, acc[i][j], 0, 0, 0);
      }
    }
    __syncthreads();
  }

  float* out32 = p.out32 + (long)split * p.slab_stride;
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int m = row0 + wm * (16 * MT) + i * 16 + (lane & 15);
    if (m >= p.M) continue;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int n0 = col0 + wn * (BN / 4) + j * 16 + 4 * (lane >> 4);
      if (n0 < p.N) epilogue_store4<VL_EPI_F32>(p, out32, m, n0, acc[i][j]);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// generic path
// ---------------------------------------------------------------------------------------------------------------
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
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b_h[j], a_l[i], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b_l[j], a_h[i], acc[i][j], 0, 0, 0);
          }
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b_h[j], a_h[i], acc[i][j], 0, 0, 0);
        }
    }
  }

  // epilogue (D^T layout, see epilogue_store4)
  float* out32 = p.out32 + (long)blockIdx.y * p.slab_stride;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = row0 + wm * 64 + i * 16 + (lane & 15);
    if (m >= p.M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n0 = col0 + wn * 64 + j * 16 + 4 * (lane >> 4);
      if (n0 < p.N) epilogue_store4<EPI>(p, out32, m, n0, acc[i][j]);
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

// ---------------------------------------------------------------------------------------------------------------
// small-M path (M = batch products: the pooled rows of the last layer, pooler / classifier head): with M = 256 the big
// tiles give 4-16 workgroups, each walking the whole K with two K-tiles in flight -- pure memory latency (141 us for
// the pooled FFN2, 3.6 GFLOP).  Here: 64 x 64 tile (4 waves as 2 x 2, wave tile 32 x 32), BK = 64, register-staged
// prefetch, and the K range split over blockIdx.y so that ~500 workgroups (several per CU) have all of the operands
// in flight at once; every workgroup stores its raw fp32 partial tile into slab y of a caller-owned workspace and a
// second launch sums the slabs in a fixed order (deterministic) and applies the epilogue.
// ---------------------------------------------------------------------------------------------------------------
constexpr int SB = 64;
constexpr int STILE_BYTES = SB * BK * 2;  // 8 KiB per operand tile
template <int NSPLIT>
__global__ __launch_bounds__(256) void gemm_small_kernel(GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* sA_hi = smem;
  unsigned char* sB_hi = smem + STILE_BYTES;
  unsigned char* sA_lo = smem + 2 * STILE_BYTES;
  unsigned char* sB_lo = smem + 3 * STILE_BYTES;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int tm = blockIdx.x / p.tiles_n, tn = blockIdx.x - tm * p.tiles_n;
  const int row0 = tm * SB, col0 = tn * SB;
  f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int kbeg = blockIdx.y * p.k_len;
  const int kend = min(p.K, kbeg + p.k_len);
  uint4 ra_hi[2], rb_hi[2], ra_lo[2], rb_lo[2];
  auto load_tile = [&](int k0) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
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
    for (int i = 0; i < 2; ++i) {
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
  if (nk > 0) load_tile(kbeg);
  const int frow = lane & 15, fk = lane >> 4;
  for (int kt = 0; kt < nk; ++kt) {
    __syncthreads();
    store_tile();
    __syncthreads();
    if (kt + 1 < nk) load_tile(kbeg + (kt + 1) * BK);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8 a_h[2], b_h[2], a_l[2], b_l[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int ra = wm * 32 + i * 16 + frow, rb = wn * 32 + i * 16 + frow;
        const int oa = lds_off(ra, kk * 4 + fk), ob = lds_off(rb, kk * 4 + fk);
        a_h[i] = *reinterpret_cast<const bf16x8*>(sA_hi + oa);
        b_h[i] = *reinterpret_cast<const bf16x8*>(sB_hi + ob);
        if (NSPLIT == 3) {
          a_l[i] = *reinterpret_cast<const bf16x8*>(sA_lo + oa);
          b_l[i] = *reinterpret_cast<const bf16x8*>(sB_lo + ob);
        }
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          if (NSPLIT == 3) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b_h[j], a_l[i], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b_l[j], a_h[i], acc[i][j], 0, 0, 0);
          }
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b_h[j], a_h[i], acc[i][j], 0, 0, 0);
        }
    }
  }
  // raw partial tile -> slab blockIdx.y of the workspace [splits][M][ldc], ldc = N rounded up to 4 (D^T layout: a lane
  // holds 4 consecutive columns of row m)
  float* out = p.out32 + (long)blockIdx.y * p.slab_stride;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int m = row0 + wm * 32 + i * 16 + (lane & 15);
    if (m >= p.M) continue;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n0 = col0 + wn * 32 + j * 16 + 4 * (lane >> 4);
      if (n0 < p.ldc)
        *reinterpret_cast<float4*>(out + (long)m * p.ldc + n0) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
    }
  }
}

// sum of the slabs (fixed order) + the epilogue; one thread per 4 consecutive columns of a row
template <int EPI>
__global__ __launch_bounds__(256) void small_epilogue_kernel(GemmArgs p, const float* __restrict__ ws, int splits, long slab,
                                                             int ldw) {
  const long n4 = ldw >> 2;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long)p.M * n4) return;
  const int m = (int)(idx / n4), n0 = (int)(idx - (long)m * n4) * 4;
  const float* src = ws + (long)m * ldw + n0;
  float4 t = *reinterpret_cast<const float4*>(src);
  for (int s = 1; s < splits; ++s) {
    const float4 u = *reinterpret_cast<const float4*>(src + (long)s * slab);
    t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
  }
  epilogue_store4<EPI>(p, p.out32, m, n0, f32x4{t.x, t.y, t.z, t.w});
}

// K ranges of the small path: ~512 workgroups, at least 2 K-tiles per workgroup
struct SmallPlan { int splits; int k_len; int ldw; };
inline SmallPlan small_plan(int64_t M, int64_t N, int64_t K) {
  const int64_t tiles = ((M + SB - 1) / SB) * ((N + SB - 1) / SB), steps = (K + BK - 1) / BK;
  int64_t splits = 512 / tiles;
  if (splits > steps / 2) splits = steps / 2;
  if (splits < 1) splits = 1;
  const int64_t per = (steps + splits - 1) / splits;
  SmallPlan pl;
  pl.k_len = (int)(per * BK);
  pl.splits = (int)((steps + per - 1) / per);
  pl.ldw = (int)((N + 3) / 4 * 4);
  return pl;
}
// the small path pays while the big tiles cannot fill ~40 % of the chip (measured switch-over, tools/small_gemm_bench.py:
// M = 2048 x N = 3072 and M = 4096 x N = 2304 are ties, below that the small path wins by 1.3-7x)
inline bool small_shape(int64_t M, int64_t N) { return ((M + 255) / 256) * ((N + 191) / 192) < 100; }

template <int NSPLIT, int EPI>
int launch_small(const GemmArgs& a, float* ws, hipStream_t stream) {
  const SmallPlan pl = small_plan(a.M, a.N, a.K);
  const size_t lds = (NSPLIT == 3 ? 4 : 2) * STILE_BYTES;
  GemmArgs s = a;
  s.tiles_m = (a.M + SB - 1) / SB; s.tiles_n = (a.N + SB - 1) / SB;
  s.out32 = ws; s.ldc = pl.ldw; s.k_len = pl.k_len; s.slab_stride = (long)a.M * pl.ldw;
  hipLaunchKernelGGL((gemm_small_kernel<NSPLIT>), dim3(s.tiles_m * s.tiles_n, pl.splits), dim3(256), lds, stream, s);
  VL_CHECK_LAUNCH("vl_gemm_nt(small)");
  const long n = (long)a.M * (pl.ldw / 4);
  hipLaunchKernelGGL((small_epilogue_kernel<EPI>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, a, ws, pl.splits,
                     (long)a.M * pl.ldw, pl.ldw);
  VL_CHECK_LAUNCH("vl_gemm_nt(small epilogue)");
  return 0;
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

template <int NSPLIT, int EPI, int BN, int WM>
int launch2w(GemmArgs a, hipStream_t stream, int splits) {
  const size_t lds = 2 * ((256 + BN) / 8) * 1024;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm2_kernel<NSPLIT, EPI, BN, WM>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return vl_set_error(-3, "vl_gemm_nt: hipFuncSetAttribute: %s", hipGetErrorString(e));
    attr_set = true;
  }
  a.tiles_m = (a.M + 255) / 256;
  a.tiles_n = (a.N + BN - 1) / BN;
  hipLaunchKernelGGL((gemm2_kernel<NSPLIT, EPI, BN, WM>), dim3(a.tiles_m * a.tiles_n, splits), dim3(WM * 256), lds,
                     stream, a);
  VL_CHECK_LAUNCH("vl_gemm_nt(fast)");
  return 0;
}
thread_local int tl_cs_rows = 0;  // scratch of one vl_gemm_nt_ex call (reported through VL_GX_COLSUM_ROWS)
template <int NSPLIT, int EPI, int CFG, bool PERSIST>
int launch3p(const GemmArgs& a, int grid, size_t lds, hipStream_t stream) {
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm3_kernel<NSPLIT, EPI, CFG, PERSIST>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return vl_set_error(-3, "vl_gemm_nt: hipFuncSetAttribute: %s", hipGetErrorString(e));
    attr_set = true;
  }
  hipLaunchKernelGGL((gemm3_kernel<NSPLIT, EPI, CFG, PERSIST>), dim3(grid), dim3(512), lds, stream, a);
  VL_CHECK_LAUNCH("vl_gemm_nt(ping-pong)");
  return 0;
}
template <int NSPLIT, int EPI, int CFG>
int launch3(GemmArgs a, hipStream_t stream) {
  constexpr int BMT = CFG == 2 ? 224 : 256, BNT = CFG == 1 ? 192 : 256;
  // two K-tiles of both operands, or (VL_GEMM_RING == 5) five half-tile slots per operand: 160 KB at 256 x 256, all of a CU's LDS
  const size_t lds = VL_GEMM_RING == 5 ? 5 * (128 + BNT / 2) * 128 : 2 * (BMT + BNT) * 128;
  a.tiles_m = (a.M + BMT - 1) / BMT;
  a.tiles_n = (a.N + BNT - 1) / BNT;
  tl_cs_rows = a.tiles_m * 2 * (CFG == 1 ? 4 : 2);  // column-sum partial rows this configuration writes
  const int tiles = a.tiles_m * a.tiles_n;
  // persistent form (VL_GX_PERSIST): instantiated for the epilogues the layer stack uses; it stages nothing through
  // LDS in its epilogue, so the K-major image / column-sum options keep the one-tile-per-workgroup form
  constexpr bool kPersistable = (NSPLIT == 3 && (EPI == VL_EPI_F32 || EPI == VL_EPI_SPLIT || EPI == VL_EPI_GELU_SPLIT)) ||
                                (NSPLIT == 1 && (EPI == VL_EPI_F32 || EPI == VL_EPI_DGELU_BF16 || EPI == VL_EPI_BF16));
  if constexpr (kPersistable) {
    if (a.persist > 0 && tiles > a.persist && !a.img && !a.cs)
      return launch3p<NSPLIT, EPI, CFG, true>(a, a.persist, lds, stream);
  }
  return launch3p<NSPLIT, EPI, CFG, false>(a, tiles, lds, stream);
}
template <int NSPLIT, int EPI, int BN>
int launch2(const GemmArgs& a, hipStream_t stream, int splits) {
  return launch2w<NSPLIT, EPI, BN, 4>(a, stream, splits);
}


// BN for the fast path: fewest "rounds x tile width" over the 256 CUs (one 512-thread workgroup per CU)
inline int pick_bn(int64_t M, int64_t N, int splits, int tile) {
  const int64_t tm = (M + 255) / 256;
  int best = 256;
  int64_t best_cost = -1;
  if (tile == 256 || tile == 192 || tile == 128) return tile;
  const int cands[3] = {256, 192, 128};
  for (int c = 0; c < 3; ++c) {
    const int bn = cands[c];
    const int64_t tiles = tm * ((N + bn - 1) / bn) * splits;
    const int64_t cost = ((tiles + 255) / 256) * bn;
    if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = bn; }
  }
  return best;
}

inline bool fast_ok(int64_t M, int64_t K, int64_t k_len, int passes, int tile) {
  const int ks = passes == 3 ? 32 : 64;
  return tile != 7 && M >= 256 && (K % ks) == 0 && (k_len % ks) == 0;
}

template <int NSPLIT, int EPI>
int launch_any(const GemmArgs& a, hipStream_t s, int splits = 1) {
  if (splits == 1 && a.ws && !a.cs && (a.tile == 8 || (a.tile == 0 && small_shape(a.M, a.N)))) {
    const SmallPlan pl = small_plan(a.M, a.N, a.K);
    if ((long)pl.splits * a.M * pl.ldw <= a.ws_floats) return launch_small<NSPLIT, EPI>(a, a.ws, s);
    if (a.tile == 8) return vl_set_error(-1, "vl_gemm_nt_ex: workspace too small for the small-M path");
  }
  const int g_pingpong = (a.tile == 0) ? 1 : (a.tile == 2 || a.tile == 3 || a.tile == 4 || a.tile == 5) ? a.tile : 0;
  if (splits == 1 && g_pingpong && a.M >= 256 && a.N >= 192 && (a.K % (NSPLIT == 3 ? 32 : 64)) == 0) {
    // tile shape by "rounds over the 256 CUs x tile area" (key 7: 1 = automatic, 2 / 3 / 5 = force 256x256 / 256x192 /
    // 224x256).  The 224-row tile exists for the two N = 3072 products (FFN1 forward, its dX backward): at
    // M = 14336 they are 672 tiles = 2.6 rounds of 256x256 but exactly 3 full rounds of 224x256.
    const int64_t tm = (a.M + 255) / 256;
    const int64_t c256 = ((tm * ((a.N + 255) / 256) + 255) / 256) * 256, c192 = ((tm * ((a.N + 191) / 192) + 255) / 256) * 192;
    // (Round 1 forced the wide tile for the 1-pass products -- the backward dX GEMMs, which share the chip with the dW
    // stream -- because it won beside the split-K dW kernels of that time; beside the long row-major dW GEMM that holds
    // 108 CUs per layer the cost model's choice wins again: N = 768 as 224 tiles of 256 x 192, 16.92 vs 17.13 ms / step.)
    const bool wide = g_pingpong == 2 || (g_pingpong != 3 && c256 <= c192);
    if constexpr ((NSPLIT == 3 && EPI == VL_EPI_GELU_SPLIT) || (NSPLIT == 1 && EPI == VL_EPI_DGELU_BF16)) {
      const int64_t c224 = (((a.M + 223) / 224) * ((a.N + 255) / 256) + 255) / 256 * 224;
      if (g_pingpong == 5 || (g_pingpong == 1 && wide && c224 < c256)) return launch3<NSPLIT, EPI, 2>(a, s);
    }
    return wide ? launch3<NSPLIT, EPI, 0>(a, s) : launch3<NSPLIT, EPI, 1>(a, s);
  }
  if (a.cs) return vl_set_error(-2, "vl_gemm_nt_ex: column sums are produced by the ping-pong kernel only (see vl_gemm_nt_path)");
  if (fast_ok(a.M, a.K, a.k_len, NSPLIT, a.tile)) {
    switch (pick_bn(a.M, a.N, splits, a.tile)) {
      case 256: return launch2<NSPLIT, EPI, 256>(a, s, splits);
      case 192: return launch2<NSPLIT, EPI, 192>(a, s, splits);
      default: return launch2<NSPLIT, EPI, 128>(a, s, splits);
    }
  }
  return launch<NSPLIT, EPI>(a, s, splits);
}

template <int NSPLIT>
int dispatch_epi(int epi, const GemmArgs& a, hipStream_t s) {
  switch (epi) {
    case VL_EPI_F32: return launch_any<NSPLIT, VL_EPI_F32>(a, s);
    case VL_EPI_GELU_SPLIT: return launch_any<NSPLIT, VL_EPI_GELU_SPLIT>(a, s);
    case VL_EPI_DGELU_BF16: return launch_any<NSPLIT, VL_EPI_DGELU_BF16>(a, s);
    case VL_EPI_BF16: return launch_any<NSPLIT, VL_EPI_BF16>(a, s);
    case VL_EPI_SPLIT: return launch_any<NSPLIT, VL_EPI_SPLIT>(a, s);
  }
  return vl_set_error(-1, "vl_gemm_nt: unknown epilogue %d", epi);
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

}  // namespace

// which kernel the automatic choice takes: 0 generic | 1 single-barrier | 2 ping-pong | 3 small-M (needs a workspace)
extern "C" int vl_gemm_nt_path(int64_t M, int64_t N, int64_t K, int passes, int has_ws) {
  if (has_ws && small_shape(M, N)) return 3;
  if (M >= 256 && N >= 192 && (K % (passes == 3 ? 32 : 64)) == 0) return 2;
  if (fast_ok(M, K, K, passes, 0)) return 1;
  return 0;
}

extern "C" int vl_gemm_nt(const void* a_hi, const void* a_lo, int64_t lda, const void* b_hi, const void* b_lo,
                          int64_t ldb, int64_t M, int64_t N, int64_t K, int passes, int epilogue,
                          const float* bias, const float* resid32, float* out32, int64_t ldc, void* out_hi,
                          void* out_lo, void* aux16, int64_t ld16, void* stream) {
  return vl_gemm_nt_ex(a_hi, a_lo, lda, b_hi, b_lo, ldb, M, N, K, passes, epilogue, bias, resid32, out32, ldc, out_hi, out_lo,
                       aux16, ld16, nullptr, stream);
}

// ... with the optional arguments of `extra` (HOST array of VL_GX_FIELDS int64, may be NULL): kernel / tile selection,
// small-M workspace, K-major image of out_hi, column-sum partials
extern "C" int vl_gemm_nt_ex(const void* a_hi, const void* a_lo, int64_t lda, const void* b_hi, const void* b_lo,
                             int64_t ldb, int64_t M, int64_t N, int64_t K, int passes, int epilogue,
                             const float* bias, const float* resid32, float* out32, int64_t ldc, void* out_hi,
                             void* out_lo, void* aux16, int64_t ld16, int64_t* extra, void* stream) {
  const int tile = extra ? (int)extra[VL_GX_TILE] : 0;
  const int64_t persist = extra ? extra[VL_GX_PERSIST] : 0;
  VL_CHECK_ARG(persist >= 0 && persist <= 1024, "vl_gemm_nt_ex: VL_GX_PERSIST must be in [0, 1024] (got %lld)", (long long)persist);
  float* ws = extra ? reinterpret_cast<float*>(static_cast<uintptr_t>(extra[VL_GX_WS])) : nullptr;
  const int64_t ws_floats = extra ? extra[VL_GX_WS_FLOATS] : 0;
  VL_CHECK_ARG(passes == 1 || passes == 3, "vl_gemm_nt: passes must be 1 or 3 (got %d)", passes);
  VL_CHECK_ARG(tile == 0 || (tile >= 2 && tile <= 8) || tile == 128 || tile == 192 || tile == 256,
               "vl_gemm_nt_ex: unknown tile selection %d", tile);
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
  GemmArgs a{};
  a.a_hi = (const bf16_raw*)a_hi; a.a_lo = (const bf16_raw*)a_lo;
  a.b_hi = (const bf16_raw*)b_hi; a.b_lo = (const bf16_raw*)b_lo;
  a.lda = lda; a.ldb = ldb; a.M = (int)M; a.N = (int)N; a.K = (int)K;
  a.bias = bias; a.resid = resid32; a.out32 = out32; a.ldc = ldc;
  a.out_hi = (bf16_raw*)out_hi; a.out_lo = (bf16_raw*)out_lo; a.aux16 = (bf16_raw*)aux16; a.ld16 = ld16;
  a.tiles_m = (int)((M + BM - 1) / BM); a.tiles_n = (int)((N + BN - 1) / BN);
  a.k_len = (int)K; a.slab_stride = 0; a.tile = tile; a.persist = (int)persist;
  VL_CHECK_ARG(tile != 8 || ws, "vl_gemm_nt_ex: tile 8 (small-M path) needs a workspace");
  VL_CHECK_ARG(!ws || ((reinterpret_cast<uintptr_t>(ws) & 15u) == 0 && ws_floats >= 0), "vl_gemm_nt_ex: workspace must be 16-byte aligned");
  a.ws = ws; a.ws_floats = ws_floats;
  if (extra && (extra[VL_GX_IMG] || extra[VL_GX_COLSUM])) {
    VL_CHECK_ARG(epilogue != VL_EPI_F32, "vl_gemm_nt_ex: the K-major image / column sums belong to the 16-bit epilogues");
    a.img = reinterpret_cast<bf16_raw*>(static_cast<uintptr_t>(extra[VL_GX_IMG]));
    a.img_n = extra[VL_GX_IMG_COLS];
    a.cs = reinterpret_cast<float*>(static_cast<uintptr_t>(extra[VL_GX_COLSUM]));
    VL_CHECK_ARG(!a.img || a.img_n >= N, "vl_gemm_nt_ex: the image needs VL_GX_IMG_COLS >= N");
    VL_CHECK_ARG(!a.cs || epilogue == VL_EPI_DGELU_BF16, "vl_gemm_nt_ex: column sums are wired for the DGELU epilogue");
  }
  tl_cs_rows = 0;
  a.vec8 = (ld16 & 7) == 0 && aligned16(bias) && aligned16(out_hi) && aligned16(out_lo) && aligned16(aux16);
  a.vec = ((ldc | ld16) & 3) == 0 && aligned16(bias) && aligned16(resid32) && aligned16(out32) &&
          ((reinterpret_cast<uintptr_t>(out_hi) | reinterpret_cast<uintptr_t>(out_lo) |
            reinterpret_cast<uintptr_t>(aux16)) & 7) == 0;
  hipStream_t s = (hipStream_t)stream;
  const int rc = passes == 3 ? dispatch_epi<3>(epilogue, a, s) : dispatch_epi<1>(epilogue, a, s);
  if (extra) extra[VL_GX_COLSUM_ROWS] = a.cs ? tl_cs_rows : 0;
  return rc;
}

// floats of workspace the small-M path wants for this shape (0: the shape never takes it)
extern "C" int64_t vl_gemm_small_ws_floats(int64_t M, int64_t N, int64_t K) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  const SmallPlan pl = small_plan(M, N, K);
  return (int64_t)pl.splits * M * pl.ldw;
}

// Split-K variant for the weight-gradient products dW[N_out,K_in] = dY^T X, whose reduction dimension is the
// B*S = 14336 rows of the batch while the output is only 36..144 tiles: `splits` workgroups per output tile each
// reduce a K-range into their own fp32 slab (plain stores), then one streaming pass sums the slabs (deterministic;
// float atomics would run at ~1.3 TB/s chip-wide and are slower here).
// splits so that (output tiles x splits) just fills the 256 CUs once with the fast path's 256 x BN tiles
extern "C" int64_t vl_gemm_splitk_plan(int64_t M, int64_t N, int64_t K) {
  if (M <= 0 || N <= 0 || K <= 0) return 1;
  int64_t best = 1;
  if (M >= 256 && K % 64 == 0) {
    const int64_t tiles = ((M + 255) / 256) * ((N + 191) / 192);
    best = 256 / tiles;
  } else {
    const int64_t tiles = ((M + 127) / 128) * ((N + 127) / 128);
    best = 640 / tiles;
  }
  const int64_t kmax = K / 256 > 0 ? K / 256 : 1;  // at least 4 K-steps per split
  if (best > kmax) best = kmax;
  if (best > 32) best = 32;
  if (best < 1) best = 1;
  return best;
}
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
  a.vec = (N & 3) == 0 && aligned16(out32) && aligned16(ws);
  a.tiles_m = (int)((M + BM - 1) / BM); a.tiles_n = (int)((N + BN - 1) / BN);
  int64_t k_len = (K + splits - 1) / splits;
  k_len = (k_len + BK - 1) / BK * BK;
  const int eff = (int)((K + k_len - 1) / k_len);
  a.k_len = (int)k_len;
  hipStream_t s = (hipStream_t)stream;
  if (eff == 1) {
    a.out32 = out32; a.slab_stride = 0; a.k_len = (int)K;
    return launch_any<1, VL_EPI_F32>(a, s, 1);
  }
  a.out32 = ws; a.slab_stride = M * N;
  if (int rc = launch_any<1, VL_EPI_F32>(a, s, eff)) return rc;
  const long n4 = (long)(M * N / 4);
  long g = (n4 + 255) / 256;
  if (g > 2048) g = 2048;
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)g), dim3(256), 0, s, ws, eff, n4, out32);
  VL_CHECK_LAUNCH("vl_gemm_nt_splitk(reduce)");
  return 0;
}

// dW[M,N] = A^T B with A [K, M] (lda), B [K, N] (ldb) row-major bf16 (the activations as they sit in HBM), K = the
// B*S batch rows, split over `splits` workgroups per output tile (see vl_gemm_nt_splitk).  Needs K % 64 == 0,
// M, N multiples of 8 and >= 16; otherwise returns -2 and the caller uses the transposing path.
namespace {
int tn_splitk_impl(const void* a, int64_t lda, const void* b, int64_t ldb, int64_t M, int64_t N, int64_t K,
                   int64_t splits, float* ws, float* out32, void* stream) {
  GemmArgs g{};
  g.a_hi = (const bf16_raw*)a; g.b_hi = (const bf16_raw*)b;
  g.lda = lda; g.ldb = ldb; g.M = (int)M; g.N = (int)N; g.K = (int)K;
  g.ldc = N; g.vec = 1;
  int64_t k_len = (K + splits - 1) / splits;
  k_len = (k_len + 63) / 64 * 64;
  const int eff = (int)((K + k_len - 1) / k_len);
  g.k_len = (int)k_len;
  const bool direct = eff == 1;  // a single K-range writes the result itself
  g.out32 = direct ? out32 : ws;
  g.slab_stride = direct ? 0 : M * N;
  const int bn = (N % 256 == 0 || N > 1024) ? 256 : 128;
  g.tiles_m = (int)((M + 255) / 256);
  g.tiles_n = (int)((N + bn - 1) / bn);
  g.splits = eff;
  hipStream_t s = (hipStream_t)stream;
  const size_t lds = 2 * (size_t)(64 * 512 + 64 * bn * 2);
  hipError_t e = bn == 256 ? hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm2_tn_kernel<256>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)
                           : hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm2_tn_kernel<128>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return vl_set_error(-3, "vl_gemm_tn_splitk: hipFuncSetAttribute: %s", hipGetErrorString(e));
  if (bn == 256)
    hipLaunchKernelGGL((gemm2_tn_kernel<256>), dim3(g.tiles_m * g.tiles_n * eff), dim3(1024), lds, s, g);
  else
    hipLaunchKernelGGL((gemm2_tn_kernel<128>), dim3(g.tiles_m * g.tiles_n * eff), dim3(1024), lds, s, g);
  VL_CHECK_LAUNCH("vl_gemm_tn_splitk");
  if (!direct) {
    const long n4 = (long)(M * N / 4);
    long gr = (n4 + 255) / 256;
    if (gr > 2048) gr = 2048;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)gr), dim3(256), 0, s, ws, eff, n4, out32);
    VL_CHECK_LAUNCH("vl_gemm_tn_splitk(reduce)");
  }
  return 0;
}
inline bool tn_shape_ok(const void* a, int64_t lda, const void* b, int64_t ldb, int64_t M, int64_t N, int64_t K,
                        const void* ws) {
  return !((K % 64) != 0 || (M & 7) || (N & 7) || M < 256 || N < 128 || (lda & 7) || (ldb & 7) || !aligned16(a) ||
           !aligned16(b) || ((M * N) & 3) || !aligned16(ws));
}
}  // namespace

extern "C" int vl_gemm_tn_splitk(const void* a, int64_t lda, const void* b, int64_t ldb, int64_t M, int64_t N,
                                 int64_t K, int64_t splits, float* ws, float* out32, void* stream) {
  VL_CHECK_ARG(a && b && out32 && M > 0 && N > 0 && K > 0 && splits >= 1 && splits <= 64 && (splits == 1 || ws),
               "vl_gemm_tn_splitk: bad arguments");
  if (!tn_shape_ok(a, lda, b, ldb, M, N, K, ws) || !aligned16(out32))
    return vl_set_error(-2, "vl_gemm_tn_splitk: shape not supported by the TN fast path");
  return tn_splitk_impl(a, lda, b, ldb, M, N, K, splits, ws, out32, stream);
}
