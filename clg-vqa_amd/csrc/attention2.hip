// vl_attn2_fwd / vl_attn2_bwd: the V&L attention core over the single stream X = [text ; boxes] on the bf16 matrix
// pipe (v_mfma_f32_16x16x32_bf16), reading the (hi, lo) bf16 split of Q/K/V that the QKV projection's epilogue writes.
//
// Reference: volta/volta/encoders.py:255-341 (BertGatedSelfAttention.forward): with UC2's has_tt = tv = vt = vv and
// shared weights the four gated score blocks, the two concatenated softmaxes, the four dropouts and the four P.V
// products are ONE multi-head attention over S = T + V tokens with the additive key mask [t_mask ; v_mask]
// (encoders.py:978-995); M3P's MultiHeadAttention (volta/volta/m3p_transformer.py:152-212) is the same op with a -inf
// key mask.  Queries at padded positions are still computed.
//
// Precision (DESIGN.md "Precision"): forward = 3-term split products x ~= hi + lo (16 significant bits per operand):
// Q.K^T as Kh.Ql + Kl.Qh + Kh.Qh and P.V with P split in registers -- fp32-grade like the projections around it;
// backward = single-pass bf16 on the hi halves, like every other backward product of the engine.
//
// One workgroup per (sample, head) -- 4 waves for S <= 80, 8 waves beyond (at S = 120 / 140 the LDS tiles allow two
// workgroups per CU: 16 resident waves instead of 8 / 4 of round 2) -- with K and V (forward) / Q, K, V, dO (backward) in LDS as
// [row][64] bf16 at a 128-byte pitch, the 16-byte chunks of a row XOR-swizzled with sw(row) = ((row >> 1) & 3) << 1.  The
// swizzle is chosen by measurement (tools/lds_probe.hip: one kernel per access pattern under SQ_LDS_BANK_CONFLICT): of eight
// candidates only this one and row & 7 are conflict-free for all three patterns -- the ds_read_b128 operand reads (16
// consecutive rows, one chunk), the transposing ds_read_b64_tr_b16 reads (8 rows x one 32-byte block) and the staging
// stores.  Round 3's first choice (the same | (row >> 3) & 1) was conflict-free on paper and measured 0.5 on the b128 reads
// (0.22 / 0.28 for the whole kernels); gemm.hip's swizzles measure 0.5 on the transposing reads.  Round 2's padded 144-byte
// pitch measured 0.36 - 0.38 SQ_LDS_BANK_CONFLICT per active LDS cycle and
// cost 12 % more LDS.  Head dim 32 (the tiny c1 config) keeps a padded 80-byte pitch.  S <= 160: a query tile's whole key
// range lives in registers (plain softmax).
//
// Register-resident probabilities.  Scores are computed TRANSPOSED, S^T[key][query] = K.Q^T: the MFMA result layout
// (lane: column = lane & 15, rows 4*(lane >> 4) + r) then gives a lane ONE query and 4 consecutive keys per key tile,
// so the softmax reductions are in-lane plus two cross-lane steps, and the same registers ARE the B operand of the
// next product contracting over keys (ctx^T = V^T.P^T forward, dQ^T = K^T.dS^T backward): P never goes through LDS.
// The 8 k-slots of a lane's operand are keys {16t + 4g + e, 16(t+1) + 4g + e} (g = lane >> 4, e < 4), a permutation of
// the 32 keys of a tile pair; the other operand (V^T / K^T) is gathered with transposing LDS reads using the same
// permutation, so the contraction is unchanged.  The backward products that contract over QUERIES (dK, dV) use the
// non-transposed layout the same way, in a second phase that owns key tiles (no cross-wave reduction, no atomics).
//
// Dropout on the probabilities: counter-based, one 32-bit hash per (query, key pair), regenerated in backward.
#include "common.h"
#include "../../include/vlhip.h"

namespace {

// timing experiments with WRONG RESULTS (variant builds only, tools/attn_exp.sh): -DVL_EXP_ATTN=<bits> takes out of the forward
// kernel 1 = the input loads, 2 = everything between staging and the output stores, 4 = the output stores
#ifdef VL_EXP_ATTN
#define VL_EXP_OFF(bit) (((VL_EXP_ATTN) & (bit)) && p.S != -1)
#else
#define VL_EXP_OFF(bit) false
#endif

// head dim DH (64: the model configs; 32: the tiny "c1" plumbing config) is a template parameter: KS = DH / 32 k-steps
// per Q.K^T product, DT = DH / 16 output column tiles, LDS row pitch DH * 2 + 16 bytes
template <int DH> struct Geo {
  static constexpr int KS = DH / 32, DT = DH / 16, PITCH = DH == 64 ? 128 : DH * 2 + 16;
  // chunk swizzle of a row (16-byte chunks; 0 for the padded layout)
  static __device__ __forceinline__ int sw(int row) { return DH == 64 ? (((row >> 1) & 3) << 1) : 0; }
  // byte offset of 16-byte chunk `chunk` of `row` (ds_read_b128 operand fragments, staging)
  static __device__ __forceinline__ int chunk_off(int row, int chunk) { return row * PITCH + ((chunk ^ sw(row)) << 4); }
  // byte offset of this lane's 8 bytes of a transposing read: columns 16 dt + 4 pp .. + 3 of `row`
  static __device__ __forceinline__ int tr_off(int row, int dt, int pp) {
    return row * PITCH + (((2 * dt + (pp >> 1)) ^ sw(row)) << 4) + 8 * (pp & 1);
  }
};

typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;

struct Attn2Args {
  const bf16_raw* qkv_hi; const bf16_raw* qkv_lo; const float* addmask;
  bf16_raw* ctx_hi; bf16_raw* ctx_lo; float* lse;
  const bf16_raw* dctx; bf16_raw* dqkv;
  int B, S, nh, H;   // H = nh * 64
  int nq;            // queries per sample that are computed / carry a gradient (S, or 1: only the pooled row is live)
  int ctx_rows;      // rows per sample of ctx / dctx (S, or 1 in the compact layout)
  float scale, p_drop, inv_keep;
  unsigned seed_lo, seed_hi, thr16;
};

__device__ __forceinline__ f32x4 mfma(bf16x8 a, bf16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ bf16x8 lds_frag(const unsigned char* p) { return *reinterpret_cast<const bf16x8*>(p); }
// 8 k-values (rows r0.., r1.. of the LDS image; 4 each) of column (lane & 15) of a 16-column block: see gemm.hip tr_frag
__device__ __forceinline__ bf16x8 tr_frag2(const unsigned char* p0, const unsigned char* p1) {
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p0);
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p1);
  const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8, v);
}
__device__ __forceinline__ bf16x8 zero_frag() {
  const s16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
  return __builtin_bit_cast(bf16x8, z);
}
__device__ __forceinline__ bf16x8 pack8(const float* x) {
  s16x8 v;
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = (short)f32_to_bf16(x[i]);
  return __builtin_bit_cast(bf16x8, v);
}
__device__ __forceinline__ void pack8_split(const float* x, bf16x8& hi, bf16x8& lo) {
  s16x8 h, l;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    bf16_raw a, b;
    split_bf16(x[i], a, b);
    h[i] = (short)a; l[i] = (short)b;
  }
  hi = __builtin_bit_cast(bf16x8, h);
  lo = __builtin_bit_cast(bf16x8, l);
}

// Output rows as 16-byte stores.  The MFMA result layout leaves a lane (row = lane & 15, g = lane >> 4) with columns 16 dt +
// 4 g .. + 3 of its row for every column tile dt: written as they lie that is an 8-byte store per tile, 32-byte pieces of
// 16 different rows per instruction, every 128-byte line of the output assembled from four partial writes (measured:
// the forward kernel's 44 MB of stores alone took 22 us).  v_permlane16_swap_b32 exchanges, between the lane pairs
// (g, g ^ 1), the halves they need from each other: after it a lane holds 8 consecutive columns 32 m + 4 g + 12 (g & 1)
// of tile pair m -- one 16-byte store, 64-byte pieces per row, half the store instructions.  All 64 lanes must be active.
__device__ __forceinline__ uint4 pair_cols(uint2 a, uint2 b) {  // a: tile 2m, b: tile 2m + 1 (4 bf16 each)
  typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
  const u32x2 x = __builtin_amdgcn_permlane16_swap(a.x, b.x, false, false);
  const u32x2 y = __builtin_amdgcn_permlane16_swap(a.y, b.y, false, false);
  return make_uint4(x[0], y[0], x[1], y[1]);
}
__device__ __forceinline__ int pair_col0(int m, int g) { return 32 * m + 4 * g + 12 * (g & 1); }
__device__ __forceinline__ uint2 pack4(const f32x4& v) {
  return make_uint2((unsigned)f32_to_bf16(v[0]) | ((unsigned)f32_to_bf16(v[1]) << 16),
                    (unsigned)f32_to_bf16(v[2]) | ((unsigned)f32_to_bf16(v[3]) << 16));
}
__device__ __forceinline__ void pack4_split(const f32x4& v, uint2& hi, uint2& lo) {
  bf16_raw h[4], l[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) split_bf16(v[r], h[r], l[r]);
  hi = make_uint2((unsigned)h[0] | ((unsigned)h[1] << 16), (unsigned)h[2] | ((unsigned)h[3] << 16));
  lo = make_uint2((unsigned)l[0] | ((unsigned)l[1] << 16), (unsigned)l[2] | ((unsigned)l[3] << 16));
}
// one row of DT column tiles (fp32 accumulators) -> bf16 at `row` (pointer to column 0 of this lane's row); `ok`: store
template <int DT>
__device__ __forceinline__ void store_row_bf16(bf16_raw* row, const f32x4 (&o)[DT], int g, bool ok) {
#pragma unroll
  for (int m = 0; m < DT / 2; ++m) {
    const uint4 v = pair_cols(pack4(o[2 * m]), pack4(o[2 * m + 1]));
    if (ok) *reinterpret_cast<uint4*>(row + pair_col0(m, g)) = v;
  }
}

// 32 random bits per (row, key pair): two 16-bit uniform draws (low half = even key, high half = odd key)
__device__ __forceinline__ unsigned hash32(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}
// (ONE hash32 per key pair: the 64-bit seed is mixed on the host (splitmix64) into the additive / xor constants, the counter is
// spread by the odd multiplier -- two rounds of hash32 per pair were 10 % of the S = 140 kernels' vector work)
__device__ __forceinline__ unsigned drop_bits(const Attn2Args& p, unsigned row, unsigned pair, unsigned pairs_per_row) {
  return hash32(((row * pairs_per_row + pair) * 0x9E3779B1u + p.seed_lo) ^ p.seed_hi);
}
__device__ __forceinline__ float keep_of(const Attn2Args& p, unsigned bits, int odd) {
  const unsigned u = odd ? (bits >> 16) : (bits & 0xFFFFu);
  return u >= p.thr16 ? p.inv_keep : 0.0f;
}

// stage rows [0, SPAD) x DH bf16 of NM matrices into their LDS tiles (zeros beyond rows[m]); src row pitch ld[m] elements.
// ALL loads of all matrices are issued before the first LDS store: written as a loop of "if (row < rows) load; store" the
// compiler emitted load -> s_waitcnt vmcnt(0) -> ds_write per chunk, i.e. 8 - 20 memory round trips one after the other at
// the head of every workgroup.  Loads are unconditional (row clamped to 0, value replaced by zeros afterwards) so that the
// whole batch is one basic block.
template <int DH, int NTHR, int SPAD, int NM> struct StageBatch {
  static constexpr int CH = DH / 8, TOT = SPAD * CH, IT = (TOT + NTHR - 1) / NTHR;
  static constexpr bool EXACT = TOT % NTHR == 0;
  uint4 v[NM][IT];
  __device__ __forceinline__ void load(const bf16_raw* const (&src)[NM], const long (&ld)[NM], const int (&rows)[NM], int tid) {
#pragma unroll
    for (int m = 0; m < NM; ++m)
#pragma unroll
      for (int i = 0; i < IT; ++i) {
        const int idx = tid + i * NTHR, row = idx / CH, c = idx - row * CH;
        const bool ok = row < rows[m];  // (idx >= TOT -> row >= SPAD >= rows)
        v[m][i] = *reinterpret_cast<const uint4*>(src[m] + (long)(ok ? row : 0) * ld[m] + c * 8);
      }
  }
  __device__ __forceinline__ void store(unsigned char* const (&dst)[NM], const int (&rows)[NM], int tid) const {
#pragma unroll
    for (int m = 0; m < NM; ++m)
#pragma unroll
      for (int i = 0; i < IT; ++i) {
        const int idx = tid + i * NTHR, row = idx / CH, c = idx - row * CH;
        if (EXACT || idx < TOT)
          *reinterpret_cast<uint4*>(dst[m] + Geo<DH>::chunk_off(row, c)) = row < rows[m] ? v[m][i] : make_uint4(0u, 0u, 0u, 0u);
      }
  }
};

// ---------------------------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------------------------
// register budgets (workgroups per CU the allocation must allow), measured with tools/attn_bench.py: backward at S <= 128
// gains 5 % from a third resident workgroup (197 -> 188 us at B = 256, S = 120), every other variant loses
// (second argument of __launch_bounds__ = waves per SIMD the register allocation must allow: NW = 8 waves x 2 workgroups per
// CU = 4; the 4-wave instances keep round 2's budgets)
// LDS admits two workgroups per CU up to 9 key tiles (S <= 144), one at 10: the register budgets follow
#ifndef VL_ATTN_FWD_MINW
#define VL_ATTN_FWD_MINW(NT, NW) ((NT) >= 10 ? (NW) / 4 : (NT) >= 8 ? (NW) / 2 : 1)
#endif
#ifndef VL_ATTN_BWD_MINW
#define VL_ATTN_BWD_MINW(NT, NW) ((NT) >= 10 ? (NW) / 4 : (NT) >= 8 ? ((NW) == 8 ? 2 : 2) : 1)
#endif
// waves per workgroup of the long-sequence instances, measured with tools/attn_bench.py at B = 256, dropout 0.1 (us, cold):
//   forward  S = 120: 8 waves 128, 4 waves 150 | S = 140: 177 / 187 | S = 156 (B 128): 130 / 181          -> 8 waves
//   backward S = 120: 8 waves 215, 4 waves 191 - 199 | S = 140: 313 / 257 - 263 | S = 156 (B 128): 174 / 259
//            -> 4 waves while two workgroups fit a CU (<= 9 key tiles; 166 - 190 VGPRs rule out 2 x 8 waves), 8 beyond
#ifndef VL_ATTN_FWD_NW
#define VL_ATTN_FWD_NW 8
#endif
#ifndef VL_ATTN_BWD_NW
#define VL_ATTN_BWD_NW(NT) ((NT) >= 10 ? 8 : 4)
#endif
template <int NT, int DH, int NW>
__global__ __launch_bounds__(NW * 64, VL_ATTN_FWD_MINW(NT, NW)) void attn2_fwd_kernel(Attn2Args p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
  constexpr int Spad = NT * 16, NP = (NT + 1) / 2;  // key-tile pairs
  constexpr int KS = Geo<DH>::KS, DT = Geo<DH>::DT, PITCH = Geo<DH>::PITCH;
  unsigned char* sKh = sm;
  unsigned char* sKl = sKh + Spad * PITCH;
  unsigned char* sVh = sKl + Spad * PITCH;
  unsigned char* sVl = sVh + Spad * PITCH;
  float* smask = reinterpret_cast<float*>(sVl + Spad * PITCH);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.x / p.nh, h = blockIdx.x - b * p.nh;
  const int S = p.S;
  const long ld = 3L * p.H;
  const bf16_raw* bh = p.qkv_hi + (long)b * S * ld + h * DH;
  const bf16_raw* bl = p.qkv_lo + (long)b * S * ld + h * DH;
  const int l15 = lane & 15, g = lane >> 4, qq = l15 >> 2, pp = lane & 3;
  const int nqt = (p.nq + 15) >> 4;
  // everything this workgroup reads before its first product is in flight at once: key mask, K / V (hi, lo), and the first
  // query tile's rows (B operand of S^T = K.Q^T: Q[q][32 ks + 8 g ..], straight into registers)
  const float mask_v = p.addmask[(long)b * S + (tid < S ? tid : 0)];
  StageBatch<DH, NW * 64, Spad, 4> st;
  const bf16_raw* const srcs[4] = {bh + p.H, bl + p.H, bh + 2 * p.H, bl + 2 * p.H};
  unsigned char* const dsts[4] = {sKh, sKl, sVh, sVl};
  const long lds4[4] = {ld, ld, ld, ld};
  const int rows4[4] = {S, S, S, S};
#ifdef VL_EXP_ATTN
  for (int m = 0; m < 4; ++m)
    for (int i = 0; i < st.IT; ++i) st.v[m][i] = make_uint4(0u, 0u, 0u, 0u);
#endif
  if (!VL_EXP_OFF(1)) st.load(srcs, lds4, rows4, tid);
  bf16x8 qh[KS], ql[KS];
#ifdef VL_EXP_ATTN
  for (int ks = 0; ks < KS; ++ks) qh[ks] = ql[ks] = zero_frag();
#endif
  auto load_q = [&](int qt) {
    const int q = qt * 16 + l15, qr = q < S ? q : 0;  // (rows past S: any finite values, their results are never stored)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      qh[ks] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(bh + (long)qr * ld + 32 * ks + 8 * g));
      ql[ks] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(bl + (long)qr * ld + 32 * ks + 8 * g));
    }
  };
  // query tiles go to waves round-robin from a per-workgroup start: when NT is not a multiple of NW (9 tiles on 8 waves) the
  // wave with one tile more is a different one -- on a different SIMD -- in the workgroups that share a CU
  const int w0 = NT % NW == 0 ? wave : (wave + (int)((blockIdx.x * 0x9E3779B1u) >> 16)) & (NW - 1);
  if (w0 < nqt && !VL_EXP_OFF(1)) load_q(w0);
  if (tid < Spad) smask[tid] = tid < S ? mask_v : -INFINITY;
  st.store(dsts, rows4, tid);
  __syncthreads();

  for (int qt = w0; qt < nqt; qt += NW) {
    const int q = qt * 16 + l15;
    f32x4 o[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (!VL_EXP_OFF(2)) {
    f32x4 sc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      sc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const int off = Geo<DH>::chunk_off(t * 16 + l15, g + 4 * ks);
        const bf16x8 kh = lds_frag(sKh + off), kl = lds_frag(sKl + off);
        sc[t] = mfma(kh, ql[ks], sc[t]);
        sc[t] = mfma(kl, qh[ks], sc[t]);
        sc[t] = mfma(kh, qh[ks], sc[t]);
      }
      asm volatile("" ::: "memory");  // keep the next tile's LDS reads here: hoisting all NT tiles' fragments costs ~100 VGPRs
    }
    // the wave's next query tile arrives under the softmax and P.V of this one (9 key tiles: no registers to spare under the
    // 128-VGPR budget of two 8-wave workgroups per CU; loaded at the end of the trip instead)
    if (NT != 9 && qt + NW < nqt && !VL_EXP_OFF(1)) load_q(qt + NW);
    // softmax over the keys of query q: this lane holds keys 16 t + 4 g + r
    float m = -INFINITY;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const float4 mk = *reinterpret_cast<const float4*>(smask + 16 * t + 4 * g);
      sc[t][0] = sc[t][0] * p.scale + mk.x; sc[t][1] = sc[t][1] * p.scale + mk.y;
      sc[t][2] = sc[t][2] * p.scale + mk.z; sc[t][3] = sc[t][3] * p.scale + mk.w;
      m = fmaxf(m, fmaxf(fmaxf(sc[t][0], sc[t][1]), fmaxf(sc[t][2], sc[t][3])));
    }
    m = fmaxf(m, __shfl_xor(m, 16, 64));
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        sc[t][r] = __expf(sc[t][r] - m);
        sum += sc[t][r];
      }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.0f / sum;
    if (g == 0 && q < S) p.lse[((long)b * p.nh + h) * S + q] = m + __logf(sum);
    const unsigned row = (unsigned)((b * p.nh + h) * S + q);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      float k0 = 1.f, k1 = 1.f, k2 = 1.f, k3 = 1.f;
      if (p.p_drop > 0.f) {
        const unsigned b0 = drop_bits(p, row, (unsigned)(8 * t + 2 * g), Spad / 2);
        const unsigned b1 = drop_bits(p, row, (unsigned)(8 * t + 2 * g + 1), Spad / 2);
        k0 = keep_of(p, b0, 0); k1 = keep_of(p, b0, 1); k2 = keep_of(p, b1, 0); k3 = keep_of(p, b1, 1);
      }
      sc[t][0] *= inv * k0; sc[t][1] *= inv * k1; sc[t][2] *= inv * k2; sc[t][3] *= inv * k3;
    }
    // ctx^T[d][q] = sum_keys V^T[d][key] P^T[key][q]
#pragma unroll
    for (int pr = 0; pr < NP; ++pr) {
      const int t0 = 2 * pr, t1 = 2 * pr + 1;
      float pv[8];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        pv[r] = sc[t0][r];
        pv[4 + r] = t1 < NT ? sc[t1 < NT ? t1 : t0][r] : 0.f;
      }
      bf16x8 ph, pl;
      pack8_split(pv, ph, pl);
      const int vr0 = 16 * t0 + 4 * g + qq;
      const int vr1 = 16 * (t1 < NT ? t1 : t0) + 4 * g + qq;  // (the slots of a missing tile hold p = 0)
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        const int r0 = Geo<DH>::tr_off(vr0, dt, pp), r1 = Geo<DH>::tr_off(vr1, dt, pp);
        const bf16x8 vh = tr_frag2(sVh + r0, sVh + r1);
        const bf16x8 vl = tr_frag2(sVl + r0, sVl + r1);
        o[dt] = mfma(vh, pl, o[dt]);
        o[dt] = mfma(vl, ph, o[dt]);
        o[dt] = mfma(vh, ph, o[dt]);
      }
      asm volatile("" ::: "memory");
    }
    }  // (VL_EXP_OFF(2))
    {
      const bool ok = q < p.nq && !VL_EXP_OFF(4);
      const long off = ((long)b * p.ctx_rows + (ok ? q : 0)) * p.H + h * DH;
#pragma unroll
      for (int m = 0; m < DT / 2; ++m) {
        uint2 h0, l0, h1, l1;
        pack4_split(o[2 * m], h0, l0);
        pack4_split(o[2 * m + 1], h1, l1);
        const uint4 vh = pair_cols(h0, h1), vl = pair_cols(l0, l1);
        if (ok) {
          *reinterpret_cast<uint4*>(p.ctx_hi + off + pair_col0(m, g)) = vh;
          *reinterpret_cast<uint4*>(p.ctx_lo + off + pair_col0(m, g)) = vl;
        }
      }
    }
    if (NT == 9 && qt + NW < nqt && !VL_EXP_OFF(1)) load_q(qt + NW);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// backward.  dO = dctx (bf16).  With P = softmax, Pd = P * keep, dP = dO.V^T, delta = rowsum(Pd * dP) (= rowsum(dO * O)),
//   dS = P * (keep * dP - delta) * scale;   dV = Pd^T dO;   dK = dS^T Q;   dQ = dS K.
// Phase A (a wave owns query tiles, transposed layout): delta and dQ.  Phase B (a wave owns key tiles, plain layout,
// scores recomputed): dK and dV accumulate in registers over the query tiles.
// ---------------------------------------------------------------------------------------------------------------
template <int NT, int DH, int NW>
__global__ __launch_bounds__(NW * 64, VL_ATTN_BWD_MINW(NT, NW)) void attn2_bwd_kernel(Attn2Args p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
  constexpr int Spad = NT * 16, NP = (NT + 1) / 2;
  constexpr int KS = Geo<DH>::KS, DT = Geo<DH>::DT, PITCH = Geo<DH>::PITCH;
  unsigned char* sQ = sm;
  unsigned char* sK = sQ + Spad * PITCH;
  unsigned char* sV = sK + Spad * PITCH;
  unsigned char* sO = sV + Spad * PITCH;  // dO
  float* smask = reinterpret_cast<float*>(sO + Spad * PITCH);
  float* slse = smask + Spad;
  float* sdelta = slse + Spad;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.x / p.nh, h = blockIdx.x - b * p.nh;
  const int S = p.S;
  const long ld = 3L * p.H;
  const bf16_raw* bh = p.qkv_hi + (long)b * S * ld + h * DH;
  const int l15 = lane & 15, g = lane >> 4, qq = l15 >> 2, pp = lane & 3;
  const int nqt = (p.nq + 15) >> 4;      // query tiles that carry a gradient
  const int nqt_all = (S + 15) >> 4;
  const unsigned row0 = (unsigned)((b * p.nh + h) * S);
  bf16_raw* dq_base = p.dqkv + (long)b * S * ld + h * DH;
  // all of the workgroup's input in flight at once (see StageBatch)
  const int kc = tid < S ? tid : 0;
  const float mask_v = p.addmask[(long)b * S + kc];
  const float lse_v = p.lse[((long)b * p.nh + h) * S + kc];
  StageBatch<DH, NW * 64, Spad, 4> st;
  const bf16_raw* const srcs[4] = {bh, bh + p.H, bh + 2 * p.H, p.dctx + (long)b * p.ctx_rows * p.H + h * DH};
  unsigned char* const dsts[4] = {sQ, sK, sV, sO};
  const long lds4[4] = {ld, ld, ld, (long)p.H};
  const int rows4[4] = {S, S, S, p.nq};
  st.load(srcs, lds4, rows4, tid);
  if (tid < Spad) {
    smask[tid] = tid < S ? mask_v : -INFINITY;
    slse[tid] = tid < S ? lse_v : INFINITY;  // +inf -> P = 0 for padded queries
    sdelta[tid] = 0.f;
  }
  st.store(dsts, rows4, tid);
  __syncthreads();
  // tiles go to waves round-robin from a per-workgroup start, phase B half a turn after phase A: with 9 tiles on 4 waves the
  // third tile of the two phases lands on different waves (5 tile times per workgroup instead of 6), and on different SIMDs
  // in the workgroups that share a CU
  const int wA = NT % NW == 0 ? wave : (wave + (int)((blockIdx.x * 0x9E3779B1u) >> 16)) & (NW - 1);
  const int wB = NT % NW == 0 ? wave : (wA + NW / 2) & (NW - 1);

  // ---- phase A: delta and dQ (transposed layout: lane = one query, keys 16 t + 4 g + r) ------------------------
  for (int qt = wA; qt < nqt_all; qt += NW) {
    const int q = qt * 16 + l15;
    if (qt >= nqt) {  // rows without a gradient: dQ = 0
      if (q < S) {
#pragma unroll
        for (int m = 0; m < DT / 2; ++m)
          *reinterpret_cast<uint4*>(dq_base + (long)q * ld + pair_col0(m, g)) = make_uint4(0u, 0u, 0u, 0u);
      }
      continue;
    }
    bf16x8 fq[KS], fo[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      fq[ks] = lds_frag(sQ + Geo<DH>::chunk_off(q, g + 4 * ks));
      fo[ks] = lds_frag(sO + Geo<DH>::chunk_off(q, g + 4 * ks));
    }
    const float lse_q = slse[q];
    f32x4 pv[NT], kd[NT];  // P and keep * dP
    float delta = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f}, dp = s;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const int off = Geo<DH>::chunk_off(t * 16 + l15, g + 4 * ks);
        s = mfma(lds_frag(sK + off), fq[ks], s);
        dp = mfma(lds_frag(sV + off), fo[ks], dp);
      }
      const float4 mk = *reinterpret_cast<const float4*>(smask + 16 * t + 4 * g);
      const float mkv[4] = {mk.x, mk.y, mk.z, mk.w};
      float kp[4] = {1.f, 1.f, 1.f, 1.f};
      if (p.p_drop > 0.f) {
        const unsigned b0 = drop_bits(p, row0 + (unsigned)q, (unsigned)(8 * t + 2 * g), Spad / 2);
        const unsigned b1 = drop_bits(p, row0 + (unsigned)q, (unsigned)(8 * t + 2 * g + 1), Spad / 2);
        kp[0] = keep_of(p, b0, 0); kp[1] = keep_of(p, b0, 1); kp[2] = keep_of(p, b1, 0); kp[3] = keep_of(p, b1, 1);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float pr = __expf(s[r] * p.scale + mkv[r] - lse_q);
        const float x = kp[r] * dp[r];
        delta += pr * x;
        pv[t][r] = pr;
        kd[t][r] = x;
      }
      asm volatile("" ::: "memory");  // (see the forward kernel: keeps the LDS fragment reads of later tiles from being hoisted)
    }
    delta += __shfl_xor(delta, 16, 64);
    delta += __shfl_xor(delta, 32, 64);
    if (g == 0) sdelta[q] = delta;
    f32x4 o[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int pr = 0; pr < NP; ++pr) {
      const int t0 = 2 * pr, t1 = (2 * pr + 1 < NT) ? 2 * pr + 1 : 2 * pr;
      const bool has1 = 2 * pr + 1 < NT;
      float dsv[8];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        dsv[r] = pv[t0][r] * (kd[t0][r] - delta) * p.scale;
        dsv[4 + r] = has1 ? pv[t1][r] * (kd[t1][r] - delta) * p.scale : 0.f;
      }
      const bf16x8 dsb = pack8(dsv);
      const int kr0 = 16 * t0 + 4 * g + qq, kr1 = 16 * t1 + 4 * g + qq;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt)
        o[dt] = mfma(tr_frag2(sK + Geo<DH>::tr_off(kr0, dt, pp), sK + Geo<DH>::tr_off(kr1, dt, pp)), dsb, o[dt]);
      asm volatile("" ::: "memory");
    }
    store_row_bf16<DT>(dq_base + (long)(q < S ? q : 0) * ld, o, g, q < S);
  }
  __syncthreads();  // sdelta complete

  // ---- phase B: dK and dV (plain layout: lane = one key, queries 16 qt + 4 g + r) ------------------------------
  const int nqp = (nqt + 1) >> 1;  // query-tile pairs
  for (int kt = wB; kt < NT; kt += NW) {
    const int key = kt * 16 + l15;
    bf16x8 fk[KS], fv[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      fk[ks] = lds_frag(sK + Geo<DH>::chunk_off(key, g + 4 * ks));
      fv[ks] = lds_frag(sV + Geo<DH>::chunk_off(key, g + 4 * ks));
    }
    const float mk = smask[key];
    f32x4 dK[DT], dV[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) { dK[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; dV[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    for (int pq = 0; pq < nqp; ++pq) {
      float pdv[8], dsv[8];
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int qt = 2 * pq + half;
        if (qt < nqt) {
          f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f}, dp = s;
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) {
            const int off = Geo<DH>::chunk_off(qt * 16 + l15, g + 4 * ks);
            s = mfma(lds_frag(sQ + off), fk[ks], s);
            dp = mfma(lds_frag(sO + off), fv[ks], dp);
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int qr = qt * 16 + 4 * g + r;
            const float pr = __expf(s[r] * p.scale + mk - slse[qr]);
            float kp = 1.f;
            if (p.p_drop > 0.f) kp = keep_of(p, drop_bits(p, row0 + (unsigned)qr, (unsigned)(key >> 1), Spad / 2), key & 1);
            pdv[4 * half + r] = pr * kp;
            dsv[4 * half + r] = pr * (kp * dp[r] - sdelta[qr]) * p.scale;
          }
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) { pdv[4 * half + r] = 0.f; dsv[4 * half + r] = 0.f; }
        }
      }
      const bf16x8 bpd = pack8(pdv), bds = pack8(dsv);
      const int qt1 = (2 * pq + 1 < nqt) ? 2 * pq + 1 : 2 * pq;  // (the slots of a missing tile hold zeros)
      const int qr0 = 32 * pq + 4 * g + qq, qr1 = 16 * qt1 + 4 * g + qq;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        const int r0 = Geo<DH>::tr_off(qr0, dt, pp), r1 = Geo<DH>::tr_off(qr1, dt, pp);
        dV[dt] = mfma(tr_frag2(sO + r0, sO + r1), bpd, dV[dt]);
        dK[dt] = mfma(tr_frag2(sQ + r0, sQ + r1), bds, dK[dt]);
      }
    }
    {
      bf16_raw* out = p.dqkv + ((long)b * S + (key < S ? key : 0)) * ld + h * DH;
      store_row_bf16<DT>(out + p.H, dK, g, key < S);
      store_row_bf16<DT>(out + 2 * p.H, dV, g, key < S);
    }
  }
}

template <int NT, int DH, int NW>
int launch_fwd(const Attn2Args& a, hipStream_t s) {
  const size_t lds = (size_t)NT * 16 * (4 * Geo<DH>::PITCH + 4);
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn2_fwd_kernel<NT, DH, NW>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return vl_set_error(-3, "vl_attn2_fwd: hipFuncSetAttribute: %s", hipGetErrorString(e));
  hipLaunchKernelGGL((attn2_fwd_kernel<NT, DH, NW>), dim3(a.B * a.nh), dim3(NW * 64), lds, s, a);
  VL_CHECK_LAUNCH("vl_attn2_fwd");
  return 0;
}
template <int NT, int DH, int NW>
int launch_bwd(const Attn2Args& a, hipStream_t s) {
  const size_t lds = (size_t)NT * 16 * (4 * Geo<DH>::PITCH + 12);
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn2_bwd_kernel<NT, DH, NW>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return vl_set_error(-3, "vl_attn2_bwd: hipFuncSetAttribute: %s", hipGetErrorString(e));
  hipLaunchKernelGGL((attn2_bwd_kernel<NT, DH, NW>), dim3(a.B * a.nh), dim3(NW * 64), lds, s, a);
  VL_CHECK_LAUNCH("vl_attn2_bwd");
  return 0;
}
// key tiles NT (Spad = 16 NT) and waves per workgroup by sequence length; S in (128, 144] -- M3P's default 20 + 100 + ... = 140
// -- has its own instance: with 9 tiles instead of 10 two workgroups fit a CU's LDS (73.7 KB each)
template <int DH> int dispatch_fwd(const Attn2Args& a, hipStream_t s) {
  if (a.S <= 64) return launch_fwd<4, DH, 4>(a, s);
  if (a.S <= 80) return launch_fwd<5, DH, 4>(a, s);
  if (a.S <= 128) return launch_fwd<8, DH, VL_ATTN_FWD_NW>(a, s);
  if (a.S <= 144) return launch_fwd<9, DH, VL_ATTN_FWD_NW>(a, s);
  return launch_fwd<10, DH, VL_ATTN_FWD_NW>(a, s);
}
template <int DH> int dispatch_bwd(const Attn2Args& a, hipStream_t s) {
  if (a.S <= 64) return launch_bwd<4, DH, 4>(a, s);
  if (a.S <= 80) return launch_bwd<5, DH, 4>(a, s);
  if (a.S <= 128) return launch_bwd<8, DH, VL_ATTN_BWD_NW(8)>(a, s);
  if (a.S <= 144) return launch_bwd<9, DH, VL_ATTN_BWD_NW(9)>(a, s);
  return launch_bwd<10, DH, VL_ATTN_BWD_NW(10)>(a, s);
}

int fill_common(const char* fn, Attn2Args& a, int64_t B, int64_t S, int64_t nh, int64_t dh, int64_t nq, float p_drop,
                uint64_t seed) {
  VL_CHECK_ARG(dh == 64 || dh == 32, "%s: head dim must be 64 or 32 (got %lld)", fn, (long long)dh);
  VL_CHECK_ARG(S >= 1 && S <= 160, "%s: sequence length T+V must be in [1,160] (got %lld)", fn, (long long)S);
  VL_CHECK_ARG(B >= 1 && nh >= 1 && B * nh < (1LL << 30) && B * nh * S < (1LL << 31), "%s: bad B=%lld nh=%lld", fn,
               (long long)B, (long long)nh);
  VL_CHECK_ARG(nq >= 1 && nq <= S, "%s: nq must be in [1, S] (got %lld)", fn, (long long)nq);
  VL_CHECK_ARG(p_drop >= 0.f && p_drop < 1.f, "%s: dropout p must be in [0,1)", fn);
  a.B = (int)B; a.S = (int)S; a.nh = (int)nh; a.H = (int)(nh * dh); a.nq = (int)nq; a.ctx_rows = (int)nq;
  a.scale = 1.0f / sqrtf((float)dh); a.p_drop = p_drop; a.inv_keep = 1.0f / (1.0f - p_drop);
  uint64_t z = seed + 0x9E3779B97F4A7C15ull;  // splitmix64: neighbouring seeds (layers, steps) -> unrelated constants
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
  a.seed_lo = (unsigned)z; a.seed_hi = (unsigned)(z >> 32);
  a.thr16 = (unsigned)(p_drop * 65536.0f + 0.5f);
  return 0;
}
inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

}  // namespace

extern "C" int vl_attn2_fwd(const void* qkv_hi, const void* qkv_lo, const float* addmask, void* ctx_hi, void* ctx_lo,
                            float* lse, int64_t B, int64_t S, int64_t nh, int64_t dh, int64_t nq, float p_drop,
                            uint64_t seed, void* stream) {
  Attn2Args a{};
  if (int rc = fill_common("vl_attn2_fwd", a, B, S, nh, dh, nq, p_drop, seed)) return rc;
  VL_CHECK_ARG(qkv_hi && qkv_lo && addmask && ctx_hi && ctx_lo && lse, "vl_attn2_fwd: null pointer");
  VL_CHECK_ARG(al16(qkv_hi) && al16(qkv_lo) && al16(ctx_hi) && al16(ctx_lo), "vl_attn2_fwd: pointers must be 16-byte aligned");
  a.qkv_hi = (const bf16_raw*)qkv_hi; a.qkv_lo = (const bf16_raw*)qkv_lo; a.addmask = addmask;
  a.ctx_hi = (bf16_raw*)ctx_hi; a.ctx_lo = (bf16_raw*)ctx_lo; a.lse = lse;
  hipStream_t s = (hipStream_t)stream;
  return dh == 64 ? dispatch_fwd<64>(a, s) : dispatch_fwd<32>(a, s);
}

extern "C" int vl_attn2_bwd(const void* qkv_hi, const float* addmask, const void* dctx16, const float* lse,
                            void* dqkv16, int64_t B, int64_t S, int64_t nh, int64_t dh, int64_t nq, float p_drop,
                            uint64_t seed, void* stream) {
  Attn2Args a{};
  if (int rc = fill_common("vl_attn2_bwd", a, B, S, nh, dh, nq, p_drop, seed)) return rc;
  VL_CHECK_ARG(qkv_hi && addmask && dctx16 && lse && dqkv16, "vl_attn2_bwd: null pointer");
  VL_CHECK_ARG(al16(qkv_hi) && al16(dctx16) && al16(dqkv16), "vl_attn2_bwd: pointers must be 16-byte aligned");
  a.qkv_hi = (const bf16_raw*)qkv_hi; a.addmask = addmask; a.dctx = (const bf16_raw*)dctx16;
  a.lse = const_cast<float*>(lse); a.dqkv = (bf16_raw*)dqkv16;
  hipStream_t s = (hipStream_t)stream;
  return dh == 64 ? dispatch_bwd<64>(a, s) : dispatch_bwd<32>(a, s);
}
