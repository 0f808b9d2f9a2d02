// vl_attn_fwd / vl_attn_bwd: the fused V&L attention core over the single stream X = [text ; boxes].
//
// Reference: volta/volta/encoders.py:255-341 (BertGatedSelfAttention.forward).  With UC2's has_tt = has_tv =
// has_vt = has_vv and shared weights, the four gated score blocks, the two concatenated softmaxes
// (cat(tt,tv) for text rows, cat(vt,vv) for box rows -- key order [text, boxes] for both, :288-307), the four
// dropouts and the four P.V products are exactly ONE multi-head attention over S = T + V tokens with the
// additive key mask [t_mask ; v_mask] (encoders.py:978-995).  Queries at padded positions are still computed.
//
// One workgroup (4 waves) per (batch, head).  K and V tiles (S x 64 fp32) are staged in LDS once and shared by
// the 4 waves; scores, softmax and P.V run on the exact-fp32 matrix pipe (v_mfma_f32_16x16x4_f32), so this
// kernel adds no rounding beyond fp32 (the QKV projection feeding it is the 3-pass bf16 GEMM).  S <= 160 means
// the whole key range of a query tile lives in registers: a plain (not online) softmax.
//
// MFMA 16x16x4 f32 operand maps (cdna guide §3): A[i = lane&15][k = lane>>4], B[k = lane>>4][j = lane&15],
// C/D: col = lane&15, row = 4*(lane>>4) + reg.
#include "common.h"
#include "../../include/vlhip.h"

namespace {

constexpr int DH = 64;
constexpr int LDT = 66;  // padded row length (floats) of the [row][64] LDS tiles: rows land on distinct banks

struct AttnArgs {
  const float* qkv; const float* addmask; bf16_raw* ctx_hi; bf16_raw* ctx_lo; float* lse;
  const float* dctx; bf16_raw* dqkv;
  int B, S, nh, H;  // H = nh*64
  float scale, p_drop, inv_keep; uint64_t seed;
};

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// stage rows [0, Spad) x 64 floats of one head slice into an LDS tile (zeros beyond S)
__device__ __forceinline__ void stage_rows(float* dst, const float* src, long ld, int S, int Spad, int tid) {
  for (int idx = tid; idx < Spad * 16; idx += 256) {
    const int row = idx >> 4, c4 = idx & 15;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row < S) v = *reinterpret_cast<const float4*>(src + (long)row * ld + c4 * 4);
    float2* d = reinterpret_cast<float2*>(dst + row * LDT + c4 * 4);
    d[0] = make_float2(v.x, v.y);
    d[1] = make_float2(v.z, v.w);
  }
}

// Forward.  One trip = the 4 waves of the workgroup each take one 16-query tile.
//  * every global load of the trip (K, V, Q) is issued up front; Q goes straight to registers in MFMA A-operand
//    order with the head dimension permuted (lane k4 owns d = 16*k4 .. 16*k4+15: one contiguous 64-byte load per lane;
//    the K operand is read from LDS with the same permutation, so the dot products are unchanged);
//  * S <= 64 (a single trip): K and V time-share ONE LDS buffer (V waits in registers while Q.K^T runs), which
//    brings the workgroup to ~35 KB of LDS -> 4 workgroups per CU instead of 2;
//  * K/V rows are padded to 65 floats: the permuted operand reads (row = lane&15, col = 16*k4 + step) hit 32
//    distinct banks.
constexpr int LDK = 65;
// V rows are padded to 80 floats: the P.V operand read (row = 4*step + k4, col = 16*dt + lane&15) then spreads the two
// k4 values of a 32-lane half over banks 0-15 / 16-31 (with 65 they overlapped: 2-way conflicts on every read); 16-byte
// aligned rows also let V be staged with one ds_write_b128 per float4.
constexpr int LDV = 80;

__device__ __forceinline__ void store_row4(float* dst, const float4& v) {  // rows are only 4-byte aligned (LDK odd)
  dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
}

template <int NT>
__global__ __launch_bounds__(256) void attn_fwd_kernel(AttnArgs p) {
  extern __shared__ __attribute__((aligned(16))) float smf[];
  constexpr int Spad = NT * 16, LDP = Spad + 2;
  constexpr bool SHARE = NT <= 4;        // single trip: K and V share one buffer
  constexpr int NLD = (Spad * 16 + 255) / 256;  // float4 loads per thread to stage one [Spad][64] matrix
  float* sK = smf;
  float* sV = SHARE ? sK : sK + Spad * LDK;  // Spad * 65 floats is a multiple of 16 bytes: V rows stay 16-byte aligned
  float* smask = sV + Spad * LDV;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float* sP = smask + Spad + wave * (16 * LDP);

  const int b = blockIdx.x / p.nh, h = blockIdx.x - b * p.nh;
  const int S = p.S;
  const long ld = 3L * p.H;
  const float* base = p.qkv + (long)b * S * ld + h * DH;
  const int i16 = lane & 15, k4 = lane >> 4;

  // ---- issue all global loads ----
  float4 rk[NLD], rv[NLD];
#pragma unroll
  for (int t = 0; t < NLD; ++t) {
    const int idx = tid + t * 256, row = idx >> 4, c4 = idx & 15;
    rk[t] = rv[t] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (idx < Spad * 16 && row < S) {
      rk[t] = *reinterpret_cast<const float4*>(base + p.H + (long)row * ld + c4 * 4);
      rv[t] = *reinterpret_cast<const float4*>(base + 2 * p.H + (long)row * ld + c4 * 4);
    }
  }
  const int nqt = (S + 15) >> 4;
  // this wave's Q tile: lane (i16, k4) holds Q[q][16*k4 .. 16*k4+15]; issued before anything waits on K/V
  float qa[16];
  auto load_q = [&](int qt) {
    const int q = qt * 16 + i16;
    float4 v4[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) v4[t] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (qt < nqt && q < S) {
#pragma unroll
      for (int t = 0; t < 4; ++t) v4[t] = *reinterpret_cast<const float4*>(base + (long)q * ld + 16 * k4 + 4 * t);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) { qa[4 * t] = v4[t].x; qa[4 * t + 1] = v4[t].y; qa[4 * t + 2] = v4[t].z; qa[4 * t + 3] = v4[t].w; }
  };
  load_q(wave);
  for (int k = tid; k < Spad; k += 256) smask[k] = k < S ? p.addmask[(long)b * S + k] : -INFINITY;
#pragma unroll
  for (int t = 0; t < NLD; ++t) {
    const int idx = tid + t * 256, row = idx >> 4, c4 = idx & 15;
    if (idx < Spad * 16) {
      store_row4(sK + row * LDK + c4 * 4, rk[t]);
      if (!SHARE) *reinterpret_cast<float4*>(sV + row * LDV + c4 * 4) = rv[t];
    }
  }

  for (int it = 0; it * 4 < nqt; ++it) {
    const int qt = it * 4 + wave;
    const bool active = qt < nqt;
    __syncthreads();  // K (and the mask) are in LDS
    f32x4 sc[NT];
    if (active) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) sc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          sc[nt] = mfma4(qa[ks], sK[(nt * 16 + i16) * LDK + 16 * k4 + ks], sc[nt]);
      }
      // dropout keep-scales: one hash per (4 consecutive queries, key) = the 4 accumulator registers of a lane
      float kp[NT][4];
      if (p.p_drop > 0.f) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          vl_dropout_scale4(p.seed, (((uint64_t)b * p.nh + h) * ((S + 3) >> 2) + (qt * 4 + k4)) * S + nt * 16 + i16,
                            p.p_drop, p.inv_keep, kp[nt]);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ql = 4 * k4 + r, q = qt * 16 + ql;
        float m = -INFINITY;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          sc[nt][r] = sc[nt][r] * p.scale + smask[nt * 16 + i16];
          m = fmaxf(m, sc[nt][r]);
        }
        m = group16_max(m);
        float sum = 0.f;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          sc[nt][r] = __expf(sc[nt][r] - m);
          sum += sc[nt][r];
        }
        sum = group16_sum(sum);
        const float inv = 1.0f / sum;
        if (i16 == 0 && q < S) p.lse[((long)b * p.nh + h) * S + q] = m + __logf(sum);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const int key = nt * 16 + i16;
          float pv = sc[nt][r] * inv;
          if (p.p_drop > 0.f && q < S && key < S) pv *= kp[nt][r];
          sP[ql * LDP + key] = pv;
        }
      }
    }
    __syncthreads();  // every wave is done with K; P tiles are written
    if (SHARE) {      // V takes over the K buffer
#pragma unroll
      for (int t = 0; t < NLD; ++t) {
        const int idx = tid + t * 256, row = idx >> 4, c4 = idx & 15;
        if (idx < Spad * 16) *reinterpret_cast<float4*>(sV + row * LDV + c4 * 4) = rv[t];
      }
      __syncthreads();
    }
    if (active) {
      f32x4 o[4];
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
      for (int ks = 0; ks < Spad / 4; ++ks) {
        const float a = sP[i16 * LDP + 4 * ks + k4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[dt] = mfma4(a, sV[(4 * ks + k4) * LDV + dt * 16 + i16], o[dt]);
      }
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int q = qt * 16 + 4 * k4 + r;
          if (q < S) {
            bf16_raw hi, lo;
            split_bf16(o[dt][r], hi, lo);
            const long off = ((long)b * S + q) * p.H + h * DH + dt * 16 + i16;
            p.ctx_hi[off] = hi;
            p.ctx_lo[off] = lo;
          }
        }
    }
    if ((it + 1) * 4 < nqt) load_q((it + 1) * 4 + wave);
    __syncthreads();  // before the next trip overwrites the P tiles (only reached when !SHARE)
  }
}

// Backward.  dO = dctx (fp32).  Per 16-query tile (all 4 waves cooperate; wave w owns key tiles w, w+4, w+8):
//   S = Q K^T, P = exp(S*scale + mask - LSE), dPd = dO V^T, Pd = P*D, delta = rowsum(dO*O),
//   dS = P * (D*dPd - delta) * scale;   dV += Pd^T dO;  dK += dS^T Q;  dQ = dS K.
// dK / dV accumulate in registers across the query tiles (no cross-workgroup reduction, no atomics).
template <int NT>
__global__ __launch_bounds__(256) void attn_bwd_kernel(AttnArgs p) {
  extern __shared__ __attribute__((aligned(16))) float smf[];
  constexpr int Spad = NT * 16, LDP = Spad + 2, MAXKT = (NT + 3) / 4;
  float* sK = smf;
  float* sV = sK + Spad * LDT;
  float* sQ = sV + Spad * LDT;
  float* sdO = sQ + 16 * LDT;
  float* sPd = sdO + 16 * LDT;
  float* sdS = sPd + 16 * LDP;
  float* smask = sdS + 16 * LDP;
  float* slse = smask + Spad;
  float* sdelta = slse + 16;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.x / p.nh, h = blockIdx.x - b * p.nh;
  const int S = p.S;
  const long ld = 3L * p.H;
  const float* base = p.qkv + (long)b * S * ld + h * DH;
  stage_rows(sK, base + p.H, ld, S, Spad, tid);
  stage_rows(sV, base + 2 * p.H, ld, S, Spad, tid);
  for (int k = tid; k < Spad; k += 256) smask[k] = k < S ? p.addmask[(long)b * S + k] : -INFINITY;

  f32x4 dK[MAXKT][4], dV[MAXKT][4];
#pragma unroll
  for (int i = 0; i < MAXKT; ++i)
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) { dK[i][dt] = f32x4{0.f, 0.f, 0.f, 0.f}; dV[i][dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }

  const int nqt = (S + 15) >> 4;
  const int i16 = lane & 15, k4 = lane >> 4;
  // software pipeline over the query tiles: the global loads of tile qt+1 (Q, dO, O, LSE: one float4 / ushort4 per
  // thread) are issued before tile qt's MFMAs and consumed at the top of the next trip, so their HBM latency is off
  // the critical path instead of being paid once per tile
  const int prow = tid >> 4, pc4 = tid & 15;
  float4 nq = make_float4(0.f, 0.f, 0.f, 0.f), nd = nq;
  ushort4 noh = make_ushort4(0, 0, 0, 0), nol = noh;
  float nlse = INFINITY;
  auto prefetch = [&](int qt) {
    const int q = qt * 16 + prow;
    nq = nd = make_float4(0.f, 0.f, 0.f, 0.f);
    noh = nol = make_ushort4(0, 0, 0, 0);
    nlse = INFINITY;  // +inf -> P = 0 for padded queries
    if (qt < nqt && q < S) {
      nq = *reinterpret_cast<const float4*>(base + (long)q * ld + pc4 * 4);
      const long off = ((long)b * S + q) * p.H + h * DH + pc4 * 4;
      nd = *reinterpret_cast<const float4*>(p.dctx + off);
      noh = *reinterpret_cast<const ushort4*>(p.ctx_hi + off);
      nol = *reinterpret_cast<const ushort4*>(p.ctx_lo + off);
      if (pc4 == 0) nlse = p.lse[((long)b * p.nh + h) * S + q];
    }
  };
  prefetch(0);
  for (int qt = 0; qt < nqt; ++qt) {
    {  // the 16 x 64 Q and dO tiles (one float4 per thread) -> LDS, and delta = rowsum(dO * O)
      float part = nd.x * (bf16_to_f32(noh.x) + bf16_to_f32(nol.x)) + nd.y * (bf16_to_f32(noh.y) + bf16_to_f32(nol.y)) +
                   nd.z * (bf16_to_f32(noh.z) + bf16_to_f32(nol.z)) + nd.w * (bf16_to_f32(noh.w) + bf16_to_f32(nol.w));
      part = group16_sum(part);  // the 16 threads of a row are 16 consecutive lanes of one wave
      float2* dq_ = reinterpret_cast<float2*>(sQ + prow * LDT + pc4 * 4);
      dq_[0] = make_float2(nq.x, nq.y); dq_[1] = make_float2(nq.z, nq.w);
      float2* dd_ = reinterpret_cast<float2*>(sdO + prow * LDT + pc4 * 4);
      dd_[0] = make_float2(nd.x, nd.y); dd_[1] = make_float2(nd.z, nd.w);
      if (pc4 == 0) {
        sdelta[prow] = part;
        slse[prow] = nlse;
      }
    }
    prefetch(qt + 1);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < MAXKT; ++i) {
      const int kt = wave + 4 * i;
      if (kt < NT) {
        f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f}, dp = s;
#pragma unroll 4
        for (int ks = 0; ks < DH / 4; ++ks) {
          s = mfma4(sQ[i16 * LDT + 4 * ks + k4], sK[(kt * 16 + i16) * LDT + 4 * ks + k4], s);
          dp = mfma4(sdO[i16 * LDT + 4 * ks + k4], sV[(kt * 16 + i16) * LDT + 4 * ks + k4], dp);
        }
        const int key = kt * 16 + i16;
        float kp[4] = {1.0f, 1.0f, 1.0f, 1.0f};
        if (p.p_drop > 0.f)  // the forward pass's keep-scales (same hash grouping: 4 consecutive queries of one key)
          vl_dropout_scale4(p.seed, (((uint64_t)b * p.nh + h) * ((S + 3) >> 2) + (qt * 4 + k4)) * S + key, p.p_drop,
                            p.inv_keep, kp);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int ql = 4 * k4 + r, q = qt * 16 + ql;
          const float pr = __expf(s[r] * p.scale + smask[key] - slse[ql]);
          float dsc = 1.0f;
          if (p.p_drop > 0.f && q < S && key < S) dsc = kp[r];
          sPd[ql * LDP + key] = pr * dsc;
          sdS[ql * LDP + key] = pr * (dsc * dp[r] - sdelta[ql]) * p.scale;
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < MAXKT; ++i) {
      const int kt = wave + 4 * i;
      if (kt < NT) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {  // 16 queries = 4 k-steps
          // which 4 queries form a step is free as long as both operands agree: lane group k4 takes query
          // ks + {0, 8, 4, 12}[k4], so the two groups of a 32-lane half read rows 8 apart = 16 banks apart (rows are
          // 66 floats) and the row reads below are conflict-free (4*ks + k4 put them 2 banks apart: 2-way conflicts)
          const int kq = ks + 8 * (k4 & 1) + 4 * (k4 >> 1);
          const float apd = sPd[kq * LDP + kt * 16 + i16];
          const float ads = sdS[kq * LDP + kt * 16 + i16];
#pragma unroll
          for (int dt = 0; dt < 4; ++dt) {
            dV[i][dt] = mfma4(apd, sdO[kq * LDT + dt * 16 + i16], dV[i][dt]);
            dK[i][dt] = mfma4(ads, sQ[kq * LDT + dt * 16 + i16], dK[i][dt]);
          }
        }
      }
    }
    {  // dQ tile: wave w computes head-dim columns [16w, 16w+16)
      f32x4 dq = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
      for (int ks = 0; ks < Spad / 4; ++ks)
        dq = mfma4(sdS[i16 * LDP + 4 * ks + k4], sK[(4 * ks + k4) * LDT + wave * 16 + i16], dq);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int q = qt * 16 + 4 * k4 + r;
        if (q < S) p.dqkv[((long)b * S + q) * ld + h * DH + wave * 16 + i16] = f32_to_bf16(dq[r]);
      }
    }
    __syncthreads();  // tiles are overwritten by the next trip
  }
#pragma unroll
  for (int i = 0; i < MAXKT; ++i) {
    const int kt = wave + 4 * i;
    if (kt < NT) {
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int key = kt * 16 + 4 * k4 + r;
          if (key < S) {
            const long off = ((long)b * S + key) * ld + h * DH + dt * 16 + i16;
            p.dqkv[off + p.H] = f32_to_bf16(dK[i][dt][r]);
            p.dqkv[off + 2 * p.H] = f32_to_bf16(dV[i][dt][r]);
          }
        }
    }
  }
}

inline size_t fwd_lds_bytes(int NT) {
  const int Spad = NT * 16, LDP = Spad + 2;
  return sizeof(float) * (size_t)((NT <= 4 ? 0 : Spad * LDK) + Spad * LDV + Spad + 4 * 16 * LDP);
}
inline size_t bwd_lds_bytes(int NT) {
  const int Spad = NT * 16, LDP = Spad + 2;
  return sizeof(float) * (size_t)(2 * Spad * LDT + 2 * 16 * LDT + 2 * 16 * LDP + Spad + 32);
}

template <int NT>
int launch_fwd(const AttnArgs& a, hipStream_t s) {
  const size_t lds = fwd_lds_bytes(NT);
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_fwd_kernel<NT>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return vl_set_error(-3, "vl_attn_fwd: hipFuncSetAttribute: %s", hipGetErrorString(e));
  hipLaunchKernelGGL((attn_fwd_kernel<NT>), dim3(a.B * a.nh), dim3(256), lds, s, a);
  VL_CHECK_LAUNCH("vl_attn_fwd");
  return 0;
}
template <int NT>
int launch_bwd(const AttnArgs& a, hipStream_t s) {
  const size_t lds = bwd_lds_bytes(NT);
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_kernel<NT>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return vl_set_error(-3, "vl_attn_bwd: hipFuncSetAttribute: %s", hipGetErrorString(e));
  hipLaunchKernelGGL((attn_bwd_kernel<NT>), dim3(a.B * a.nh), dim3(256), lds, s, a);
  VL_CHECK_LAUNCH("vl_attn_bwd");
  return 0;
}

int check_common(const char* fn, int64_t B, int64_t S, int64_t nh, int64_t dh, float p_drop) {
  VL_CHECK_ARG(dh == DH, "%s: head dim must be 64 (got %lld)", fn, (long long)dh);
  VL_CHECK_ARG(S >= 1 && S <= 160, "%s: sequence length T+V must be in [1,160] (got %lld)", fn, (long long)S);
  VL_CHECK_ARG(B >= 1 && nh >= 1 && B * nh < (1LL << 30), "%s: bad B=%lld nh=%lld", fn, (long long)B, (long long)nh);
  VL_CHECK_ARG(p_drop >= 0.f && p_drop < 1.f, "%s: dropout p must be in [0,1)", fn);
  return 0;
}

}  // namespace

extern "C" int vl_attn_fwd(const float* qkv32, const float* addmask, void* ctx_hi, void* ctx_lo, float* lse,
                           int64_t B, int64_t S, int64_t nh, int64_t dh, float p_drop, uint64_t seed, void* stream) {
  if (int rc = check_common("vl_attn_fwd", B, S, nh, dh, p_drop)) return rc;
  VL_CHECK_ARG(qkv32 && addmask && ctx_hi && ctx_lo && lse, "vl_attn_fwd: null pointer");
  AttnArgs a{};
  a.qkv = qkv32; a.addmask = addmask; a.ctx_hi = (bf16_raw*)ctx_hi; a.ctx_lo = (bf16_raw*)ctx_lo; a.lse = lse;
  a.B = (int)B; a.S = (int)S; a.nh = (int)nh; a.H = (int)(nh * DH);
  a.scale = 1.0f / sqrtf((float)dh); a.p_drop = p_drop; a.inv_keep = 1.0f / (1.0f - p_drop); a.seed = seed;
  hipStream_t s = (hipStream_t)stream;
  if (S <= 64) return launch_fwd<4>(a, s);
  if (S <= 80) return launch_fwd<5>(a, s);
  if (S <= 128) return launch_fwd<8>(a, s);
  return launch_fwd<10>(a, s);
}

extern "C" int vl_attn_bwd(const float* qkv32, const float* addmask, const void* ctx_hi, const void* ctx_lo,
                           const float* dctx32, const float* lse, void* dqkv16, int64_t B, int64_t S, int64_t nh,
                           int64_t dh, float p_drop, uint64_t seed, void* stream) {
  if (int rc = check_common("vl_attn_bwd", B, S, nh, dh, p_drop)) return rc;
  VL_CHECK_ARG(qkv32 && addmask && ctx_hi && ctx_lo && dctx32 && lse && dqkv16, "vl_attn_bwd: null pointer");
  AttnArgs a{};
  a.qkv = qkv32; a.addmask = addmask; a.ctx_hi = (bf16_raw*)ctx_hi; a.ctx_lo = (bf16_raw*)ctx_lo;
  a.lse = const_cast<float*>(lse); a.dctx = dctx32; a.dqkv = (bf16_raw*)dqkv16;
  a.B = (int)B; a.S = (int)S; a.nh = (int)nh; a.H = (int)(nh * DH);
  a.scale = 1.0f / sqrtf((float)dh); a.p_drop = p_drop; a.inv_keep = 1.0f / (1.0f - p_drop); a.seed = seed;
  hipStream_t s = (hipStream_t)stream;
  if (S <= 64) return launch_bwd<4>(a, s);
  if (S <= 80) return launch_bwd<5>(a, s);
  if (S <= 128) return launch_bwd<8>(a, s);
  return launch_bwd<10>(a, s);
}
