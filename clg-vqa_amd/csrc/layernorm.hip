// vl_ln_fwd / vl_ln_bwd: (dropout +) residual + LayerNorm with wavefront reductions.
//
// Reference: BertLayerNorm = apex FusedLayerNorm when built (volta/volta/encoders.py:44-47; CUDA kernels
// volta/apex/csrc/layer_norm_cuda_kernel.cu:279-322 cuApplyLayerNorm, :403-520 cuComputePartGradGammaBeta /
// cuComputeGradGammaBeta, :522-637 cuComputeGradInput) else the TF-style Python module (encoders.py:49-62):
// biased variance, epsilon inside the sqrt.  Fused here with the eager ops around it in
// BertGatedSelfOutput / BertGatedOutput (encoders.py:411-425, :553-567: dropout(dense) + input -> LN) and in
// UC2Embeddings (embeddings.py:653-666: LN -> dropout; sum of three terms -> LN).
//
// HBM-bound: one wave64 per row, the row (H = NV*256 floats) lives in registers as NV float4 per lane, mean
// and variance are two wave reductions (two-pass, like the reference), all traffic is 16 B per lane.
// Backward keeps per-lane column partials (dgamma, dbeta, dbias) in registers across the rows a wave walks,
// then one LDS combine per workgroup and a second tiny kernel sums the per-workgroup partials (deterministic;
// no atomics).
#include "common.h"
#include "../../include/vlhip.h"

namespace {

constexpr uint64_t POST_SALT = 0x5bd1e9955bd1e995ull;

struct LnArgs {
  float* y; const float* resid; const float* addvec; long addvec_rows; const float* row_pre; const float* row_post;
  const float* gamma; const float* beta; float eps;
  float* out32; bf16_raw* out_hi; bf16_raw* out_lo; float* mean; float* rstd;
  long M; int H; long group, out_stride, out_off;
  float p_pre, inv_pre, p_post, inv_post; uint64_t seed;
  // compact-row calls (only a subset of the rows of the full [M_full, H] problem is live, e.g. the pooled row of every
  // sample in the last layer): row r here is row r * orig_stride of the full problem -- the dropout counter and the
  // row masks are indexed by that ORIGINAL row (bit-identical to the dense run), resid is read at row r * resid_stride
  long orig_stride, resid_stride;
  // residual recomputed instead of read (vl_ln_fwd_rr): the residual IS the output of an earlier LayerNorm whose saved z /
  // mean / rstd are in memory anyway (backward needs them) -- resid[row][c] = rgamma[c] * ((rz[row][c] - rmean[row]) *
  // rrstd[row]) + rbeta[c] (* rrow_post[row]), the very expression that LayerNorm stored; it then need not store its fp32
  // output at all (44 MB per launch at c2)
  const float* rz; const float* rmean; const float* rrstd; const float* rgamma; const float* rbeta; const float* rrow_post;
  // backward
  const float* dy; const float* z; float* dz; bf16_raw* dpre16; float* dpre32; float* ws; int nblk;
};

__device__ __forceinline__ long map_row(const LnArgs& p, long r) {
  return (r / p.group) * p.out_stride + p.out_off + (r % p.group);
}

template <int NV>
__global__ __launch_bounds__(256) void ln_fwd_kernel(LnArgs p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float invH = 1.0f / (float)p.H;
  for (long r = (long)blockIdx.x * 4 + wave; r < p.M; r += (long)gridDim.x * 4) {
    float4 z[NV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = (i * 64 + lane) * 4;
      if (c >= p.H) { z[i] = make_float4(0.f, 0.f, 0.f, 0.f); continue; }  // H = 128: half of the wave is idle
      const long e = r * p.H + c;
      const long eo_ = r * p.orig_stride * p.H + c;  // element index in the full problem (dropout counter)
      float4 v = *reinterpret_cast<const float4*>(p.y + e);
      if (p.p_pre > 0.f) {
        float ks_[4];
        vl_dropout_scale4(p.seed, (uint64_t)eo_ >> 2, p.p_pre, p.inv_pre, ks_);
        v.x *= ks_[0]; v.y *= ks_[1]; v.z *= ks_[2]; v.w *= ks_[3];
      }
      if (p.resid) {
        const float4 q = *reinterpret_cast<const float4*>(p.resid + r * p.resid_stride * p.H + c);
        v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w;
      } else if (p.rz) {
        const long rr = r * p.resid_stride;
        const float4 zz = *reinterpret_cast<const float4*>(p.rz + rr * p.H + c);
        const float4 g = *reinterpret_cast<const float4*>(p.rgamma + c);
        const float4 bt = *reinterpret_cast<const float4*>(p.rbeta + c);
        const float rmu = p.rmean[rr], rrs = p.rrstd[rr];
        float4 q;
        q.x = g.x * ((zz.x - rmu) * rrs) + bt.x;
        q.y = g.y * ((zz.y - rmu) * rrs) + bt.y;
        q.z = g.z * ((zz.z - rmu) * rrs) + bt.z;
        q.w = g.w * ((zz.w - rmu) * rrs) + bt.w;
        if (p.rrow_post) {
          const float rq = p.rrow_post[rr];
          q.x *= rq; q.y *= rq; q.z *= rq; q.w *= rq;
        }
        v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w;
      }
      if (p.addvec) {
        const float4 q = *reinterpret_cast<const float4*>(p.addvec + (r % p.addvec_rows) * p.H + c);
        v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w;
      }
      if (p.row_pre) {
        const float rp = p.row_pre[r * p.orig_stride];
        v.x *= rp; v.y *= rp; v.z *= rp; v.w *= rp;
      }
      if (p.p_pre > 0.f || p.resid || p.rz || p.addvec || p.row_pre) *reinterpret_cast<float4*>(p.y + e) = v;  // z saved in place
      z[i] = v;
      s += (v.x + v.y) + (v.z + v.w);
    }
    const float mu = wave_sum(s) * invH;
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      if ((i * 64 + lane) * 4 >= p.H) continue;
      const float a = z[i].x - mu, b = z[i].y - mu, c = z[i].z - mu, d = z[i].w - mu;
      ss += (a * a + b * b) + (c * c + d * d);
    }
    const float rs = 1.0f / sqrtf(wave_sum(ss) * invH + p.eps);
    if (lane == 0) { p.mean[r] = mu; p.rstd[r] = rs; }
    const long orow = map_row(p, r);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = (i * 64 + lane) * 4;
      if (c >= p.H) continue;
      const float4 g = *reinterpret_cast<const float4*>(p.gamma + c);
      const float4 bt = *reinterpret_cast<const float4*>(p.beta + c);
      float4 o;
      o.x = g.x * ((z[i].x - mu) * rs) + bt.x;
      o.y = g.y * ((z[i].y - mu) * rs) + bt.y;
      o.z = g.z * ((z[i].z - mu) * rs) + bt.z;
      o.w = g.w * ((z[i].w - mu) * rs) + bt.w;
      if (p.p_post > 0.f) {
        const long e = r * p.orig_stride * p.H + c;
        float ks_[4];
        vl_dropout_scale4(p.seed ^ POST_SALT, (uint64_t)e >> 2, p.p_post, p.inv_post, ks_);
        o.x *= ks_[0]; o.y *= ks_[1]; o.z *= ks_[2]; o.w *= ks_[3];
      }
      if (p.row_post) {
        const float rq = p.row_post[r * p.orig_stride];
        o.x *= rq; o.y *= rq; o.z *= rq; o.w *= rq;
      }
      const long eo = orow * p.H + c;
      if (p.out32) *reinterpret_cast<float4*>(p.out32 + eo) = o;
      if (p.out_hi) {
        ushort4 hi, lo;
        split_bf16(o.x, hi.x, lo.x); split_bf16(o.y, hi.y, lo.y);
        split_bf16(o.z, hi.z, lo.z); split_bf16(o.w, hi.w, lo.w);
        *reinterpret_cast<ushort4*>(p.out_hi + eo) = hi;
        if (p.out_lo) *reinterpret_cast<ushort4*>(p.out_lo + eo) = lo;
      }
    }
  }
}

// Backward.  A wave walks TWO rows per trip (rows r and r + rows_per_trip/2): the loads of both are issued before either
// is reduced, i.e. 12 KB of loads in flight per wave instead of 6 -- the kernel is a latency-bound stream (one dependent
// load -> two wave reductions -> store chain per row) and was running at 2.8 TB/s with one row per trip.
#ifndef VL_LN_BWD_MINWG
#define VL_LN_BWD_MINWG 1  // (A/B builds: workgroups per CU the register allocation must allow)
#endif
#ifndef VL_LN_BWD_RPT
#define VL_LN_BWD_RPT 4    // rows per trip and wave (loads of all of them are issued before the first is reduced)
#endif
template <int NV>
__global__ __launch_bounds__(256, VL_LN_BWD_MINWG) void ln_bwd_kernel(LnArgs p) {
  __shared__ float red[3][4][NV * 256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float invH = 1.0f / (float)p.H;
  float4 ag[NV], ab[NV], ap[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) ag[i] = ab[i] = ap[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  constexpr int RPT = VL_LN_BWD_RPT;
  float4 gam[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (i * 64 + lane) * 4;
    gam[i] = c < p.H ? *reinterpret_cast<const float4*>(p.gamma + c) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const long half = (long)gridDim.x * 4;  // rows one pass of the grid covers
  for (long r0 = (long)blockIdx.x * 4 + wave; r0 < p.M; r0 += RPT * half) {
    float4 dy[RPT][NV], zz[RPT][NV];
    long rr[RPT];
    bool ok[RPT];
    long orow[RPT];
    float mus[RPT], rss[RPT], rqs[RPT], rps[RPT];
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
      rr[k] = r0 + k * half;
      ok[k] = rr[k] < p.M;
      if (!ok[k]) rr[k] = r0;  // (loads a valid row; its results are discarded)
      // original (un-compacted) row index: the dropout RNG and the row masks are indexed by it
      orow[k] = map_row(p, rr[k]);
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c >= p.H) { dy[k][i] = zz[k][i] = make_float4(0.f, 0.f, 0.f, 0.f); continue; }
        dy[k][i] = *reinterpret_cast<const float4*>(p.dy + orow[k] * p.H + c);
        zz[k][i] = *reinterpret_cast<const float4*>(p.z + rr[k] * p.H + c);
      }
      // (the row statistics ride with the row loads: fetched inside the per-row section below, each would expose a full
      // memory latency per row)
      mus[k] = p.mean[rr[k]];
      rss[k] = p.rstd[rr[k]];
      rqs[k] = p.row_post ? p.row_post[rr[k] * p.orig_stride] : 1.f;
      rps[k] = p.row_pre ? p.row_pre[rr[k] * p.orig_stride] : 1.f;
    }
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
      if (!ok[k]) continue;
      const long r = rr[k];
      const float mu = mus[k], rs = rss[k];
      float4 xh[NV];
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c >= p.H) { xh[i] = make_float4(0.f, 0.f, 0.f, 0.f); continue; }
        const long e = r * p.orig_stride * p.H + c;
        float4 d = dy[k][i];
        if (p.p_post > 0.f) {
          float ks_[4];
          vl_dropout_scale4(p.seed ^ POST_SALT, (uint64_t)e >> 2, p.p_post, p.inv_post, ks_);
          d.x *= ks_[0]; d.y *= ks_[1]; d.z *= ks_[2]; d.w *= ks_[3];
        }
        if (p.row_post) {
          const float rq = rqs[k];
          d.x *= rq; d.y *= rq; d.z *= rq; d.w *= rq;
        }
        const float4 z = zz[k][i];
        const float4 g = gam[i];
        float4 x;
        x.x = (z.x - mu) * rs; x.y = (z.y - mu) * rs; x.z = (z.z - mu) * rs; x.w = (z.w - mu) * rs;
        ag[i].x += d.x * x.x; ag[i].y += d.y * x.y; ag[i].z += d.z * x.z; ag[i].w += d.w * x.w;
        ab[i].x += d.x; ab[i].y += d.y; ab[i].z += d.z; ab[i].w += d.w;
        d.x *= g.x; d.y *= g.y; d.z *= g.z; d.w *= g.w;  // now dL/dxhat
        s1 += (d.x + d.y) + (d.z + d.w);
        s2 += (d.x * x.x + d.y * x.y) + (d.z * x.z + d.w * x.w);
        dy[k][i] = d; xh[i] = x;
      }
      const float c1 = wave_sum(s1) * invH, c2 = wave_sum(s2) * invH;
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c >= p.H) continue;
        const long e = r * p.H + c;
        float4 dz;
        dz.x = (dy[k][i].x - c1 - xh[i].x * c2) * rs;
        dz.y = (dy[k][i].y - c1 - xh[i].y * c2) * rs;
        dz.z = (dy[k][i].z - c1 - xh[i].z * c2) * rs;
        dz.w = (dy[k][i].w - c1 - xh[i].w * c2) * rs;
        if (p.row_pre) {
          const float rp = rps[k];
          dz.x *= rp; dz.y *= rp; dz.z *= rp; dz.w *= rp;
        }
        if (p.dz) *reinterpret_cast<float4*>(p.dz + e) = dz;
        float4 dp = dz;
        if (p.p_pre > 0.f) {
          float ks_[4];
          vl_dropout_scale4(p.seed, (uint64_t)(r * p.orig_stride * p.H + c) >> 2, p.p_pre, p.inv_pre, ks_);
          dp.x *= ks_[0]; dp.y *= ks_[1]; dp.z *= ks_[2]; dp.w *= ks_[3];
        }
        if (p.dpre32) *reinterpret_cast<float4*>(p.dpre32 + e) = dp;
        if (p.dpre16) {
          ushort4 h;
          h.x = f32_to_bf16(dp.x); h.y = f32_to_bf16(dp.y); h.z = f32_to_bf16(dp.z); h.w = f32_to_bf16(dp.w);
          *reinterpret_cast<ushort4*>(p.dpre16 + e) = h;
        }
        ap[i].x += dp.x; ap[i].y += dp.y; ap[i].z += dp.z; ap[i].w += dp.w;
      }
    }
  }
  // combine the 4 waves' column partials, then one partial row-set per workgroup
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (i * 64 + lane) * 4;
    *reinterpret_cast<float4*>(&red[0][wave][c]) = ag[i];
    *reinterpret_cast<float4*>(&red[1][wave][c]) = ab[i];
    *reinterpret_cast<float4*>(&red[2][wave][c]) = ap[i];
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < 3 * p.H; idx += 256) {
    const int k = idx / p.H, c = idx - k * p.H;
    p.ws[((long)blockIdx.x * 3 + k) * p.H + c] = (red[k][0][c] + red[k][1][c]) + (red[k][2][c] + red[k][3][c]);
  }
}

// out_k[c] = sum_blk ws[blk][k][c]; 256 threads = 16 columns x 16 row-groups, LDS combine.  blockIdx.y selects one
// of up to two independent reductions (the two LayerNorms of a transformer layer share one launch).
struct LnReduceSet { const float* ws; int nblk; float* dgamma; float* dbeta; float* dbias; };
struct LnReduceArgs { LnReduceSet set[2]; int H; int accumulate; };
__global__ __launch_bounds__(256) void ln_bwd_reduce_kernel(LnReduceArgs a) {
  __shared__ float red[16][17];
  const LnReduceSet& r = a.set[blockIdx.y];
  const int H = a.H;
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int idx = blockIdx.x * 16 + tx;  // column in [0, 3H)
  float s = 0.f;
  if (idx < 3 * H)
    for (int b = ty; b < r.nblk; b += 16) s += r.ws[(long)b * 3 * H + idx];
  red[ty][tx] = s;
  __syncthreads();
  if (ty == 0 && idx < 3 * H) {
    float t = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) t += red[j][tx];
    const int k = idx / H, c = idx - k * H;
    float* dst = k == 0 ? r.dgamma : (k == 1 ? r.dbeta : r.dbias);
    if (dst) dst[c] = a.accumulate ? dst[c] + t : t;
  }
}

#ifndef VL_LN_BWD_BLOCKS
#define VL_LN_BWD_BLOCKS 512  // round 2 (2 rows per trip, row statistics fetched per row), ms / step beside the dW GEMM, same box:
                              // 192: 16.76, 256: 16.62, 320: 16.64, 384: 16.55, 512: 16.78, 768: 16.63, 1024 / 2048: worse (partials);
                              // round 3 (4 rows per trip, statistics / gamma hoisted; profiles/r03_ab_log.txt): kernel alone warm /
                              // cold 384: 39.6 / 53.2 us, 512: 36.8 / 51.6; step 384: 16.73 / 16.74, 512: 16.68
#endif
constexpr int LN_BWD_BLOCKS = VL_LN_BWD_BLOCKS;
int nblk_for(int64_t M) {
  // one partial row-set (3 x H floats) per workgroup: enough workgroups to keep >= 4 waves per SIMD in flight (the
  // kernel is a latency-bound stream, one row per wave at a time), few enough that the partials stay << the data
  int64_t n = (M + 3) / 4;
  return (int)(n < LN_BWD_BLOCKS ? n : LN_BWD_BLOCKS);
}

template <int NV>
int launch_fwd(const LnArgs& a, hipStream_t s) {
  int64_t n = (a.M + 3) / 4;
  if (n > 4096) n = 4096;
  hipLaunchKernelGGL((ln_fwd_kernel<NV>), dim3((unsigned)n), dim3(256), 0, s, a);
  VL_CHECK_LAUNCH("vl_ln_fwd");
  return 0;
}
template <int NV>
int launch_bwd(const LnArgs& a, hipStream_t s, float* dgamma, float* dbeta, float* dbias) {
  hipLaunchKernelGGL((ln_bwd_kernel<NV>), dim3(a.nblk), dim3(256), 0, s, a);
  VL_CHECK_LAUNCH("vl_ln_bwd");
  if (!dgamma && !dbeta && !dbias) return 0;  // partials only: the caller sums them with vl_ln_bwd_reduce
  LnReduceArgs ra{};
  ra.set[0] = LnReduceSet{a.ws, a.nblk, dgamma, dbeta, dbias};
  ra.H = a.H;
  hipLaunchKernelGGL(ln_bwd_reduce_kernel, dim3((3 * a.H + 15) / 16, 1), dim3(256), 0, s, ra);
  VL_CHECK_LAUNCH("vl_ln_bwd(reduce)");
  return 0;
}

int check_shape(const char* fn, int64_t M, int64_t H, int64_t group, float p_pre, float p_post) {
  VL_CHECK_ARG(M > 0 && H > 0 && (H == 128 || (H % 256 == 0 && H <= 2048 && (H / 256 <= 4 || H == 1536 || H == 2048))),
               "%s: H must be 128,256,512,768,1024,1536 or 2048 (got %lld)", fn, (long long)H);
  VL_CHECK_ARG(group > 0, "%s: group must be > 0", fn);
  VL_CHECK_ARG(p_pre >= 0.f && p_pre < 1.f && p_post >= 0.f && p_post < 1.f, "%s: dropout p must be in [0,1)", fn);
  return 0;
}

}  // namespace

extern "C" int vl_ln_fwd(float* y32_z32, const float* resid32, const float* addvec, int64_t addvec_rows,
                         const float* row_pre, const float* row_post, const float* gamma, const float* beta, float eps, float* out32, void* out_hi, void* out_lo, float* mean,
                         float* rstd, int64_t M, int64_t H, int64_t group, int64_t out_stride, int64_t out_off,
                         float p_pre, float p_post, uint64_t seed, int64_t orig_row_stride, int64_t resid_row_stride,
                         void* stream) {
  return vl_ln_fwd_rr(y32_z32, resid32, nullptr, addvec, addvec_rows, row_pre, row_post, gamma, beta, eps, out32, out_hi, out_lo, mean,
                      rstd, M, H, group, out_stride, out_off, p_pre, p_post, seed, orig_row_stride, resid_row_stride, stream);
}

extern "C" int vl_ln_fwd_rr(float* y32_z32, const float* resid32, const int64_t* resid_ln, const float* addvec, int64_t addvec_rows,
                            const float* row_pre, const float* row_post, const float* gamma, const float* beta, float eps,
                            float* out32, void* out_hi, void* out_lo, float* mean, float* rstd, int64_t M, int64_t H, int64_t group,
                            int64_t out_stride, int64_t out_off, float p_pre, float p_post, uint64_t seed,
                            int64_t orig_row_stride, int64_t resid_row_stride, void* stream) {
  if (int rc = check_shape("vl_ln_fwd", M, H, group, p_pre, p_post)) return rc;
  VL_CHECK_ARG(y32_z32 && gamma && beta && mean && rstd && (out32 || out_hi), "vl_ln_fwd: null pointer");
  LnArgs a{};
  VL_CHECK_ARG(!addvec || addvec_rows >= 1, "vl_ln_fwd: addvec_rows must be >= 1");
  a.y = y32_z32; a.resid = resid32; a.addvec = addvec; a.addvec_rows = addvec_rows; a.row_pre = row_pre;
  a.row_post = row_post; a.gamma = gamma; a.beta = beta; a.eps = eps;
  if (resid_ln) {  // {z32, mean, rstd, gamma, beta, row_post (may be 0)} of the LayerNorm whose output is the residual
    VL_CHECK_ARG(!resid32, "vl_ln_fwd_rr: pass the residual OR the LayerNorm to recompute it from");
    VL_CHECK_ARG(resid_ln[0] && resid_ln[1] && resid_ln[2] && resid_ln[3] && resid_ln[4], "vl_ln_fwd_rr: null pointer in resid_ln");
    a.rz = reinterpret_cast<const float*>(resid_ln[0]); a.rmean = reinterpret_cast<const float*>(resid_ln[1]);
    a.rrstd = reinterpret_cast<const float*>(resid_ln[2]); a.rgamma = reinterpret_cast<const float*>(resid_ln[3]);
    a.rbeta = reinterpret_cast<const float*>(resid_ln[4]); a.rrow_post = reinterpret_cast<const float*>(resid_ln[5]);
    VL_CHECK_ARG(a.rz != y32_z32, "vl_ln_fwd_rr: the residual's z buffer must not be this call's y / z buffer");
  }
  a.out32 = out32; a.out_hi = (bf16_raw*)out_hi; a.out_lo = (bf16_raw*)out_lo; a.mean = mean; a.rstd = rstd;
  a.M = M; a.H = (int)H; a.group = group; a.out_stride = out_stride; a.out_off = out_off;
  a.p_pre = p_pre; a.inv_pre = 1.f / (1.f - p_pre); a.p_post = p_post; a.inv_post = 1.f / (1.f - p_post);
  a.seed = seed;
  VL_CHECK_ARG(orig_row_stride >= 1 && resid_row_stride >= 1, "vl_ln_fwd: row strides must be >= 1");
  a.orig_stride = orig_row_stride; a.resid_stride = resid_row_stride;
  hipStream_t s = (hipStream_t)stream;
  switch ((H + 255) / 256) {
    case 1: return launch_fwd<1>(a, s);
    case 2: return launch_fwd<2>(a, s);
    case 3: return launch_fwd<3>(a, s);
    case 4: return launch_fwd<4>(a, s);
    case 6: return launch_fwd<6>(a, s);
    default: return launch_fwd<8>(a, s);
  }
}

extern "C" int64_t vl_ln_bwd_ws_floats(int64_t M, int64_t H) { return (int64_t)nblk_for(M) * 3 * H; }

extern "C" int vl_ln_bwd(const float* dy32, const float* z32, const float* mean, const float* rstd,
                         const float* gamma, const float* row_pre, const float* row_post, float* dz32, void* dpre16, float* dpre32, float* dgamma, float* dbeta,
                         float* dbias, float* partial_ws, int64_t M, int64_t H, int64_t group, int64_t out_stride,
                         int64_t out_off, float p_pre, float p_post, uint64_t seed, int64_t orig_row_stride, void* stream) {
  if (int rc = check_shape("vl_ln_bwd", M, H, group, p_pre, p_post)) return rc;
  VL_CHECK_ARG(orig_row_stride >= 1, "vl_ln_bwd: orig_row_stride must be >= 1");
  VL_CHECK_ARG(dy32 && z32 && mean && rstd && gamma && partial_ws, "vl_ln_bwd: null pointer");
  LnArgs a{};
  a.dy = dy32; a.z = z32; a.mean = const_cast<float*>(mean); a.rstd = const_cast<float*>(rstd); a.gamma = gamma;
  a.row_pre = row_pre; a.row_post = row_post;
  a.dz = dz32; a.dpre16 = (bf16_raw*)dpre16; a.dpre32 = dpre32; a.ws = partial_ws; a.nblk = nblk_for(M);
  a.M = M; a.H = (int)H; a.group = group; a.out_stride = out_stride; a.out_off = out_off;
  a.p_pre = p_pre; a.inv_pre = 1.f / (1.f - p_pre); a.p_post = p_post; a.inv_post = 1.f / (1.f - p_post);
  a.seed = seed;
  a.orig_stride = orig_row_stride; a.resid_stride = 1;
  hipStream_t s = (hipStream_t)stream;
  switch ((H + 255) / 256) {
    case 1: return launch_bwd<1>(a, s, dgamma, dbeta, dbias);
    case 2: return launch_bwd<2>(a, s, dgamma, dbeta, dbias);
    case 3: return launch_bwd<3>(a, s, dgamma, dbeta, dbias);
    case 4: return launch_bwd<4>(a, s, dgamma, dbeta, dbias);
    case 6: return launch_bwd<6>(a, s, dgamma, dbeta, dbias);
    default: return launch_bwd<8>(a, s, dgamma, dbeta, dbias);
  }
}

// Second half of vl_ln_bwd on its own (any stream that is ordered after the vl_ln_bwd call which filled partial_ws with
// dgamma = dbeta = dbias = NULL): the column sums are only needed by the optimizer, so the training engine runs this
// 15-us launch on the weight-gradient stream instead of between the kernels of the backward critical path.
extern "C" int vl_ln_bwd_reduce(const float* partial_ws, int64_t M, int64_t H, float* dgamma, float* dbeta,
                                float* dbias, void* stream) {
  if (int rc = check_shape("vl_ln_bwd_reduce", M, H, 1, 0.f, 0.f)) return rc;
  VL_CHECK_ARG(partial_ws, "vl_ln_bwd_reduce: null workspace");
  LnReduceArgs ra{};
  ra.set[0] = LnReduceSet{partial_ws, nblk_for(M), dgamma, dbeta, dbias};
  ra.H = (int)H;
  hipLaunchKernelGGL(ln_bwd_reduce_kernel, dim3((unsigned)((3 * H + 15) / 16), 1), dim3(256), 0, (hipStream_t)stream, ra);
  VL_CHECK_LAUNCH("vl_ln_bwd_reduce");
  return 0;
}

// Two such reductions in one launch (the two LayerNorms of a transformer layer: 12 launches of ~20 us less per step).
extern "C" int vl_ln_bwd_reduce2(const float* ws_a, int64_t M_a, float* dgamma_a, float* dbeta_a, float* dbias_a,
                                 const float* ws_b, int64_t M_b, float* dgamma_b, float* dbeta_b, float* dbias_b,
                                 int64_t H, int accumulate, void* stream) {
  if (int rc = check_shape("vl_ln_bwd_reduce2", M_a, H, 1, 0.f, 0.f)) return rc;
  VL_CHECK_ARG(ws_a && ws_b && M_b > 0, "vl_ln_bwd_reduce2: bad arguments");
  LnReduceArgs ra{};
  ra.set[0] = LnReduceSet{ws_a, nblk_for(M_a), dgamma_a, dbeta_a, dbias_a};
  ra.set[1] = LnReduceSet{ws_b, nblk_for(M_b), dgamma_b, dbeta_b, dbias_b};
  ra.H = (int)H;
  ra.accumulate = accumulate;
  hipLaunchKernelGGL(ln_bwd_reduce_kernel, dim3((unsigned)((3 * H + 15) / 16), 2), dim3(256), 0, (hipStream_t)stream, ra);
  VL_CHECK_LAUNCH("vl_ln_bwd_reduce2");
  return 0;
}
