// Shared device helpers for the vlhip kernels (gfx950 / CDNA4 only: wave64, MFMA, 160 KiB LDS).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef unsigned short bf16_raw;

#define VL_WAVE 64

// ---- error plumbing (host side; implemented in api.hip) -------------------------------------
int vl_set_error(int code, const char* fmt, ...);
#define VL_CHECK_ARG(cond, ...) do { if (!(cond)) return vl_set_error(-1, __VA_ARGS__); } while (0)
#define VL_CHECK_LAUNCH(name) do { hipError_t e_ = hipGetLastError(); \
    if (e_ != hipSuccess) return vl_set_error(-3, "%s: launch failed: %s", name, hipGetErrorString(e_)); } while (0)

// ---- bf16 helpers -----------------------------------------------------------------------------
__device__ __forceinline__ float bf16_to_f32(bf16_raw h) { return __uint_as_float(((unsigned)h) << 16); }
__device__ __forceinline__ bf16_raw f32_to_bf16(float f) {  // round-to-nearest-even, NaN preserved
  __bf16 b = (__bf16)f;
  return __builtin_bit_cast(bf16_raw, b);
}
// x ~= hi + lo with hi = bf16(x), lo = bf16(x - hi): 16 significant bits in total.
__device__ __forceinline__ void split_bf16(float x, bf16_raw& hi, bf16_raw& lo) {
  hi = f32_to_bf16(x);
  lo = f32_to_bf16(x - bf16_to_f32(hi));
}

// ---- erf-GELU (reference volta/encoders.py:131-137: x * 0.5 * (1 + erf(x / sqrt 2))) -----------------------------
// erf by Abramowitz-Stegun 7.1.26 (|abs err| <= 1.5e-7, i.e. fp32-grade for the activation): one v_exp + one v_rcp
// instead of libm's ~40-instruction erff.  exp(-x^2/2) is shared between the cdf and the pdf in the gradient.
__device__ __forceinline__ void vl_cdf_pdf(float x, float& cdf, float& e) {
  const float ax = fabsf(x) * 0.70710678118654752f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));  // v_rcp_f32 (1 ulp), not the IEEE division sequence
  e = __expf(-0.5f * x * x);  // = exp(-(x/sqrt2)^2)
  float poly = fmaf(1.061405429f, t, -1.453152027f);
  poly = fmaf(poly, t, 1.421413741f);
  poly = fmaf(poly, t, -0.284496736f);
  poly = fmaf(poly, t, 0.254829592f);
  // (__fmul_rn: the product must not be contracted into the subtraction below -- which instantiation the optimizer
  // contracts is not stable, and the epilogues of all tile configurations are tested bit-identical)
  const float h = __fmul_rn(0.5f * poly * t, e);  // 0.5 * (1 - erf(|x|/sqrt2))
  cdf = x >= 0.f ? 1.0f - h : h;
}
__device__ __forceinline__ float gelu_erf(float x) {
  float cdf, e;
  vl_cdf_pdf(x, cdf, e);
  return x * cdf;
}
// value and derivative together (the forward epilogue stores the derivative for the backward pass: one fma more there,
// ~25 VALU ops per element less in the GELU' epilogue of the backward GEMM)
__device__ __forceinline__ void gelu_erf_both(float x, float& y, float& dy) {
  float cdf, e;
  vl_cdf_pdf(x, cdf, e);
  y = x * cdf;
  dy = fmaf(x * 0.3989422804014327f, e, cdf);
}
__device__ __forceinline__ float gelu_erf_grad(float x) {
  float cdf, e;
  vl_cdf_pdf(x, cdf, e);
  return fmaf(x * 0.3989422804014327f, e, cdf);
}

// ---- counter-based RNG for dropout ------------------------------------------------------------
// One 32-bit draw per (seed, stream offset, element index); the backward pass regenerates the same
// keep-mask instead of storing it.  splitmix64-style finaliser: plenty for Bernoulli masks.
__device__ __forceinline__ uint32_t vl_rand_u32(uint64_t seed, uint64_t idx) {
  uint64_t z = seed + idx * 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z = z ^ (z >> 31);
  return (uint32_t)(z >> 32);
}
// keep-scale: 1/(1-p) when kept, 0 when dropped (p == 0 -> always 1)
__device__ __forceinline__ float vl_dropout_scale(uint64_t seed, uint64_t idx, float p, float inv_keep) {
  if (p <= 0.f) return 1.0f;
  const float u = (float)(vl_rand_u32(seed, idx) >> 8) * (1.0f / 16777216.0f);
  return u >= p ? inv_keep : 0.0f;
}

// Four keep-scales from ONE 64-bit draw: element t of the aligned group `gidx` uses bits [16t, 16t+16) of the hash
// (drop probability quantised to 1/65536).  A lane of the LayerNorm kernels holds 4 consecutive elements of a row and
// a lane of the attention kernels 4 consecutive queries of one key, so this is one splitmix64 (~25 VALU ops, most of the
// per-element work of the attention soft-max loop) per 4 elements instead of per element.
__device__ __forceinline__ void vl_dropout_scale4(uint64_t seed, uint64_t gidx, float p, float inv_keep, float s[4]) {
  uint64_t z = seed + gidx * 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z = z ^ (z >> 31);
  const uint32_t thr = (uint32_t)(p * 65536.0f + 0.5f);
  const uint32_t lo = (uint32_t)z, hi = (uint32_t)(z >> 32);
  s[0] = (lo & 0xFFFFu) >= thr ? inv_keep : 0.0f;
  s[1] = (lo >> 16) >= thr ? inv_keep : 0.0f;
  s[2] = (hi & 0xFFFFu) >= thr ? inv_keep : 0.0f;
  s[3] = (hi >> 16) >= thr ? inv_keep : 0.0f;
}

// ---- wave reductions (64 lanes) ------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float group16_max(float v) {  // across lanes sharing lane>>4
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float group16_sum(float v) {
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
