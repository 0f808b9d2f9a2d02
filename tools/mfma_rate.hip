// Sustained MFMA issue rate and package power for the two bf16 MFMA shapes (no memory traffic): one 8-wave workgroup per
// CU (128 KB of LDS requested so that exactly one fits), every wave issuing independent MFMAs back to back.
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_rate tools/mfma_rate.hip     Run: ./mfma_rate <shape 16|32> <seconds> [workgroups]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int SHAPE>
__global__ __launch_bounds__(512) void mfma_loop(float* out, int iters) {
  extern __shared__ unsigned char smem[];
  const int lane = threadIdx.x & 63;
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (lane + i)); b[i] = (__bf16)(0.002f * (lane - i)); }
  float r = 0.f;
  if (SHAPE == 16) {
    f32x4 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
    for (int i = 0; i < 16; ++i) r += acc[i][0] + acc[i][3];
  } else {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int rep = 0; rep < 2; ++rep)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
    for (int i = 0; i < 4; ++i) r += acc[i][0] + acc[i][15];
  }
  if (r == 123.456f) out[threadIdx.x] = r + smem[0];
}

int main(int argc, char** argv) {
  const int shape = argc > 1 ? atoi(argv[1]) : 16;
  const double seconds = argc > 2 ? atof(argv[2]) : 3.0;
  const int wgs = argc > 3 ? atoi(argv[3]) : 256;
  float* out;
  hipMalloc(&out, 4096);
  const size_t lds = 128 * 1024;
  hipFuncSetAttribute(reinterpret_cast<const void*>(&mfma_loop<16>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&mfma_loop<32>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const int iters = 20000;  // per launch: 16 (8) MFMAs of 16384 (32768) flop * 64... per wave and iteration
  const double flop_per_launch = (double)wgs * 8 * iters * (shape == 16 ? 16 * 2.0 * 16 * 16 * 32 : 8 * 2.0 * 32 * 32 * 16);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  double total_ms = 0; int launches = 0;
  while (total_ms < seconds * 1e3) {
    hipEventRecord(e0);
    for (int k = 0; k < 10; ++k) {
      if (shape == 16) hipLaunchKernelGGL(mfma_loop<16>, dim3(wgs), dim3(512), lds, 0, out, iters);
      else hipLaunchKernelGGL(mfma_loop<32>, dim3(wgs), dim3(512), lds, 0, out, iters);
    }
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    total_ms += ms; launches += 10;
    if (launches % 50 == 0) printf("shape %d  %d workgroups: %.0f TFLOP/s\n", shape, wgs, flop_per_launch * 10 / (ms * 1e-3) / 1e12), fflush(stdout);
  }
  printf("shape %d  %d workgroups: average %.0f TFLOP/s over %.1f s\n", shape, wgs, flop_per_launch * launches / (total_ms * 1e-3) / 1e12, total_ms * 1e-3);
  return 0;
}
