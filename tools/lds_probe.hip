// LDS bank-conflict probe for the access patterns of attention2.hip (SQ_LDS_BANK_CONFLICT / SQ_LDS_ACTIVE was
// 0.22 / 0.28 for the attention kernels although every pattern is conflict-free on paper).  One kernel per pattern, each a
// loop of that access only, so that a PMC pass attributes conflicts to a pattern:
//   hipcc --offload-arch=gfx950 -O3 -o tools/lds_probe tools/lds_probe.hip
//   rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d gpurun_out/lds -- tools/lds_probe
#include <hip/hip_runtime.h>
#include <cstdio>

typedef __attribute__((ext_vector_type(4))) short s16x4;
constexpr int PITCH = 128, SPAD = 64, ITER = 2000;

// chunk swizzles under test (16-byte chunks of a 128-byte row): 0 = attention2.hip round 3, 1 = gemm.hip lds_off, 2 = row & 7,
// 3 = pairs only (no low bit), 4 = none, 5 = gemm.hip swz3 (ping-pong kernel), 6 / 7: candidates
template <int SW> __device__ __forceinline__ int sw(int row) {
  return SW == 0 ? ((((row >> 1) & 3) << 1) | ((row >> 3) & 1)) : SW == 1 ? ((row >> 1) & 7) : SW == 2 ? (row & 7)
       : SW == 3 ? (((row >> 1) & 3) << 1)
       : SW == 5 ? (((row >> 1) & 1) | (((row >> 3) & 1) << 1) | ((((row >> 4) ^ (row >> 2)) & 1) << 2))
       : SW == 6 ? (((row >> 1) & 1) | (((row >> 2) & 1) << 1) | (((row >> 3) & 1) << 2))
       : SW == 7 ? (((row >> 2) & 1) | (((row >> 1) & 1) << 1) | (((row >> 3) & 1) << 2)) : 0;
}
template <int SW> __device__ __forceinline__ int chunk_off(int row, int chunk) { return row * PITCH + ((chunk ^ sw<SW>(row)) << 4); }
template <int SW> __device__ __forceinline__ int tr_off(int row, int dt, int pp) {
  return row * PITCH + (((2 * dt + (pp >> 1)) ^ sw<SW>(row)) << 4) + 8 * (pp & 1);
}

template <int SW>
__global__ __launch_bounds__(256) void probe_write_b128(unsigned* out) {
  __shared__ __attribute__((aligned(16))) unsigned char sm[4 * SPAD * PITCH];
  const int tid = threadIdx.x;
  uint4 v = make_uint4(tid, 1, 2, 3);
  for (int it = 0; it < ITER; ++it) {
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int idx = tid + i * 256, row = idx >> 3, c = idx & 7;
        *reinterpret_cast<uint4*>(sm + m * SPAD * PITCH + chunk_off<SW>(row, c)) = v;
      }
    v.x += 1;
    __syncthreads();
  }
  out[blockIdx.x * 256 + tid] = *reinterpret_cast<unsigned*>(sm + 4 * tid);
}

template <int SW>
__global__ __launch_bounds__(256) void probe_read_b128(unsigned* out) {
  __shared__ __attribute__((aligned(16))) unsigned char sm[4 * SPAD * PITCH];
  const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, g = lane >> 4;
  for (int i = tid; i < 4 * SPAD * PITCH / 4; i += 256) reinterpret_cast<unsigned*>(sm)[i] = i;
  __syncthreads();
  unsigned acc = 0;
  for (int it = 0; it < ITER; ++it) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        // (inline assembly: with plain loads the compiler narrows a read whose components are not all used)
        typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
        u32x4 a, b;
        const unsigned oa = (unsigned)(size_t)(sm + chunk_off<SW>(t * 16 + l15, g + 4 * ks));
        asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:8192\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(a), "=&v"(b) : "v"(oa) : "memory");
        acc += a[0] + b[1];
      }
    asm volatile("" ::: "memory");
  }
  out[blockIdx.x * 256 + tid] = acc;
}

template <int SW>
__global__ __launch_bounds__(256) void probe_read_tr(unsigned* out) {
  __shared__ __attribute__((aligned(16))) unsigned char sm[4 * SPAD * PITCH];
  const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, g = lane >> 4, qq = l15 >> 2, pp = lane & 3;
  for (int i = tid; i < 4 * SPAD * PITCH / 4; i += 256) reinterpret_cast<unsigned*>(sm)[i] = i;
  __syncthreads();
  int acc = 0;
  for (int it = 0; it < ITER; ++it) {
#pragma unroll
    for (int pr = 0; pr < 2; ++pr)
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        const int r0 = 32 * pr + 4 * g + qq, r1 = r0 + 16;
        const int o0 = tr_off<SW>(r0, dt, pp), o1 = tr_off<SW>(r1, dt, pp);
        const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(sm + o0));
        const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(sm + o1));
        acc += a[0] + b[1];
      }
    asm volatile("" ::: "memory");
  }
  out[blockIdx.x * 256 + tid] = (unsigned)acc;
}

__global__ __launch_bounds__(256) void probe_read_mask(float* out) {
  __shared__ __attribute__((aligned(16))) float sm[3 * SPAD];
  const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, g = lane >> 4;
  for (int i = tid; i < 3 * SPAD; i += 256) sm[i] = (float)i;
  __syncthreads();
  float acc = 0.f;
  for (int it = 0; it < ITER; ++it) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const float4 mk = *reinterpret_cast<const float4*>(sm + 16 * t + 4 * g);   // key mask of a key tile (forward, phase A)
      acc += mk.x + mk.w;
#pragma unroll
      for (int r = 0; r < 4; ++r) acc += sm[SPAD + t * 16 + 4 * g + r] + sm[2 * SPAD + t * 16 + 4 * g + r];  // lse / delta (phase B)
    }
    acc += sm[l15];  // one value per lane column (slse[q], smask[key])
    asm volatile("" ::: "memory");
  }
  out[blockIdx.x * 256 + tid] = acc;
}

int main() {
  unsigned* out;
  if (hipMalloc(&out, 256 * 256 * 4) != hipSuccess) return 1;
#define RUN3(SW)                                                                      \
  hipLaunchKernelGGL(probe_write_b128<SW>, dim3(256), dim3(256), 0, 0, out);           \
  hipLaunchKernelGGL(probe_read_b128<SW>, dim3(256), dim3(256), 0, 0, out);            \
  hipLaunchKernelGGL(probe_read_tr<SW>, dim3(256), dim3(256), 0, 0, out);
  for (int rep = 0; rep < 2; ++rep) {
    RUN3(0) RUN3(1) RUN3(2) RUN3(3) RUN3(4) RUN3(5) RUN3(6) RUN3(7)
    hipLaunchKernelGGL(probe_read_mask, dim3(256), dim3(256), 0, 0, reinterpret_cast<float*>(out));
  }
  const hipError_t e = hipDeviceSynchronize();
  printf("lds_probe: %s\n", hipGetErrorString(e));
  return e == hipSuccess ? 0 : 1;
}
