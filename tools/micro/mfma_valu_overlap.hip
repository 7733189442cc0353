// Does VALU work of the SAME wave overlap its MFMAs on gfx950?  One wave per SIMD, a loop of MFMAs (two independent
// accumulator chains) with K independent v_fma per MFMA woven in.  hipcc --offload-arch=gfx950 -O3 ... ; ./a.out
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NV, bool MF>
__global__ __launch_bounds__(256, 1) void k(float* out, int iters) {
  bf16x8 a = {}, b = {};
  a[0] = (__bf16)1.0f; b[0] = (__bf16)(1.0f + threadIdx.x * 1e-3f);
  f32x16 c0 = {}, c1 = {};
  float x[8];
  for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 0.001f + i;
  const float m = 1.0001f, d = 0.5f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (MF) c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int v = 0; v < NV; ++v) x[(u + v) & 7] = __builtin_fmaf(x[(u + v) & 7], m, d);
      __builtin_amdgcn_sched_barrier(0);
      if (MF) c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int v = 0; v < NV; ++v) x[(u + v + 3) & 7] = __builtin_fmaf(x[(u + v + 3) & 7], m, d);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float s = 0;
  for (int i = 0; i < 16; ++i) s += c0[i] + c1[i];
  for (int i = 0; i < 8; ++i) s += x[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NV, bool MF>
float run(float* out, int iters) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<NV, MF>), dim3(256), dim3(256), 0, 0, out, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<NV, MF>), dim3(256), dim3(256), 0, 0, out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms * 1e3f;
}

int main() {
  float* out; hipMalloc(&out, 256 * 256 * 4);
  const int iters = 4000;  // 16 MFMAs per iteration
  printf("MFMA only            : %8.1f us\n", run<0, true>(out, iters));
  printf("VALU only  (4 / slot): %8.1f us\n", run<4, false>(out, iters));
  printf("MFMA + VALU (4 / slot): %8.1f us\n", run<4, true>(out, iters));
  printf("VALU only  (7 / slot): %8.1f us\n", run<7, false>(out, iters));
  printf("MFMA + VALU (7 / slot): %8.1f us\n", run<7, true>(out, iters));
  printf("VALU only  (12 / slot): %8.1f us\n", run<12, false>(out, iters));
  printf("MFMA + VALU (12 / slot): %8.1f us\n", run<12, true>(out, iters));
  return 0;
}
