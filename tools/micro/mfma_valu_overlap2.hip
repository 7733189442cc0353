// Do the MFMAs of ONE wave overlap the VALU work of ANOTHER wave on the same SIMD (gfx950)?  512 threads per workgroup,
// one workgroup per CU: waves 0-3 (one per SIMD) run an MFMA loop, waves 4-7 (their SIMD mates) a v_fma loop.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// mode: 1 = MFMA waves only work, 2 = VALU waves only work, 3 = both
__global__ __launch_bounds__(512, 2) void k(float* out, int iters, int mode) {
  const int wave = threadIdx.x >> 6;
  float s = 0;
  if (wave < 4) {
    if (mode & 1) {
      bf16x8 a = {}, b = {};
      a[0] = (__bf16)1.0f; b[0] = (__bf16)(1.0f + threadIdx.x * 1e-3f);
      f32x16 c0 = {}, c1 = {};
      for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
          c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
        }
      for (int i = 0; i < 16; ++i) s += c0[i] + c1[i];
    }
  } else if (mode & 2) {
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 0.001f + i;
    const float m = 1.0001f, d = 0.5f;
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int u = 0; u < 16; ++u)
#pragma unroll
        for (int v = 0; v < 8; ++v) x[v] = __builtin_fmaf(x[v], m, d);   // 128 v_fma per iteration (16 MFMAs in the other waves)
    for (int i = 0; i < 8; ++i) s += x[i];
  }
  out[blockIdx.x * 512 + threadIdx.x] = s;
}

float run(float* out, int iters, int mode) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, out, iters, mode);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, out, iters, mode);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms * 1e3f;
}

int main() {
  float* out; hipMalloc(&out, 256 * 512 * 4);
  const int iters = 4000;
  printf("MFMA waves only (16 MFMA / iteration)     : %8.1f us\n", run(out, iters, 1));
  printf("VALU waves only (128 v_fma / iteration)    : %8.1f us\n", run(out, iters, 2));
  printf("both, on the same SIMDs                    : %8.1f us\n", run(out, iters, 3));
  return 0;
}
