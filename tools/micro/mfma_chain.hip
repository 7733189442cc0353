// mfma_chain.hip - issue interval of DEPENDENT MFMAs on gfx950: one wave per SIMD (256 threads per CU, one workgroup per CU)
// runs N MFMAs as 1, 2 or 4 independent accumulator chains; cycles by s_memtime around the loop.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_chain.hip -o tools/micro/mfma_chain && tools/micro/mfma_chain
// *Measured* (round 5): v_mfma_f32_32x32x16_bf16 32.0 cycles per MFMA with ONE accumulator chain in one wave (2, 4 chains: the
// same; two waves per SIMD: 32 per SIMD, the older wave served first); v_mfma_f32_16x16x32_bf16 17.0 with one chain, 20.0 / 23.8
// with 2 / 4 chains in ONE wave, 16.1 - 16.7 per SIMD with 2 or 4 waves.  A single dependent chain is not what holds the
// 32 x 32 x 16 kernels (gemm_vocab.hip, gemm_store32.hip) at 54 cycles per MFMA: the partner wave's vector work is.
#include <hip/hip_runtime.h>
#include <cstdio>
#pragma clang diagnostic ignored "-Wunused-value"
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int CHAINS, int BIG, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k(float* out, unsigned long long* cyc, int iters) {
  bf16x8 a = {}, b = {};
  a[0] = (__bf16)1.0f; b[0] = (__bf16)(1.0f + threadIdx.x * 1e-3f);
  f32x16 c[4] = {};
  f32x4 d[4] = {};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      if constexpr (BIG) c[u % CHAINS] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c[u % CHAINS], 0, 0, 0);
      else d[u % CHAINS] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, d[u % CHAINS], 0, 0, 0);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int i = 0; i < 4; ++i) { for (int j = 0; j < 16; ++j) s += c[i][j]; for (int j = 0; j < 4; ++j) s += d[i][j]; }
  out[blockIdx.x * 64 * WAVES + threadIdx.x] = s;
  if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) cyc[threadIdx.x >> 6] = t1 - t0;  // per wave: the oldest wave of a SIMD wins the arbitration
}

template <int CHAINS, int BIG, int WAVES>
void run(float* out, unsigned long long* cyc) {
  const int iters = 2000;
  for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((k<CHAINS, BIG, WAVES>), dim3(256), dim3(64 * WAVES), 0, 0, out, cyc, iters);
  hipDeviceSynchronize();
  unsigned long long h[16];
  hipMemcpy(h, cyc, 8 * WAVES, hipMemcpyDeviceToHost);
  unsigned long long mx = 0, mn = ~0ull;
  for (int w = 0; w < WAVES; ++w) { mx = h[w] > mx ? h[w] : mx; mn = h[w] < mn ? h[w] : mn; }
  printf("%s, %d chain(s), %d wave(s) per SIMD: %6.1f .. %6.1f ticks per MFMA of a wave (slowest: %5.1f per SIMD)\n", BIG ? "32x32x16" : "16x16x32", CHAINS,
         WAVES / 4, (double)mn / (iters * 16.0), (double)mx / (iters * 16.0), (double)mx / (iters * 16.0) / (WAVES / 4));
}

int main() {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 8 * 16);
  run<1, 1, 4>(out, cyc); run<2, 1, 4>(out, cyc); run<4, 1, 4>(out, cyc);
  run<1, 1, 8>(out, cyc); run<2, 1, 8>(out, cyc);
  run<1, 0, 4>(out, cyc); run<2, 0, 4>(out, cyc); run<4, 0, 4>(out, cyc);
  run<1, 0, 8>(out, cyc); run<2, 0, 8>(out, cyc); run<4, 0, 8>(out, cyc); run<1, 0, 16>(out, cyc); run<4, 0, 16>(out, cyc);
  return 0;
}
