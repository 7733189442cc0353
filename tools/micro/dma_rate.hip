// dma_rate.hip - what the L2 -> LDS path of a 256 x 256 GEMM tile delivers on gfx950, by the shape of the 1-KB pieces
// one wave instruction moves (tools only, not in the library).
//   hipcc --offload-arch=gfx950 -O3 tools/micro/dma_rate.hip -o tools/micro/dma_rate && tools/micro/dma_rate
// One workgroup of 8 waves per output tile (tm, tn) of C[M, N] = A[M, K] W[N, K]^T, tiles in the order of
// csrc/gemm_tile.hip (bands of 8 row tiles per XCD run): it fetches its 256 A rows and 256 W rows, K bf16 each, in steps;
// every wave keeps DEPTH pieces in flight (counted vmcnt) and nothing else happens (no LDS reads, no MFMA).
//   ROWB = bytes of a row a piece takes: 64 (16 rows per piece), 128 (8), 256 (4), 1024 (1)
//   REG  = 1: global_load_dwordx4 into registers instead of LDS-DMA
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int ROWB, int DEPTH, int REG, int SWZ = 0>
__global__ __launch_bounds__(512) void dma_kernel(const unsigned char* A, const unsigned char* W, int tiles_m, int tiles_n, int K,
                                                  float* sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  int tm, tn;
  {
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
    const int t = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
    const int band = 8 * tiles_n, b = t / band, o = t - b * band;
    const int rows = min(8, tiles_m - b * 8);
    tm = b * 8 + o % rows;
    tn = o / rows;
  }
  constexpr int RPP = 1024 / ROWB;     // rows per piece
  constexpr int LPR = ROWB / 16;       // lanes per row
  constexpr int PPW = 512 / RPP / 8;   // pieces per wave and step (A and W together: 512 rows)
  const int64_t ldb = (int64_t)K * 2;
  const unsigned char* src[PPW];
#pragma unroll
  for (int i = 0; i < PPW; ++i) {
    const int R = (wave * PPW + i) * RPP + lane / LPR;  // 0 .. 511: A rows, then W rows
    src[i] = (R < 256 ? A + (int64_t)(tm * 256 + R) * ldb : W + (int64_t)(tn * 256 + R - 256) * ldb) + (((lane % LPR) ^ (SWZ ? ((lane / LPR) & (LPR - 1)) : 0)) * 16);
  }
  const int steps = (K * 2) / ROWB;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  f32x4 ring[REG ? DEPTH : 1];
  int slot = 0;
  for (int s = 0; s < steps; ++s) {
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      if constexpr (REG) {
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(ring[i % DEPTH]) : "v"(src[i] + (int64_t)s * ROWB) : "memory");
      } else {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[i] + (int64_t)s * ROWB),
                                         (__attribute__((address_space(3))) void*)(smem + (wave * 16 + slot) * 1024), 16, 0, 0);
        slot = (slot + 1) & 15;
      }
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH - 1) : "memory");  // a rolling window of DEPTH pieces
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if constexpr (REG) {
#pragma unroll
    for (int i = 0; i < (REG ? DEPTH : 1); ++i) acc += ring[i];
  }
  if (acc[0] == 12345.678f) sink[0] = acc[1];
}

template <int ROWB, int DEPTH, int REG, int SWZ = 0>
void run(const unsigned char* A, const unsigned char* W, int M, int N, int K, float* sink) {
  const int tiles_m = M / 256, tiles_n = N / 256;
  const int lds = REG ? 0 : 128 * 1024;
  if (lds) CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&dma_kernel<ROWB, DEPTH, REG, SWZ>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  float best = 1e30f;
  for (int rep = 0; rep < 4; ++rep) {
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((dma_kernel<ROWB, DEPTH, REG, SWZ>), dim3(tiles_m * tiles_n), dim3(512), lds, 0, A, W, tiles_m, tiles_n, K, sink);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    if (rep && ms < best) best = ms;
  }
  const double bytes = (double)tiles_m * tiles_n * 512.0 * K * 2;
  printf("M=%d N=%d K=%d  piece %2d rows x %4d B%s  depth %2d pieces/wave (%3d KB/CU)  %s: %8.1f us  %6.2f TB/s  %5.1f GB/s per CU\n", M, N, K,
         1024 / ROWB, ROWB, SWZ ? " chunks permuted in the row" : "", DEPTH, DEPTH * 8, REG ? "registers" : "LDS-DMA  ", best * 1e3, bytes / best / 1e9, bytes / best / 1e6 / 256);
}


// ---- a weight image that EVERY workgroup streams (csrc/gemm_ln.hip: the packed W of the row-panel GEMMs, 32 KB per K
// step from L2): by how a 1-KB piece is cut out of the 32-KB stage.  MODE 0: 1 KB contiguous (care_pack_ln_weight's
// order); 1: four runs of 256 B, 8 KB apart; 2: eight runs of 128 B, 4 KB apart.  LOADERS of the 8 waves issue.
template <int MODE, int LOADERS>
__global__ __launch_bounds__(512) void packed_kernel(const unsigned char* W, int nk, int blocks_per_wg, float* sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (wave >= LOADERS) return;
  constexpr int PPW = 32 / LOADERS;  // pieces per wave and stage
  const unsigned lane_off = MODE == 0 ? lane * 16u : MODE == 1 ? (unsigned)(lane >> 4) * 32u * 256u + (lane & 15) * 16u
                                                               : (unsigned)(lane >> 3) * 32u * 128u + (lane & 7) * 16u;
  const unsigned piece_step = MODE == 0 ? 1024u : MODE == 1 ? 256u : 128u;
  int slot = 0;
  for (int b = 0; b < blocks_per_wg; ++b)
    for (int s = 0; s < nk; ++s) {
      const unsigned char* st = W + (size_t)s * 32768;
#pragma unroll
      for (int i = 0; i < PPW; ++i) {
        const int q = wave * PPW + i;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(st + q * piece_step + lane_off),
                                         (__attribute__((address_space(3))) void*)(smem + (wave * 16 + slot) * 1024), 16, 0, 0);
        slot = (slot + 1) & 15;
        asm volatile("s_waitcnt vmcnt(11)" ::: "memory");
      }
    }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (lane == 99) sink[0] = 1.f;
}

template <int MODE, int LOADERS>
void run_packed(const unsigned char* W, float* sink) {
  const int nk = 64, per = 28;  // K = 2048; 28 row blocks per workgroup (917504 rows / 128 / 256)
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&packed_kernel<MODE, LOADERS>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  float best = 1e30f;
  for (int rep = 0; rep < 4; ++rep) {
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((packed_kernel<MODE, LOADERS>), dim3(256), dim3(512), 128 * 1024, 0, W, nk, per, sink);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    if (rep && ms < best) best = ms;
  }
  const double bytes = 256.0 * per * nk * 32768.0;
  printf("packed 2-MB weight image, every workgroup streams it 28 times, %d loader waves, pieces %s: %8.1f us  %6.2f TB/s  (%5.1f cycles of a CU per 1-KB piece at 2.1 GHz)\n",
         LOADERS, MODE == 0 ? "1 x 1024 B        " : MODE == 1 ? "4 x 256 B, 8 KB apart" : "8 x 128 B, 4 KB apart", best * 1e3, bytes / best / 1e9,
         best * 1e-3 * 2.1e9 / (per * nk * 32.0));
}

int main() {
  setvbuf(stdout, nullptr, _IONBF, 0);
  const int M = 65536, N = 4096, K = 1024;  // buffers sized for 65536 x 4096 (A) and 4096 x 4096 (W) elements
  unsigned char *A, *W;
  float* sink;
  CK(hipMalloc(&A, (size_t)65536 * 4096 * 2));
  CK(hipMalloc(&W, (size_t)4096 * 4096 * 2));
  CK(hipMalloc(&sink, 4096));
  CK(hipMemset(A, 1, (size_t)65536 * 4096 * 2));
  CK(hipMemset(W, 1, (size_t)4096 * 4096 * 2));
  run<64, 6, 0>(A, W, M, N, K, sink);
  run<64, 12, 0>(A, W, M, N, K, sink);
  run<128, 6, 0>(A, W, M, N, K, sink);
  run<128, 12, 0>(A, W, M, N, K, sink);
  run<256, 12, 0>(A, W, M, N, K, sink);
  run<1024, 12, 0>(A, W, M, N, K, sink);
  run<64, 12, 0, 1>(A, W, M, N, K, sink);
  run<128, 12, 0, 1>(A, W, M, N, K, sink);
  run<64, 12, 0>(A, W, 65536, 2048, 4096, sink);
  run<128, 12, 0>(A, W, 65536, 2048, 4096, sink);
  run<128, 12, 0, 1>(A, W, 65536, 2048, 4096, sink);
  run<64, 4, 1>(A, W, M, N, K, sink);
  run<128, 4, 1>(A, W, M, N, K, sink);
  run<128, 8, 1>(A, W, M, N, K, sink);
  // a small problem: one tile per CU, operands hot in the caches after the first repetition
  run<64, 12, 0>(A, W, 4096, 4096, K, sink);
  run<128, 12, 0>(A, W, 4096, 4096, K, sink);
  run_packed<0, 4>(W, sink); run_packed<1, 4>(W, sink); run_packed<2, 4>(W, sink);
  run_packed<0, 8>(W, sink); run_packed<1, 8>(W, sink); run_packed<2, 8>(W, sink);
  return 0;
}
