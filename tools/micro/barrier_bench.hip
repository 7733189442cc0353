// barrier_bench.hip - cost of a grid barrier between resident workgroups on gfx950 (tools only, not in the library).
//   hipcc --offload-arch=gfx950 -O3 tools/micro/barrier_bench.hip -o tools/micro/barrier_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// mode bits: 1 = every wave fences (else thread 0 only); 2 = pollers read a flag the last arriver publishes;
//            4 = two levels (8 sub-counters by blockIdx % 8, then one); 8 = no fences at all
__device__ __forceinline__ void do_sleep(int s) { if (s >= 8) __builtin_amdgcn_s_sleep(8); else if (s >= 1) __builtin_amdgcn_s_sleep(1); }

template <int MODE>
__global__ __launch_bounds__(256) void bar_kernel(unsigned* sync, int iters, int sleep, float* sink) {
  unsigned target = 0, gen = 0;
  const unsigned nb = gridDim.x;
  float acc = 0.f;
  for (int it = 0; it < iters; ++it) {
    // a little dependent "work": one store + one load of another block's slot
    sink[blockIdx.x * 64 + (threadIdx.x & 63)] = acc + it;
    if (MODE & 1) __threadfence();
    __syncthreads();
    target += nb;
    gen += 1;
    if (threadIdx.x == 0) {
      if (!(MODE & 1) && !(MODE & 8)) __threadfence();
      if (MODE & 4) {
        const unsigned grp = blockIdx.x & 7, ngrp = (nb + 7 - grp) / 8;  // blocks with this residue
        unsigned old = __hip_atomic_fetch_add(sync + 64 + grp * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old == gen * ngrp - 1) {  // last of the group
          unsigned o2 = __hip_atomic_fetch_add(sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          const unsigned ngroups = nb < 8 ? nb : 8;
          if (o2 == gen * ngroups - 1) __hip_atomic_store(sync + 32, gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        while (__hip_atomic_load(sync + 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gen) do_sleep(sleep);
      } else if (MODE & 2) {
        unsigned old = __hip_atomic_fetch_add(sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old == target - 1) __hip_atomic_store(sync + 32, gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(sync + 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gen) do_sleep(sleep);
      } else {
        __hip_atomic_fetch_add(sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) do_sleep(sleep);
      }
      if (!(MODE & 1) && !(MODE & 8)) __threadfence();
    }
    __syncthreads();
    if (MODE & 1) __threadfence();
    acc += sink[((blockIdx.x + 1) % nb) * 64 + (threadIdx.x & 63)];
  }
  if (acc == 12345.678f) sink[0] = acc;
}

template <int MODE>
void run(int nb, int iters, int sleep, unsigned* sync, float* sink) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int rep = 0; rep < 2; ++rep) {
    CK(hipMemset(sync, 0, 4096));
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(bar_kernel<MODE>, dim3(nb), dim3(256), 0, 0, sync, iters, sleep, sink);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    if (rep == 1) printf("mode %2d blocks %3d sleep %2d: %.2f us / barrier\n", MODE, nb, sleep, ms * 1e3 / iters);
  }
}

int main() {
  unsigned* sync; float* sink;
  CK(hipMalloc(&sync, 4096)); CK(hipMalloc(&sink, 256 * 64 * 4 * 2));
  CK(hipMemset(sink, 0, 256 * 64 * 4 * 2));
  const int iters = 2000;
  for (int nb : {8, 32, 64, 128, 256}) {
    for (int sleep : {0, 1, 8}) {
      run<0>(nb, iters, sleep, sync, sink);
      run<1>(nb, iters, sleep, sync, sink);
      run<2>(nb, iters, sleep, sync, sink);
      run<3>(nb, iters, sleep, sync, sink);
      run<4>(nb, iters, sleep, sync, sink);
      run<8>(nb, iters, sleep, sync, sink);
      run<10>(nb, iters, sleep, sync, sink);
      run<12>(nb, iters, sleep, sync, sink);
    }
  }
  return 0;
}
