// care_common.h - shared device helpers for the gfx950 kernels of libcare_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/care_hip.h"

// The 16-bit storage type of the library (weights, activation mirrors, K/V caches, MFMA operands).  ONE source, two
// libraries: libcare_hip.so with bfloat16 (8 significand bits: compute mode `bf16`) and libcare_hip_f16.so with IEEE half
// (-DCARE_H16_FP16: 11 significand bits at the same bytes and the same MFMA rate: compute mode `fp16`, care_amd/build.py).
// `bf16_t` / `bf16x8` are the historical names of that type in the kernels; read them as "the library's 16-bit float".
// Accumulators, LayerNorm / softmax statistics and the residual stream are fp32 in both.  The split-fp16 products of the
// concept embedder and of fp16x3 mode name _Float16 explicitly and are the same in both libraries.
#ifdef CARE_H16_FP16
typedef _Float16 h16_t;
#define CARE_H16_NAME "fp16"
#else
typedef __bf16 h16_t;
#define CARE_H16_NAME "bf16"
#endif
typedef h16_t bf16_t;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef h16_t bf16x8 __attribute__((ext_vector_type(8)));
typedef h16_t bf16x4 __attribute__((ext_vector_type(4)));
typedef float care_f32x16 __attribute__((ext_vector_type(16)));
typedef short care_s16x4 __attribute__((ext_vector_type(4)));

// The library's MFMAs on its 16-bit type (the trailing cbsz / abid / blgp arguments of the builtins are always 0 here and
// are accepted only so that call sites read like the builtin): v_mfma_f32_{16x16x32,32x32x16,16x16x16}_{bf16,f16}.
__device__ __forceinline__ f32x4 care_mfma_16x16x32_h16(bf16x8 a, bf16x8 b, f32x4 c, int = 0, int = 0, int = 0) {
#ifdef CARE_H16_FP16
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
#else
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
#endif
}
__device__ __forceinline__ care_f32x16 care_mfma_32x32x16_h16(bf16x8 a, bf16x8 b, care_f32x16 c, int = 0, int = 0, int = 0) {
#ifdef CARE_H16_FP16
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
#else
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
#endif
}
// (operands as four 16-bit lanes in a short4, the form the bf16 _1k builtin takes)
__device__ __forceinline__ f32x4 care_mfma_16x16x16_h16(care_s16x4 a, care_s16x4 b, f32x4 c, int = 0, int = 0, int = 0) {
#ifdef CARE_H16_FP16
  typedef _Float16 hx4 __attribute__((ext_vector_type(4)));
  return __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(hx4, a), __builtin_bit_cast(hx4, b), c, 0, 0, 0);
#else
  return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0);
#endif
}

#define CARE_WAVE 64

static inline int care_launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}

static inline bool care_aligned16(const void* p) { return (((uintptr_t)p) & 15u) == 0; }

// Raise a kernel's dynamic-LDS limit, once per (kernel, device): `done` is a per-kernel bit mask
// over device ordinals.  Safe from any thread: the attribute call is idempotent, so two racing first
// callers at worst both make it.  Returns 0 or the hipError_t.
#include <atomic>
static inline int care_allow_dynamic_lds(const void* kernel, int bytes, std::atomic<unsigned long long>& done) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return (int)e;
  const unsigned long long bit = 1ull << (dev & 63);
  if (done.load(std::memory_order_acquire) & bit) return 0;
  e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e != hipSuccess) return (int)e;
  done.fetch_or(bit, std::memory_order_release);
  return 0;
}

__device__ __forceinline__ float care_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__device__ __forceinline__ float care_wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// Wave-wide maximum through the DPP data path: an inclusive max-scan along each row of 16 lanes
// (row_shr 1, 2, 4, 8), then row_bcast15 / row_bcast31 carry the row results to lane 63, which is
// broadcast.  ~8 VALU operations; the xor-shuffle version is 6 DEPENDENT ds_bpermute round trips
// through the LDS crossbar (~100 cycles each).  All 64 lanes must be active.
__device__ __forceinline__ float care_wave_max_dpp(float v) {
#define CARE_DPP_MAX(CTRL, ROWMASK)                                                                       \
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v),          \
                                                                     __builtin_bit_cast(int, v), CTRL, ROWMASK, \
                                                                     0xF, false)))
  CARE_DPP_MAX(0x111, 0xF);  // row_shr:1
  CARE_DPP_MAX(0x112, 0xF);  // row_shr:2
  CARE_DPP_MAX(0x114, 0xF);  // row_shr:4
  CARE_DPP_MAX(0x118, 0xF);  // row_shr:8 -> lane 15 of every row holds the row maximum
  CARE_DPP_MAX(0x142, 0xA);  // row_bcast15 into rows 1 and 3
  CARE_DPP_MAX(0x143, 0xC);  // row_bcast31 into rows 2 and 3 -> lane 63 holds the wave maximum
#undef CARE_DPP_MAX
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// Lane exchanges through the DPP data path (no LDS-crossbar round trip), for reductions:
//   care_dpp_x1 / _x2 : the value of lane ^ 1 / lane ^ 2 (quad_perm);
//   care_dpp_m8       : the value of the mirrored lane of the 8-lane group (row_half_mirror) - after
//                       the x1, x2 steps of a sum / max every quad holds one value, so this is the
//                       lane ^ 4 step of the reduction, bit for bit;
//   care_dpp_x8       : the value of lane ^ 8 (row_ror:8).
template <int CTRL>
__device__ __forceinline__ float care_dpp(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float care_dpp_x1(float v) { return care_dpp<0xB1>(v); }   // quad_perm [1,0,3,2]
__device__ __forceinline__ float care_dpp_x2(float v) { return care_dpp<0x4E>(v); }   // quad_perm [2,3,0,1]
__device__ __forceinline__ float care_dpp_m8(float v) { return care_dpp<0x141>(v); }  // row_half_mirror
__device__ __forceinline__ float care_dpp_x8(float v) { return care_dpp<0x128>(v); }  // row_ror:8

// load 8 consecutive elements of a K/V row as fp32
__device__ __forceinline__ void care_load8(const float* p, float (&o)[8]) {
  float4 a = *reinterpret_cast<const float4*>(p);
  float4 b = *reinterpret_cast<const float4*>(p + 4);
  o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w;
  o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
}
__device__ __forceinline__ void care_load8(const bf16_t* p, float (&o)[8]) {
  bf16x8 v = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
  for (int i = 0; i < 8; ++i) o[i] = (float)v[i];
}
