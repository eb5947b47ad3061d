// Shared device/host helpers for the scenesplat_hip C-ABI library (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <stdint.h>

#define SS_OK 0
#define SS_ERR_ARG 1
#define SS_ERR_LAUNCH 2
#define SS_ERR_WORKSPACE 3

#define SS_F32 0
#define SS_BF16 1

// Launch + immediate status check.  The sticky HIP error is cleared first: the host process
// (PyTorch) may have left an unrelated non-fatal error (e.g. hipErrorNotReady from an event
// query) in the thread's error slot.
#define SS_LAUNCH(kern, grid, block, shmem, stream, ...)                       \
  do {                                                                          \
    (void)hipGetLastError();                                                    \
    hipLaunchKernelGGL(kern, grid, block, shmem, stream, __VA_ARGS__);          \
    if (hipGetLastError() != hipSuccess) return SS_ERR_LAUNCH;                  \
  } while (0)
#define SS_CHECK_LAUNCH() do { } while (0)

static inline int ss_div_up(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

typedef __attribute__((ext_vector_type(8))) short bf16x8_t;   // 8 bf16 = 4 VGPRs (MFMA A/B fragment)
typedef __attribute__((ext_vector_type(4))) short bf16x4_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;     // 16x16 MFMA accumulator
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

__device__ __forceinline__ float bf16_to_f32(unsigned short v) {
  return __uint_as_float(((unsigned int)v) << 16);
}
// round-to-nearest-even; NaN stays NaN (plain cast lowers to v_cvt_pk_bf16_f32 on gfx950)
__device__ __forceinline__ unsigned short f32_to_bf16(float f) {
  __hip_bfloat16 b = __float2bfloat16(f);
  return *reinterpret_cast<unsigned short*>(&b);
}
// two f32 -> packed bf16x2 with ONE v_cvt_pk_bf16_f32 (RNE; NaN stays NaN)
typedef __attribute__((ext_vector_type(2))) float ss_f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 ss_bf16x2_t;
__device__ __forceinline__ unsigned int pack_bf16x2(float lo, float hi) {
  ss_f32x2_t v = {lo, hi};
  return __builtin_bit_cast(unsigned int, __builtin_convertvector(v, ss_bf16x2_t));
}

template <typename T> struct ElemIO;
template <> struct ElemIO<float> {
  static __device__ __forceinline__ float load(const float* p) { return *p; }
  static __device__ __forceinline__ void store(float* p, float v) { *p = v; }
};
template <> struct ElemIO<unsigned short> {  // bf16 storage
  static __device__ __forceinline__ float load(const unsigned short* p) { return bf16_to_f32(*p); }
  static __device__ __forceinline__ void store(unsigned short* p, float v) { *p = f32_to_bf16(v); }
};

__device__ __forceinline__ float wave_reduce_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_reduce_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
