// Internal helpers shared by the HIP translation units of libvlnce_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/vlnce_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

void vlnce_set_error(const char* fmt, ...);

#define VLNCE_CHECK_ARG(cond, ...)      \
  do {                                  \
    if (!(cond)) {                      \
      vlnce_set_error(__VA_ARGS__);     \
      return 1;                         \
    }                                   \
  } while (0)

#define VLNCE_CHECK_LAUNCH(name)                                              \
  do {                                                                        \
    hipError_t e__ = hipGetLastError();                                       \
    if (e__ != hipSuccess) {                                                  \
      vlnce_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
      return 2;                                                               \
    }                                                                         \
  } while (0)

static inline int ceil_div(long a, long b) { return (int)((a + b - 1) / b); }

__device__ __forceinline__ float apply_act(float v, int act) {
  if (act == VLNCE_ACT_RELU) return v > 0.f ? v : 0.f;
  if (act == VLNCE_ACT_SIGMOID) return 1.f / (1.f + __expf(-v));
  if (act == VLNCE_ACT_TANH) return tanhf(v);
  return v;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
