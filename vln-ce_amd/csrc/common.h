// Internal helpers shared by the HIP translation units of libvlnce_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "common_opts.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));


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

// Zero-fill as a KERNEL, never hipMemset*Async: inside a captured HIP graph the runtime's memset
// nodes were observed to race with the neighbouring kernels when the same buffer is zeroed and
// accumulated into repeatedly (a T-step rollout re-uses one split-K output ~200 times per
// graph: run-to-run different results, eager correct).  A kernel node orders like any other.
template <int UNUSED = 0>
__global__ __launch_bounds__(256) void vlnce_zero_kernel(float* __restrict__ p, long ld, int cols,
                                                         long total) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long r = i / cols;
    p[r * ld + (i - r * cols)] = 0.f;
  }
}
static inline void vlnce_zero(float* p, long rows, int cols, long ld, hipStream_t s) {
  const long total = rows * cols;
  long g = (total + 255) / 256;
  if (g > 2048) g = 2048;
  if (g < 1) g = 1;
  hipLaunchKernelGGL(vlnce_zero_kernel<0>, dim3((unsigned)g), dim3(256), 0, s, p, ld, cols, total);
}

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
