// Definitions shared by the convolution / GEMM translation units (igemm.hip, conv_p3.hip).
#pragma once
#include "common.h"
#include <stdlib.h>
#include <type_traits>

namespace vlnce_detail {

constexpr int BK = 32;
constexpr int LDP = 36;  // LDS row pitch in floats (32 + 4 pad)

enum { A_IM2COL_V4 = 0, A_IM2COL_S = 1, A_TRANS = 2, A_BUF = 3 };
enum { B_NK_V4 = 0, B_NK_S = 1, B_KN = 2, B_BUF = 3, B_IM2COL = 4 };

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int BUF_OOB = (int)0x80000000;  // voffset beyond any buffer: the load returns zeros

struct IgemmParams {
  const float* A;
  const float* B;
  float* C;
  int M, N, K;
  int H, W, Cin, KH, KW, stride, pad, Ho, Wo;  // im2col geometry (plain GEMM: 1x1 "image" row)
  int lda, ldb, ldc;
  const float* in_scale;
  const float* in_shift;
  const float* in_center;  // optional: x' = (x - center) * scale + shift
  int in_relu;
  // dual-input prologue (1x1 convolutions through the buffer loaders only):
  //   x' = act((A - center)*scale + shift + ((A2 - center2)*scale2 + shift2  |  A2))
  // and, when side_out is set, the n-tile-0 workgroups store x' to side_out[m, 0..K)
  const float* A2;
  const float* in2_scale;
  const float* in2_shift;
  const float* in2_center;
  float* side_out;
  const void* Bsplit;  // conv_x3_kernel: the weights as three bf16 planes [3][N*K]
  const void* Bfrag;   // conv_p3_kernel: the weights as MFMA B fragments (vlnce_conv2d_pack_weights)
  int p3_rows;         // conv_p3_kernel: patch rows allocated per LDS buffer (multiple of 32)
  const float* scale;
  const float* shift;
  const float* residual;
  int ldr;
  int act;
  int accumulate;
  float* stat_partial;
  int tiles_m, tiles_n;
  int splitk;  // > 1: blockIdx.y owns a K range and atomically adds into a pre-zeroed C
  int stat_rows;  // rows per statistics partial (vlnce_conv2d_tile_rows)
  long a_bytes, b_bytes, c_bytes;  // extents of A / B / C for the buffer descriptors
};

__device__ __forceinline__ f32x4 ldg4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }

// BatchNorm / GroupNorm partial statistics of one wave's accumulator sub-tile (rows x NT*32
// columns).  Lane (half, l31) holds, per 32x32 MFMA tile, column l31 and rows
// (r&3) + 8*(r>>2) + 4*half.  Two passes over the registers: column sums -> sub-tile mean ->
// sum of squared deviations (Chan/Welford form, merged later in fp64).
template <int MT, int NT>
__device__ __forceinline__ void wave_stats(const f32x16 (&acc)[MT][NT], float* stat_partial,
                                           int part_row, int rows_left, int rows_full, int col0,
                                           int N, int half, int l31) {
  const int rows_valid = min(rows_full, rows_left);
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (row < rows_valid) s += acc[i][j][r];
      }
    s += __shfl_xor(s, 32, 64);
    const float mean = rows_valid > 0 ? s / (float)rows_valid : 0.f;
    float m2 = 0.f;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        const float d = acc[i][j][r] - mean;
        if (row < rows_valid) m2 += d * d;
      }
    m2 += __shfl_xor(m2, 32, 64);
    const int col = col0 + j * 32 + l31;
    if (half == 0 && col < N && rows_left > 0) {
      float* dst = stat_partial + ((long)part_row * N + col) * 2;
      dst[0] = s;
      dst[1] = m2;
    }
  }
}

// per-wave partials at a granularity of `rows` = 16 or 32 pixels (GroupNorm over samples of 16 /
// 32 / ... pixels: partial tiles must not straddle samples).  Block b of 16 rows lives in MFMA
// tile i = b / 2, accumulator registers [8 * (b % 2), +8) of both half-waves.
template <int MT, int NT>
__device__ __forceinline__ void wave_stats_fine(const f32x16 (&acc)[MT][NT], float* stat_partial,
                                                int rows, int row0, int M, int col0, int N,
                                                int half, int l31) {
  const int nblk = MT * 32 / rows;
  for (int b = 0; b < nblk; ++b) {
    const int r_first = b * rows;                  // first row of the block inside the wave tile
    const int left = M - (row0 + r_first);
    const int valid = min(rows, left);
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
          if (row >= r_first && row < r_first + valid) s += acc[i][j][r];
        }
      s += __shfl_xor(s, 32, 64);
      const float mean = valid > 0 ? s / (float)valid : 0.f;
      float m2 = 0.f;
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
          const float d = acc[i][j][r] - mean;
          if (row >= r_first && row < r_first + valid) m2 += d * d;
        }
      m2 += __shfl_xor(m2, 32, 64);
      const int col = col0 + j * 32 + l31;
      if (half == 0 && col < N && left > 0) {
        float* dst = stat_partial + ((long)((row0 + r_first) / rows) * N + col) * 2;
        dst[0] = s;
        dst[1] = m2;
      }
    }
  }
}

// the same for ONE 32-row MFMA block (NT 32x32 tiles side by side)
template <int NT>
__device__ __forceinline__ void wave_stats_block(const f32x16 (&acc)[NT], float* stat_partial,
                                                 int part_row, int rows_left, int col0, int N,
                                                 int half, int l31) {
  const int rows_valid = min(32, rows_left);
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r)
      if ((r & 3) + 8 * (r >> 2) + 4 * half < rows_valid) s += acc[j][r];
    s += __shfl_xor(s, 32, 64);
    const float mean = rows_valid > 0 ? s / (float)rows_valid : 0.f;
    float m2 = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float d = acc[j][r] - mean;
      if ((r & 3) + 8 * (r >> 2) + 4 * half < rows_valid) m2 += d * d;
    }
    m2 += __shfl_xor(m2, 32, 64);
    const int col = col0 + j * 32 + l31;
    if (half == 0 && col < N && rows_left > 0) {
      float* dst = stat_partial + ((long)part_row * N + col) * 2;
      dst[0] = s;
      dst[1] = m2;
    }
  }
}

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ int x3_peek(const int* flag) {
  return __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void x3_wait(const int* flag, int seen, int need) {
  while (seen < need) {
    __builtin_amdgcn_s_sleep(1);
    seen = x3_peek(flag);
  }
  asm volatile("" ::: "memory");
}
__device__ __forceinline__ void x3_signal(int* flag) {
  // one lane's ds_add_u32 (the caller masks to lane 0); written out because the compiler's
  // atomic optimiser wraps a wave-uniform add in a ballot / mbcnt sequence
  typedef __attribute__((address_space(3))) int lds_int;
  const unsigned addr = (unsigned)(__UINTPTR_TYPE__)(lds_int*)flag;
  asm volatile("ds_add_u32 %0, %1" ::"v"(addr), "v"(1) : "memory");
}

// CUs of the device, rounded down to a multiple of 8 (one persistent workgroup per CU)
static inline int x3_cus() {
  static const int cus = [] {
    int dev = 0, n = 256;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
      n = 256;
    return n >= 8 ? (n / 8) * 8 : 8;
  }();
  return cus;
}

// convolution arithmetic: 1 = fp32 operands split into three bf16 planes, six products on the
// bf16 matrix pipe (conv_p3_kernel / conv_x3_kernel; fp32-class result, see the kernels'
// headers); 0 = v_mfma_f32_32x32x2_f32 everywhere (igemm_kernel).  Option "conv_math".
static inline int conv_math() { return vlnce_opt(VLNCE_OPT_CONV_MATH) != 0; }

// conv_p3.hip: the patch-resident bf16-plane convolution.  Returns -1 when the problem is not
// one it covers (the caller falls through to conv_x3_kernel / igemm_kernel), else a C-ABI status.
int p3_try_launch(const IgemmParams& p, hipStream_t stream);

}  // namespace vlnce_detail
