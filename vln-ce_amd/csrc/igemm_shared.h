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
  int math;            // plane format of Bsplit / Bfrag and arithmetic of the plane kernels (MATH_*)
  int p3_rows;         // conv_p3_kernel: patch rows allocated per LDS buffer (multiple of 32)
  const float* scale;
  const float* shift;
  const float* residual;
  int ldr;
  int act;
  int accumulate;
  float* stat_partial;
  vlnce_bn_sums bn;  // bn.acc != nullptr: the launch adds its BatchNorm column sums to bn.acc
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
                                           int N, int half, int l31, float post = 1.f) {
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
      dst[0] = s * post;
      dst[1] = m2 * post * post;
    }
  }
}

// per-wave partials at a granularity of `rows` = 16 or 32 pixels (GroupNorm over samples of 16 /
// 32 / ... pixels: partial tiles must not straddle samples).  Block b of 16 rows lives in MFMA
// tile i = b / 2, accumulator registers [8 * (b % 2), +8) of both half-waves.
template <int MT, int NT>
__device__ __forceinline__ void wave_stats_fine(const f32x16 (&acc)[MT][NT], float* stat_partial,
                                                int rows, int row0, int M, int col0, int N,
                                                int half, int l31, float post = 1.f) {
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
        dst[0] = s * post;
        dst[1] = m2 * post * post;
      }
    }
  }
}

// the same for ONE 32-row MFMA block (NT 32x32 tiles side by side)
template <int NT>
__device__ __forceinline__ void wave_stats_block(const f32x16 (&acc)[NT], float* stat_partial,
                                                 int part_row, int rows_left, int col0, int N,
                                                 int half, int l31, float post = 1.f) {
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
      dst[0] = s * post;
      dst[1] = m2 * post * post;
    }
  }
}

// ---- train-mode BatchNorm statistics added by the convolution (vlnce_bn_sums) -----------------
// A wave's running {sum x, sum x^2} of its NT x 32 output columns over the tiles it has finished
// for one column tile; flushed to IgemmParams::bn.acc with fp64 atomics when the column tile
// changes and at the end of the launch.  Every lane of a half-wave pair holds the same values
// (the block sums are completed with a shuffle across the halves); half 0 flushes.
template <int NT>
struct WaveBn {
  double s[NT], q[NT];
  int col0;   // first column of the wave's current column block, -1 = nothing accumulated
};
template <int NT>
__device__ __forceinline__ void wave_bn_reset(WaveBn<NT>& w) {
#pragma unroll
  for (int j = 0; j < NT; ++j) w.s[j] = w.q[j] = 0.0;
  w.col0 = -1;
}
// The sums are kept in VLNCE_BN_SHARDS copies, a workgroup adds to copy blockIdx.x % SHARDS: with
// one copy the 256 workgroups of a launch queue up on each address (~12 ns per device-scope
// atomic: 3-5 us at the very end of the kernel, where nothing hides it).
template <int NT>
__device__ __forceinline__ void wave_bn_flush(WaveBn<NT>& w, double* acc, int N, int half, int l31) {
  acc += (long)(blockIdx.x % VLNCE_BN_SHARDS) * N * 2;
  if (w.col0 >= 0 && half == 0) {
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int col = w.col0 + j * 32 + l31;
      if (col < N) {
        unsafeAtomicAdd(acc + 2 * col, w.s[j]);
        unsafeAtomicAdd(acc + 2 * col + 1, w.q[j]);
      }
    }
  }
  wave_bn_reset(w);
}
// one 32-row MFMA block (NT 32x32 tiles side by side) of raw accumulators: rows_left = rows of it
// that exist (<= 0: none)
template <int NT>
__device__ __forceinline__ void wave_bn_block(const f32x16 (&acc)[NT], WaveBn<NT>& w, int rows_left,
                                              int half, float post = 1.f) {
  const int rows_valid = min(32, rows_left);
  if (rows_valid <= 0) return;   // (wave-uniform)
  const double inv_n = 1.0 / (double)rows_valid;
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r)
      if ((r & 3) + 8 * (r >> 2) + 4 * half < rows_valid) s += acc[j][r];
    s += __shfl_xor(s, 32, 64);
    const float mean = s / (float)rows_valid;
    float m2 = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float d = acc[j][r] - mean;
      if ((r & 3) + 8 * (r >> 2) + 4 * half < rows_valid) m2 += d * d;
    }
    m2 += __shfl_xor(m2, 32, 64);
    // (post: the power-of-two scale the accumulators carry -- MATH_F16X3 -- exact in fp64)
    w.s[j] += (double)s * (double)post;
    w.q[j] += ((double)m2 + (double)s * (double)s * inv_n) * ((double)post * (double)post);
  }
}
// a wave's MT x NT blocks of one finished tile: col0 = its first column, rows_left = rows of the
// wave's sub-tile that exist
template <int MT, int NT>
__device__ __forceinline__ void wave_bn_tile(const f32x16 (&acc)[MT][NT], WaveBn<NT>& w, double* gacc,
                                             int col0, int N, int rows_left, int half, int l31,
                                             float post = 1.f) {
  if (w.col0 != col0) {
    wave_bn_flush(w, gacc, N, half, l31);
    w.col0 = col0;
  }
#pragma unroll
  for (int i = 0; i < MT; ++i) wave_bn_block<NT>(acc[i], w, rows_left - i * 32, half, post);
}
// (Measured and dropped, round 4: finishing the statistics INSIDE the convolution -- ticket per
// workgroup, the last one reads the sums back with atomic exchanges and writes the vectors -- costs
// the launch three dependent device-scope round trips (atomics acknowledged, ticket returned,
// exchanges returned: +12 us per launch, as much as the separate finalize launch it removed;
// with a __threadfence() in front of the ticket +40 us: buffer_wbl2 / buffer_inv by every wave
// behind a kernel that has just written up to 268 MB).  The convolution only ADDS; a one-workgroup
// kernel behind it turns the sums into the vectors: profiles/archive/r04_y_*.)

// Register epilogue of the bf16-plane kernels for a wave's MT x NT blocks of 32 x 32 outputs:
// y = act(acc * scale[col] + shift[col] [+ residual[row, col]]), one store = 2 rows x 32 columns.
// e_voff[j]: lane byte offset of (first row of the wave tile + 4 * half, column of block j), or
// BUF_OOB; rows_left: rows of the wave tile that exist, counted from row 4 * half.
// The residual (eval-mode block ends: relu(bn(conv) + identity), torchvision BasicBlock /
// Bottleneck.forward) has the raster of the output (ldr == ldc is a dispatch condition), so its
// loads use the stores' offsets; they are fetched per 32-row block.
template <int MT, int NT>
__device__ __forceinline__ void wave_epilogue(f32x16 (&acc)[MT][NT], const float (&e_sc)[NT],
                                              const float (&e_sh)[NT], const int (&e_voff)[NT],
                                              int rows_left, int ldc, int act, bool has_res,
                                              __amdgpu_buffer_rsrc_t rsrc_c,
                                              __amdgpu_buffer_rsrc_t rsrc_r, bool clear) {
  if (has_res) {
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      float rv[NT][16];   // (one block at a time: a second set spills in the 128-register kernels)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rw = i * 32 + (r & 3) + 8 * (r >> 2);
#pragma unroll
        for (int j = 0; j < NT; ++j)
          rv[j][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                         rsrc_r, rw < rows_left ? e_voff[j] : BUF_OOB, rw * ldc * 4, 0));
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rw = i * 32 + (r & 3) + 8 * (r >> 2);
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const float v = apply_act(acc[i][j][r] * e_sc[j] + e_sh[j] + rv[j][r], act);
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsrc_c,
                                                rw < rows_left ? e_voff[j] : BUF_OOB, rw * ldc * 4, 0);
          if (clear) acc[i][j][r] = 0.f;
        }
      }
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int rw = i * 32 + (r & 3) + 8 * (r >> 2);
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const float v = apply_act(acc[i][j][r] * e_sc[j] + e_sh[j], act);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsrc_c,
                                              rw < rows_left ? e_voff[j] : BUF_OOB, rw * ldc * 4, 0);
        if (clear) acc[i][j][r] = 0.f;
      }
    }
}

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// ---- arithmetic of the plane kernels (option "conv_math", vlnce_prologue.w_format) -------------
// MATH_BF16X6: every fp32 operand = three bf16 planes x1 + x2 + x3 (exact, 8 + 8 + 8 mantissa bits,
//   fp32's exponent range); a product = six plane products on v_mfma_f32_32x32x16_bf16.
// MATH_F16X3 (round 6): a = a1 + a2, a1 = fp16(a), a2 = fp16((a - a1) * 2^11) / 2^11 (11 + 11
//   mantissa bits + the sign of a2: |a - a1 - a2| <= 2^-22 |a|); a product = THREE plane products
//   a1 b1 + a1 b2 + a2 b1 on v_mfma_f32_32x32x16_f16 (the dropped a2 b2 <= 2^-22 |a b|), all three
//   accumulated at the common scale 2^11 -- A planes {a1, a2 * 2^11}, B planes {b1 * 2^11,
//   b2 * 2^11, b1} -- so that the low planes stay out of fp16's subnormal range, and the
//   accumulator is multiplied by 2^-11 (exact) where it leaves the registers.  Against an fp64
//   convolution this is as close as the six-product form (the fp32 accumulation dominates both:
//   half the accumulator updates), at half the matrix-pipe time and two thirds of the operand
//   bytes.  Range: |a| < 65504, |b| < 32 (outside: inf / NaN in the output, never a wrong finite
//   value); operands far below fp16's normal range (gradients) lose relative precision -- the
//   data-gradient launches of the trainable encoders stay on MATH_BF16X6.
enum { MATH_F32 = 0, MATH_BF16X6 = 1, MATH_F16X3 = 2 };
template <int MATH>
struct Planes;
template <>
struct Planes<MATH_BF16X6> {
  static constexpr int NA = 3, NB = 3, NP = 6;
  static constexpr int ROW = 208;   // bytes of a patch row: NA planes x 32 x 2 B + 16 pad (13 x 16 B)
  static constexpr float POST = 1.f;
  static constexpr int PA[6] = {2, 1, 0, 1, 0, 0};   // smallest products first
  static constexpr int PB[6] = {0, 1, 2, 0, 1, 0};
};
template <>
struct Planes<MATH_F16X3> {
  static constexpr int NA = 2, NB = 3, NP = 3;
  static constexpr int ROW = 144;   // 9 x 16 B: consecutive rows are conflict-free as well
  static constexpr float POST = 1.f / 2048.f;
  static constexpr int PA[6] = {1, 0, 0, 0, 0, 0};   // a2 b1, a1 b2, a1 b1 (at scale 2^11)
  static constexpr int PB[6] = {2, 1, 0, 0, 0, 0};
};
template <int MATH>
__device__ __forceinline__ f32x16 plane_mfma(bf16x8 a, bf16x8 b, f32x16 c) {
  if constexpr (MATH == MATH_F16X3)
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a),
                                                  __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
// two consecutive fp32 values -> one 32-bit word (two 16-bit plane values) per A plane
template <int MATH>
__device__ __forceinline__ void split_pair(float v0, float v1, unsigned (&w)[Planes<MATH>::NA]) {
  typedef float f32x2_ __attribute__((ext_vector_type(2)));
  if constexpr (MATH == MATH_F16X3) {
    typedef _Float16 f16x2_ __attribute__((ext_vector_type(2)));
    const f32x2_ v = {v0, v1};
    const f16x2_ h = __builtin_convertvector(v, f16x2_);   // round to nearest even
    w[0] = __builtin_bit_cast(unsigned, h);
    const f32x2_ r = {(v0 - (float)h[0]) * 2048.f, (v1 - (float)h[1]) * 2048.f};   // exact
    w[1] = __builtin_bit_cast(unsigned, __builtin_convertvector(r, f16x2_));
  } else {
    typedef __bf16 bf16x2_ __attribute__((ext_vector_type(2)));
    f32x2_ v = {v0, v1};
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const unsigned hb = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_));
      w[q] = hb;
      if (q < 2) {
        v[0] -= __builtin_bit_cast(float, hb << 16);
        v[1] -= __builtin_bit_cast(float, hb & 0xffff0000u);
      }
    }
  }
}
// one weight -> its three B-plane values (16-bit words)
template <int MATH>
__device__ __forceinline__ void split_weight(float v, unsigned short (&o)[3]) {
  if constexpr (MATH == MATH_F16X3) {
    const _Float16 h = (_Float16)v;
    o[0] = __builtin_bit_cast(unsigned short, (_Float16)((float)h * 2048.f));   // exact unless |v| >= 32
    o[1] = __builtin_bit_cast(unsigned short, (_Float16)((v - (float)h) * 2048.f));
    o[2] = __builtin_bit_cast(unsigned short, h);
  } else {
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const __bf16 hb = (__bf16)v;   // round to nearest even
      o[q] = __builtin_bit_cast(unsigned short, hb);
      v -= (float)hb;
    }
  }
}

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

// dynamic LDS a workgroup of this device may ask for (163840 on gfx950: one workgroup per CU)
static inline int x3_lds_max() {
  static const int v = [] {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&n, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) != hipSuccess || n <= 0)
      n = 65536;
    return n;
  }();
  return v;
}

// convolution arithmetic, option "conv_math": 0 = v_mfma_f32_32x32x2_f32 everywhere (igemm_kernel);
// non-zero = the plane kernels (conv_p3 / u3 / s3 / m3 / x3) in the arithmetic of the planes the
// launch hands over (vlnce_prologue.w_format -> IgemmParams::math: MATH_BF16X6 or MATH_F16X3).
static inline int conv_math() { return vlnce_opt(VLNCE_OPT_CONV_MATH) != 0; }

// conv_p3.hip: the patch-resident bf16-plane convolution.  Returns -1 when the problem is not
// one it covers (the caller falls through to conv_x3_kernel / igemm_kernel), else a C-ABI status.
int p3_try_launch(const IgemmParams& p, hipStream_t stream);
int m3_try_launch(const IgemmParams& p, hipStream_t stream);   // conv_m3.hip
// wgrad_x6.hip: the weight gradient on the 16-bit pipe (three bf16 planes); -1 = not covered
int wgrad_x6_try_launch(const float* x, const float* dy, float* dw, const vlnce_conv_desc* d,
                        const float* dy_up, const float* dy_down, int accumulate,
                        hipStream_t stream);

}  // namespace vlnce_detail
