// fp32 implicit-GEMM convolution / general GEMM on v_mfma_f32_32x32x2_f32.
//
//   C[M,N] = epilogue( A[M,K] * B[K,N] )
//
// A is gathered on the fly (im2col of a channels-last image, or a plain /
// transposed matrix), B is an nn.Linear-style [N,K] weight or a [K,N] matrix.
// Design (MI355X_MICROARCH / cdna_hip_programming guides):
//   * 256 threads = 4 wave64; block tile BM x BN, K-tile 32; each wave owns a
//     (BM/WM) x (BN/WN) sub-tile made of 32x32 MFMA tiles (16 acc VGPRs each).
//   * operands go global -> registers -> LDS (register staging: the im2col
//     gather needs zero-fill + an optional per-channel prologue, which LDS-DMA
//     cannot do), double-buffered LDS, ONE barrier per K-tile; the global loads
//     of tile t+1 are issued before the MFMAs of tile t and written to LDS after.
//   * LDS rows are K-contiguous with a 36-float pitch: every lane fetches its
//     MFMA operands for four k-steps with one conflict-free ds_read_b128.
//     The k order inside a group of 8 is permuted identically for A and B
//     (lanes 0-31 take k 0..3, lanes 32-63 take k 4..7), which a dot product
//     does not care about.
//   * exact fp32: the MFMA is a k-ordered fmaf chain, no reduced precision.
//   * blockIdx -> tile map is XCD-aware: each XCD (private L2) gets a
//     contiguous run of tiles with the N-tile index fastest, so blocks that
//     share an A row-panel hit the same L2.
#include "common.h"
#include <stdlib.h>
#include <type_traits>

namespace {

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
  const float* scale;
  const float* shift;
  const float* residual;
  int ldr;
  int act;
  int accumulate;
  float* stat_partial;
  int tiles_m, tiles_n;
  int splitk;  // > 1: blockIdx.y owns a K range and atomically adds into a pre-zeroed C
  int pk_tiles;  // conv_dma_kernel: tiles one workgroup walks before it retires
  int stat_rows;  // conv_dma_kernel: rows per statistics partial (vlnce_conv2d_tile_rows)
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

// CIN_C / KW_C: compile-time Cin and KW for the scalar im2col loader (0 = runtime values);
// the 7x7 stems (Cin 3 / 1) use them so k -> (r, q, ci) is multiply-shift, not a division.
template <int BM, int BN, int WM, int WN, int AMODE, int BMODE, int CIN_C = 0, int KW_C = 0,
          int DUAL = 0>
__global__ __launch_bounds__(256, 2) void igemm_kernel(IgemmParams p) {
  static_assert(!DUAL || AMODE == A_BUF, "dual-input prologue: buffer-loader path only");
  constexpr int WTM = BM / WM;  // rows per wave
  constexpr int WTN = BN / WN;
  constexpr int MT = WTM / 32;
  constexpr int NT = WTN / 32;
  constexpr int A_TILE = BM * LDP;
  constexpr int B_TILE = BN * LDP;
  constexpr int STAGE = A_TILE + B_TILE;
  static_assert(WM * WN == 4, "4 waves");
  static_assert(MT >= 1 && NT >= 1, "tile");

  extern __shared__ __attribute__((aligned(16))) float smem[];

  // ---- XCD-aware tile mapping (bijective for any grid size)
  int tile;
  {
    const int bid = blockIdx.x, nwg = gridDim.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tile_m = tile / p.tiles_n;
  const int tile_n = tile - tile_m * p.tiles_n;
  const int m0 = tile_m * BM;
  const int n0 = tile_n * BN;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WN;
  const int wn = wave - wm * WN;
  const int half = lane >> 5;
  const int l31 = lane & 31;

  // K-tile range of this workgroup (split-K along blockIdx.y)
  const int KT_all = (p.K + BK - 1) / BK;
  const int kt_per = (KT_all + p.splitk - 1) / p.splitk;
  const int kt0 = blockIdx.y * kt_per;
  const int kt1 = min(KT_all, kt0 + kt_per);
  if (kt0 >= kt1) return;

  // ------------------------------------------------------------------ A loader state
  constexpr int A_ROWS = BM / 32;  // row-major modes: 32 rows x 8 float4 per pass
  const int lrow = tid >> 3;
  const int lk4 = (tid & 7) * 4;
  int a_off[A_ROWS];
  int a_hw[A_ROWS];
  // current (r, q, ci) of this thread's first k in the K-tile (im2col v4 mode)
  int k_r = 0, k_q = 0, k_ci = 0;
  if constexpr (AMODE == A_IM2COL_V4 || AMODE == A_IM2COL_S) {
    const int HoWo = p.Ho * p.Wo;
#pragma unroll
    for (int i = 0; i < A_ROWS; ++i) {
      const int m = m0 + i * 32 + lrow;
      if (m < p.M) {
        const int img = m / HoWo;
        const int rem = m - img * HoWo;
        const int ho = rem / p.Wo;
        const int wo = rem - ho * p.Wo;
        const int hi0 = ho * p.stride - p.pad;
        const int wi0 = wo * p.stride - p.pad;
        a_off[i] = (img * p.H + hi0) * p.W + wi0;
        a_hw[i] = (hi0 << 16) | (wi0 & 0xffff);
      } else {
        a_off[i] = 0;
        a_hw[i] = (int)0x80008000;  // hi0 = wi0 = -32768: never valid
      }
    }
    if constexpr (AMODE == A_IM2COL_V4) {
      const int kfirst = kt0 * BK + lk4;
      const int tap = kfirst / p.Cin;
      k_ci = kfirst - tap * p.Cin;
      k_r = tap / p.KW;
      k_q = tap - k_r * p.KW;
    }
  }

  f32x4 a_reg[A_ROWS];
  constexpr int B_ROWS = BN / 32;
  f32x4 b_reg[B_ROWS];

  // ---- buffer-descriptor loaders (A_BUF / B_BUF): the hot path.  Per K-tile the only
  // per-lane work is one bit test + select per row: the tap offset is a wave-uniform SGPR
  // (soffset), rows outside the image / matrix get an out-of-range voffset and the hardware
  // bounds check returns zeros (no exec-mask branches, no 64-bit address math).
  __amdgpu_buffer_rsrc_t rsrc_a, rsrc_b, rsrc_a2;
  f32x4 a2_reg[DUAL ? A_ROWS : 1];
  f32x4 pro2_s = {1.f, 1.f, 1.f, 1.f}, pro2_t = {0.f, 0.f, 0.f, 0.f}, pro2_c = {0.f, 0.f, 0.f, 0.f};
  int a_kcur = 0;  // K offset (= input channel of a 1x1 conv) of the staged tile
  int a_voff[A_ROWS];
  unsigned a_taps[A_ROWS];  // bit t: filter tap t of this output pixel reads inside the image
  int b_voff[B_ROWS];
  int u_r = 0, u_q = 0, u_ci = 0;  // wave-uniform tap state of the next tile to fetch
  if constexpr (AMODE == A_BUF) {
    const long bias = ((long)p.pad * p.W + p.pad) * p.lda * 4;  // keeps voffsets non-negative
    rsrc_a = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(p.A)) - bias, 0, (int)(p.a_bytes + bias),
        0x00020000);
    if constexpr (DUAL)
      rsrc_a2 = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<char*>(reinterpret_cast<const char*>(p.A2)) - bias, 0,
          (int)(p.a_bytes + bias), 0x00020000);
    const int HoWo = p.Ho * p.Wo;
#pragma unroll
    for (int i = 0; i < A_ROWS; ++i) {
      const int m = m0 + i * 32 + lrow;
      a_voff[i] = BUF_OOB;
      a_taps[i] = 0;
      if (m < p.M) {
        const int img = m / HoWo;
        const int rem = m - img * HoWo;
        const int ho = rem / p.Wo;
        const int wo = rem - ho * p.Wo;
        const int hi0 = ho * p.stride - p.pad, wi0 = wo * p.stride - p.pad;
        a_voff[i] = (int)((((long)(img * p.H + hi0 + p.pad) * p.W + wi0 + p.pad) * p.lda + lk4) * 4);
        unsigned mask = 0;
        for (int r = 0; r < p.KH; ++r)
          for (int q = 0; q < p.KW; ++q)
            if ((unsigned)(hi0 + r) < (unsigned)p.H && (unsigned)(wi0 + q) < (unsigned)p.W)
              mask |= 1u << (r * p.KW + q);
        a_taps[i] = mask;
      }
    }
    const int kfirst = kt0 * BK;
    const int tap = kfirst / p.Cin;
    u_ci = kfirst - tap * p.Cin;
    u_r = tap / p.KW;
    u_q = tap - u_r * p.KW;
  }
  if constexpr (BMODE == B_BUF) {
    rsrc_b = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(p.B)), 0, (int)p.b_bytes, 0x00020000);
#pragma unroll
    for (int i = 0; i < B_ROWS; ++i) {
      const int n = n0 + i * 32 + lrow;
      b_voff[i] = n < p.N ? (int)(((long)n * p.ldb + lk4) * 4) : BUF_OOB;
    }
  }

  // ---- weight-gradient GEMM (B_IM2COL): C[Cout, K] = dY^T[Cout, M] * im2col(X)[M, K].  The
  // reduction index is the output pixel m; this block's N range [n0, n0+BN) of K = (r, q, ci)
  // lies inside one filter tap when Cin % BN == 0 (vector path), else elements are decoded
  // one by one (7x7 stems).  Each thread keeps the (img, ho, wo) of its rows and advances
  // them by 32 pixels per K-tile.
  constexpr int WG_TPR = BN / 4;          // threads per reduction row
  constexpr int WG_KPP = 256 / WG_TPR;    // reduction rows per pass
  int wg_img[B_ROWS], wg_ho[B_ROWS], wg_wo[B_ROWS];
  if constexpr (BMODE == B_IM2COL) {
    const int HoWo = p.Ho * p.Wo;
#pragma unroll
    for (int i = 0; i < B_ROWS; ++i) {
      const int m = kt0 * BK + i * WG_KPP + tid / WG_TPR;
      wg_img[i] = m / HoWo;
      const int rem = m - wg_img[i] * HoWo;
      wg_ho[i] = rem / p.Wo;
      wg_wo[i] = rem - wg_ho[i] * p.Wo;
    }
  }

  // The operand transform (x*s+t, ReLU) is applied when the staged registers are written to
  // LDS, i.e. AFTER the MFMAs of the current tile: applying it right after the loads would
  // put the global-load latency in front of the MFMAs instead of behind them.
  f32x4 pro_s, pro_t;       // V4 mode: scale/shift of this thread's 4 channels (current stage)
  f32x4 pro_c = {0.f, 0.f, 0.f, 0.f};
  float pro_ec[4] = {0.f, 0.f, 0.f, 0.f};
  unsigned a_okmask = 0;    // bit i (V4) / bit 4*i+e (scalar): element is inside the image
  float pro_es[4], pro_et[4];

  auto load_a = [&](int k0) {
    if constexpr (AMODE == A_BUF) {
      const int tap = u_r * p.KW + u_q;
      const int soff = ((u_r * p.W + u_q) * p.lda + u_ci) * 4;
      if (p.in_scale != nullptr) {
        pro_s = ldg4(p.in_scale + u_ci + lk4);
        pro_t = ldg4(p.in_shift + u_ci + lk4);
        if (p.in_center) pro_c = ldg4(p.in_center + u_ci + lk4);
      }
      if constexpr (DUAL) {
        a_kcur = u_ci;
        if (p.in2_scale != nullptr) {
          pro2_s = ldg4(p.in2_scale + u_ci + lk4);
          pro2_t = ldg4(p.in2_shift + u_ci + lk4);
          if (p.in2_center) pro2_c = ldg4(p.in2_center + u_ci + lk4);
        }
      }
      a_okmask = 0;
#pragma unroll
      for (int i = 0; i < A_ROWS; ++i) {
        const unsigned ok = (a_taps[i] >> tap) & 1u;
        a_okmask |= ok << i;
        a_reg[i] = __builtin_bit_cast(
            f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, ok ? a_voff[i] : BUF_OOB, soff, 0));
        if constexpr (DUAL)
          a2_reg[i] = __builtin_bit_cast(
              f32x4,
              __builtin_amdgcn_raw_buffer_load_b128(rsrc_a2, ok ? a_voff[i] : BUF_OOB, soff, 0));
      }
      // advance the uniform tap state by one K-tile (Cin % 32 == 0: at most one wrap)
      u_ci += BK;
      if (u_ci >= p.Cin) {
        u_ci = 0;
        if (++u_q == p.KW) {
          u_q = 0;
          ++u_r;
        }
      }
    } else if constexpr (AMODE == A_IM2COL_V4) {
      const bool tap_ok = k_r < p.KH;
      if (p.in_scale != nullptr && tap_ok) {
        pro_s = ldg4(p.in_scale + k_ci);
        pro_t = ldg4(p.in_shift + k_ci);
        if (p.in_center) pro_c = ldg4(p.in_center + k_ci);
      }
      a_okmask = 0;
#pragma unroll
      for (int i = 0; i < A_ROWS; ++i) {
        const int hi = (a_hw[i] >> 16) + k_r;
        const int wi = (int)(short)(a_hw[i] & 0xffff) + k_q;
        const bool ok = tap_ok && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (ok) {
          const long pix = (long)a_off[i] + k_r * p.W + k_q;
          v = ldg4(p.A + pix * p.lda + k_ci);
          a_okmask |= 1u << i;
        }
        a_reg[i] = v;
      }
      // advance (r, q, ci) by one K-tile
      k_ci += BK;
      while (k_ci >= p.Cin) {
        k_ci -= p.Cin;
        if (++k_q == p.KW) {
          k_q = 0;
          ++k_r;
        }
      }
    } else if constexpr (AMODE == A_IM2COL_S) {
      int er[4], eq[4], eci[4];
      const int cin = CIN_C > 0 ? CIN_C : p.Cin;
      const int kw = KW_C > 0 ? KW_C : p.KW;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int k = k0 + lk4 + e;
        const int tap = k / cin;
        eci[e] = k - tap * cin;
        er[e] = tap / kw;
        eq[e] = tap - er[e] * kw;
        pro_es[e] = 1.f;
        pro_et[e] = 0.f;
        pro_ec[e] = 0.f;
        if (p.in_scale != nullptr && k < p.K) {
          pro_es[e] = p.in_scale[eci[e]];
          pro_et[e] = p.in_shift[eci[e]];
          if (p.in_center) pro_ec[e] = p.in_center[eci[e]];
        }
      }
      a_okmask = 0;
#pragma unroll
      for (int i = 0; i < A_ROWS; ++i) {
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int hi = (a_hw[i] >> 16) + er[e];
          const int wi = (int)(short)(a_hw[i] & 0xffff) + eq[e];
          const bool ok =
              er[e] < p.KH && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
          float x = 0.f;
          if (ok) {
            const long pix = (long)a_off[i] + er[e] * p.W + eq[e];
            x = p.A[pix * p.lda + eci[e]];
            a_okmask |= 1u << (4 * i + e);
          }
          v[e] = x;
        }
        a_reg[i] = f32x4{v[0], v[1], v[2], v[3]};
      }
    } else {  // A_TRANS: A[m][k] stored at A[k*lda + m]; tile read as BK x BM, m contiguous
      constexpr int TPR = BM / 4;        // threads per k-row
      constexpr int KPP = 256 / TPR;     // k rows per pass
      const int km = tid / TPR;
      const int m4 = (tid - km * TPR) * 4;
      const bool vec = ((p.lda & 3) == 0) && ((p.M & 3) == 0);
#pragma unroll
      for (int i = 0; i < A_ROWS; ++i) {
        const int k = k0 + i * KPP + km;
        const int m = m0 + m4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (k < p.K) {
          const float* src = p.A + (long)k * p.lda + m;
          if (vec && m + 3 < p.M) {
            v = ldg4(src);
          } else {
            if (m + 0 < p.M) v.x = src[0];
            if (m + 1 < p.M) v.y = src[1];
            if (m + 2 < p.M) v.z = src[2];
            if (m + 3 < p.M) v.w = src[3];
          }
        }
        a_reg[i] = v;
      }
    }
  };

  auto load_b = [&](int k0) {
    if constexpr (BMODE == B_BUF) {
#pragma unroll
      for (int i = 0; i < B_ROWS; ++i)
        b_reg[i] = __builtin_bit_cast(
            f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_b, b_voff[i], k0 * 4, 0));
    } else if constexpr (BMODE == B_NK_V4) {
      const int k = k0 + lk4;
#pragma unroll
      for (int i = 0; i < B_ROWS; ++i) {
        const int n = n0 + i * 32 + lrow;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (n < p.N && k < p.K) v = ldg4(p.B + (long)n * p.ldb + k);
        b_reg[i] = v;
      }
    } else if constexpr (BMODE == B_NK_S) {
      const int k = k0 + lk4;
#pragma unroll
      for (int i = 0; i < B_ROWS; ++i) {
        const int n = n0 + i * 32 + lrow;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (n < p.N) {
          const float* src = p.B + (long)n * p.ldb + k;
          if (k + 0 < p.K) v.x = src[0];
          if (k + 1 < p.K) v.y = src[1];
          if (k + 2 < p.K) v.z = src[2];
          if (k + 3 < p.K) v.w = src[3];
        }
        b_reg[i] = v;
      }
    } else if constexpr (BMODE == B_IM2COL) {
      const int n4 = (tid % WG_TPR) * 4;
      const bool vec = (p.Cin % BN) == 0;  // whole N tile inside one tap, 16-byte aligned
      const int n = n0 + n4;
      int tap = 0, ci = 0;
      if (vec) {
        tap = n0 / p.Cin;
        ci = n - tap * p.Cin;
      }
#pragma unroll
      for (int i = 0; i < B_ROWS; ++i) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        // p.K is the reduction length here (number of output pixels N*Ho*Wo)
        const bool row_ok = ((long)(wg_img[i] * p.Ho + wg_ho[i]) * p.Wo + wg_wo[i]) < (long)p.K;
        if (row_ok) {
          if (vec) {
            const int r = tap / p.KW, q = tap - r * p.KW;
            const int hi = wg_ho[i] * p.stride - p.pad + r;
            const int wi = wg_wo[i] * p.stride - p.pad + q;
            if (n < p.N && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W)
              v = ldg4(p.B + (((long)wg_img[i] * p.H + hi) * p.W + wi) * p.ldb + ci);
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int ne = n + e;
              if (ne < p.N) {
                const int tp = ne / p.Cin;
                const int ce = ne - tp * p.Cin;
                const int r = tp / p.KW, q = tp - r * p.KW;
                const int hi = wg_ho[i] * p.stride - p.pad + r;
                const int wi = wg_wo[i] * p.stride - p.pad + q;
                if ((unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W)
                  v[e] = p.B[(((long)wg_img[i] * p.H + hi) * p.W + wi) * p.ldb + ce];
              }
            }
          }
        }
        b_reg[i] = v;
        // advance this row by one K-tile (32 output pixels)
        wg_wo[i] += BK;
        while (wg_wo[i] >= p.Wo) {
          wg_wo[i] -= p.Wo;
          if (++wg_ho[i] == p.Ho) {
            wg_ho[i] = 0;
            ++wg_img[i];
          }
        }
      }
    } else {  // B_KN: B[k][n] at B[k*ldb + n]
      constexpr int TPR = BN / 4;
      constexpr int KPP = 256 / TPR;
      const int kn = tid / TPR;
      const int n4 = (tid - kn * TPR) * 4;
      const bool vec = ((p.ldb & 3) == 0) && ((p.N & 3) == 0);
#pragma unroll
      for (int i = 0; i < B_ROWS; ++i) {
        const int k = k0 + i * KPP + kn;
        const int n = n0 + n4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (k < p.K) {
          const float* src = p.B + (long)k * p.ldb + n;
          if (vec && n + 3 < p.N) {
            v = ldg4(src);
          } else {
            if (n + 0 < p.N) v.x = src[0];
            if (n + 1 < p.N) v.y = src[1];
            if (n + 2 < p.N) v.z = src[2];
            if (n + 3 < p.N) v.w = src[3];
          }
        }
        b_reg[i] = v;
      }
    }
  };

  auto store_ab = [&](float* stage) {
    float* As = stage;
    float* Bs = stage + A_TILE;
    if constexpr (AMODE == A_TRANS) {
      constexpr int TPR = BM / 4;
      constexpr int KPP = 256 / TPR;
      const int km = tid / TPR;
      const int m4 = (tid - km * TPR) * 4;
#pragma unroll
      for (int i = 0; i < A_ROWS; ++i) {
        const int kk = i * KPP + km;
        As[(m4 + 0) * LDP + kk] = a_reg[i].x;
        As[(m4 + 1) * LDP + kk] = a_reg[i].y;
        As[(m4 + 2) * LDP + kk] = a_reg[i].z;
        As[(m4 + 3) * LDP + kk] = a_reg[i].w;
      }
    } else {
      if (p.in_scale != nullptr) {
#pragma unroll
        for (int i = 0; i < A_ROWS; ++i) {
          f32x4 v = a_reg[i];
          if constexpr (AMODE == A_IM2COL_V4 || AMODE == A_BUF) {
            v = (v - pro_c) * pro_s + pro_t;
            if constexpr (DUAL) v += (a2_reg[i] - pro2_c) * pro2_s + pro2_t;
            if (p.in_relu) {
              v.x = fmaxf(v.x, 0.f);
              v.y = fmaxf(v.y, 0.f);
              v.z = fmaxf(v.z, 0.f);
              v.w = fmaxf(v.w, 0.f);
            }
            if (!((a_okmask >> i) & 1u)) v = f32x4{0.f, 0.f, 0.f, 0.f};
            if constexpr (DUAL) {
              // the materialised block output: written once, by the workgroups of n-tile 0
              if (p.side_out != nullptr && n0 == 0 && ((a_okmask >> i) & 1u))
                *reinterpret_cast<f32x4*>(p.side_out + (long)(m0 + i * 32 + lrow) * p.lda +
                                          a_kcur + lk4) = v;
            }
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              float x = (v[e] - pro_ec[e]) * pro_es[e] + pro_et[e];
              if (p.in_relu) x = fmaxf(x, 0.f);
              v[e] = ((a_okmask >> (4 * i + e)) & 1u) ? x : 0.f;
            }
          }
          a_reg[i] = v;
        }
      }
#pragma unroll
      for (int i = 0; i < A_ROWS; ++i)
        *reinterpret_cast<f32x4*>(As + (i * 32 + lrow) * LDP + lk4) = a_reg[i];
    }
    if constexpr (BMODE == B_KN || BMODE == B_IM2COL) {
      constexpr int TPR = BN / 4;
      constexpr int KPP = 256 / TPR;
      const int kn = tid / TPR;
      const int n4 = (tid - kn * TPR) * 4;
#pragma unroll
      for (int i = 0; i < B_ROWS; ++i) {
        const int kk = i * KPP + kn;
        Bs[(n4 + 0) * LDP + kk] = b_reg[i].x;
        Bs[(n4 + 1) * LDP + kk] = b_reg[i].y;
        Bs[(n4 + 2) * LDP + kk] = b_reg[i].z;
        Bs[(n4 + 3) * LDP + kk] = b_reg[i].w;
      }
    } else {
#pragma unroll
      for (int i = 0; i < B_ROWS; ++i)
        *reinterpret_cast<f32x4*>(Bs + (i * 32 + lrow) * LDP + lk4) = b_reg[i];
    }
  };

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int KT = kt1 - kt0;
  load_a(kt0 * BK);
  load_b(kt0 * BK);
  store_ab(smem);
  __syncthreads();

  for (int t = 0; t < KT; ++t) {
    float* cur = smem + (t & 1) * STAGE;
    const bool more = (t + 1) < KT;
    if (more) {
      load_a((kt0 + t + 1) * BK);
      load_b((kt0 + t + 1) * BK);
    }
    const float* Aw = cur + (wm * WTM + l31) * LDP + 4 * half;
    const float* Bw = cur + A_TILE + (wn * WTN + l31) * LDP + 4 * half;
#pragma unroll
    for (int g = 0; g < BK / 8; ++g) {
      f32x4 af[MT], bf[NT];
#pragma unroll
      for (int i = 0; i < MT; ++i) af[i] = *reinterpret_cast<const f32x4*>(Aw + i * 32 * LDP + 8 * g);
#pragma unroll
      for (int j = 0; j < NT; ++j) bf[j] = *reinterpret_cast<const f32x4*>(Bw + j * 32 * LDP + 8 * g);
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j)
#ifdef IGEMM_DBG_NOMFMA
            acc[i][j][0] += af[i][e] * bf[j][e];
#else
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][e], bf[j][e], acc[i][j], 0, 0, 0);
#endif
    }
    if (more) store_ab(smem + ((t + 1) & 1) * STAGE);
    __syncthreads();
  }

  // ------------------------------------------------------------------ BN statistics of the raw tile
  // one partial per WAVE sub-tile (WTM rows x WTN columns): {sum, M2 about the sub-tile mean},
  // index (tile_m * WM + wm).  No cross-wave reduction and no barrier here; the finalize
  // kernels merge the partials in fp64.
#ifdef IGEMM_DBG_NOSTATS  // bisection builds: -DIGEMM_DBG_NOSTATS / _NOSTORE / _NOMFMA
  if (false) {
#else
  if (p.stat_partial != nullptr) {
#endif
    if (p.stat_rows > 0 && p.stat_rows < WTM)
      wave_stats_fine<MT, NT>(acc, p.stat_partial, p.stat_rows, m0 + wm * WTM, p.M, n0 + wn * WTN,
                              p.N, half, l31);
    else
      wave_stats<MT, NT>(acc, p.stat_partial, tile_m * WM + wm, p.M - (m0 + wm * WTM), WTM,
                         n0 + wn * WTN, p.N, half, l31);
  }

  // ------------------------------------------------------------------ epilogue
#ifdef IGEMM_DBG_NOSTORE
  if (acc[0][0][0] != 123456.f) return;  // (keeps the accumulators alive)
#endif
  if (p.splitk > 1) {
    // partial sums of this K range: plain atomic accumulation (C was zeroed by the host entry)
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int col = n0 + wn * WTN + j * 32 + l31;
      if (col >= p.N) continue;
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = m0 + wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
          if (row < p.M) atomicAdd(p.C + (long)row * p.ldc + col, acc[i][j][r]);
        }
    }
    return;
  }
  const bool vec_out = ((p.N & 3) == 0) && ((p.ldc & 3) == 0) &&
                       ((reinterpret_cast<uintptr_t>(p.C) & 15) == 0) &&
                       (!p.residual || (((p.ldr & 3) == 0) &&
                                        (reinterpret_cast<uintptr_t>(p.residual) & 15) == 0));
  if (vec_out) {
    // stage the accumulator tile through LDS so every lane stores 16 contiguous bytes of a row
    // (the MFMA layout would give 4-byte stores: store-issue bound on the wide layers)
    constexpr int LDC = BN + 4;
    float* Ct = smem;
    __syncthreads();  // all waves are done with the K-loop / statistics scratch
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
          Ct[row * LDC + wn * WTN + j * 32 + l31] = acc[i][j][r];
        }
    __syncthreads();
    constexpr int TPR = BN / 4;        // threads per output row
    constexpr int RPP = 256 / TPR;     // rows per pass
    const int c4 = (tid % TPR) * 4;
    const int col = n0 + c4;
    if (col < p.N) {
      f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
      if (p.scale) sc = *reinterpret_cast<const f32x4*>(p.scale + col);
      if (p.shift) sh = *reinterpret_cast<const f32x4*>(p.shift + col);
#pragma unroll 4
      for (int rr = tid / TPR; rr < BM; rr += RPP) {
        const int row = m0 + rr;
        if (row >= p.M) break;
        f32x4 v = *reinterpret_cast<const f32x4*>(Ct + rr * LDC + c4);
        v = v * sc + sh;
        if (p.residual) v += *reinterpret_cast<const f32x4*>(p.residual + (long)row * p.ldr + col);
        v.x = apply_act(v.x, p.act);
        v.y = apply_act(v.y, p.act);
        v.z = apply_act(v.z, p.act);
        v.w = apply_act(v.w, p.act);
        float* dst = p.C + (long)row * p.ldc + col;
        if (p.accumulate) v += *reinterpret_cast<const f32x4*>(dst);
        *reinterpret_cast<f32x4*>(dst) = v;
      }
    }
    return;
  }
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int col = n0 + wn * WTN + j * 32 + l31;
    if (col >= p.N) continue;
    const float sc = p.scale ? p.scale[col] : 1.f;
    const float sh = p.shift ? p.shift[col] : 0.f;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (row < p.M) {
          float v = acc[i][j][r] * sc + sh;
          if (p.residual) v += p.residual[(long)row * p.ldr + col];
          v = apply_act(v, p.act);
          float* dst = p.C + (long)row * p.ldc + col;
          if (p.accumulate) v += *dst;
          *dst = v;
        }
      }
    }
  }
}

// ====================================================================================
// conv_dma_kernel: the hot convolution path (channels-last, Cin % 16 == 0, [N,K] weights).
//
// Measured on the one-tile-per-workgroup kernel above (profiles/r02_c_*): layer time ~= time of
// the data movement alone + time of the MFMAs alone, i.e. the two never overlapped -- every
// K-tile pays "wait for the staged loads, write LDS, barrier, read fragments" with the matrix
// pipe idle (~400-600 cycles per K-tile), and every tile pays a cold prologue and an epilogue
// whose stores must drain before the workgroup can retire.  This kernel removes those seams:
//   * operands go global -> LDS by DMA (buffer_load ... lds): no staging registers, no LDS
//     write pass, and a ring of 3 K-tiles of 32 channels in flight.  LDS rows are the
//     unpadded 128-byte K-runs the DMA writes (wave-linear), chunk-swizzled on the SOURCE
//     address ((row/2)&7 xor chunk) so that the ds_read_b128 fragment reads are conflict-free;
//   * the K loop is rotated: the MFMA operands of the next group (also across K-tiles and
//     across output tiles) are read while the current group's MFMAs issue, and the ONE barrier
//     per K-tile sits in the middle of the MFMA stream ("stage t+1 has landed for everybody,
//     stage t-1 is free") -- nothing but barrier skew is left between MFMAs;
//   * the prologue of the A operand (previous layer's BatchNorm + ReLU, zero padding AFTER it)
//     is applied to the fragments in registers; its per-channel vectors ride in the stage;
//   * a workgroup walks a strided list of output tiles (bounded: it retires after pk_tiles so
//     that kernels of side streams get CU slots); the DMA ring runs across tile boundaries;
//   * epilogue straight from the accumulator registers, no LDS, no barrier: statistics partials
//     per wave, then scale/shift/activation into a register copy whose 4-byte row-segment
//     stores (2 full 128-byte lines per instruction) are issued ONE PER MFMA during the next
//     tile's first K-tile (counted vmcnt: the ring never waits for a store).
template <int BM, int BN, int PRO>
__global__ __launch_bounds__(256, 2) void conv_dma_kernel(IgemmParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int NSTAGE = 3;
  constexpr int DBK = 32;                   // channels per K-tile (one 128-byte LDS row)
  constexpr int NG = DBK / 8;               // MFMA operand groups (8 channels) per K-tile
  constexpr int WTM = BM / 2, WTN = BN / 2, MT = WTM / 32, NT = WTN / 32;
  constexpr int RB = 128, RPI = 8;          // row bytes; rows per wave-wide DMA instruction
  constexpr int A_BYTES = BM * RB, B_BYTES = BN * RB, PV_BYTES = PRO ? 384 : 0;
  constexpr int STAGE_BYTES = A_BYTES + B_BYTES + PV_BYTES;
  constexpr int A_INSTR = BM / RPI / 4, B_INSTR = BN / RPI / 4;  // DMA instructions per wave
  constexpr int GM = 4 * MT * NT;           // MFMAs (= deferred stores) per operand group
  constexpr int H2 = 2 * GM;                // ... per half K-tile
  constexpr int NS = 16 * MT * NT;          // accumulator registers = stores per wave per tile
  static_assert(NS == NG * GM, "a tile's stores cover exactly one K-tile of MFMAs");
  typedef __attribute__((address_space(3))) void lds_void;

  extern __shared__ __attribute__((aligned(16))) char dsm[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int half = lane >> 5, l31 = lane & 31;

  // ---- this workgroup's tile list (see launch_dma): XCD-contiguous, stride 64, pk_tiles long
  constexpr int PK_RES = 64;
  const int ntiles = p.tiles_m * p.tiles_n;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int per_xcd = (ntiles + 7) >> 3;
  const int round = slot / PK_RES;
  const int x0 = xcd * per_xcd + round * (PK_RES * p.pk_tiles);
  const int t_end = min(min(ntiles, (xcd + 1) * per_xcd), x0 + PK_RES * p.pk_tiles);
  const int first_tile = x0 + (slot - round * PK_RES);
  if (first_tile >= t_end) return;
  const int KT = p.K / DBK;
  const int HoWo = p.Ho * p.Wo;
  const bool taps_matter = PRO && (p.KH * p.KW > 1);

  // ---- buffer descriptors (wave-uniform: kernel arguments only)
  const long bias = ((long)p.pad * p.W + p.pad) * p.lda * 4;  // keeps voffsets non-negative
  const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<char*>(reinterpret_cast<const char*>(p.A)) - bias, 0, (int)(p.a_bytes + bias),
      0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_b = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<char*>(reinterpret_cast<const char*>(p.B)), 0, (int)p.b_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_c = __builtin_amdgcn_make_buffer_rsrc(
      reinterpret_cast<char*>(p.C), 0, (int)p.c_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rsrc_ps = rsrc_b, rsrc_pt = rsrc_b, rsrc_pc = rsrc_b;
  if constexpr (PRO) {
    rsrc_ps = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in_scale), 0, p.Cin * 4, 0x00020000);
    rsrc_pt = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in_shift), 0, p.Cin * 4, 0x00020000);
    rsrc_pc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in_center), 0, p.Cin * 4, 0x00020000);
  }

  // ------------------------------------------------------------------ loader (DMA) side
  // instruction j of this wave fills LDS rows [(j*4 + wave)*8, +8): lane -> row lane/8,
  // PHYSICAL 16-byte chunk lane%8, which holds LOGICAL chunk (lane%8) ^ ((row/2)&7)
  const int drow = lane >> 3, dchunk = lane & 7;
  int a_voff[A_INSTR], b_voff[B_INSTR];
  unsigned a_taps[A_INSTR];
  int u_r = 0, u_q = 0, u_ci = 0;  // filter tap / channel of the next K-tile to fetch
  int l_tile = first_tile, l_kt = 0;
  bool l_more = true;

  auto tap_mask = [&](int m, int& voff_out, int chunk) -> unsigned {
    // pixel of output row m: byte offset of its (tap 0, channel chunk) and the valid-tap bits
    voff_out = BUF_OOB;
    if (m >= p.M) return 0u;
    const int img = m / HoWo;
    const int rem = m - img * HoWo;
    const int ho = rem / p.Wo;
    const int wo = rem - ho * p.Wo;
    const int hi0 = ho * p.stride - p.pad, wi0 = wo * p.stride - p.pad;
    voff_out = (int)((((long)(img * p.H + hi0 + p.pad) * p.W + wi0 + p.pad) * p.lda + chunk * 4) * 4);
    unsigned mask = 0;
    for (int r = 0; r < p.KH; ++r)
      for (int q = 0; q < p.KW; ++q)
        if ((unsigned)(hi0 + r) < (unsigned)p.H && (unsigned)(wi0 + q) < (unsigned)p.W)
          mask |= 1u << (r * p.KW + q);
    return mask;
  };

  auto loader_setup = [&](int t) {
    const int tm = t / p.tiles_n;
    const int lm0 = tm * BM, ln0 = (t - tm * p.tiles_n) * BN;
#pragma unroll
    for (int j = 0; j < A_INSTR; ++j) {
      const int row = (j * 4 + wave) * RPI + drow;
      a_taps[j] = tap_mask(lm0 + row, a_voff[j], dchunk ^ ((row >> 1) & 7));
    }
#pragma unroll
    for (int j = 0; j < B_INSTR; ++j) {
      const int row = (j * 4 + wave) * RPI + drow;
      const int n = ln0 + row;
      b_voff[j] = n < p.N ? (int)(((long)n * p.ldb + (dchunk ^ ((row >> 1) & 7)) * 4) * 4) : BUF_OOB;
    }
    u_r = u_q = u_ci = 0;
  };

  // DMA of the loader's next K-tile into ring slot `buf`; moves on to the next tile of the
  // list when the current one is exhausted
  auto issue = [&](int buf) {
    char* base = dsm + buf * STAGE_BYTES;
    const int tap = u_r * p.KW + u_q;
    const int soff = ((u_r * p.W + u_q) * p.lda + u_ci) * 4;
    const int koff = (tap * p.Cin + u_ci) * 4;
#pragma unroll
    for (int j = 0; j < A_INSTR; ++j) {
      const bool ok = (a_taps[j] >> tap) & 1u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (lds_void*)(base + (j * 4 + wave) * 1024), 16,
                                               ok ? a_voff[j] : BUF_OOB, soff, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < B_INSTR; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(
          rsrc_b, (lds_void*)(base + A_BYTES + (j * 4 + wave) * 1024), 16, b_voff[j], koff, 0, 0);
    if constexpr (PRO) {
      // the 32 channels' scale | shift | center (128 bytes each) ride in the stage; every wave
      // writes the same bytes (keeps the per-wave DMA count uniform for the counted waits)
      if (lane < 8) {
        const int po = u_ci * 4 + lane * 16;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_ps, (lds_void*)(base + A_BYTES + B_BYTES), 16, po, 0, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_pt, (lds_void*)(base + A_BYTES + B_BYTES + 128), 16, po, 0, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_pc, (lds_void*)(base + A_BYTES + B_BYTES + 256), 16, po, 0, 0, 0);
      }
    }
    u_ci += DBK;
    if (u_ci >= p.Cin) {
      u_ci = 0;
      if (++u_q == p.KW) {
        u_q = 0;
        ++u_r;
      }
    }
    if (++l_kt == KT) {
      l_kt = 0;
      l_tile += PK_RES;
      if (l_tile < t_end) loader_setup(l_tile);
      else l_more = false;
    }
  };

  // ------------------------------------------------------------------ compute side
  // fragment byte offsets inside a stage, per operand group (chunk swizzle folded in)
  int a_rd[NG][MT], b_rd[NG][NT];
#pragma unroll
  for (int g = 0; g < NG; ++g) {
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int row = wm * WTM + i * 32 + l31;
      a_rd[g][i] = row * RB + (((2 * g + half) ^ ((row >> 1) & 7)) << 4);
    }
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int row = wn * WTN + j * 32 + l31;
      b_rd[g][j] = A_BYTES + row * RB + (((2 * g + half) ^ ((row >> 1) & 7)) << 4);
    }
  }
  const float relu_floor = p.in_relu ? 0.f : -INFINITY;

  f32x4 fa[2][MT], fb[2][NT];       // operand fragments: [group parity]
  f32x4 pvs[2], pvt[2], pvc[2];     // prologue vectors of the group (PRO)
  unsigned f_taps[MT], nf_taps[MT]; // valid-tap bits of this lane's fragment rows (this / next tile)
#pragma unroll
  for (int i = 0; i < MT; ++i) f_taps[i] = nf_taps[i] = 0xffffffffu;

  auto frag_taps = [&](int t, unsigned (&out)[MT]) {
    const int tm = t / p.tiles_n;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      int dummy;
      out[i] = tap_mask(tm * BM + wm * WTM + i * 32 + l31, dummy, 0);
    }
  };

  auto read_frags = [&](const char* stage, int g, int slot_) {  // g, slot_: compile-time
#pragma unroll
    for (int i = 0; i < MT; ++i)
      fa[slot_][i] = *reinterpret_cast<const f32x4*>(stage + a_rd[g][i]);
#pragma unroll
    for (int j = 0; j < NT; ++j)
      fb[slot_][j] = *reinterpret_cast<const f32x4*>(stage + b_rd[g][j]);
    if constexpr (PRO) {
      const char* pv = stage + A_BYTES + B_BYTES + ((2 * g + half) << 4);
      pvs[slot_] = *reinterpret_cast<const f32x4*>(pv);
      pvt[slot_] = *reinterpret_cast<const f32x4*>(pv + 128);
      pvc[slot_] = *reinterpret_cast<const f32x4*>(pv + 256);
    }
  };

  // x' = max((x - c) * s + t, floor), zero where the filter tap reads padding
  auto transform = [&](int slot_, int tap, const unsigned (&taps)[MT]) {
    if constexpr (PRO) {
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        f32x4 v = (fa[slot_][i] - pvc[slot_]) * pvs[slot_] + pvt[slot_];
        v.x = fmaxf(v.x, relu_floor);
        v.y = fmaxf(v.y, relu_floor);
        v.z = fmaxf(v.z, relu_floor);
        v.w = fmaxf(v.w, relu_floor);
        if (taps_matter && !((taps[i] >> tap) & 1u)) v = f32x4{0.f, 0.f, 0.f, 0.f};
        fa[slot_][i] = v;
      }
    }
  };

  f32x16 acc[MT][NT];
  float pend[NS];     // the previous tile's epilogue values, stored one per MFMA
  int st_voff[NT];        // this lane's byte offset inside 32-column block j of a row (or OOB)
#pragma unroll
  for (int j = 0; j < NT; ++j) st_voff[j] = BUF_OOB;
  int st_soff = 0;        // scalar byte offset of the pending tile's wave sub-tile origin
  const int ldc4 = p.ldc * 4;

  auto store_one = [&](int s) {  // s: compile-time index into pend[] = (j, i, r)
    const int r = s & 15, i = (s >> 4) % MT, j = (s >> 4) / MT;
    const int soff = st_soff + (i * 32 + (r & 3) + 8 * (r >> 2)) * ldc4 + j * 128;
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, pend[s]), rsrc_c, st_voff[j], soff, 0);
  };

  // one K-tile.  PHASE: 0 nothing pending | 1 the K-tile after a tile end: the pending tile's NS
  // stores ride on its NS MFMAs | 2 the K-tile after that: no stores, but H2 of them are younger
  // than the DMA it waits for
  int buf = 0;            // ring slot of the current K-tile
  int c_tap = 0, c_ci = 0;
  bool have_next;         // a K-tile follows this one (in this or the next tile)

#define VLNCE_MFMA_GROUP(SLOT, STORE_BASE, WITH_STORES)                                          \
  _Pragma("unroll") for (int e = 0; e < 4; ++e) _Pragma("unroll") for (int i = 0; i < MT; ++i)   \
      _Pragma("unroll") for (int j = 0; j < NT; ++j) {                                           \
    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[SLOT][i][e], fb[SLOT][j][e], acc[i][j],  \
                                                     0, 0, 0);                                   \
    if constexpr (WITH_STORES) store_one((STORE_BASE) + (e * MT + i) * NT + j);                  \
  }

  auto ktile = [&](auto phase_tag, bool last_of_tile) {
    constexpr int PHASE = decltype(phase_tag)::value;
    constexpr bool ST = PHASE == 1;
    const char* cur = dsm + buf * STAGE_BYTES;
    const int nbuf = buf + 1 == NSTAGE ? 0 : buf + 1;
    const int pbuf = buf == 0 ? NSTAGE - 1 : buf - 1;
    // ---- groups 0 and 1 (group 0's operands are already in slot 0)
    read_frags(cur, 1, 1);
    VLNCE_MFMA_GROUP(0, 0, ST)
    transform(1, c_tap, f_taps);
    read_frags(cur, 2, 0);
    VLNCE_MFMA_GROUP(1, GM, ST)
    transform(0, c_tap, f_taps);
    // ---- the stage after this one has landed (mine), then everybody's; stage t-1 is free
    if (have_next) {
      if constexpr (PHASE != 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(H2) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    if (l_more) issue(pbuf);
    // ---- groups 2 and 3; group 0 of the next K-tile is fetched behind group 3's MFMAs
    read_frags(cur, 3, 1);
    VLNCE_MFMA_GROUP(0, 2 * GM, ST)
    transform(1, c_tap, f_taps);
    int n_tap = c_tap, n_ci = c_ci + DBK;
    if (n_ci >= p.Cin) {
      n_ci = 0;
      ++n_tap;
    }
    if (last_of_tile) n_tap = 0;
    if (have_next) read_frags(dsm + nbuf * STAGE_BYTES, 0, 0);
    VLNCE_MFMA_GROUP(1, 3 * GM, ST)
    if (have_next) {
      if (last_of_tile) transform(0, 0, nf_taps);
      else transform(0, n_tap, f_taps);
    }
    c_tap = n_tap;
    c_ci = last_of_tile ? 0 : n_ci;
    buf = nbuf;
  };

  // ------------------------------------------------------------------ fill the ring
  loader_setup(first_tile);
  if (taps_matter) frag_taps(first_tile, f_taps);
#pragma unroll
  for (int s = 0; s < NSTAGE - 1; ++s)
    if (l_more) issue(s);
  // stage 0 (the older of the two in flight) has landed
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(A_INSTR + B_INSTR + (PRO ? 3 : 0)) : "memory");
  __builtin_amdgcn_s_barrier();
  read_frags(dsm, 0, 0);
  transform(0, 0, f_taps);

  bool pending = false;
  int tile = first_tile;
  while (true) {
    const int tile_m = tile / p.tiles_n;
    const int m0 = tile_m * BM;
    const int n0 = (tile - tile_m * p.tiles_n) * BN;
    const int next_tile = tile + PK_RES;
    const bool more_tiles = next_tile < t_end;
    if (taps_matter && more_tiles) frag_taps(next_tile, nf_taps);

#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    for (int kt = 0; kt < KT; ++kt) {
      const bool last = kt + 1 == KT;
      have_next = !last || more_tiles;
      if (pending && kt == 0) ktile(std::integral_constant<int, 1>{}, last);
      else if (pending && kt == 1) ktile(std::integral_constant<int, 2>{}, last);
      else ktile(std::integral_constant<int, 0>{}, last);
    }

    // ---- statistics of the raw tile: per-wave partials (rows of p.stat_rows), no barrier
    if (p.stat_partial != nullptr) {
      if (MT == 2 && p.stat_rows == 32) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
          wave_stats_block<NT>(acc[i], p.stat_partial, (m0 + wm * WTM) / 32 + i,
                               p.M - (m0 + wm * WTM + i * 32), n0 + wn * WTN, p.N, half, l31);
      } else {
        wave_stats<MT, NT>(acc, p.stat_partial, tile_m * 2 + wm, p.M - (m0 + wm * WTM), WTM,
                           n0 + wn * WTN, p.N, half, l31);
      }
    }

    // ---- epilogue values into the pending registers; their stores ride on the next tile
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int col = n0 + wn * WTN + j * 32 + l31;
      const bool cok = col < p.N;
      const float sc = (p.scale && cok) ? p.scale[col] : 1.f;
      const float sh = (p.shift && cok) ? p.shift[col] : 0.f;
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          pend[(j * MT + i) * 16 + r] = apply_act(acc[i][j][r] * sc + sh, p.act);
    }
    // rows >= M fall outside the output's buffer descriptor and are dropped by the hardware
    // bounds check; a 32-column block at or beyond N (N % 32 == 0, host-checked) is masked here
#pragma unroll
    for (int j = 0; j < NT; ++j)
      st_voff[j] = (n0 + wn * WTN + j * 32 + l31) < p.N ? (4 * half * p.ldc + l31) * 4 : BUF_OOB;
    st_soff = __builtin_amdgcn_readfirstlane(((m0 + wm * WTM) * p.ldc + n0 + wn * WTN) * 4);
    pending = true;

    if (!more_tiles) break;
    tile = next_tile;
    if (taps_matter) {
#pragma unroll
      for (int i = 0; i < MT; ++i) f_taps[i] = nf_taps[i];
    }
  }
  // the last tile's stores
#pragma unroll
  for (int s = 0; s < NS; ++s) store_one(s);
#undef VLNCE_MFMA_GROUP
#endif
}

// y = act(y + shift): second pass of a split-K GEMM that has a bias / activation
__global__ __launch_bounds__(256) void bias_act_kernel(float* __restrict__ c, int ldc, int M, int N,
                                                       const float* __restrict__ shift, int act) {
  const long total = (long)M * N;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long)gridDim.x * blockDim.x) {
    const long row = i / N;
    const int col = (int)(i - row * N);
    float* q = c + row * ldc + col;
    *q = apply_act(*q + (shift ? shift[col] : 0.f), act);
  }
}

template <int BM, int BN, int WM, int WN, int AMODE, int BMODE, int CIN_C = 0, int KW_C = 0,
          int DUAL = 0>
int launch(const IgemmParams& p, hipStream_t stream) {
  constexpr int smem_bytes = 2 * (BM + BN) * LDP * (int)sizeof(float);
  static_assert(BM * (BN + 4) * (int)sizeof(float) <= smem_bytes, "epilogue tile must fit");
  auto kern = igemm_kernel<BM, BN, WM, WN, AMODE, BMODE, CIN_C, KW_C, DUAL>;
  static bool attr_set = false;  // per instantiation
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, smem_bytes);
    if (e != hipSuccess) {
      vlnce_set_error("igemm: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
      return 2;
    }
    attr_set = true;
  }
  IgemmParams q = p;
  q.tiles_m = ceil_div(p.M, BM);
  q.tiles_n = ceil_div(p.N, BN);
  if (q.splitk < 1) q.splitk = 1;
  const long nwg = (long)q.tiles_m * q.tiles_n;
  if (nwg <= 0 || nwg > 0x7fffffffL) {
    vlnce_set_error("igemm: bad grid %ld", nwg);
    return 1;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)nwg, (unsigned)q.splitk), dim3(256), smem_bytes, stream,
                     q);
  VLNCE_CHECK_LAUNCH("igemm");
  return 0;
}

// tile choice: biggest tile that still yields >= ~2 workgroups per CU
struct TileChoice {
  int bm, bn;
};
TileChoice choose_tile(long M, int N) {
  static const int force = getenv("VLNCE_IGEMM_TILE") ? atoi(getenv("VLNCE_IGEMM_TILE")) : 0;
  if (force == 1) return {128, 128};  // tuning knob (scripts/convbench.py)
  if (force == 2) return {128, 64};
  if (force == 3) return {64, 64};
  const long want = 512;
  auto tiles = [&](int bm, int bn) { return (long)ceil_div(M, bm) * ceil_div(N, bn); };
  if (N > 64 && M > 64 && tiles(128, 128) >= want) return {128, 128};
  // N <= 64: 64x64 tiles (4 workgroups per CU out of phase) measured 5-15 % ahead of 128x64
  // on the layer1 shapes (scripts/convbench.py, VLNCE_IGEMM_TILE sweep)
  if (N > 64 && M > 64 && tiles(128, 64) >= want) return {128, 64};
  return {64, 64};
}

template <int AMODE, int BMODE>
int dispatch_tiles(const IgemmParams& p, hipStream_t s) {
  const TileChoice t = choose_tile(p.M, p.N);
  if (t.bm == 128 && t.bn == 128) return launch<128, 128, 2, 2, AMODE, BMODE>(p, s);
  if (t.bm == 128 && t.bn == 64) return launch<128, 64, 2, 2, AMODE, BMODE>(p, s);
  return launch<64, 64, 2, 2, AMODE, BMODE>(p, s);
}
int dispatch_dual(const IgemmParams& p, hipStream_t s) {
  const TileChoice t = choose_tile(p.M, p.N);
  if (t.bm == 128 && t.bn == 128) return launch<128, 128, 2, 2, A_BUF, B_BUF, 0, 0, 1>(p, s);
  if (t.bm == 128 && t.bn == 64) return launch<128, 64, 2, 2, A_BUF, B_BUF, 0, 0, 1>(p, s);
  return launch<64, 64, 2, 2, A_BUF, B_BUF, 0, 0, 1>(p, s);
}
template <int AMODE, int BMODE>
int dispatch_small(const IgemmParams& p, hipStream_t s) {
  return launch<64, 64, 2, 2, AMODE, BMODE>(p, s);
}
// buffer-descriptor hot path: channels-last im2col / row-major A with Cin % 32 == 0, at most
// 32 filter taps, [N,K] weights, operands below 2 GiB
bool buf_ok(const IgemmParams& p) {
  static const bool off = getenv("VLNCE_IGEMM_NOBUF") != nullptr;
  const long bias = ((long)p.pad * p.W + p.pad) * p.lda * 4;
  return !off && (p.Cin % 32 == 0) && (p.K % 32 == 0) && (p.lda % 4 == 0) && (p.ldb % 4 == 0) &&
         p.KH * p.KW <= 32 && p.a_bytes + bias < 0x7fffffffL && p.b_bytes < 0x7fffffffL;
}
// 7x7 stems: scalar loaders with compile-time Cin / KW
template <int CIN_C>
int dispatch_stem(const IgemmParams& p, hipStream_t s) {
  const TileChoice t = choose_tile(p.M, p.N);
  if (t.bm == 128) return launch<128, 64, 2, 2, A_IM2COL_S, B_NK_S, CIN_C, 7>(p, s);
  return launch<64, 64, 2, 2, A_IM2COL_S, B_NK_S, CIN_C, 7>(p, s);
}

// split-K factor for a plain GEMM with few output tiles and a long reduction
int choose_splitk(const IgemmParams& p) {
  static const bool off = getenv("VLNCE_IGEMM_NO_SPLITK") != nullptr;  // diagnostic switch
  if (off) return 1;
  const long tiles = (long)ceil_div(p.M, 64) * ceil_div(p.N, 64);
  const int KT = ceil_div(p.K, BK);
  if (tiles >= 128 || KT < 8) return 1;
  long s = (256 + tiles - 1) / tiles;
  if (s > KT / 2) s = KT / 2;
  if (s > 64) s = 64;
  return s < 2 ? 1 : (int)s;
}

void fill_epilogue(IgemmParams& p, const vlnce_epilogue* e) {
  p.scale = e ? e->scale : nullptr;
  p.shift = e ? e->shift : nullptr;
  p.residual = e ? e->residual : nullptr;
  p.ldr = e ? e->ldr : 0;
  p.act = e ? e->act : 0;
  p.accumulate = e ? e->accumulate : 0;
  p.stat_partial = e ? e->stat_partial : nullptr;
}

bool aligned16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }

// conv_dma launch: 64 workgroups per XCD per round (2 per CU), each walking pk_tiles tiles
__device__ float vlnce_zero_vec[4096];

template <int BM, int BN, int PRO>
int launch_dma(const IgemmParams& p, hipStream_t stream) {
  constexpr int smem_bytes = 3 * ((BM + BN) * 128 + (PRO ? 384 : 0));
  auto kern = conv_dma_kernel<BM, BN, PRO>;
  static bool attr_set = false;  // per instantiation
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, smem_bytes);
    if (e != hipSuccess) {
      vlnce_set_error("conv_dma: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
      return 2;
    }
    attr_set = true;
  }
  IgemmParams q = p;
  q.tiles_m = ceil_div(p.M, BM);
  q.tiles_n = ceil_div(p.N, BN);
  q.splitk = 1;
  const long ntiles = (long)q.tiles_m * q.tiles_n;
  if (ntiles <= 0 || ntiles > 0x7fffffffL) {
    vlnce_set_error("conv_dma: bad tile count %ld", ntiles);
    return 1;
  }
  // tiles per workgroup (tuning knob; 1 = one tile per workgroup, large = fully persistent)
  static const int pk_tiles = getenv("VLNCE_PK_TILES") ? atoi(getenv("VLNCE_PK_TILES")) : 8;
  q.pk_tiles = pk_tiles < 1 ? 1 : pk_tiles;
  const long per_xcd = (ntiles + 7) / 8;
  const long rounds = (per_xcd + 64L * q.pk_tiles - 1) / (64L * q.pk_tiles);
  // the last round of an XCD may be short: only as many workgroups as it has tiles
  const long last = per_xcd - (rounds - 1) * 64L * q.pk_tiles;
  const long slots = (rounds - 1) * 64 + (last < 64 ? last : 64);
  hipLaunchKernelGGL(kern, dim3((unsigned)(8 * slots)), dim3(256), smem_bytes, stream, q);
  VLNCE_CHECK_LAUNCH("conv_dma");
  return 0;
}

template <int PRO>
int dispatch_dma(const IgemmParams& p, hipStream_t s) {
  // 128x64 is the largest tile: 128x128 would need accumulators + the deferred-store copy +
  // double-buffered fragments = more than the 256 registers two waves per SIMD leave each.
  // The M extent (hence the statistics granularity, vlnce_conv2d_tile_rows) is choose_tile's.
  // p.stat_rows (what vlnce_conv2d_tile_rows told the caller) is 64 or 32 here (dma_ok)
  if ((long)ceil_div(p.M, 128) * ceil_div(p.N, 64) >= 512) return launch_dma<128, 64, PRO>(p, s);
  if (p.stat_partial && p.stat_rows != 32) {
    vlnce_set_error("conv_dma: inconsistent statistics granularity");
    return 1;
  }
  return launch_dma<64, 64, PRO>(p, s);
}

// what conv_dma_kernel covers: 32-channel K-tiles inside one filter tap, at least 2 of them,
// 32-column output blocks, plain epilogue (scale / shift / activation, statistics)
bool dma_ok(const IgemmParams& p) {
  // opt-in while it is slower than the register-staged kernel (profiles/r02_e_*)
  static const bool off = getenv("VLNCE_IGEMM_DMA") == nullptr;
  const long bias = ((long)p.pad * p.W + p.pad) * p.lda * 4;
  return !off && (p.Cin % 32 == 0) && p.K >= 64 && (p.lda % 4 == 0) && (p.ldb % 4 == 0) &&
         (p.N % 32 == 0) && p.KH * p.KW <= 32 && p.a_bytes + bias < 0x7fffffffL &&
         p.b_bytes < 0x7fffffffL && p.c_bytes < 0x7fffffffL && !p.residual && !p.accumulate &&
         !p.A2 && !p.side_out && p.Cin <= 4096 && !(p.stat_partial && p.stat_rows < 32);
}

}  // namespace

// rows per statistics partial: the M extent of one wave's sub-tile (BM / 2 = 64 or 32), halved
// down to 16 while it does not divide a sample's pixel count (GroupNorm needs partials that do
// not straddle samples; habitat's depth trunk ends at 4x4 = 16 pixels per sample)
static int stat_rows_for(const vlnce_conv_desc* d) {
  const long M = (long)d->N * d->Ho * d->Wo;
  int r = choose_tile(M, d->Cout).bm / 2;
  const int hw = d->Ho * d->Wo;
  while (r > 16 && hw % r != 0) r /= 2;
  return r;
}

extern "C" int vlnce_conv2d_tile_rows(const vlnce_conv_desc* d) { return stat_rows_for(d); }

extern "C" int vlnce_conv2d_tiles_m(const vlnce_conv_desc* d) {
  const long M = (long)d->N * d->Ho * d->Wo;
  return ceil_div(M, stat_rows_for(d));
}

extern "C" int vlnce_conv2d_fwd(const float* x, const float* w, float* y, const vlnce_conv_desc* d,
                                const vlnce_prologue* pro, const vlnce_epilogue* epi,
                                vlnce_stream_t stream) {
  VLNCE_CHECK_ARG(x && w && y && d, "conv2d_fwd: null argument");
  VLNCE_CHECK_ARG(d->N > 0 && d->H > 0 && d->W > 0 && d->Cin > 0 && d->Cout > 0,
                  "conv2d_fwd: bad shape");
  VLNCE_CHECK_ARG(d->H < 32768 && d->W < 32768, "conv2d_fwd: H/W must be < 32768");
  const int ho = (d->H + 2 * d->pad - d->KH) / d->stride + 1;
  const int wo = (d->W + 2 * d->pad - d->KW) / d->stride + 1;
  VLNCE_CHECK_ARG(ho == d->Ho && wo == d->Wo, "conv2d_fwd: Ho/Wo mismatch (%d,%d) vs (%d,%d)", ho,
                  wo, d->Ho, d->Wo);
  const long M = (long)d->N * d->Ho * d->Wo;
  VLNCE_CHECK_ARG(M < 0x7fffffffL && (long)d->N * d->H * d->W < 0x7fffffffL,
                  "conv2d_fwd: too many pixels for 32-bit row indices");
  IgemmParams p{};
  p.A = x;
  p.B = w;
  p.C = y;
  p.M = (int)M;
  p.N = d->Cout;
  p.K = d->KH * d->KW * d->Cin;
  p.H = d->H;
  p.W = d->W;
  p.Cin = d->Cin;
  p.KH = d->KH;
  p.KW = d->KW;
  p.stride = d->stride;
  p.pad = d->pad;
  p.Ho = d->Ho;
  p.Wo = d->Wo;
  p.lda = d->ldx ? d->ldx : d->Cin;
  p.ldb = p.K;
  p.ldc = d->ldy ? d->ldy : d->Cout;
  p.in_scale = pro ? pro->in_scale : nullptr;
  p.in_shift = pro ? pro->in_shift : nullptr;
  p.in_center = pro ? pro->in_center : nullptr;
  p.in_relu = pro ? pro->in_relu : 0;
  p.A2 = pro ? pro->x2 : nullptr;
  p.in2_scale = pro ? pro->in2_scale : nullptr;
  p.in2_shift = pro ? pro->in2_shift : nullptr;
  p.in2_center = pro ? pro->in2_center : nullptr;
  p.side_out = pro ? pro->side_out : nullptr;
  VLNCE_CHECK_ARG((p.in_scale == nullptr) == (p.in_shift == nullptr),
                  "conv2d_fwd: in_scale and in_shift must come together");
  fill_epilogue(p, epi);
  const bool v4 = (d->Cin % 4 == 0) && (p.lda % 4 == 0) && aligned16(x) && aligned16(w) &&
                  (!p.in_scale || (aligned16(p.in_scale) && aligned16(p.in_shift))) &&
                  (!p.in_center || aligned16(p.in_center));
  p.splitk = 1;
  p.a_bytes = (((long)d->N * d->H * d->W - 1) * p.lda + d->Cin) * 4;
  p.b_bytes = (long)d->Cout * p.K * 4;
  p.c_bytes = ((M - 1) * p.ldc + d->Cout) * 4;
  p.stat_rows = stat_rows_for(d);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (p.A2 != nullptr || p.side_out != nullptr) {
    VLNCE_CHECK_ARG(p.A2 && p.in_scale && d->KH == 1 && d->KW == 1 && d->stride == 1 &&
                        d->pad == 0 && v4 && buf_ok(p) && aligned16(p.A2) &&
                        (!p.side_out || aligned16(p.side_out)) &&
                        ((p.in2_scale == nullptr) == (p.in2_shift == nullptr)) &&
                        (!p.in2_scale || (aligned16(p.in2_scale) && aligned16(p.in2_shift))) &&
                        (!p.in2_center || (p.in2_scale && aligned16(p.in2_center))),
                    "conv2d_fwd: the dual-input prologue needs x2 + in_scale on a 1x1/stride-1/"
                    "pad-0 convolution with Cin %% 32 == 0 and 16-byte aligned operands");
    return dispatch_dual(p, s);
  }
  if (v4 && dma_ok(p)) {
    if (p.in_scale == nullptr) return dispatch_dma<0>(p, s);
    if (p.in_center == nullptr) {  // the kernel always subtracts a centre: hand it zeros
      static float* zeros = nullptr;
      if (!zeros && hipGetSymbolAddress(reinterpret_cast<void**>(&zeros), HIP_SYMBOL(vlnce_zero_vec)) != hipSuccess) {
        vlnce_set_error("conv2d_fwd: no zero vector");
        return 2;
      }
      p.in_center = zeros;
    }
    return dispatch_dma<1>(p, s);
  }
  if (v4 && buf_ok(p)) {
    // Small batches (act() at num_envs 1..8, eval BatchNorm folded into scale/shift): a late
    // ResNet layer is a handful of 64x64 tiles with a reduction of up to 144 K-tiles -- one
    // workgroup walking them alone is pure latency (20-50 us per layer).  Split the reduction
    // over blockIdx.y with atomic accumulation into a zeroed output and apply the epilogue
    // (scale/shift/residual/activation) in a second, row-wise pass.
    const int sk = (!p.stat_partial && !p.accumulate && p.ldc == p.N &&
                    ((p.scale && p.shift) || (!p.scale && !p.residual)))
                       ? choose_splitk(p)
                       : 1;
    if (sk > 1) {
      const float* scale = p.scale;
      const float* shift = p.shift;
      const float* residual = p.residual;
      const int act = p.act, ldr = p.ldr;
      p.scale = p.shift = p.residual = nullptr;
      p.act = 0;
      p.splitk = sk;
      vlnce_zero(y, M, p.N, p.ldc, s);
      if (int rc = dispatch_small<A_BUF, B_BUF>(p, s)) return rc;
      if (scale) {
        VLNCE_CHECK_ARG(!residual || ldr == p.N, "conv2d_fwd: split-K needs a contiguous residual");
        return vlnce_scale_shift_act(y, scale, shift, nullptr, 0, residual, y, M, p.N, act, stream);
      }
      if (shift || act) {
        const long work = M * p.N;
        const int grid = (int)((work + 255) / 256 > 2048 ? 2048 : (work + 255) / 256);
        hipLaunchKernelGGL(bias_act_kernel, dim3(grid), dim3(256), 0, s, y, p.ldc, (int)M, p.N, shift, act);
        VLNCE_CHECK_LAUNCH("conv2d_fwd bias/act");
      }
      return 0;
    }
    return dispatch_tiles<A_BUF, B_BUF>(p, s);
  }
  if (v4) return dispatch_tiles<A_IM2COL_V4, B_NK_V4>(p, s);
  if (d->KW == 7 && d->Cin == 3) return dispatch_stem<3>(p, s);
  if (d->KW == 7 && d->Cin == 1) return dispatch_stem<1>(p, s);
  return dispatch_tiles<A_IM2COL_S, B_NK_S>(p, s);
}

extern "C" int vlnce_gemm(const float* A, int lda, int transA, const float* B, int ldb, int transB,
                          float* C, int ldc, int M, int N, int K, const vlnce_epilogue* epi,
                          vlnce_stream_t stream) {
  VLNCE_CHECK_ARG(A && B && C, "gemm: null argument");
  VLNCE_CHECK_ARG(M > 0 && N > 0 && K > 0, "gemm: bad shape %d %d %d", M, N, K);
  VLNCE_CHECK_ARG(!(epi && epi->stat_partial), "gemm: stat_partial is a conv-only option");
  IgemmParams p{};
  p.A = A;
  p.B = B;
  p.C = C;
  p.M = M;
  p.N = N;
  p.K = K;
  // plain matrix as a 1x1-conv over an "image" of M pixels with K channels, folded
  // into rows of 1024 pixels so the loader's 16-bit (h, w) fields never overflow
  VLNCE_CHECK_ARG((long)M < 32767L * 1024L, "gemm: M too large");
  p.W = M < 1024 ? M : 1024;
  p.H = ceil_div(M, p.W);
  p.Cin = K;
  p.KH = p.KW = 1;
  p.stride = 1;
  p.pad = 0;
  p.Ho = p.H;
  p.Wo = p.W;
  p.lda = lda;
  p.ldb = ldb;
  p.ldc = ldc;
  fill_epilogue(p, epi);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  // split-K: few output tiles + long reduction (tail GEMMs at num_envs rows, dW GEMMs).
  // The K ranges add atomically into a zeroed C; bias / activation run as a second pass.
  p.splitk = 1;
  // split-K: K ranges accumulate with atomics.  C += A B (accumulate without an activation) is
  // the same thing minus the zero-fill -- the recurrent dgrad of a T-step rollout,
  // dh += dgates W_hh with 5 x 512 outputs and K = 1536, is 8 workgroups otherwise.
  const bool plain = !p.scale && !p.residual && !(p.accumulate && p.act);
  if (plain) p.splitk = choose_splitk(p);
  const float* bias2 = nullptr;
  int act2 = 0;
  if (p.splitk > 1) {
    bias2 = p.shift;
    act2 = p.act;
    p.shift = nullptr;
    p.act = 0;
    if (!p.accumulate) vlnce_zero(C, M, N, ldc, s);
    p.accumulate = 0;
  }
  int rc;
  if (!transA) {
    const bool av4 = (K % 4 == 0) && (lda % 4 == 0) && aligned16(A);
    if (!transB) {
      const bool bv4 = (K % 4 == 0) && (ldb % 4 == 0) && aligned16(B);
      p.a_bytes = (((long)M - 1) * lda + K) * 4;
      p.b_bytes = (((long)N - 1) * ldb + K) * 4;
      if (av4 && bv4 && buf_ok(p))
        rc = p.splitk > 1 ? dispatch_small<A_BUF, B_BUF>(p, s) : dispatch_tiles<A_BUF, B_BUF>(p, s);
      else if (av4 && bv4)
        rc = p.splitk > 1 ? dispatch_small<A_IM2COL_V4, B_NK_V4>(p, s)
                          : dispatch_tiles<A_IM2COL_V4, B_NK_V4>(p, s);
      else
        rc = dispatch_small<A_IM2COL_S, B_NK_S>(p, s);
    } else {
      VLNCE_CHECK_ARG(aligned16(B), "gemm: B must be 16-byte aligned");
      if (av4)
        rc = p.splitk > 1 ? dispatch_small<A_IM2COL_V4, B_KN>(p, s)
                          : dispatch_tiles<A_IM2COL_V4, B_KN>(p, s);
      else
        rc = dispatch_small<A_IM2COL_S, B_KN>(p, s);
    }
  } else {
    VLNCE_CHECK_ARG(transB, "gemm: transA requires transB (only A^T * B^T-stored form is built)");
    VLNCE_CHECK_ARG(aligned16(A) && aligned16(B), "gemm: operands must be 16-byte aligned");
    rc = p.splitk > 1 ? dispatch_small<A_TRANS, B_KN>(p, s) : dispatch_tiles<A_TRANS, B_KN>(p, s);
  }
  if (rc != 0) return rc;
  if (p.splitk > 1 && (bias2 || act2)) {
    const long work = (long)M * N;
    const int grid = (int)((work + 255) / 256 > 2048 ? 2048 : (work + 255) / 256);
    hipLaunchKernelGGL(bias_act_kernel, dim3(grid), dim3(256), 0, s, C, ldc, M, N, bias2, act2);
    VLNCE_CHECK_LAUNCH("gemm bias/act");
  }
  return 0;
}

// dW[Cout, KH, KW, Cin] = sum over output pixels of dY[m, co] * im2col(X)[m, (r,q,ci)]
extern "C" int vlnce_conv2d_wgrad(const float* x, const float* dy, float* dw_ohwi,
                                  const vlnce_conv_desc* d, vlnce_stream_t stream) {
  VLNCE_CHECK_ARG(x && dy && dw_ohwi && d, "conv2d_wgrad: null argument");
  const long Mrows = (long)d->N * d->Ho * d->Wo;
  VLNCE_CHECK_ARG(Mrows > 0 && Mrows < 0x7fffffffL, "conv2d_wgrad: bad shape");
  IgemmParams p{};
  p.A = dy;
  p.B = x;
  p.C = dw_ohwi;
  p.M = d->Cout;
  p.N = d->KH * d->KW * d->Cin;
  p.K = (int)Mrows;
  p.H = d->H;
  p.W = d->W;
  p.Cin = d->Cin;
  p.KH = d->KH;
  p.KW = d->KW;
  p.stride = d->stride;
  p.pad = d->pad;
  p.Ho = d->Ho;
  p.Wo = d->Wo;
  p.lda = d->ldy ? d->ldy : d->Cout;
  p.ldb = d->ldx ? d->ldx : d->Cin;
  p.ldc = p.N;
  VLNCE_CHECK_ARG(aligned16(dy) && aligned16(x) && aligned16(dw_ohwi),
                  "conv2d_wgrad: operands must be 16-byte aligned");
  fill_epilogue(p, nullptr);
  const long tiles = (long)ceil_div(p.M, 64) * ceil_div(p.N, 64);
  const int KT = ceil_div(p.K, BK);
  long sk = (1024 + tiles - 1) / tiles;
  if (sk > KT / 4) sk = KT / 4;
  if (sk > 512) sk = 512;
  p.splitk = sk < 2 ? 1 : (int)sk;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (p.splitk > 1) {
    vlnce_zero(dw_ohwi, 1, p.M * p.N, (long)p.M * p.N, s);
  }
  return launch<64, 64, 2, 2, A_TRANS, B_IM2COL>(p, s);
}
