// Implicit-GEMM convolution / general GEMM.  Two kernels live here:
//   * igemm_kernel: exact fp32 on v_mfma_f32_32x32x2_f32 (described right below) -- every GEMM of
//     the trainable tail, the 7x7 stems, the handful-of-tiles layers, VLNCE_CONV_MATH=f32;
//   * conv_x3_kernel (further down): fp32 operands split into three bf16 planes, six plane
//     products on v_mfma_f32_32x32x16_bf16 -- the 1x1 / strided convolutions of the frozen trunks
//     that conv_p3.hip (conv_p3_kernel / conv_u3_kernel / conv_s3_kernel) does not take.
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
//     (Operands that are k-major in memory -- transposed A, [K,N] B, the weight
//     gradient's im2col -- stay k-major in LDS, [k][tile + 8]: one 16-byte write
//     per staged float4, one float per lane and MFMA on the read side.)
//     The k order inside a group of 8 is permuted identically for A and B
//     (lanes 0-31 take k 0..3, lanes 32-63 take k 4..7), which a dot product
//     does not care about.
//   * exact fp32: the MFMA is a k-ordered fmaf chain, no reduced precision.
//   * blockIdx -> tile map is XCD-aware: each XCD (private L2) gets a
//     contiguous run of tiles with the N-tile index fastest, so blocks that
//     share an A row-panel hit the same L2.
#include "igemm_shared.h"

using namespace vlnce_detail;

namespace {


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
  constexpr bool A_KMAJOR = AMODE == A_TRANS;
  constexpr bool B_KMAJOR = BMODE == B_KN || BMODE == B_IM2COL;
  constexpr int LDT_A = BM + 8, LDT_B = BN + 8;  // k-major LDS images: [BK][B? + 8]
  constexpr int A_TILE = BM * LDP;  // (BK * LDT <= B? * LDP: the k-major image fits the same space)
  constexpr int B_TILE = BN * LDP;
  static_assert(BK * LDT_A <= A_TILE && BK * LDT_B <= B_TILE, "k-major LDS image");
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
      // a k-major operand stays k-major in LDS ([k][BM + 8]): the staged float4 (four
      // consecutive m of one k) is ONE conflict-free 16-byte write, and the fragment reads below
      // take one float per lane and MFMA from consecutive m = consecutive banks (the pitch puts
      // the other half-wave's k + 4 on the other 32 banks).  Writing it transposed into the
      // [m][k] image of the row-major modes was four 4-byte writes, 4-way bank-conflicted, and
      // bound the weight-gradient GEMMs (profiles/archive/r03_i_*).
      constexpr int TPR = BM / 4;
      constexpr int KPP = 256 / TPR;
      const int km = tid / TPR;
      const int m4 = (tid - km * TPR) * 4;
#pragma unroll
      for (int i = 0; i < A_ROWS; ++i)
        *reinterpret_cast<f32x4*>(As + (i * KPP + km) * LDT_A + m4) = a_reg[i];
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
      for (int i = 0; i < B_ROWS; ++i)  // k-major in LDS, see A_TRANS above
        *reinterpret_cast<f32x4*>(Bs + (i * KPP + kn) * LDT_B + n4) = b_reg[i];
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
    const float* Aw = A_KMAJOR ? cur + (4 * half) * LDT_A + wm * WTM + l31
                               : cur + (wm * WTM + l31) * LDP + 4 * half;
    const float* Bw = B_KMAJOR ? cur + A_TILE + (4 * half) * LDT_B + wn * WTN + l31
                               : cur + A_TILE + (wn * WTN + l31) * LDP + 4 * half;
#pragma unroll
    for (int g = 0; g < BK / 8; ++g) {
      f32x4 af[MT], bf[NT];
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        if constexpr (A_KMAJOR) {
#pragma unroll
          for (int e = 0; e < 4; ++e) af[i][e] = Aw[(8 * g + e) * LDT_A + i * 32];
        } else {
          af[i] = *reinterpret_cast<const f32x4*>(Aw + i * 32 * LDP + 8 * g);
        }
      }
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        if constexpr (B_KMAJOR) {
#pragma unroll
          for (int e = 0; e < 4; ++e) bf[j][e] = Bw[(8 * g + e) * LDT_B + j * 32];
        } else {
          bf[j] = *reinterpret_cast<const f32x4*>(Bw + j * 32 * LDP + 8 * g);
        }
      }
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
// conv_x3_kernel: the hot convolution path (channels-last, Cin % 32 == 0, [N,K] weights).
//
// fp32 convolution on the bf16 matrix pipe.  gfx950 has no fp32-rate matrix instruction beyond
// v_mfma_f32_32x32x2_f32 (157 TF/s, = the vector rate); v_mfma_f32_32x32x16_bf16 is 16x faster.
// Every fp32 operand is split EXACTLY into three bf16 planes (round to nearest at each step),
//     x = x1 + x2 + x3   (8 + 8 + 8 mantissa bits, same exponent range as fp32),
// and a product a*b is the six plane products of order <= 2^-16,
//     a1 b1 + (a1 b2 + a2 b1) + (a2 b2 + a1 b3 + a3 b1),
// each exact in the fp32 accumulator; the dropped terms are <= 2^-26 relative.  Round 2 split by
// TRUNCATION (dropped terms 2^-24): measured against fp64 that was rms 3.4-5.0e-7 on the 3x3
// layers = 1.1-1.2x the fp32-MFMA kernel above and 2-3x torch's own fp32 convolution (1.6-1.7e-7;
// profiles/archive/r02_o_conv_accuracy_*.txt) -- fp32-class, NOT equal to it as round 2's text claimed.
// With the round-to-nearest split: profiles/archive/r03_*_conv_accuracy*.txt.  Six
// bf16 MFMAs of 32 cycles replace eight fp32 MFMAs of 64 cycles per 32x32x16 block: 2.7x less
// matrix-pipe time.
//
// What the fp32-MFMA kernel taught (profiles/archive/r02_c_*): with every wave doing "load, transform,
// write LDS, barrier, read fragments, MFMA", the matrix pipe idles while the wave does anything
// else, and the co-resident workgroup runs in lock-step, so nothing covers it.  Here the roles
// are split.  A workgroup is 16 waves, four per SIMD, one workgroup per CU:
//   * waves 8-15, the PRODUCERS (two per SIMD: one wave alone issues one instruction per ~4
//     cycles, not enough for ~400 instructions per K-tile): buffer-load the A (im2col) and B
//     (weight) K-tiles into a register ring several tiles ahead, apply the operand prologue
//     (previous layer's BatchNorm + ReLU, the dual-input block end, zero padding after it),
//     split to bf16 planes and write them to LDS;
//   * waves 0-7, the MATRIX waves (two per SIMD, taking turns on the matrix pipe: while one
//     waits for LDS or a counter the other issues): ds_read_b128 fragments, MFMAs, and the
//     epilogue of their 64x32 sub-tile straight from the accumulator registers.
// The SIMD's matrix pipe (matrix wave) and its VALU / memory pipes (producer waves) run side by
// side by hardware arbitration, not by compiler scheduling.
// LDS: two stages of {A, B} x 3 planes x rows x 80 bytes (32 bf16 + 16 bytes pad: the
// ds_read_b128 fragment reads are conflict-free).  The hand-over is two LDS counters instead of
// s_barrier (measured: a barrier per K-tile costs the matrix pipe ~200 idle cycles of skew):
// per stage, `full` counts its writes (8 per K-tile), `empty` the matrix waves done reading it
// (8 per K-tile); whoever is ahead never waits.  The LDS unit executes one wave's operations in
// order, so "data writes, then counter add" needs no wait in between.
// A workgroup walks a strided list of output tiles; the K-tile stream (and the producers'
// register ring) runs across tile boundaries, so the prologue of a tile (addresses, first loads)
// and its epilogue are covered by the neighbours' MFMAs.
constexpr int X3_PITCH = 80;      // bytes per LDS row of one plane
constexpr int X3_PRODUCERS = 8;  // producer waves; the matrix waves are WM x WN = 8 (or 4)

template <int ROWS, int DUAL>
struct X3Staged {
  f32x4 a[ROWS];
  f32x4 a2[DUAL ? ROWS : 1];
  unsigned ok;
  int m0;
};

// x (4 consecutive k of one row) -> the three planes' 8-byte LDS words.  Round-to-nearest split
// (v_cvt_pk_bf16_f32): x = x1 + x2 + x3 exactly, |x2| <= 2^-9 |x|, |x3| <= 2^-18 |x|, so the three
// dropped cross terms are <= 2^-26 relative (truncation, rounds 2: 2^-24); same instruction count.
// (MATH_F16X3: two fp16 A planes, see Planes<> in igemm_shared.h)
template <int MATH>
__device__ __forceinline__ void x3_split_store(f32x4 x, char* row_ptr, int plane_bytes) {
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  constexpr int NA = Planes<MATH>::NA;
  unsigned w0[NA], w1[NA];
  split_pair<MATH>(x[0], x[1], w0);
  split_pair<MATH>(x[2], x[3], w1);
#pragma unroll
  for (int q = 0; q < NA; ++q) *reinterpret_cast<u32x2*>(row_ptr + q * plane_bytes) = u32x2{w0[q], w1[q]};
}

template <int BM, int BN, int WM, int WN, int DUAL, int MATH>
__global__ __launch_bounds__((WM * WN + X3_PRODUCERS) * 64) void conv_x3_kernel(IgemmParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
  typedef Planes<MATH> PL;
  constexpr int NA = PL::NA;
  constexpr int X3_MATRIX = WM * WN;  // matrix waves (8: two per SIMD take turns on the pipe)
  constexpr int WTM = BM / WM, WTN = BN / WN, MT = WTM / 32, NT = WTN / 32;
  constexpr int A_PLANE = BM * X3_PITCH, B_PLANE = BN * X3_PITCH;
  constexpr int A_BYTES = NA * A_PLANE, B_BYTES = 3 * B_PLANE;
  constexpr int STAGE_BYTES = A_BYTES + B_BYTES;
  constexpr int PR = X3_PRODUCERS * 64 / 8;          // rows per producer pass (8 float4 each)
  constexpr int A_ROWS = BM / PR;                    // producer passes over the A tile
  static_assert(MT >= 1 && NT >= 1 && A_ROWS >= 1 && BN <= X3_PRODUCERS * 16, "tile");

  extern __shared__ __attribute__((aligned(16))) char xsm[];
  // counters per STAGE: waves are at most one K-tile apart, so a single running count could be
  // reached by fast waves' next-K-tile adds while a slow wave's are missing; a stage's count
  // cannot (its next use waits for everybody's current one)
  int* const full = reinterpret_cast<int*>(xsm + 2 * STAGE_BYTES);  // [2]
  int* const empty = full + 2;                                       // [2]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5;
  const int l31 = lane & 31;
  const int KT = p.K / BK;

  // ---- this workgroup's tiles: virtual block ids blockIdx.x + r * gridDim.x through the
  // XCD-aware map of igemm_kernel (gridDim.x is a multiple of 8 or the whole tile count)
  const int ntiles = p.tiles_m * p.tiles_n;
  const int my_tiles = (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
  const int G_total = my_tiles * KT;  // K-tiles this workgroup streams
  auto tile_of = [&](int round, int& m0, int& n0) {
    const int v = blockIdx.x + round * gridDim.x;
    const int q = ntiles >> 3, r = ntiles & 7, xcd = v & 7, idx = v >> 3;
    const int tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    const int tm = tile / p.tiles_n;
    m0 = tm * BM;
    n0 = (tile - tm * p.tiles_n) * BN;
  };

  if (tid < 4) full[tid] = 0;
  __syncthreads();

  if (wave >= X3_MATRIX) {
    // ================================================================ producer waves
    const int ptid = tid - X3_MATRIX * 64;
    const int lrow = ptid >> 3;
    const int lk4 = (ptid & 7) * 4;
    const long bias = ((long)p.pad * p.W + p.pad) * p.lda * 4;  // keeps voffsets non-negative
    const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(p.A)) - bias, 0, (int)(p.a_bytes + bias),
        0x00020000);
    __amdgpu_buffer_rsrc_t rsrc_a2 = rsrc_a;
    if constexpr (DUAL)
      rsrc_a2 = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<char*>(reinterpret_cast<const char*>(p.A2)) - bias, 0,
          (int)(p.a_bytes + bias), 0x00020000);
    // B comes pre-split (vlnce_conv2d_split_weights): three planes of N*K bf16, row n = K
    // contiguous; thread -> (row ptid / 4, 16-byte chunk ptid % 4 of the K-tile's 64 bytes)
    const int plane_b = (int)(p.b_bytes >> 1);  // bytes of one plane
    const __amdgpu_buffer_rsrc_t rsrc_b = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(p.Bsplit)), 0, 3 * plane_b, 0x00020000);
    const int brow = ptid >> 2, bchunk = ptid & 3;
    const bool has_pro = p.in_scale != nullptr;
    const bool pad_matters = p.pad > 0;
    const float relu_floor = p.in_relu ? 0.f : -__builtin_huge_valf();  // max(x, -inf) = x
    const int HoWo = p.Ho * p.Wo;

    // ---- load cursor: the tile whose K-tiles are being fetched
    int a_voff[A_ROWS], b_voff;
    unsigned a_taps[A_ROWS];  // bit t: filter tap t of this output pixel reads inside the image
    int l_round = 0, l_m0 = 0, l_n0 = 0, l_kt = 0;
    int u_r = 0, u_q = 0, u_ci = 0, u_k = 0;  // wave-uniform (tap, channel, k) of the next fetch
    auto setup_tile = [&](int round) {
      int m0, n0;
      tile_of(round, m0, n0);
      l_m0 = m0;
      l_n0 = n0;
#pragma unroll
      for (int i = 0; i < A_ROWS; ++i) {
        const int m = m0 + i * PR + lrow;
        a_voff[i] = BUF_OOB;
        a_taps[i] = 0;
        if (m < p.M) {
          const int img = m / HoWo;
          const int rem = m - img * HoWo;
          const int ho = rem / p.Wo;
          const int wo = rem - ho * p.Wo;
          const int hi0 = ho * p.stride - p.pad, wi0 = wo * p.stride - p.pad;
          a_voff[i] =
              (int)((((long)(img * p.H + hi0 + p.pad) * p.W + wi0 + p.pad) * p.lda + lk4) * 4);
          unsigned mask = 0;
          for (int r = 0; r < p.KH; ++r)
            for (int q = 0; q < p.KW; ++q)
              if ((unsigned)(hi0 + r) < (unsigned)p.H && (unsigned)(wi0 + q) < (unsigned)p.W)
                mask |= 1u << (r * p.KW + q);
          a_taps[i] = mask;
        }
      }
      b_voff = (brow < BN && n0 + brow < p.N) ? ((n0 + brow) * p.ldb + bchunk * 8) * 2 : BUF_OOB;
      u_r = u_q = u_ci = u_k = 0;
    };

    // Register ring of NSET staged K-tiles: NSET-1 tiles of loads are in flight while one is
    // transformed and written to LDS (8 producer waves per CU: 1 K-tile each = 8 in flight).  Loads past the last K-tile are issued out of range (the
    // hardware returns zeros, no traffic).
    constexpr int NSET = 2;  // (3 measured 1-4 % slower: the registers it costs spill)
    typedef X3Staged<A_ROWS, DUAL> Staged;
    Staged st[NSET];
    u32x4 sb[NSET][3];
    // The prologue vectors of a K-tile (its 32 input channels) are not part of the ring: one copy,
    // fetched right after the previous K-tile was written (they are L1/L2 hits: every row of
    // every workgroup reads the same few KB), a K-tile time before they are used.
    f32x4 ps = {0.f, 0.f, 0.f, 0.f}, pt = ps, pc = ps, p2s = {1.f, 1.f, 1.f, 1.f}, pt2 = ps, p2c = ps;
    int s_ci = 0;  // first input channel of the K-tile to be written next
    auto load_vec = [&]() {
      if (has_pro) {
        ps = ldg4(p.in_scale + s_ci + lk4);
        pt = ldg4(p.in_shift + s_ci + lk4);
        if (p.in_center) pc = ldg4(p.in_center + s_ci + lk4);
        if constexpr (DUAL) {
          if (p.in2_scale != nullptr) {
            p2s = ldg4(p.in2_scale + s_ci + lk4);
            pt2 = pt + ldg4(p.in2_shift + s_ci + lk4);
            if (p.in2_center) p2c = ldg4(p.in2_center + s_ci + lk4);
          }
        }
      }
    };

    auto load = [&](Staged& s, u32x4 (&b)[3], bool live) {
      const int tap = u_r * p.KW + u_q;
      const int soff = ((u_r * p.W + u_q) * p.lda + u_ci) * 4;
      if constexpr (DUAL) s.m0 = l_n0 == 0 ? l_m0 : -1;  // side_out rows, or -1: not this tile's job
      s.ok = 0;
#pragma unroll
      for (int i = 0; i < A_ROWS; ++i) {
        const unsigned ok = live ? ((a_taps[i] >> tap) & 1u) : 0u;
        s.ok |= ok << i;
        s.a[i] = __builtin_bit_cast(
            f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, ok ? a_voff[i] : BUF_OOB, soff, 0));
        if constexpr (DUAL)
          s.a2[i] = __builtin_bit_cast(
              f32x4,
              __builtin_amdgcn_raw_buffer_load_b128(rsrc_a2, ok ? a_voff[i] : BUF_OOB, soff, 0));
      }
#pragma unroll
      for (int q = 0; q < 3; ++q)
        b[q] = __builtin_amdgcn_raw_buffer_load_b128(
            rsrc_b, (live && b_voff != BUF_OOB) ? b_voff + q * plane_b : BUF_OOB, u_k * 2, 0);
      // advance the cursor by one K-tile; past the tile's last one, move to the next tile
      u_k += BK;
      u_ci += BK;
      if (u_ci >= p.Cin) {
        u_ci = 0;
        if (++u_q == p.KW) {
          u_q = 0;
          ++u_r;
        }
      }
      if (++l_kt == KT) {
        l_kt = 0;
        if (++l_round < my_tiles) setup_tile(l_round);
      }
    };

    auto stash = [&](const Staged& s, const u32x4 (&b)[3], char* stage) {
      char* arow = stage + lrow * X3_PITCH + lk4 * 2;
#pragma unroll
      for (int i = 0; i < A_ROWS; ++i) {
        f32x4 v = s.a[i];
        if (has_pro) {
          // (scalar fma / max per element: packed fp32 VALU beside MFMAs costs more issue time
          // than the two plain instructions it replaces -- MI355X_MICROARCH.md)
          if constexpr (DUAL) {
            if (p.in2_scale != nullptr) {  // downsample branch: its own BatchNorm (pt2 = pt + p2t)
#pragma unroll
              for (int e = 0; e < 4; ++e)
                v[e] = fmaxf(fmaf(v[e] - pc[e], ps[e], fmaf(s.a2[i][e] - p2c[e], p2s[e], pt2[e])),
                             relu_floor);
            } else {  // identity skip: added as is
#pragma unroll
              for (int e = 0; e < 4; ++e)
                v[e] = fmaxf(fmaf(v[e] - pc[e], ps[e], pt[e]) + s.a2[i][e], relu_floor);
            }
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(fmaf(v[e] - pc[e], ps[e], pt[e]), relu_floor);
          }
          // zero padding comes AFTER the transform (rows past M are never stored or counted)
          if (pad_matters && !((s.ok >> i) & 1u)) v = f32x4{0.f, 0.f, 0.f, 0.f};
          if constexpr (DUAL) {
            // the materialised block output: written once, by the workgroups of n-tile 0
            // (1x1 convolution: every tile of one tile_m row has the same rows)
            if (p.side_out != nullptr && s.m0 >= 0 && ((s.ok >> i) & 1u))
              *reinterpret_cast<f32x4*>(p.side_out + (long)(s.m0 + i * PR + lrow) * p.lda + s_ci +
                                        lk4) = v;
          }
        }
        x3_split_store<MATH>(v, arow + i * PR * X3_PITCH, A_PLANE);
      }
      if (brow < BN) {
        char* bdst = stage + A_BYTES + brow * X3_PITCH + bchunk * 16;
#pragma unroll
        for (int q = 0; q < 3; ++q) *reinterpret_cast<u32x4*>(bdst + q * B_PLANE) = b[q];
      }
    };

    // K-tile g of the stream goes to stage g & 1 once the matrix waves are done with K-tile
    // g - 2; iteration g: issue the loads of K-tile g + NSET - 1, write K-tile g
#ifdef X3_DBG_TIME
    long long d_pl = 0, d_pw = 0, d_pm = 0, d_ps = 0;
    const long long d_p0 = clock64();
#endif
    setup_tile(0);
    load_vec();
#pragma unroll
    for (int j = 0; j < NSET - 1; ++j) load(st[j], sb[j], j < G_total);
    for (int g0 = 0; g0 < G_total; g0 += NSET) {
#pragma unroll
      for (int j = 0; j < NSET; ++j) {
        const int g = g0 + j;
        if (g < G_total) {
#ifdef X3_DBG_TIME
          const long long d_0 = clock64();
#endif
          const int seen = x3_peek(empty + (g & 1));
          load(st[(j + NSET - 1) % NSET], sb[(j + NSET - 1) % NSET], g + NSET - 1 < G_total);
#ifdef X3_DBG_TIME
          const long long d_1 = clock64();
#endif
          x3_wait(empty + (g & 1), seen, X3_MATRIX * (g >> 1));  // K-tile g-2 has been read
#ifdef X3_DBG_TIME
          const long long d_2 = clock64();
          asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NSET - 1) * (A_ROWS * (1 + DUAL) + 3)) : "memory");
          const long long d_3 = clock64();
#endif
          stash(st[j], sb[j], xsm + (g & 1) * STAGE_BYTES);
          if (lane == 0) x3_signal(full + (g & 1));  // (in LDS order behind this wave's stage writes)
          s_ci += BK;
          if (s_ci >= p.Cin) s_ci = 0;
          if (g + 1 < G_total) load_vec();
#ifdef X3_DBG_TIME
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          const long long d_4 = clock64();
          d_pl += d_1 - d_0; d_pw += d_2 - d_1; d_pm += d_3 - d_2; d_ps += d_4 - d_3;
#endif
        }
      }
    }
#ifdef X3_DBG_TIME
    if (blockIdx.x == 8 && (tid == X3_MATRIX * 64 || tid == X3_MATRIX * 64 + 448))
      printf("x3 producer wave %d: total %lld: issue loads %lld, wait for matrix waves %lld, wait for "
             "data %lld, transform+write %lld\n", wave, (long long)(clock64() - d_p0), d_pl, d_pw, d_pm, d_ps);
#endif
  } else {
    // ================================================================ matrix waves
    const int wm = wave / WN, wn = wave % WN;
    struct Frag {
      bf16x8 a[MT][NA];
      bf16x8 b[NT][3];
    };
    const char* abase = xsm + (wm * WTM + l31) * X3_PITCH + half * 16;
    const char* bbase = xsm + A_BYTES + (wn * WTN + l31) * X3_PITCH + half * 16;
    auto read = [&](Frag& f, int stage, int slab) {
      const int off = stage * STAGE_BYTES + slab * 32;
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int q = 0; q < NA; ++q)
          f.a[i][q] = *reinterpret_cast<const bf16x8*>(abase + off + q * A_PLANE + i * 32 * X3_PITCH);
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int q = 0; q < 3; ++q)
          f.b[j][q] = *reinterpret_cast<const bf16x8*>(bbase + off + q * B_PLANE + j * 32 * X3_PITCH);
    };
    f32x16 acc[MT][NT];
    auto mma = [&](const Frag& f) {
#pragma unroll
      for (int q = 0; q < PL::NP; ++q)
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j)
            acc[i][j] = plane_mfma<MATH>(f.a[i][PL::PA[q]], f.b[j][PL::PB[q]], acc[i][j]);
    };
    const __amdgpu_buffer_rsrc_t rsrc_c = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<char*>(p.C), 0, (int)p.c_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_r = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(p.residual ? p.residual : p.C)), 0,
        (int)p.c_bytes, 0x00020000);   // (ldr == ldc: the output's extent)

#ifdef X3_DBG_TIME
    const long long d_c0 = clock64(), d_w0 = wall_clock64();
    long long d_wait = 0;
#endif
    Frag fa, fb;
    WaveBn<NT> wbn;   // BatchNorm finished in this launch (p.bn): the wave's running column sums
    wave_bn_reset(wbn);
    // the SIMD's VALU issue port is shared with the producer waves: the MFMAs must win it the
    // moment the matrix pipe frees up, the producers take the slots in between
    __builtin_amdgcn_s_setprio(3);
    x3_wait(full, x3_peek(full), X3_PRODUCERS);  // K-tile 0 is written (stage 0's first)
    read(fa, 0, 0);
    int g = 0;
#ifdef X3_DBG_TIME
    const long long d_c1 = clock64();
#endif
    for (int round = 0; round < my_tiles; ++round) {
      int m0, n0;
      tile_of(round, m0, n0);
      // epilogue vectors of this wave's columns (loaded now, used after the K loop)
      float e_sc[NT], e_sh[NT];
      int e_voff[NT];
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int col = n0 + wn * WTN + j * 32 + l31;
        const bool ok = col < p.N;
        e_sc[j] = ((ok && p.scale) ? p.scale[col] : 1.f) * PL::POST;
        e_sh[j] = (ok && p.shift) ? p.shift[col] : 0.f;
        e_voff[j] = ok ? (int)((((long)(m0 + wm * WTM + 4 * half)) * p.ldc + col) * 4) : BUF_OOB;
      }
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

      for (int t = 0; t < KT; ++t, ++g) {
        // (the scheduling fences keep the LDS reads of the NEXT slab in front of the current
        // slab's MFMAs; left alone the compiler sinks them behind and the pipe waits on LDS)
        const int seen = x3_peek(full + ((g + 1) & 1));
        read(fb, g & 1, 1);
        __builtin_amdgcn_sched_barrier(0);
        mma(fa);
        __builtin_amdgcn_sched_barrier(0);
        if (g + 1 < G_total) {
#ifdef X3_DBG_TIME
          const long long d_a = clock64();
#endif
          x3_wait(full + ((g + 1) & 1), seen, X3_PRODUCERS * (((g + 1) >> 1) + 1));  // K-tile g+1 is written
#ifdef X3_DBG_TIME
          d_wait += clock64() - d_a;
#endif
          read(fa, (g + 1) & 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        // this wave's reads of K-tile g are complete once fb has arrived (LDS returns in order)
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(NA * MT + 3 * NT) : "memory");
        if (lane == 0) x3_signal(empty + (g & 1));
        __builtin_amdgcn_sched_barrier(0);
        mma(fb);
        __builtin_amdgcn_sched_barrier(0);
      }

      // -------------------------------------------------------------- statistics
      if (p.bn.acc != nullptr) {
        wave_bn_tile<MT, NT>(acc, wbn, p.bn.acc, n0 + wn * WTN, p.N, p.M - (m0 + wm * WTM), half, l31,
                             PL::POST);
        if (round == my_tiles - 1) wave_bn_flush(wbn, p.bn.acc, p.N, half, l31);  // in front of the stores
      } else if (p.stat_partial != nullptr) {
        const int tile_m = m0 / BM;
        if (p.stat_rows == 32 && MT > 1) {
          // one partial per 32x32 MFMA block row: exactly the 16 accumulator registers of a lane
#pragma unroll
          for (int i = 0; i < MT; ++i)
            wave_stats_block<NT>(acc[i], p.stat_partial, (m0 + wm * WTM) / 32 + i,
                                 p.M - (m0 + wm * WTM + i * 32), n0 + wn * WTN, p.N, half, l31,
                                 PL::POST);
        } else if (p.stat_rows > 0 && p.stat_rows < WTM)
          wave_stats_fine<MT, NT>(acc, p.stat_partial, p.stat_rows, m0 + wm * WTM, p.M,
                                  n0 + wn * WTN, p.N, half, l31, PL::POST);
        else
          wave_stats<MT, NT>(acc, p.stat_partial, tile_m * WM + wm, p.M - (m0 + wm * WTM), WTM,
                             n0 + wn * WTN, p.N, half, l31, PL::POST);
      }
      // -------------------------------------------------------------- epilogue from registers
      // one store = 2 rows x 32 columns = two full 128-byte lines; rows past M fall outside
      // the buffer (the row bound is folded into the descriptor's extent) only for the last
      // tile row, where the lane offset is sent out of range instead
      const int rows_left = p.M - (m0 + wm * WTM + 4 * half);
      wave_epilogue<MT, NT>(acc, e_sc, e_sh, e_voff, rows_left, p.ldc, p.act, p.residual != nullptr,
                            rsrc_c, rsrc_r, false);
    }
    __builtin_amdgcn_s_setprio(0);
#ifdef X3_DBG_TIME
    if (blockIdx.x == 8 && tid == 0) {
      const long long c = clock64() - d_c0, w = wall_clock64() - d_w0;
      printf("x3 tiles %d KT %d: first K-tile after %lld cycles; total %lld cycles = %lld ticks of 100 MHz "
             "(%.2f GHz); waiting for producers %lld\n", my_tiles, KT, d_c1 - d_c0, c, w,
             (double)c / (double)w * 0.1, d_wait);
    }
#endif
  }
#endif
}

// statistics partials {sum, M2 about the block mean} of a finished [M, N] output, per block of
// `rows` output pixels and column: the second pass of a split-K convolution that feeds a
// BatchNorm / GroupNorm (same layout as the convolution kernels' epilogue partials)
__global__ __launch_bounds__(256) void rows_stats_kernel(const float* __restrict__ y, int ldc, int M,
                                                         int N, int rows,
                                                         float* __restrict__ partial) {
  const int col = blockIdx.y * 64 + (threadIdx.x & 63);
  const int blk = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int r0 = blk * rows;
  if (col >= N || r0 >= M) return;
  const int n = min(rows, M - r0);
  const float* q = y + (long)r0 * ldc + col;
  float v[32];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 32; ++i) {
    v[i] = i < n ? q[(long)i * ldc] : 0.f;
    s += v[i];
  }
  const float mean = s / (float)n;
  float m2 = 0.f;
#pragma unroll
  for (int i = 0; i < 32; ++i) {
    const float d = v[i] - mean;
    if (i < n) m2 += d * d;
  }
  float* dst = partial + ((long)blk * N + col) * 2;
  dst[0] = s;
  dst[1] = m2;
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
  const int force = vlnce_opt(VLNCE_OPT_IGEMM_TILE);
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
  const bool off = vlnce_opt(VLNCE_OPT_IGEMM_NOBUF) != 0;
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
int choose_splitk(const IgemmParams& p, bool long_reduction_form = false) {
  const bool off = vlnce_opt(VLNCE_OPT_IGEMM_NO_SPLITK) != 0;  // diagnostic switch
  if (off) return 1;
  const long tiles = (long)ceil_div(p.M, 64) * ceil_div(p.N, 64);
  const int KT = ceil_div(p.K, BK);
  // dW = dz^T x of a sequence-mode batch (reduction over T*N*P rows, e.g. 512 x 2112 x 8000 for
  // rgb_kv at 500 rows x 16 positions): a few hundred tiles with one workgroup per CU walk a
  // 250-step K loop with nothing to hide its latency behind -- four workgroups per CU as in
  // vlnce_conv2d_wgrad (364 -> ~190 us there)
  if (long_reduction_form && KT >= 32 && tiles >= 128 && tiles < 1024) {
    long s = (1024 + tiles - 1) / tiles;
    if (s > KT / 4) s = KT / 4;
    return s < 2 ? 1 : (int)s;
  }
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
  p.bn = vlnce_bn_sums{};
  if (e && e->bn) p.bn = *e->bn;
}

bool aligned16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }

// conv_x3_kernel launch: one workgroup (16 or 12 waves) per CU, walking tiles
template <int BM, int BN, int WM, int WN, int DUAL, int MATH>
int launch_x3(const IgemmParams& p, hipStream_t stream) {
  constexpr int smem_bytes = 2 * (Planes<MATH>::NA * BM + 3 * BN) * X3_PITCH + 16;  // two stages + 4 counters
  constexpr int threads = (WM * WN + X3_PRODUCERS) * 64;
  auto kern = conv_x3_kernel<BM, BN, WM, WN, DUAL, MATH>;
  static bool attr_set = false;  // per instantiation
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, smem_bytes);
    if (e != hipSuccess) {
      vlnce_set_error("conv_x3: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
      return 2;
    }
    attr_set = true;
  }
  IgemmParams q = p;
  q.tiles_m = ceil_div(p.M, BM);
  q.tiles_n = ceil_div(p.N, BN);
  q.splitk = 1;
  const long nwg = (long)q.tiles_m * q.tiles_n;
  if (nwg <= 0 || nwg > 0x7fffffffL) {
    vlnce_set_error("conv_x3: bad grid %ld", nwg);
    return 1;
  }
  // one workgroup per CU walks tiles a grid apart (a multiple of 8: the XCD of a tile is fixed)
  const int cus = x3_cus();
  const unsigned grid = nwg <= cus ? (unsigned)nwg : (unsigned)cus;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), smem_bytes, stream, q);
  VLNCE_CHECK_LAUNCH("conv_x3");
  return 0;
}

// what conv_x3_kernel covers: the buffer-descriptor hot path (buf_ok) with whole 32-channel
// K-tiles and a plain epilogue (scale / shift / activation, statistics).  Tile: the largest of
// 128x128, 64x128, 128x64, 64x64 that keeps >= 80 % of the CUs busy over the rounds a workgroup
// per CU needs (tiles / (rounds * CUs)).
struct X3Plan {
  int bm, bn;
};
bool x3_plan(const IgemmParams& p, X3Plan* out) {
  if (!conv_math() || !p.Bsplit || !buf_ok(p) || p.splitk > 1) return false;
  if (p.accumulate || p.c_bytes >= 0x7fffffffL) return false;
  if (p.residual && (p.ldr != p.ldc || p.stat_partial || p.bn.acc)) return false;  // (register epilogue)
  const int force = vlnce_opt(VLNCE_OPT_X3_TILE);  // tuning
  const X3Plan cand[4] = {{128, 128}, {64, 128}, {128, 64}, {64, 64}};
  if (force >= 1 && force <= 4) {
    *out = cand[force - 1];
    return true;
  }
  const int cus = x3_cus();
  double best = 0.0;
  for (const X3Plan& c : cand) {
    if (c.bn == 128 && p.N <= 64) continue;
    const long tiles = (long)ceil_div(p.M, c.bm) * ceil_div(p.N, c.bn);
    const long rounds = (tiles + cus - 1) / cus;
    const double eff = (double)tiles / (double)(rounds * cus);
    if (eff >= 0.8) {
      *out = c;
      return true;
    }
    if (eff > best) {
      best = eff;
      *out = c;
    }
  }
  return best >= 0.4;  // below that the problem is a handful of tiles: split-K / small-tile path
}
template <int DUAL, int MATH>
int dispatch_x3_(const IgemmParams& p, const X3Plan& t, hipStream_t s) {
  if (t.bm == 128 && t.bn == 128) return launch_x3<128, 128, 2, 4, DUAL, MATH>(p, s);
  if (t.bm == 64 && t.bn == 128) return launch_x3<64, 128, 2, 4, DUAL, MATH>(p, s);
  if (t.bm == 128 && t.bn == 64) return launch_x3<128, 64, 4, 2, DUAL, MATH>(p, s);
  return launch_x3<64, 64, 2, 2, DUAL, MATH>(p, s);
}
template <int DUAL>
int dispatch_x3(const IgemmParams& p, const X3Plan& t, hipStream_t s) {
  return p.math == MATH_F16X3 ? dispatch_x3_<DUAL, MATH_F16X3>(p, t, s)
                              : dispatch_x3_<DUAL, MATH_BF16X6>(p, t, s);
}

}  // namespace

// rows per statistics partial: 32 output pixels (the M extent of the smallest wave sub-tile of
// any convolution kernel here), 16 when a sample's pixel count is not a multiple of 32
// (GroupNorm needs partials that do not straddle samples; habitat's depth trunk ends at 4x4 = 16
// pixels per sample).  A function of the descriptor only: the caller sizes the partial buffer
// before it knows which kernel runs.
static int stat_rows_for(const vlnce_conv_desc* d) {
  const int hw = d->Ho * d->Wo;
  return hw % 32 == 0 ? 32 : 16;
}

extern "C" int vlnce_conv2d_tile_rows(const vlnce_conv_desc* d) { return stat_rows_for(d); }

extern "C" int vlnce_conv2d_tiles_m(const vlnce_conv_desc* d) {
  const long M = (long)d->N * d->Ho * d->Wo;
  return ceil_div(M, stat_rows_for(d));
}

// w[i] -> planes[q][i], q = 0..2: the exact three-way (round-to-nearest) bf16 split of
// conv_x3_kernel's B operand
template <int MATH>
__global__ __launch_bounds__(256) void split_weights_kernel(const float* __restrict__ w,
                                                            unsigned short* __restrict__ planes,
                                                            long count) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < count; i += gridDim.x * 256L) {
    unsigned short o[3];   // MATH_BF16X6: w == plane0 + plane1 + plane2 exactly
    split_weight<MATH>(w[i], o);
#pragma unroll
    for (int q = 0; q < 3; ++q) planes[q * count + i] = o[q];
  }
}

extern "C" int vlnce_conv2d_split_weights(const float* w, void* planes, long count, int format,
                                          vlnce_stream_t stream) {
  VLNCE_CHECK_ARG(w && planes && count > 0, "conv2d_split_weights: bad argument");
  VLNCE_CHECK_ARG(format == MATH_BF16X6 || format == MATH_F16X3,
                  "conv2d_split_weights: format must be 1 (three bf16 planes) or 2 (fp16 planes)");
  const long blocks = (count + 255) / 256;
  hipLaunchKernelGGL(format == MATH_F16X3 ? split_weights_kernel<MATH_F16X3>
                                          : split_weights_kernel<MATH_BF16X6>,
                     dim3((unsigned)(blocks > 4096 ? 4096 : blocks)),
                     dim3(256), 0, reinterpret_cast<hipStream_t>(stream), w,
                     reinterpret_cast<unsigned short*>(planes), count);
  VLNCE_CHECK_LAUNCH("conv2d_split_weights");
  return 0;
}

// {sum x, sum x^2} per channel (fp64, added by the convolution's workgroups) -> the pending
// normalisation and the running statistics; leaves the sums zero for the next launch.  One
// workgroup per 256 channels: ~2 us + a launch boundary behind the convolution instead of
// finalize (+ coarsen) over thousands of tile moments.
__global__ __launch_bounds__(256) void bn_sums_finalize_kernel(
    double* __restrict__ acc, int M, int N, const float* __restrict__ gamma,
    const float* __restrict__ beta, float eps, float momentum, float* running_mean,
    float* running_var, float* scale_out, float* shift_out, float* mean_out, float* rstd_out) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= N) return;
  double S = 0.0, Q = 0.0;
#pragma unroll
  for (int k = 0; k < VLNCE_BN_SHARDS; ++k) {
    double* q = acc + ((long)k * N + c) * 2;
    S += q[0];
    Q += q[1];
    q[0] = 0.0;
    q[1] = 0.0;
  }
  const double mean = S / (double)M;
  double m2 = Q - S * mean;           // sum (x - mean)^2
  if (m2 < 0.0) m2 = 0.0;
  const double var = m2 / (double)M;  // biased, used for normalisation
  const float rstd = (float)(1.0 / sqrt(var + (double)eps));
  const float g = gamma ? gamma[c] : 1.f;
  const float sc = g * rstd;
  scale_out[c] = sc;
  if (shift_out) shift_out[c] = (beta ? beta[c] : 0.f) - (float)mean * sc;
  if (mean_out) mean_out[c] = (float)mean;
  if (rstd_out) rstd_out[c] = rstd;
  if (running_mean) {
    const double unbiased = M > 1 ? m2 / (double)(M - 1) : var;
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
  }
}

// tile moments {sum, M2 about the tile mean} of a kernel without the sums epilogue -> the same sums:
// thread = (strip of 32 tiles, channel), channels fastest (coalesced 8-byte reads)
__global__ __launch_bounds__(256) void bn_partials_to_sums_kernel(const float* __restrict__ partial,
                                                                  int tiles_m, int tile_rows, int M,
                                                                  int N, double* __restrict__ acc) {
  const int c = blockIdx.y * 64 + (threadIdx.x & 63);
  const int t0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 32;
  if (c >= N || t0 >= tiles_m) return;
  double S = 0.0, Q = 0.0;
  for (int t = t0; t < min(t0 + 32, tiles_m); ++t) {
    const float2 v = *reinterpret_cast<const float2*>(partial + ((long)t * N + c) * 2);
    const int nt = min(tile_rows, M - t * tile_rows);
    S += (double)v.x;
    Q += (double)v.y + (double)v.x * (double)v.x / (double)nt;
  }
  double* q = acc + ((long)(blockIdx.x % VLNCE_BN_SHARDS) * N + c) * 2;
  unsafeAtomicAdd(q, S);
  unsafeAtomicAdd(q + 1, Q);
}

// vlnce_bn_sums.workspace: the tile moments of a convolution kernel WITHOUT the sums epilogue
extern "C" long vlnce_conv2d_bn_workspace_bytes(const vlnce_conv_desc* d) {
  if (!d || d->Cout <= 0) return 0;
  return (((long)vlnce_conv2d_tiles_m(d) * d->Cout * 2 * 4) + 255) / 256 * 256;
}

extern "C" int vlnce_bn_finalize_sums(double* acc, int M, int C, const float* gamma, const float* beta,
                                      float eps, float momentum, float* running_mean,
                                      float* running_var, float* scale_out, float* shift_out,
                                      float* mean_out, float* rstd_out, vlnce_stream_t stream) {
  VLNCE_CHECK_ARG(acc && scale_out && M > 0 && C > 0, "bn_finalize_sums: bad argument");
  VLNCE_CHECK_ARG((running_mean == nullptr) == (running_var == nullptr),
                  "bn_finalize_sums: running stats must come together");
  hipLaunchKernelGGL(bn_sums_finalize_kernel, dim3(ceil_div(C, 256)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), acc, M, C, gamma, beta, eps, momentum,
                     running_mean, running_var, scale_out, shift_out, mean_out, rstd_out);
  VLNCE_CHECK_LAUNCH("bn_finalize_sums");
  return 0;
}

// which kernel the calling thread's last vlnce_conv2d_fwd went to (bench.py prices the bf16-pipe
// launches and the fp32-MFMA launches against their own peaks)
static thread_local int g_last_path = -1;
extern "C" int vlnce_conv2d_last_path(void) { return g_last_path; }

extern "C" int vlnce_conv2d_fwd(const float* x, const float* w, float* y, const vlnce_conv_desc* d,
                                const vlnce_prologue* pro, const vlnce_epilogue* epi,
                                vlnce_stream_t stream) {
  VLNCE_CHECK_ARG(x && w && y && d, "conv2d_fwd: null argument");
  VLNCE_CHECK_ARG(d->N > 0 && d->H > 0 && d->W > 0 && d->Cin > 0 && d->Cout > 0,
                  "conv2d_fwd: bad shape");
  const VlnceOptScope opt_scope(pro ? pro->options : nullptr);   // this launch's dispatch options
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
  p.Bsplit = pro ? pro->w_split : nullptr;
  p.Bfrag = pro ? pro->w_frag : nullptr;
  p.math = (pro && pro->w_format) ? pro->w_format : MATH_BF16X6;
  VLNCE_CHECK_ARG(p.math == MATH_BF16X6 || p.math == MATH_F16X3,
                  "conv2d_fwd: w_format must be 0 / 1 (three bf16 planes) or 2 (fp16 planes)");
  VLNCE_CHECK_ARG((p.in_scale == nullptr) == (p.in_shift == nullptr),
                  "conv2d_fwd: in_scale and in_shift must come together");
  fill_epilogue(p, epi);
  // ---- train-mode BatchNorm statistics added by the launch (vlnce_bn_sums).  The kernels
  // without that epilogue (fp32-MFMA kernel, split-K) write tile moments into the workspace
  // instead and a reduction kernel behind them adds those to the sums: same result for the caller.
  const vlnce_bn_sums* const bnf = epi ? epi->bn : nullptr;
  if (bnf != nullptr) {
    VLNCE_CHECK_ARG(bnf->acc != nullptr, "conv2d_fwd: bn needs acc");
    VLNCE_CHECK_ARG(!p.stat_partial && !p.scale && !p.shift && !p.residual && !p.act && !p.accumulate,
                    "conv2d_fwd: bn excludes stat_partial / scale / shift / residual / act / accumulate");
    VLNCE_CHECK_ARG(bnf->workspace && bnf->workspace_bytes >= vlnce_conv2d_bn_workspace_bytes(d) &&
                        aligned16(bnf->workspace),
                    "conv2d_fwd: bn workspace too small (vlnce_conv2d_bn_workspace_bytes)");
  }
  // hand the statistics to the tile-moment path (for a kernel without the sums epilogue) ...
  auto bn_to_partials = [&]() {
    if (bnf != nullptr) {
      p.bn = vlnce_bn_sums{};
      p.stat_partial = static_cast<float*>(bnf->workspace);
    }
  };
  auto bn_sums_behind = [&](int rc) -> int { return rc; };   // the kernel added the sums itself
  // ... and reduce the moments into the sums behind the convolution
  auto bn_finalize_behind = [&](int rc) -> int {
    if (rc != 0 || bnf == nullptr) return rc;
    const int tiles = vlnce_conv2d_tiles_m(d);
    hipLaunchKernelGGL(bn_partials_to_sums_kernel, dim3(ceil_div(tiles, 128), ceil_div(d->Cout, 64)),
                       dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       static_cast<const float*>(bnf->workspace), tiles, stat_rows_for(d), (int)M,
                       d->Cout, bnf->acc);
    VLNCE_CHECK_LAUNCH("conv2d_fwd bn moments -> sums");
    return 0;
  };
  const bool v4 = (d->Cin % 4 == 0) && (p.lda % 4 == 0) && aligned16(x) && aligned16(w) &&
                  (!p.in_scale || (aligned16(p.in_scale) && aligned16(p.in_shift))) &&
                  (!p.in_center || aligned16(p.in_center));
  p.splitk = 1;
  p.a_bytes = (((long)d->N * d->H * d->W - 1) * p.lda + d->Cin) * 4;
  p.b_bytes = (long)d->Cout * p.K * 4;
  p.c_bytes = ((M - 1) * p.ldc + d->Cout) * 4;
  p.stat_rows = stat_rows_for(d);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  g_last_path = VLNCE_CONV_PATH_F32;
  if (p.A2 != nullptr || p.side_out != nullptr) {
    VLNCE_CHECK_ARG(p.A2 && p.in_scale && d->KH == 1 && d->KW == 1 && d->stride == 1 &&
                        d->pad == 0 && v4 && buf_ok(p) && aligned16(p.A2) &&
                        (!p.side_out || aligned16(p.side_out)) &&
                        ((p.in2_scale == nullptr) == (p.in2_shift == nullptr)) &&
                        (!p.in2_scale || (aligned16(p.in2_scale) && aligned16(p.in2_shift))) &&
                        (!p.in2_center || (p.in2_scale && aligned16(p.in2_center))),
                    "conv2d_fwd: the dual-input prologue needs x2 + in_scale on a 1x1/stride-1/"
                    "pad-0 convolution with Cin %% 32 == 0 and 16-byte aligned operands");
    g_last_path = VLNCE_CONV_PATH_P3;
    if (const int rc = p3_try_launch(p, s); rc >= 0) return bn_sums_behind(rc);
    g_last_path = VLNCE_CONV_PATH_X3;
    if (X3Plan t; x3_plan(p, &t)) return bn_sums_behind(dispatch_x3<1>(p, t, s));
    g_last_path = VLNCE_CONV_PATH_F32;
    bn_to_partials();
    return bn_finalize_behind(dispatch_dual(p, s));
  }
  if (v4 && buf_ok(p)) {
    // small launches (the depth trunk, any layer at a few environments): conv_m3_kernel
    g_last_path = VLNCE_CONV_PATH_M3;
    if (const int rc = m3_try_launch(p, s); rc >= 0) return bn_sums_behind(rc);
    g_last_path = VLNCE_CONV_PATH_F32;
    // Small batches (act() at num_envs 1..8, eval BatchNorm folded into scale/shift): a late
    // ResNet layer is a handful of 64x64 tiles with a reduction of up to 144 K-tiles -- one
    // workgroup walking them alone is pure latency (20-50 us per layer).  Split the reduction
    // over blockIdx.y with atomic accumulation into a zeroed output and apply the epilogue
    // (scale/shift/residual/activation) in a second, row-wise pass.
    const int sk = (!p.accumulate && p.ldc == p.N && p.stat_rows <= 32 &&
                    ((p.scale && p.shift) || (!p.scale && !p.residual)))
                       ? choose_splitk(p)
                       : 1;
    if (sk > 1) {
      bn_to_partials();
      const float* scale = p.scale;
      const float* shift = p.shift;
      const float* residual = p.residual;
      float* stat_partial = p.stat_partial;
      const int act = p.act, ldr = p.ldr;
      p.scale = p.shift = p.residual = nullptr;
      p.stat_partial = nullptr;
      p.act = 0;
      p.splitk = sk;
      vlnce_zero(y, M, p.N, p.ldc, s);
      if (int rc = dispatch_small<A_BUF, B_BUF>(p, s)) return rc;
      if (stat_partial) {  // statistics of the RAW sums, as the one-pass kernels' epilogues take them
        const int nblk = ceil_div((int)M, p.stat_rows);
        hipLaunchKernelGGL(rows_stats_kernel, dim3(ceil_div(nblk, 4), ceil_div(p.N, 64)), dim3(256),
                           0, s, y, p.ldc, (int)M, p.N, p.stat_rows, stat_partial);
        VLNCE_CHECK_LAUNCH("conv2d_fwd split-K statistics");
      }
      if (bnf != nullptr) return bn_finalize_behind(0);
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
    g_last_path = VLNCE_CONV_PATH_P3;
    if (const int rc = p3_try_launch(p, s); rc >= 0) return bn_sums_behind(rc);
    g_last_path = VLNCE_CONV_PATH_X3;
    if (X3Plan t; x3_plan(p, &t)) return bn_sums_behind(dispatch_x3<0>(p, t, s));
    g_last_path = VLNCE_CONV_PATH_F32;
    bn_to_partials();
    return bn_finalize_behind(dispatch_tiles<A_BUF, B_BUF>(p, s));
  }
  bn_to_partials();
  if (v4) return bn_finalize_behind(dispatch_tiles<A_IM2COL_V4, B_NK_V4>(p, s));
  if (d->KW == 7 && d->Cin == 3) return bn_finalize_behind(dispatch_stem<3>(p, s));
  if (d->KW == 7 && d->Cin == 1) return bn_finalize_behind(dispatch_stem<1>(p, s));
  return bn_finalize_behind(dispatch_tiles<A_IM2COL_S, B_NK_S>(p, s));
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
  if (plain) p.splitk = choose_splitk(p, transA != 0);
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
                                  const vlnce_conv_desc* d, const float* dy_pow2, int P,
                                  int accumulate, vlnce_stream_t stream) {
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
  VLNCE_CHECK_ARG(!dy_pow2 || P > 0, "conv2d_wgrad: dy_pow2 needs P > 0");
  fill_epilogue(p, nullptr);
  // round 6: three bf16 planes on the 16-bit pipe (wgrad_x6_kernel) where it covers the layer;
  // option "wgrad_tile" = 1 keeps every layer on the fp32-MFMA kernel below (A/B)
  if (vlnce_opt(VLNCE_OPT_WGRAD_TILE) != 1)
    if (const int rc = wgrad_x6_try_launch(x, dy, dw_ohwi, d, dy_pow2, dy_pow2 ? dy_pow2 + P : nullptr,
                                           accumulate, reinterpret_cast<hipStream_t>(stream)); rc >= 0)
      return rc;
  // option "wgrad_tile" = 128: 128x128 tiles where both output dimensions allow.  Measured slower on
  // the trainable-encoder step (46.7 vs 45.0 ms, profiles/archive/r03_g_*): fewer workgroups per
  // split-K slice, and the transposed-operand LDS writes do not get cheaper.  Default 64.
  const int tile_pref = vlnce_opt(VLNCE_OPT_WGRAD_TILE);
  const bool big = tile_pref >= 128 && p.M >= 128 && p.N >= 128;
  const int T = big ? 128 : 64;
  const long tiles = (long)ceil_div(p.M, T) * ceil_div(p.N, T);
  const int KT = ceil_div(p.K, BK);
  long sk = ((big ? 512 : 1024) + tiles - 1) / tiles;
  if (sk > KT / 4) sk = KT / 4;
  if (sk > 512) sk = 512;
  p.splitk = sk < 2 ? 1 : (int)sk;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (p.splitk > 1) {
    if (!accumulate) vlnce_zero(dw_ohwi, 1, p.M * p.N, (long)p.M * p.N, s);
  } else if (accumulate) {
    p.accumulate = 1;
  }
  if (big) return launch<128, 128, 2, 2, A_TRANS, B_IM2COL>(p, s);
  return launch<64, 64, 2, 2, A_TRANS, B_IM2COL>(p, s);
}
