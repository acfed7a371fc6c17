// conv_p3_kernel: fp32 convolution on the bf16 matrix pipe with a PATCH-RESIDENT A operand.
//
// Same arithmetic as conv_x3_kernel (igemm.hip): every fp32 operand is split exactly into three
// bf16 planes and a product is the six plane products of order <= 2^-16 on
// v_mfma_f32_32x32x16_bf16 with fp32 accumulation -- here with a round-to-nearest split
// (v_cvt_pk_bf16_f32), so the dropped cross terms are <= 2^-26 relative.
//
// What round 2's PMC passes showed for conv_x3_kernel: the matrix waves wait for the producer
// waves about half the time, because the operand transform (previous layer's BatchNorm + ReLU,
// zero padding, three-way split, LDS writes: ~40 VALU instructions per float4) is redone for
// every filter tap (9x for a 3x3 convolution) and for every N-tile, and it shares the SIMD's issue
// port with the MFMAs.  This kernel removes both redundancies and the B operand's LDS round trip:
//   * A: the workgroup owns BM consecutive output pixels.  For a stride-1 convolution their
//     receptive field is ONE contiguous range of the zero-padded input in padded-linear pixel
//     order u = (img * Hp + hp) * Wp + wp:  tap (r, q) of output pixel m reads u0(m) + r * Wp + q.
//     Per 32-channel chunk the producers transform that range ONCE into an LDS patch
//     (rows = padded pixels incl. the zero halo, 208 bytes = 3 planes x 32 bf16 + 16 pad), and
//     the matrix waves read their A fragments for every tap from it with a wave-uniform row
//     offset -- implicit im2col out of LDS: 2.1x instead of 9x transformed elements at 64x64,
//     1.4x at 16x16.  1x1 convolutions (any stride) are the degenerate case: patch row = output
//     pixel (GATHER mode).
//   * B: the weights are pre-packed once per parameter version into MFMA fragment order
//     (vlnce_conv2d_pack_weights: [n-block][k-slab][plane][lane][8 bf16], k ordered
//     chunk-major / tap / 16-slab to match the loop here), so a matrix wave fetches a B fragment
//     with one fully coalesced 1 KB buffer_load_dwordx4 straight into registers, one slab
//     ahead -- no LDS bytes, no producer work, no hand-over for B.  The weight set of a layer
//     lives in L2 / MALL.
//   * wide tiles (BN up to 256) where the layer has the channels: the transform per MFMA halves
//     again for the 1x1 convolutions.
// A workgroup is 8 matrix waves (two per SIMD taking turns on the matrix pipe) + 4 producer waves
// (one per SIMD), 168 VGPRs each, one workgroup per CU, persistent over a strided tile list; the
// hand-over is per patch buffer (two of them) through LDS counters as in conv_x3_kernel.
#include "igemm_shared.h"

using namespace vlnce_detail;

namespace vlnce_detail {
namespace {

constexpr int P3_ROW = 208;      // bytes per patch row (13 x 16 B: consecutive rows are conflict-free)
constexpr int P3_PRODUCERS = 4;  // producer waves
constexpr int P3_RING = 4;       // producer items (32 rows x 32 channels) in flight

enum { P3_GATHER = 0, P3_DENSE = 1 };

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

// Wait for an LDS counter of the hand-over.  Waits are between the waves of ONE workgroup, i.e.
// microseconds; a protocol bug must become a launch error, not a hung GPU: after ~0.2 s of
// spinning the wave traps.
__device__ __forceinline__ void p3_wait(const int* flag, int need) {
  int seen = x3_peek(flag);
  for (int spins = 0; seen < need; ++spins) {
    __builtin_amdgcn_s_sleep(1);
    seen = x3_peek(flag);
    if (spins > (1 << 22)) __builtin_trap();
  }
  asm volatile("" ::: "memory");
}

// x (4 consecutive k of one patch row) -> the three planes' 8-byte words, round-to-nearest split
__device__ __forceinline__ void p3_split_store(f32x4 x, char* row_ptr) {
  u32x2 w[3];
#pragma unroll
  for (int pr = 0; pr < 2; ++pr) {
    f32x2 v = {x[2 * pr], x[2 * pr + 1]};
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const unsigned hb = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
      w[q][pr] = hb;
      if (q < 2) {
        v[0] -= __builtin_bit_cast(float, hb << 16);
        v[1] -= __builtin_bit_cast(float, hb & 0xffff0000u);
      }
    }
  }
#pragma unroll
  for (int q = 0; q < 3; ++q) *reinterpret_cast<u32x2*>(row_ptr + q * 64) = w[q];
}

template <int BM, int BN, int WM, int WN, int DUAL, int MODE>
__global__ __launch_bounds__((WM * WN + P3_PRODUCERS) * 64) void conv_p3_kernel(IgemmParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int MATRIX = WM * WN;
  constexpr int WTM = BM / WM, WTN = BN / WN, MT = WTM / 32, NT = WTN / 32;
  static_assert(MT >= 1 && NT >= 1 && WTM % 32 == 0 && WTN % 32 == 0, "tile");
  static_assert(!DUAL || MODE == P3_GATHER, "dual-input prologue: 1x1 convolutions only");

  extern __shared__ __attribute__((aligned(16))) char xsm[];
  const int pbuf = p.p3_rows * P3_ROW;                         // bytes of one patch buffer
  int* const pfull = reinterpret_cast<int*>(xsm + 2 * pbuf);   // [2] producer waves done writing
  int* const pempty = pfull + 2;                               // [2] matrix waves done reading

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5;
  const int l31 = lane & 31;
  const int NC = p.Cin / 32;      // channel chunks
  const int T = p.KH * p.KW;      // filter taps
  const int HoWo = p.Ho * p.Wo;
  const int Hp = p.H + 2 * p.pad, Wp = p.W + 2 * p.pad;

  // ---- this workgroup's tiles: virtual block ids blockIdx.x + r * gridDim.x through the
  // XCD-aware map of igemm_kernel (gridDim.x is a multiple of 8 or the whole tile count)
  const int ntiles = p.tiles_m * p.tiles_n;
  const int my_tiles = (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
  auto tile_of = [&](int round, int& m0, int& n0) {
    const int v = blockIdx.x + round * gridDim.x;
    const int q = ntiles >> 3, r = ntiles & 7, xcd = v & 7, idx = v >> 3;
    const int tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    const int tm = tile / p.tiles_n;
    m0 = tm * BM;
    n0 = (tile - tm * p.tiles_n) * BN;
  };
  // padded-linear index of output pixel m's tap (0, 0)
  auto u0_of = [&](int m) {
    const int img = m / HoWo;
    const int rem = m - img * HoWo;
    const int ho = rem / p.Wo;
    return (img * Hp + ho) * Wp + (rem - ho * p.Wo);
  };
  // rows of the patch of the tile at m0 (DENSE: padded pixels; GATHER: output pixels)
  auto patch_rows = [&](int m0) {
    const int mlast = min(m0 + BM, p.M) - 1;
    if constexpr (MODE == P3_DENSE) return u0_of(mlast) - u0_of(m0) + (p.KH - 1) * Wp + p.KW;
    return mlast - m0 + 1;
  };

  if (tid < 4) pfull[tid] = 0;
  __syncthreads();

  if (wave >= MATRIX) {
    // ================================================================ producer waves
    const int ptid = tid - MATRIX * 64;
    const int lrow = ptid >> 3;          // row inside a 32-row pass
    const int lk4 = (ptid & 7) * 4;      // first of this thread's 4 channels inside the chunk
    const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(p.A)), 0, (int)p.a_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t rsrc_a2 = rsrc_a;
    if constexpr (DUAL)
      rsrc_a2 = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<char*>(reinterpret_cast<const char*>(p.A2)), 0, (int)p.a_bytes, 0x00020000);
    const bool has_pro = p.in_scale != nullptr;
    const float relu_floor = p.in_relu ? 0.f : -__builtin_huge_valf();  // max(x, -inf) = x
    const int n_img = p.M / HoWo;
    const bool linear = p.stride == 1;   // GATHER: input pixel index = output pixel index

    struct Item {
      f32x4 a;
      f32x4 a2;
      unsigned ok;
    };
    Item st[P3_RING];

    // ---- load cursor: (tile, chunk, pass) of the next item to fetch, plus this thread's row
    int l_round = 0, l_c = 0, l_p = 0, l_np = 0, l_m0 = 0;
    int l_img = 0, l_h = 0, l_w = 0;       // DENSE: padded (img, hp, wp) of the row; GATHER: (img, ho, wo)
    int l_img0 = 0, l_h0 = 0, l_w0 = 0;    // ... of pass 0 (restored at every chunk)
    int l_rows = 0;
    auto l_setup = [&](int round) {
      int m0, n0;
      tile_of(round, m0, n0);
      l_m0 = m0;
      l_rows = patch_rows(m0);
      l_np = (l_rows + 31) >> 5;
      if constexpr (MODE == P3_DENSE) {
        const int u = u0_of(m0) + lrow;
        l_img0 = u / (Hp * Wp);
        const int rem = u - l_img0 * (Hp * Wp);
        l_h0 = rem / Wp;
        l_w0 = rem - l_h0 * Wp;
      } else {
        const int m = m0 + lrow;
        l_img0 = m / HoWo;
        const int rem = m - l_img0 * HoWo;
        l_h0 = rem / p.Wo;
        l_w0 = rem - l_h0 * p.Wo;
      }
      l_img = l_img0;
      l_h = l_h0;
      l_w = l_w0;
    };
    auto load = [&](Item& s) {
      int voff = BUF_OOB;
      if (l_round < my_tiles) {
        const int j = l_p * 32 + lrow;
        if constexpr (MODE == P3_DENSE) {
          const int hi = l_h - p.pad, wi = l_w - p.pad;
          if (j < l_rows && l_img < n_img && (unsigned)hi < (unsigned)p.H &&
              (unsigned)wi < (unsigned)p.W)
            voff = (((l_img * p.H + hi) * p.W + wi) * p.lda + lk4) * 4;
        } else {
          if (j < l_rows)
            voff = ((linear ? l_m0 + j : (l_img * p.H + l_h * p.stride) * p.W + l_w * p.stride) *
                        p.lda + lk4) * 4;
        }
      }
      s.ok = voff != BUF_OOB;
      const int soff = l_c * 128;
      s.a = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, voff, soff, 0));
      if constexpr (DUAL)
        s.a2 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_a2, voff, soff, 0));
      if (l_round >= my_tiles) return;
      // advance: next pass of this chunk, else next chunk, else next tile
      if (++l_p < l_np) {
        if (MODE == P3_DENSE || !linear) {
          const int wrap_w = MODE == P3_DENSE ? Wp : p.Wo, wrap_h = MODE == P3_DENSE ? Hp : p.Ho;
          l_w += 32;
          while (l_w >= wrap_w) {
            l_w -= wrap_w;
            if (++l_h == wrap_h) {
              l_h = 0;
              ++l_img;
            }
          }
        }
      } else {
        l_p = 0;
        l_img = l_img0;
        l_h = l_h0;
        l_w = l_w0;
        if (++l_c == NC) {
          l_c = 0;
          if (++l_round < my_tiles) l_setup(l_round);
        }
      }
    };

    // ---- store cursor
    int s_round = 0, s_c = 0, s_p = 0, s_np = 0, s_m0 = 0, s_side = 0, s_h = 0;
    auto s_setup = [&](int round) {
      int m0, n0;
      tile_of(round, m0, n0);
      s_m0 = m0;
      s_side = n0 == 0;   // the materialised block output: written once, by the n-tile-0 workgroups
      s_np = (patch_rows(m0) + 31) >> 5;
    };
    // prologue vectors of a chunk (its 32 input channels): `cur` in use, `nxt` in flight a chunk ahead
    struct Vec {
      f32x4 s, t, c, s2, t2, c2;
    };
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f}, one4 = {1.f, 1.f, 1.f, 1.f};
    Vec cur = {one4, zero4, zero4, one4, zero4, zero4}, nxt = cur;
    auto load_vec = [&](Vec& v, int ci) {
      if (has_pro) {
        v.s = ldg4(p.in_scale + ci + lk4);
        v.t = ldg4(p.in_shift + ci + lk4);
        if (p.in_center) v.c = ldg4(p.in_center + ci + lk4);
        if constexpr (DUAL) {
          if (p.in2_scale != nullptr) {
            v.s2 = ldg4(p.in2_scale + ci + lk4);
            v.t2 = ldg4(p.in2_shift + ci + lk4);
            if (p.in2_center) v.c2 = ldg4(p.in2_center + ci + lk4);
          }
        }
      }
    };
    auto store = [&](const Item& s) {
      char* const buf = xsm + (s_h & 1) * pbuf;
      if (s_p == 0) {
        cur = nxt;
        if constexpr (DUAL) {
          if (p.in2_scale != nullptr) cur.t2 = cur.t + cur.t2;   // both shifts in one add
        }
        load_vec(nxt, s_c + 1 < NC ? (s_c + 1) * 32 : 0);
        p3_wait(pempty + (s_h & 1), MATRIX * (s_h >> 1));  // chunk h-2 has been read
      }
      f32x4 v = s.a;
      if (has_pro) {
        // (scalar fma / max per element: packed fp32 VALU beside MFMAs costs more issue time than
        // the two plain instructions it replaces -- MI355X_MICROARCH.md)
        if constexpr (DUAL) {
          if (p.in2_scale != nullptr) {  // downsample branch: its own BatchNorm
#pragma unroll
            for (int e = 0; e < 4; ++e)
              v[e] = fmaxf(fmaf(v[e] - cur.c[e], cur.s[e],
                                fmaf(s.a2[e] - cur.c2[e], cur.s2[e], cur.t2[e])), relu_floor);
          } else {  // identity skip: added as is
#pragma unroll
            for (int e = 0; e < 4; ++e)
              v[e] = fmaxf(fmaf(v[e] - cur.c[e], cur.s[e], cur.t[e]) + s.a2[e], relu_floor);
          }
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            v[e] = fmaxf(fmaf(v[e] - cur.c[e], cur.s[e], cur.t[e]), relu_floor);
        }
        // zero padding / rows past M come AFTER the transform
        if (!s.ok) v = zero4;
        if constexpr (DUAL) {
          if (p.side_out != nullptr && s_side && s.ok)
            *reinterpret_cast<f32x4*>(p.side_out + (long)(s_m0 + s_p * 32 + lrow) * p.lda +
                                      s_c * 32 + lk4) = v;
        }
      }
      p3_split_store(v, buf + (s_p * 32 + lrow) * P3_ROW + lk4 * 2);
      if (++s_p == s_np) {
        if (lane == 0) x3_signal(pfull + (s_h & 1));  // (in LDS order behind this wave's writes)
        s_p = 0;
        ++s_h;
        if (++s_c == NC) {
          s_c = 0;
          if (++s_round < my_tiles) s_setup(s_round);
        }
      }
    };

    l_setup(0);
    s_setup(0);
    load_vec(nxt, 0);
#pragma unroll
    for (int j = 0; j < P3_RING - 1; ++j) load(st[j]);
    while (s_round < my_tiles) {
#pragma unroll
      for (int j = 0; j < P3_RING; ++j) {
        if (s_round < my_tiles) {
          load(st[(j + P3_RING - 1) % P3_RING]);
          store(st[j]);
        }
      }
    }
  } else {
    // ================================================================ matrix waves
    const int wm = wave / WN, wn = wave % WN;
    const int KS3 = (p.K / 16) * 3072;   // bytes of one n-block's fragments (all k-slabs, 3 planes)
    const __amdgpu_buffer_rsrc_t rsrc_b = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(p.Bfrag)), 0, (int)((long)p.N * p.K * 6),
        0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_c = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<char*>(p.C), 0, (int)p.c_bytes, 0x00020000);

    bf16x8 fa[MT][3];
    bf16x8 b0[NT][3], b1[NT][3];
    f32x16 acc[MT][NT];
    auto loadB = [&](bf16x8 (&b)[NT][3], const int (&vb)[NT], int soff) {
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int q = 0; q < 3; ++q)
          b[j][q] = __builtin_bit_cast(
              bf16x8, __builtin_amdgcn_raw_buffer_load_b128(
                          rsrc_b, vb[j] == BUF_OOB ? BUF_OOB : vb[j] + q * 1024, soff, 0));
    };
    auto mma = [&](const bf16x8 (&b)[NT][3]) {
      constexpr int PA[6] = {2, 1, 0, 1, 0, 0};  // smallest products first
      constexpr int PB[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
      for (int q = 0; q < 6; ++q)
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][PA[q]], b[j][PB[q]],
                                                                acc[i][j], 0, 0, 0);
    };
    auto vb_of = [&](int n0, int (&vb)[NT]) {
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int nb = (n0 + wn * WTN) / 32 + j;
        vb[j] = nb * 32 < p.N ? nb * KS3 + lane * 16 : BUF_OOB;
      }
    };

    // the SIMD's VALU issue port is shared with the producer wave: the MFMAs must win it the
    // moment the matrix pipe frees up
    __builtin_amdgcn_s_setprio(3);
    int h = 0;
    int vb[NT], vbn[NT];
    {
      int m0, n0;
      tile_of(0, m0, n0);
      vb_of(n0, vb);
      loadB(b0, vb, 0);
    }
    for (int round = 0; round < my_tiles; ++round) {
      int m0, n0;
      tile_of(round, m0, n0);
      vb_of(n0, vb);
      if (round + 1 < my_tiles) {
        int m1, n1;
        tile_of(round + 1, m1, n1);
        vb_of(n1, vbn);
      } else {
#pragma unroll
        for (int j = 0; j < NT; ++j) vbn[j] = BUF_OOB;
      }
      // LDS byte offset of this lane's patch row for tap (0, 0), per 32-row MFMA block
      int a_row[MT];
      {
        const int u_lo = MODE == P3_DENSE ? u0_of(m0) : m0;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          const int m = min(m0 + wm * WTM + i * 32 + l31, p.M - 1);
          a_row[i] = ((MODE == P3_DENSE ? u0_of(m) : m) - u_lo) * P3_ROW + half * 16;
        }
      }
      // epilogue vectors of this wave's columns (loaded now, used after the K loop)
      float e_sc[NT], e_sh[NT];
      int e_voff[NT];
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int col = n0 + wn * WTN + j * 32 + l31;
        const bool ok = col < p.N;
        e_sc[j] = (ok && p.scale) ? p.scale[col] : 1.f;
        e_sh[j] = (ok && p.shift) ? p.shift[col] : 0.f;
        e_voff[j] = ok ? (int)((((long)(m0 + wm * WTM + 4 * half)) * p.ldc + col) * 4) : BUF_OOB;
      }
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

      int ks3 = 0;  // byte offset of the current k-slab pair inside an n-block's fragments
      for (int c = 0; c < NC; ++c, ++h) {
        p3_wait(pfull + (h & 1), P3_PRODUCERS * ((h >> 1) + 1));
        const int bufoff = (h & 1) * pbuf;
        int tr = 0, tq = 0;
        for (int t = 0; t < T; ++t) {
          const char* const abase = xsm + bufoff + (tr * Wp + tq) * P3_ROW;
          const bool last_of_chunk = t == T - 1;
          const bool last_of_tile = last_of_chunk && c == NC - 1;
          // ---- k-slab 0 of this (chunk, tap): B fragments in b0 (fetched a slab ago)
          loadB(b1, vb, ks3 + 3072);
#pragma unroll
          for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int q = 0; q < 3; ++q)
              fa[i][q] = *reinterpret_cast<const bf16x8*>(abase + a_row[i] + q * 64);
          __builtin_amdgcn_sched_barrier(0);
          mma(b0);
          __builtin_amdgcn_sched_barrier(0);
          // ---- k-slab 1: B fragments in b1; b0 <- the next step's (or the next tile's first)
          if (last_of_tile) loadB(b0, vbn, 0);
          else loadB(b0, vb, ks3 + 6144);
#pragma unroll
          for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int q = 0; q < 3; ++q)
              fa[i][q] = *reinterpret_cast<const bf16x8*>(abase + a_row[i] + q * 64 + 32);
          if (last_of_chunk) {
            // this wave's reads of the patch buffer are complete once they have all returned
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane == 0) x3_signal(pempty + (h & 1));
          }
          __builtin_amdgcn_sched_barrier(0);
          mma(b1);
          __builtin_amdgcn_sched_barrier(0);
          ks3 += 6144;
          if (++tq == p.KW) {
            tq = 0;
            ++tr;
          }
        }
      }

      // -------------------------------------------------------------- statistics partials
      if (p.stat_partial != nullptr) {
        const int tile_m = m0 / BM;
        if (p.stat_rows == 32 && MT > 1) {
#pragma unroll
          for (int i = 0; i < MT; ++i)
            wave_stats_block<NT>(acc[i], p.stat_partial, (m0 + wm * WTM) / 32 + i,
                                 p.M - (m0 + wm * WTM + i * 32), n0 + wn * WTN, p.N, half, l31);
        } else if (p.stat_rows > 0 && p.stat_rows < WTM)
          wave_stats_fine<MT, NT>(acc, p.stat_partial, p.stat_rows, m0 + wm * WTM, p.M,
                                  n0 + wn * WTN, p.N, half, l31);
        else
          wave_stats<MT, NT>(acc, p.stat_partial, tile_m * WM + wm, p.M - (m0 + wm * WTM), WTM,
                             n0 + wn * WTN, p.N, half, l31);
      }
      // -------------------------------------------------------------- epilogue from registers
      // one store = 2 rows x 32 columns = two full 128-byte lines; rows past M get an
      // out-of-range lane offset
      const int rows_left = p.M - (m0 + wm * WTM + 4 * half);
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int rw = i * 32 + (r & 3) + 8 * (r >> 2);
          const int soff = rw * p.ldc * 4;
#pragma unroll
          for (int j = 0; j < NT; ++j) {
            const float v = apply_act(acc[i][j][r] * e_sc[j] + e_sh[j], p.act);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsrc_c,
                                                  rw < rows_left ? e_voff[j] : BUF_OOB, soff, 0);
          }
        }
    }
    __builtin_amdgcn_s_setprio(0);
  }
#endif
}

// w_ohwi [N][KH][KW][Cin] fp32 -> B fragments [N/32][K/16][3][64 lanes][8 bf16]: k-slab
// ks = ((chunk * T + tap) * 2 + s) holds input channels chunk*32 + s*16 + [0, 16) of that tap;
// lane (l31, half) holds output channel nb*32 + l31, channels half*8 + [0, 8) of the slab;
// plane q is the q-th term of the exact round-to-nearest three-way bf16 split.
__global__ __launch_bounds__(256) void pack_weights_kernel(const float* __restrict__ w,
                                                           unsigned short* __restrict__ frag,
                                                           int N, int T, int Cin) {
  const long total = (long)(N / 32) * (T * Cin / 16) * 64;  // (nb, ks, lane) triples
  const int KS = T * Cin / 16;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += gridDim.x * 256L) {
    const int lane = (int)(i & 63);
    const long rest = i >> 6;
    const int ks = (int)(rest % KS);
    const int nb = (int)(rest / KS);
    const int s = ks & 1, ct = ks >> 1;
    const int t = ct % T, c = ct / T;
    const int n = nb * 32 + (lane & 31);
    const int ci = c * 32 + s * 16 + (lane >> 5) * 8;
    const float* src = w + ((long)n * T + t) * Cin + ci;
    unsigned short out[3][8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float v = src[e];
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const __bf16 hb = (__bf16)v;  // round to nearest even
        out[q][e] = __builtin_bit_cast(unsigned short, hb);
        v -= (float)hb;
      }
    }
    unsigned short* dst = frag + ((rest * 3) * 64 + lane) * 8;
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
      for (int e = 0; e < 8; ++e) dst[(long)q * 512 + e] = out[q][e];
  }
}

template <int BM, int BN, int WM, int WN, int DUAL, int MODE>
int launch_p3(const IgemmParams& p, int rows_alloc, hipStream_t stream) {
  const int smem_bytes = 2 * rows_alloc * P3_ROW + 16;  // two patch buffers + 4 counters
  constexpr int threads = (WM * WN + P3_PRODUCERS) * 64;
  auto kern = conv_p3_kernel<BM, BN, WM, WN, DUAL, MODE>;
  static int attr_bytes = 0;  // per instantiation: the largest dynamic LDS size enabled so far
  if (smem_bytes > attr_bytes) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
    if (e != hipSuccess) {
      vlnce_set_error("conv_p3: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
      return 2;
    }
    attr_bytes = 163840;
  }
  IgemmParams q = p;
  q.tiles_m = ceil_div(p.M, BM);
  q.tiles_n = ceil_div(p.N, BN);
  q.splitk = 1;
  q.p3_rows = rows_alloc;
  const long nwg = (long)q.tiles_m * q.tiles_n;
  if (nwg <= 0 || nwg > 0x7fffffffL) {
    vlnce_set_error("conv_p3: bad grid %ld", nwg);
    return 1;
  }
  const int cus = x3_cus();
  const unsigned grid = nwg <= cus ? (unsigned)nwg : (unsigned)cus;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), smem_bytes, stream, q);
  VLNCE_CHECK_LAUNCH("conv_p3");
  return 0;
}

struct P3Tile {
  int bm, bn;
};

// patch rows (rounded up to whole 32-row passes) the BM-pixel tiles of this problem need: the
// exact maximum over the tiles (the pattern of tile starts repeats with the image, so at most
// Ho*Wo / gcd(BM, Ho*Wo) tiles are looked at)
int p3_rows_for(const IgemmParams& p, int bm, bool dense) {
  if (!dense) return bm;
  const int Hp = p.H + 2 * p.pad, Wp = p.W + 2 * p.pad;
  const int howo = p.Ho * p.Wo;
  auto u0 = [&](long m) {
    const long img = m / howo, rem = m - img * howo, ho = rem / p.Wo;
    return (img * Hp + ho) * Wp + (rem - ho * p.Wo);
  };
  long rows = 0;
  for (long m0 = 0; m0 < p.M; m0 += bm) {
    if (m0 > 0 && m0 % howo == 0) break;  // same tile starts as from m0 = 0 on
    const long mlast = (m0 + bm < p.M ? m0 + bm : p.M) - 1;
    const long r = u0(mlast) - u0(m0) + (p.KH - 1) * Wp + p.KW;
    if (r > rows) rows = r;
  }
  return (int)((rows + 31) / 32 * 32);
}

template <int DUAL, int MODE>
int dispatch_p3(const IgemmParams& p, const P3Tile& t, int rows, hipStream_t s) {
  if (t.bm == 128 && t.bn == 256) return launch_p3<128, 256, 2, 4, DUAL, MODE>(p, rows, s);
  if (t.bm == 64 && t.bn == 256) return launch_p3<64, 256, 2, 4, DUAL, MODE>(p, rows, s);
  if (t.bm == 128 && t.bn == 128) return launch_p3<128, 128, 2, 4, DUAL, MODE>(p, rows, s);
  if (t.bm == 64 && t.bn == 128) return launch_p3<64, 128, 2, 4, DUAL, MODE>(p, rows, s);
  if (t.bm == 128 && t.bn == 64) return launch_p3<128, 64, 4, 2, DUAL, MODE>(p, rows, s);
  return launch_p3<64, 64, 2, 2, DUAL, MODE>(p, rows, s);
}

}  // namespace

int p3_try_launch(const IgemmParams& p, hipStream_t stream) {
  static const int mode_env = getenv("VLNCE_P3") ? atoi(getenv("VLNCE_P3")) : 1;  // 0 = off
  static const int force = getenv("VLNCE_P3_TILE") ? atoi(getenv("VLNCE_P3_TILE")) : 0;  // tuning
  if (!mode_env || !conv_math() || !p.Bfrag) return -1;
  if (p.Cin % 32 != 0 || p.N % 32 != 0 || p.lda % 4 != 0 || p.splitk > 1) return -1;
  if (p.residual || p.accumulate || p.c_bytes >= 0x7fffffffL || p.a_bytes >= 0x7fffffffL) return -1;
  if ((long)p.N * p.K * 6 >= 0x7fffffffL) return -1;
  const bool one = p.KH == 1 && p.KW == 1 && p.pad == 0;
  const bool dense = !one;
  if (dense && p.stride != 1) return -1;
  const bool dual = p.A2 != nullptr || p.side_out != nullptr;
  if (dual && !(one && p.stride == 1)) return -1;
  if (mode_env == 2 && !dense) return -1;  // VLNCE_P3=2: only the patch (KxK) layers
  if (mode_env == 3 && dense) return -1;   // VLNCE_P3=3: only the 1x1 layers

  const P3Tile cand[6] = {{128, 256}, {64, 256}, {128, 128}, {64, 128}, {128, 64}, {64, 64}};
  const int cus = x3_cus();
  P3Tile pick{0, 0};
  int pick_rows = 0;
  double best = 0.0;
  for (int ci = 0; ci < 6; ++ci) {
    const P3Tile& c = cand[ci];
    if (force >= 1 && force <= 6 && ci != force - 1) continue;
    if (c.bn > 64 && p.N < c.bn && !(force >= 1 && force <= 6)) continue;
    const int rows = p3_rows_for(p, c.bm, dense);
    if (2L * rows * P3_ROW + 16 > 163840) continue;
    const long tiles = (long)ceil_div(p.M, c.bm) * ceil_div(p.N, c.bn);
    const long rounds = (tiles + cus - 1) / cus;
    const double eff = (double)tiles / (double)(rounds * cus);
    if (eff > best) {
      best = eff;
      pick = c;
      pick_rows = rows;
    }
    if (eff >= 0.8) break;
  }
  if (pick.bm == 0 || (best < 0.4 && !(force >= 1 && force <= 6))) return -1;
  if (dual) return dispatch_p3<1, P3_GATHER>(p, pick, pick_rows, stream);
  if (dense) return dispatch_p3<0, P3_DENSE>(p, pick, pick_rows, stream);
  return dispatch_p3<0, P3_GATHER>(p, pick, pick_rows, stream);
}

}  // namespace vlnce_detail

extern "C" long vlnce_conv2d_pack_bytes(const vlnce_conv_desc* d) {
  if (!d || d->Cin <= 0 || d->Cout <= 0 || d->Cin % 32 != 0 || d->Cout % 32 != 0) return 0;
  return (long)d->Cout * d->KH * d->KW * d->Cin * 6;
}

extern "C" int vlnce_conv2d_pack_weights(const float* w_ohwi, void* frag, const vlnce_conv_desc* d,
                                         vlnce_stream_t stream) {
  VLNCE_CHECK_ARG(w_ohwi && frag && d, "conv2d_pack_weights: null argument");
  VLNCE_CHECK_ARG(vlnce_conv2d_pack_bytes(d) > 0,
                  "conv2d_pack_weights: needs Cin %% 32 == 0 and Cout %% 32 == 0");
  const int T = d->KH * d->KW;
  const long total = (long)(d->Cout / 32) * (T * d->Cin / 16) * 64;
  const long blocks = (total + 255) / 256;
  hipLaunchKernelGGL(pack_weights_kernel, dim3((unsigned)(blocks > 8192 ? 8192 : blocks)), dim3(256),
                     0, reinterpret_cast<hipStream_t>(stream), w_ohwi,
                     reinterpret_cast<unsigned short*>(frag), d->Cout, T, d->Cin);
  VLNCE_CHECK_LAUNCH("conv2d_pack_weights");
  return 0;
}
