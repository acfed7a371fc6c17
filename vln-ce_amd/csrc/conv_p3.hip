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
#include <type_traits>

using namespace vlnce_detail;

namespace vlnce_detail {
namespace {

// bytes per patch row: Planes<MATH>::ROW (208 = 13 x 16 B / 144 = 9 x 16 B: consecutive rows are
// conflict-free for ds_read_b128)
constexpr int P3_PRODUCERS = 4;  // producer waves
#ifndef U3_B_AHEAD_64
#define U3_B_AHEAD_64 2    // conv_u3_kernel: B fragments requested a chunk ahead (see BA in the kernel)
#endif
#ifndef U3_B_AHEAD_128
#define U3_B_AHEAD_128 1
#endif
#ifndef U3_RAW_BATCH
#define U3_RAW_BATCH 1   // conv_u3_kernel: raw-row chunks per request burst (2: the round-6 experiment below)
#endif
constexpr int U3_MAX_CIN = 4096;   // conv_u3_kernel: input channels whose prologue vectors fit its LDS
// patch rows the KxK producers address: 12 row groups of 32 (3 items of 128), their byte offsets in
// twelve registers.  (Round 6, measured and dropped: with the 144-byte rows of the fp16 planes the LDS
// would hold 568 rows, i.e. 256-row tiles on 64x64 maps -- but 16 offsets in registers pushed the
// workgroup's register allocation (the maximum over both roles) over the edge, the MATRIX waves'
// k-loop reloaded two spilled values per 24 MFMAs behind an s_waitcnt vmcnt(0), and every 3x3
// layer ran 1.5-2.7x slower: profiles/r6_06; the offsets in LDS instead, read back with one
// ds_read_b128 per item, left 120 spilled registers: profiles/r6_07; sixteen offsets only in the
// instantiations whose matrix waves hold <= 2 accumulator blocks made exactly those instantiations
// 1.9-2.1x slower -- l1 3x3 100 -> 207 us, ResNet-18's 64-channel 3x3 at 416 frames 683 -> 1335 us --
// and left the others alone: profiles/r06_f_*.  The 64-channel 3x3 layers on 64x64 maps therefore
// stay on 128x64 tiles (one accumulator block per wave), 185-195 TF/s against 300-330 TF/s for the
// 128- to 512-channel layers.)
constexpr int P3_MAX_ROWS = 384;
#ifndef P3_DENSE_VEC_GLOBAL   // (A/B switch: the KxK producers' prologue vectors as per-chunk global loads)
#define P3_DENSE_VEC_LDS 1
#else
#define P3_DENSE_VEC_LDS 0
#endif
#ifndef P3_MPRIO
#define P3_MPRIO 1   // s_setprio of the matrix waves
#endif
#ifndef P3_PPRIO
#define P3_PPRIO 2   // s_setprio of the producer waves: measured, a producer-bound layer
                     // (3x3 64->64 at 64x64) runs 140 -> 119 us when the producers win the issue port
#endif

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

// x (4 consecutive k of one patch row) -> the A planes' 8-byte words, round-to-nearest split
template <int MATH>
__device__ __forceinline__ void p3_split_store(f32x4 x, char* row_ptr) {
  constexpr int NA = Planes<MATH>::NA;
  unsigned w0[NA], w1[NA];
  split_pair<MATH>(x[0], x[1], w0);
  split_pair<MATH>(x[2], x[3], w1);
#pragma unroll
  for (int q = 0; q < NA; ++q) *reinterpret_cast<u32x2*>(row_ptr + q * 64) = u32x2{w0[q], w1[q]};
}

template <int BM, int BN, int WM, int WN, int DUAL, int MODE, int MATH>
__global__ __launch_bounds__((WM * WN + P3_PRODUCERS) * 64) void conv_p3_kernel(IgemmParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
  typedef Planes<MATH> PL;
  constexpr int P3_ROW = PL::ROW, NA = PL::NA;
  constexpr int MATRIX = WM * WN;
  constexpr int WTM = BM / WM, WTN = BN / WN, MT = WTM / 32, NT = WTN / 32;
  static_assert(MT >= 1 && NT >= 1 && WTM % 32 == 0 && WTN % 32 == 0, "tile");
  static_assert(!DUAL || MODE == P3_GATHER, "dual-input prologue: 1x1 convolutions only");

  // 1x1 convolutions have short reductions: their epilogue stores must not sit in front of any
  // load of the same wave (vmcnt counts stores too, in order), so there the matrix waves issue NO
  // vector loads: the producers bring a chunk's B fragments into LDS with LDS-DMA (fragment
  // order = lane order, 1 KB per instruction) next to the patch, under the same hand-over.  KxK
  // convolutions (long reductions, large patches) fetch B straight from L2 into registers.
  constexpr bool B_LDS = MODE == P3_GATHER;
  constexpr int BST = B_LDS ? BN * 192 : 0;   // bytes of one B stage: BN/32 n-blocks x 2 slabs x 3 KB
  extern __shared__ __attribute__((aligned(16))) char xsm[];
  const int pbuf = p.p3_rows * P3_ROW;                         // bytes of one patch buffer
  char* const bstage = xsm + 2 * pbuf;                         // [2][BST]
  int* const pfull = reinterpret_cast<int*>(bstage + 2 * BST);  // [2] producer waves done writing
  int* const pempty = pfull + 2;                               // [2] matrix waves done reading
  // KxK form: the prologue vectors [3][Cin] behind the counters (conv_u3_kernel's arrangement)
  float* const vlds = reinterpret_cast<float*>(pfull + 4);

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
  if constexpr (MODE == P3_DENSE && P3_DENSE_VEC_LDS) {
    const bool pro = p.in_scale != nullptr;
    for (int ch = tid; ch < p.Cin; ch += (MATRIX + P3_PRODUCERS) * 64) {
      vlds[ch] = pro ? p.in_scale[ch] : 1.f;
      vlds[p.Cin + ch] = pro ? p.in_shift[ch] : 0.f;
      vlds[2 * p.Cin + ch] = (pro && p.in_center) ? p.in_center[ch] : 0.f;
    }
  }
  __syncthreads();

  if (wave >= MATRIX) {
    // ================================================================ producer waves
    __builtin_amdgcn_s_setprio(P3_PPRIO);
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
    auto next_vec = [&](int c_next) {  // at the first row group of a chunk
      cur = nxt;
      if constexpr (DUAL) {
        if (p.in2_scale != nullptr) cur.t2 = cur.t + cur.t2;  // both shifts in one add
      }
      load_vec(nxt, c_next < NC ? c_next * 32 : 0);
    };
    // one thread's float4 of a patch row: prologue, zero padding, (block output), split, LDS write
    auto transform_store = [&](f32x4 v, f32x4 v2, bool ok, char* dst, float* side_dst) {
      if (has_pro) {
        // (scalar fma / max per element: packed fp32 VALU beside MFMAs costs more issue time than
        // the two plain instructions it replaces -- MI355X_MICROARCH.md)
        if constexpr (DUAL) {
          if (p.in2_scale != nullptr) {  // downsample branch: its own BatchNorm
#pragma unroll
            for (int e = 0; e < 4; ++e)
              v[e] = fmaxf(fmaf(v[e] - cur.c[e], cur.s[e],
                                fmaf(v2[e] - cur.c2[e], cur.s2[e], cur.t2[e])), relu_floor);
          } else {  // identity skip: added as is
#pragma unroll
            for (int e = 0; e < 4; ++e)
              v[e] = fmaxf(fmaf(v[e] - cur.c[e], cur.s[e], cur.t[e]) + v2[e], relu_floor);
          }
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            v[e] = fmaxf(fmaf(v[e] - cur.c[e], cur.s[e], cur.t[e]), relu_floor);
        }
        // zero padding / rows past M come AFTER the transform
        if (!ok) v = zero4;
        if constexpr (DUAL) {
          if (side_dst != nullptr && ok) *reinterpret_cast<f32x4*>(side_dst) = v;
        }
      }
      p3_split_store<MATH>(v, dst);
    };
#ifdef P3_DBG_TIME
    long long d_wait = 0, d_vmw = 0, d_tr = 0, d_ld = 0;
    const long long d_t0 = clock64();
#endif

    if constexpr (MODE == P3_GATHER) {
      // ---- 1x1 convolutions: patch row = output pixel, BM / 32 row groups per chunk, all of a
      // chunk's loads in flight one chunk ahead (two register sets)
      constexpr int NP = BM / 32;
      constexpr int NPI = NP < 4 ? NP : 4;   // row groups per staged item (register budget)
      constexpr int PARTS = NP / NPI;        // items per chunk: 1, or 2 for 256-row tiles
      static_assert(PARTS == 1 || PARTS == 2, "tile height");
      struct Staged {
        f32x4 a[NPI];
        f32x4 a2[DUAL ? NPI : 1];
        unsigned ok;
        int m0;  // first row of the tile if this workgroup writes side_out for it, else -1
      };
      Staged st[2];
      int a_voff[NP];
      unsigned a_ok = 0;
      int l_round = 0, l_c = 0, l_m0 = 0, l_side = 0;
      auto setup_tile = [&](int round) {
        int m0, n0;
        tile_of(round, m0, n0);
        l_m0 = m0;
        l_side = n0 == 0;  // the materialised block output: written once, by the n-tile-0 workgroups
        a_ok = 0;
#pragma unroll
        for (int i = 0; i < NP; ++i) {
          const int m = m0 + i * 32 + lrow;
          a_voff[i] = BUF_OOB;
          if (m < p.M) {
            int pix = m;
            if (!linear) {
              const int img = m / HoWo;
              const int rem = m - img * HoWo;
              const int ho = rem / p.Wo;
              pix = (img * p.H + ho * p.stride) * p.W + (rem - ho * p.Wo) * p.stride;
            }
            a_voff[i] = (pix * p.lda + lk4) * 4;
            a_ok |= 1u << i;
          }
        }
      };
      // item `it` of the stream is part it % PARTS of chunk it / PARTS; it lives in set it & 1, so
      // with two parts the part index equals the set index (static register indexing)
      auto load = [&](Staged& s, int part, bool live) {
        const int soff = l_c * 128;
        s.ok = live ? (a_ok >> (part * NPI)) : 0u;
        s.m0 = l_side ? l_m0 : -1;
#pragma unroll
        for (int i = 0; i < NPI; ++i) {
          const int vo = live ? a_voff[part * NPI + i] : BUF_OOB;
          s.a[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, vo, soff, 0));
          if constexpr (DUAL)
            s.a2[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_a2, vo, soff, 0));
        }
        if (part == PARTS - 1 && ++l_c == NC) {
          l_c = 0;
          if (++l_round < my_tiles) setup_tile(l_round);
        }
      };
      int s_c = 0, s_round = 0, s_n0 = 0;
      const int KS3 = (p.K / 16) * 3072;   // bytes of one n-block's fragments
      const __amdgpu_buffer_rsrc_t rsrc_b = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<char*>(reinterpret_cast<const char*>(p.Bfrag)), 0, (int)((long)p.N * p.K * 6),
          0x00020000);
      auto stash = [&](const Staged& s, int part, int g) {   // g = chunk index of the stream
        char* const buf = xsm + (g & 1) * pbuf + (part * NPI * 32 + lrow) * P3_ROW + lk4 * 2;
        if (part == 0) {
          next_vec(s_c + 1);
#ifdef P3_DBG_TIME
          const long long d_a = clock64();
#endif
          p3_wait(pempty + (g & 1), MATRIX * (g >> 1));  // chunk g-2 has been read
#ifdef P3_DBG_TIME
          d_wait += clock64() - d_a;
#endif
        }
#pragma unroll
        for (int i = 0; i < NPI; ++i) {
          float* side = nullptr;
          if constexpr (DUAL) {
            if (p.side_out != nullptr && s.m0 >= 0)
              side = p.side_out + (long)(s.m0 + (part * NPI + i) * 32 + lrow) * p.lda + s_c * 32 + lk4;
          }
          transform_store(s.a[i], s.a2[DUAL ? i : 0], (s.ok >> i) & 1u, buf + i * 32 * P3_ROW, side);
        }
        if (part == PARTS - 1) {
          // this chunk's B fragments: BN/32 runs of 6 KB (2 k-slabs x 3 planes of one n-block),
          // 1 KB per instruction, instructions dealt round-robin to the producer waves
          const int pw = wave - MATRIX;
          typedef __attribute__((address_space(3))) void lds_void;
#pragma unroll
          for (int k = 0; k < 6 * (BN / 32) / P3_PRODUCERS; ++k) {
            const int f = k * P3_PRODUCERS + pw;     // fragment slot inside the stage
            const int nbl = f / 6, part6 = f - nbl * 6;
            const int nb = s_n0 / 32 + nbl;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(
                rsrc_b, (lds_void*)(bstage + (g & 1) * BST + f * 1024), 16,
                nb * 32 < p.N ? lane * 16 : BUF_OOB, nb * KS3 + s_c * 6144 + part6 * 1024, 0, 0);
          }
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the DMA has landed
          if (lane == 0) x3_signal(pfull + (g & 1));  // (in LDS order behind this wave's writes)
          if (++s_c == NC) {
            s_c = 0;
            if (++s_round < my_tiles) {
              int m0;
              tile_of(s_round, m0, s_n0);
            }
          }
        }
      };
      {
        int m0;
        tile_of(0, m0, s_n0);
      }
      const int I_total = my_tiles * NC * PARTS;
      setup_tile(0);
      load_vec(nxt, 0);
      load(st[0], 0, true);
      for (int it = 0; it < I_total; it += 2) {
        load(st[1], PARTS == 2 ? 1 : 0, it + 1 < I_total);
        stash(st[0], 0, it / PARTS);
        if (it + 1 < I_total) {
          load(st[0], 0, it + 2 < I_total);
          stash(st[1], PARTS == 2 ? 1 : 0, (it + 1) / PARTS);
        }
      }
    } else {
      // ---- KxK stride-1 convolutions: the patch is a run of padded-linear pixels, fetched and
      // transformed as a stream of items of 4 row groups (128 rows x 32 channels), two register
      // sets: 4 - 8 loads per thread in flight
      struct Staged {
        f32x4 a[4];
        unsigned ok;
      };
      Staged st[2];
      // load cursor: (tile, chunk, part) of the next item to fetch.  The byte offsets of this
      // thread's (up to 12) patch rows are decoded once per tile -- every chunk re-reads the same
      // pixels 32 channels further on (soffset)
      // (scalars and wave-uniform branches: a runtime-indexed array would live in scratch memory)
      int l_round = 0, l_c = 0, l_part = 0, l_nparts = 0;
      int r0 = 0, r1 = 0, r2 = 0, r3 = 0, r4 = 0, r5 = 0, r6 = 0, r7 = 0, r8 = 0, r9 = 0, r10 = 0,
          r11 = 0;   // byte offsets of this thread's rows in row groups 0..11 (p3_rows <= P3_MAX_ROWS)
      unsigned l_ok = 0;
      auto l_setup = [&](int round) {
        int m0, n0;
        tile_of(round, m0, n0);
        const int rows = patch_rows(m0);
        l_nparts = (rows + 127) >> 7;
        const int u = u0_of(m0) + lrow;
        int img = u / (Hp * Wp);
        const int rem = u - img * (Hp * Wp);
        int hh = rem / Wp;
        int ww = rem - hh * Wp;
        l_ok = 0;
        auto next_row = [&](int g, int& vo) {
          const int hi = hh - p.pad, wi = ww - p.pad;
          const bool ok = g * 32 + lrow < rows && img < n_img && (unsigned)hi < (unsigned)p.H &&
                          (unsigned)wi < (unsigned)p.W;
          vo = ok ? (((img * p.H + hi) * p.W + wi) * p.lda + lk4) * 4 : BUF_OOB;
          l_ok |= (ok ? 1u : 0u) << g;
          ww += 32;
          while (ww >= Wp) {
            ww -= Wp;
            if (++hh == Hp) {
              hh = 0;
              ++img;
            }
          }
        };
        next_row(0, r0); next_row(1, r1); next_row(2, r2); next_row(3, r3);
        next_row(4, r4); next_row(5, r5); next_row(6, r6); next_row(7, r7);
        next_row(8, r8); next_row(9, r9); next_row(10, r10); next_row(11, r11);
      };
      auto load = [&](Staged& s) {
        const bool live = l_round < my_tiles;
        const int soff = l_c * 128;
        s.ok = live ? (l_ok >> (l_part * 4)) & 15u : 0u;
#define P3_LD(V) \
  __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, live ? (V) : BUF_OOB, soff, 0))
        if (l_part == 0) {           // (wave-uniform branches, scalar variables: no indexed array)
          s.a[0] = P3_LD(r0); s.a[1] = P3_LD(r1); s.a[2] = P3_LD(r2); s.a[3] = P3_LD(r3);
        } else if (l_part == 1) {
          s.a[0] = P3_LD(r4); s.a[1] = P3_LD(r5); s.a[2] = P3_LD(r6); s.a[3] = P3_LD(r7);
        } else {
          s.a[0] = P3_LD(r8); s.a[1] = P3_LD(r9); s.a[2] = P3_LD(r10); s.a[3] = P3_LD(r11);
        }
#undef P3_LD
        if (!live) return;
        if (++l_part == l_nparts) {  // next chunk of this tile, else next tile
          l_part = 0;
          if (++l_c == NC) {
            l_c = 0;
            if (++l_round < my_tiles) l_setup(l_round);
          }
        }
      };
      // store cursor
      int s_round = 0, s_c = 0, s_part = 0, s_nparts = 0, s_h = 0;
      auto s_setup = [&](int round) {
        int m0, n0;
        tile_of(round, m0, n0);
        s_nparts = (patch_rows(m0) + 127) >> 7;
      };
      auto stash = [&](const Staged& s) {
        char* const buf = xsm + (s_h & 1) * pbuf + lk4 * 2;
        if (s_part == 0) {
#if P3_DENSE_VEC_LDS
          const float* const vp = vlds + s_c * 32 + lk4;   // this chunk's vectors (LDS, staged once)
          cur.s = *reinterpret_cast<const f32x4*>(vp);
          cur.t = *reinterpret_cast<const f32x4*>(vp + p.Cin);
          cur.c = *reinterpret_cast<const f32x4*>(vp + 2 * p.Cin);
#else
          next_vec(s_c + 1);
#endif
#ifdef P3_DBG_TIME
          const long long d_a = clock64();
#endif
          p3_wait(pempty + (s_h & 1), MATRIX * (s_h >> 1));  // chunk h-2 has been read
#ifdef P3_DBG_TIME
          d_wait += clock64() - d_a;
#endif
        }
#ifdef P3_DBG_TIME
        const long long d_b = clock64();
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // this item's 4 loads (4 newer ones fly)
        const long long d_c = clock64();
        d_vmw += d_c - d_b;
#endif
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int j = (s_part * 4 + i) * 32 + lrow;
          if (j < p.p3_rows)  // (the last item may reach past the rows the buffer holds)
            transform_store(s.a[i], zero4, (s.ok >> i) & 1u, buf + j * P3_ROW, nullptr);
        }
#ifdef P3_DBG_TIME
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        d_tr += clock64() - d_c;
#endif
        if (++s_part == s_nparts) {
          if (lane == 0) x3_signal(pfull + (s_h & 1));  // (in LDS order behind this wave's writes)
          s_part = 0;
          ++s_h;
          if (++s_c == NC) {
            s_c = 0;
            if (++s_round < my_tiles) s_setup(s_round);
          }
        }
      };
      l_setup(0);
      s_setup(0);
#if !P3_DENSE_VEC_LDS
      load_vec(nxt, 0);
#endif
      load(st[0]);
      while (s_round < my_tiles) {
#ifdef P3_DBG_TIME
        const long long d_l = clock64();
#endif
        load(st[1]);
#ifdef P3_DBG_TIME
        d_ld += clock64() - d_l;
#endif
        stash(st[0]);
        if (s_round < my_tiles) {
          load(st[0]);
          stash(st[1]);
        }
      }
    }
#ifdef P3_DBG_TIME
    if (blockIdx.x == 8 && ptid == 0)
      printf("p3 producer: total %lld cycles, waiting for the matrix waves %lld; KxK form: waiting for "
             "loads %lld, transform + LDS writes %lld, issuing every other item's loads %lld\n",
             (long long)(clock64() - d_t0), d_wait, d_vmw, d_tr, d_ld);
#endif
  } else {
    // ================================================================ matrix waves
    const int wm = wave / WN, wn = wave % WN;
    const int KS3 = (p.K / 16) * 3072;   // bytes of one n-block's fragments (all k-slabs, 3 planes)
    const __amdgpu_buffer_rsrc_t rsrc_b = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(p.Bfrag)), 0, (int)((long)p.N * p.K * 6),
        0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_c = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<char*>(p.C), 0, (int)p.c_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_r = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(p.residual ? p.residual : p.C)), 0,
        (int)p.c_bytes, 0x00020000);   // (ldr == ldc: the output's extent)

    // (Measured and dropped, round 4: a second A-fragment set so that the LDS reads of k-slab
    // s + 1 are issued in front of the MFMAs of slab s -- branch-free steady loop, fragments
    // carried as 128-bit integer vectors -- changes no 3x3 layer by more than 2 % on the same box
    // (profiles/r04_t_*).  Bisection builds of this loop at 256 channels, 16x16 frames: 100 us as
    // is, 96 without any hand-over wait, 84 without the B-fragment loads, 85 without the A reads,
    // 69 with neither -- against 58 us of pure MFMA issue at the 1.9 GHz the launch runs at
    // (profiles/r04_u_*): what the loop loses is the issue cost of its 9 memory instructions per
    // 12 MFMAs (768 B of operands per MFMA at these per-wave tiles), not their latency.)
    // (Round 6, measured and dropped as well: the B fragments of the one- / two-block waves two
    // (chunk, tap) steps ahead in four sets -- what paid in conv_u3: the 64-channel 3x3 layer
    // 178 -> 156 TF/s with no spill, the other instances spill 198 registers under the 168-register
    // cap of a 12-wave workgroup; profiles/r06_v_conv_p3_b_fragments_two_steps_ahead.txt.)
    bf16x8 fa[MT][NA];
    bf16x8 b0[NT][3], b1[NT][3];
    f32x16 acc[MT][NT];
    auto loadB = [&](bf16x8 (&b)[NT][3], const int (&vb)[NT], int soff) {
      if constexpr (!B_LDS) {
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
          for (int q = 0; q < 3; ++q)
            b[j][q] = __builtin_bit_cast(
                bf16x8, __builtin_amdgcn_raw_buffer_load_b128(
                            rsrc_b, vb[j] == BUF_OOB ? BUF_OOB : vb[j] + q * 1024, soff, 0));
      }
    };
    // B_LDS: k-slab s of the chunk in stage `st`, this wave's n-blocks, fragment (= lane) order
    auto readB = [&](bf16x8 (&b)[NT][3], int st, int s) {
      if constexpr (B_LDS) {
        const char* base = bstage + st * BST + ((wn * NT) * 2 + s) * 3072 + lane * 16;
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
          for (int q = 0; q < 3; ++q)
            b[j][q] = *reinterpret_cast<const bf16x8*>(base + j * 6144 + q * 1024);
      }
    };
    auto mma = [&](const bf16x8 (&b)[NT][3]) {
#pragma unroll
      for (int q = 0; q < PL::NP; ++q)
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j)
            acc[i][j] = plane_mfma<MATH>(fa[i][PL::PA[q]], b[j][PL::PB[q]], acc[i][j]);
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
    __builtin_amdgcn_s_setprio(P3_MPRIO);
#ifdef P3_DBG_TIME
    long long d_pf = 0, d_vm = 0, d_lg = 0;
    const long long d_t0 = clock64(), d_w0 = wall_clock64();
#endif
    int h = 0;
    int vb[NT], vbn[NT];
    WaveBn<NT> wbn;   // BatchNorm finished in this launch (p.bn): the wave's running column sums
    wave_bn_reset(wbn);
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

      int ks3 = 0;  // byte offset of the current k-slab pair inside an n-block's fragments
      for (int c = 0; c < NC; ++c, ++h) {
#ifdef P3_DBG_TIME
        const long long d_a = clock64();
#endif
        p3_wait(pfull + (h & 1), P3_PRODUCERS * ((h >> 1) + 1));
#ifdef P3_DBG_TIME
        d_pf += clock64() - d_a;
#endif
        const int bufoff = (h & 1) * pbuf;
        int tr = 0, tq = 0;
        for (int t = 0; t < T; ++t) {
          const char* const abase = xsm + bufoff + (tr * Wp + tq) * P3_ROW;
          const bool last_of_chunk = t == T - 1;
          const bool last_of_tile = last_of_chunk && c == NC - 1;
          // ---- k-slab 0 of this (chunk, tap): B fragments in b0 (fetched a slab ago)
          loadB(b1, vb, ks3 + 3072);
          readB(b0, h & 1, 0);
#pragma unroll
          for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int q = 0; q < NA; ++q)
              fa[i][q] = *reinterpret_cast<const bf16x8*>(abase + a_row[i] + q * 64);
          __builtin_amdgcn_sched_barrier(0);
#ifdef P3_DBG_TIME
          {
            const long long d_0 = clock64();
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(B_LDS ? 0 : NT * 3) : "memory");
            const long long d_1 = clock64();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            d_vm += d_1 - d_0;
            d_lg += clock64() - d_1;
          }
          __builtin_amdgcn_sched_barrier(0);
#endif
          mma(b0);
          __builtin_amdgcn_sched_barrier(0);
          // ---- k-slab 1: B fragments in b1; b0 <- the next step's (or the next tile's first)
          if (last_of_tile) loadB(b0, vbn, 0);
          else loadB(b0, vb, ks3 + 6144);
          readB(b1, h & 1, 1);
#pragma unroll
          for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int q = 0; q < NA; ++q)
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

      // -------------------------------------------------------------- statistics
      if (p.bn.acc != nullptr) {
        wave_bn_tile<MT, NT>(acc, wbn, p.bn.acc, n0 + wn * WTN, p.N, p.M - (m0 + wm * WTM), half, l31,
                             PL::POST);
        if (round == my_tiles - 1) wave_bn_flush(wbn, p.bn.acc, p.N, half, l31);  // in front of the stores
      } else if (p.stat_partial != nullptr) {
        const int tile_m = m0 / BM;
        if (p.stat_rows == 32 && MT > 1) {
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
      // one store = 2 rows x 32 columns = two full 128-byte lines; rows past M get an
      // out-of-range lane offset
      const int rows_left = p.M - (m0 + wm * WTM + 4 * half);
      wave_epilogue<MT, NT>(acc, e_sc, e_sh, e_voff, rows_left, p.ldc, p.act, p.residual != nullptr,
                            rsrc_c, rsrc_r, false);
    }
    __builtin_amdgcn_s_setprio(0);
#ifdef P3_DBG_TIME
    if (blockIdx.x == 8 && (tid == 0 || tid == 64 * (MATRIX - 1))) {
      const long long c = clock64() - d_t0, w = wall_clock64() - d_w0;
      printf("p3 matrix wave %d: tiles %d chunks %d taps %d: total %lld cycles = %lld ticks of 100 MHz "
             "(%.2f GHz); waiting for the patch %lld, slab-0 waits: B fragments %lld, A fragments %lld\n",
             wave, my_tiles, NC, T, c, w, (double)c / (double)w * 0.1, d_pf, d_vm, d_lg);
    }
#endif
  }
#endif
}

// ====================================================================================
// conv_u3_kernel: 1x1 convolutions with NO producer waves.
//
// What the in-kernel timers of conv_p3_kernel / conv_x3_kernel show for the 1x1 layers (60 % of
// the trunks' convolution time): a dedicated producer wave next to two MFMA-issuing waves of its
// SIMD gets one instruction in 15-20 cycles -- it has no second wave to hide its own dependency
// and LDS latencies behind, and the matrix waves own the issue port -- so the matrix waves wait
// for the patch.  Here every wave does both jobs, and the compiler interleaves them in ONE
// instruction stream: 8 waves (two per SIMD, 256 VGPRs each), wave w owns output columns
// [32 w, 32 w + 32) of a BM x 256 tile for ALL BM rows (MT = BM / 32 MFMA blocks: a B fragment
// fetched from L2 feeds MT MFMAs), and, between the MFMAs of K-chunk g, transforms its 1/8 share
// of the rows of K-chunk g + 1 (BatchNorm + ReLU prologue, block end, three-way bf16 split) into
// the other patch buffer.  One raw s_barrier per K-chunk (48 MFMAs per wave) swaps the buffers;
// raw A rows are fetched two chunks ahead into registers, B fragments one k-slab ahead.
// (Round 6, measured and dropped: a ring of three / four raw-row sets -- inside a trunk the rows come
// from HBM, not from the Infinity Cache scripts/convbench.py keeps them in (--rotate: 1024 -> 256 block
// end 48 us cache-hot, 69 us from HBM) -- is 5-19 % SLOWER on the 128-row single-input form, cache-hot
// and from HBM alike, and changes nothing on the 64-row forms: profiles/r06_i_*.)
// WAVES = 8: two waves per SIMD, 256 registers each, a wave owns BM x 32 outputs;
// WAVES = 4: ONE wave per SIMD with the whole 512-register file, a wave owns BM x 64 outputs and
// double-buffers its A fragments (no partner wave to hide LDS latency behind).
// LINEAR: stride 1 (input pixel = output pixel).  A compile-time flag: as a runtime one the
// strided path's division constants stayed live through the chunk loop, were spilled, and were
// reloaded from scratch behind every chunk's MFMAs -- each reload followed by an
// s_waitcnt vmcnt(0) that drained the wave's whole prefetch queue (profiles/archive/r03_n_*).
// DUAL: 0 = one input; 1 = block end with an identity skip (second input added as is); 2 = block
// end whose skip path has its own BatchNorm.  Compile-time: the identity form carries half the
// prologue vectors and its chunk body has no branch.
template <int BM, int DUAL, int WAVES, int LINEAR, int MATH>
__global__ __launch_bounds__(WAVES * 64) void conv_u3_kernel(IgemmParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
  typedef Planes<MATH> PL;
  constexpr int P3_ROW = PL::ROW, NA = PL::NA;
  constexpr int BN = 256, MT = BM / 32, NT = 8 / WAVES;
  constexpr int RG = WAVES * 8;                  // rows per group of the transform's thread map
  constexpr int NPT = BM / RG;                   // float4 of a K-chunk's A rows per thread
  constexpr bool ADB = WAVES == 4;               // A fragments double-buffered across the k-slabs
  static_assert(NPT >= 1 && MT >= 1 && (WAVES == 8 || WAVES == 4), "tile");
  constexpr int PBUF = BM * P3_ROW;
  extern __shared__ __attribute__((aligned(16))) char xsm[];  // [2][PBUF] + prologue vectors [NV][Cin]
  // The prologue vectors (scale / shift / centre per input channel, twice for a skip path with its
  // own BatchNorm) live in LDS for the whole launch (round 6).  As per-chunk global loads into a
  // `cur` / `nxt` register pair they cost 24-48 VGPRs (the 128-row instances spilled) and sat in
  // the wave's one in-order vmcnt queue between the raw-row and B-fragment prefetches.
  constexpr int NV = DUAL == 2 ? 6 : 3;
  float* const vlds = reinterpret_cast<float*>(xsm + 2 * PBUF);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5;
  const int l31 = lane & 31;
  const int NC = p.Cin / 32;
  const int HoWo = p.Ho * p.Wo;
  const int trow = tid >> 3;          // this thread's row inside a group of RG rows
  const int lk4 = (tid & 7) * 4;

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
  const int G = my_tiles * NC;  // K-chunks this workgroup streams

  const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<char*>(reinterpret_cast<const char*>(p.A)), 0, (int)p.a_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rsrc_a2 = rsrc_a;
  if constexpr (DUAL)
    rsrc_a2 = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(p.A2)), 0, (int)p.a_bytes, 0x00020000);
  const int KS3 = (p.K / 16) * 3072;
  const __amdgpu_buffer_rsrc_t rsrc_b = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<char*>(reinterpret_cast<const char*>(p.Bfrag)), 0, (int)((long)p.N * p.K * 6),
      0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_c = __builtin_amdgcn_make_buffer_rsrc(
      reinterpret_cast<char*>(p.C), 0, (int)p.c_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_r = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<char*>(reinterpret_cast<const char*>(p.residual ? p.residual : p.C)), 0,
      (int)p.c_bytes, 0x00020000);   // (ldr == ldc: the output's extent)
  const bool has_pro = p.in_scale != nullptr;
  const float relu_floor = p.in_relu ? 0.f : -__builtin_huge_valf();
  constexpr bool linear = LINEAR != 0;

  // ---------------------------------------------------------------- the A side (every thread)
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  for (int ch = tid; ch < p.Cin; ch += WAVES * 64) {   // neutral vectors where there is no prologue
    vlds[ch] = has_pro ? p.in_scale[ch] : 1.f;
    const float t1 = has_pro ? p.in_shift[ch] : 0.f;
    vlds[p.Cin + ch] = t1;
    vlds[2 * p.Cin + ch] = (has_pro && p.in_center) ? p.in_center[ch] : 0.f;
    if constexpr (DUAL == 2) {
      vlds[3 * p.Cin + ch] = p.in2_scale[ch];
      vlds[4 * p.Cin + ch] = t1 + p.in2_shift[ch];   // both shifts in one add
      vlds[5 * p.Cin + ch] = p.in2_center ? p.in2_center[ch] : 0.f;
    }
  }
  __syncthreads();
  struct Raw {
    f32x4 a[NPT];
    f32x4 a2[DUAL ? NPT : 1];
    unsigned ok;
    int m0;  // first row of the tile if this workgroup writes side_out for it, else -1
    int ci;  // first input channel of the chunk
  };
  int a_voff[NPT];
  unsigned a_ok = 0;
  int l_round = 0, l_c = 0, l_m0 = 0, l_side = 0;
  auto setup_tile = [&](int round) {
    int m0, n0;
    tile_of(round, m0, n0);
    l_m0 = m0;
    l_side = n0 == 0;
    a_ok = 0;
#pragma unroll
    for (int i = 0; i < NPT; ++i) {
      const int m = m0 + i * RG + trow;
      a_voff[i] = BUF_OOB;
      if (m < p.M) {
        int pix = m;
        if constexpr (!linear) {
          const int img = m / HoWo;
          const int rem = m - img * HoWo;
          const int ho = rem / p.Wo;
          pix = (img * p.H + ho * p.stride) * p.W + (rem - ho * p.Wo) * p.stride;
        }
        a_voff[i] = (pix * p.lda + lk4) * 4;
        a_ok |= 1u << i;
      }
    }
  };
  auto load_raw = [&](Raw& r) {   // (branch-free; advance_raw() moves the cursor afterwards)
    const bool live = l_round < my_tiles;
    const int soff = l_c * 128;
    r.ok = live ? a_ok : 0u;
    r.m0 = l_side ? l_m0 : -1;
    r.ci = l_c * 32;
#pragma unroll
    for (int i = 0; i < NPT; ++i) {
      const int vo = live ? a_voff[i] : BUF_OOB;
      r.a[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, vo, soff, 0));
      if constexpr (DUAL)
        r.a2[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_a2, vo, soff, 0));
    }
  };
  auto advance_raw = [&]() {
    if (l_round < my_tiles && ++l_c == NC) {
      l_c = 0;
      if (++l_round < my_tiles) setup_tile(l_round);
    }
  };
  // prologue + split of this thread's float4s of one chunk into patch buffer `buf`.  Branch-free
  // in the single-input form (neutral vectors where the convolution has no prologue: x*1+0 and
  // max(x, -inf) are exact), so that it shares ONE basic block with the MFMAs of the chunk and
  // the scheduler can interleave the two.
  auto transform = [&](const Raw& r, char* buf) {
    struct {
      f32x4 s, t, c, s2, t2, c2;
    } cur;
    const float* const vp = vlds + r.ci + lk4;
    cur.s = *reinterpret_cast<const f32x4*>(vp);
    cur.t = *reinterpret_cast<const f32x4*>(vp + p.Cin);
    cur.c = *reinterpret_cast<const f32x4*>(vp + 2 * p.Cin);
    if constexpr (DUAL == 2) {
      cur.s2 = *reinterpret_cast<const f32x4*>(vp + 3 * p.Cin);
      cur.t2 = *reinterpret_cast<const f32x4*>(vp + 4 * p.Cin);
      cur.c2 = *reinterpret_cast<const f32x4*>(vp + 5 * p.Cin);
    }
#pragma unroll
    for (int i = 0; i < NPT; ++i) {
      f32x4 v = r.a[i];
      const bool ok = (r.ok >> i) & 1u;
      if constexpr (DUAL) {
        if constexpr (DUAL == 2) {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            v[e] = fmaxf(fmaf(v[e] - cur.c[e], cur.s[e],
                              fmaf(r.a2[i][e] - cur.c2[e], cur.s2[e], cur.t2[e])), relu_floor);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            v[e] = fmaxf(fmaf(v[e] - cur.c[e], cur.s[e], cur.t[e]) + r.a2[i][e], relu_floor);
        }
        if (!ok) v = zero4;
        if (p.side_out != nullptr && r.m0 >= 0 && ok)
          *reinterpret_cast<f32x4*>(p.side_out + (long)(r.m0 + i * RG + trow) * p.lda + r.ci + lk4) = v;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[e] = fmaxf(fmaf(v[e] - cur.c[e], cur.s[e], cur.t[e]), relu_floor);
          v[e] = ok ? v[e] : 0.f;
        }
      }
      p3_split_store<MATH>(v, buf + (i * RG + trow) * P3_ROW + lk4 * 2);
    }
  };

  // ---------------------------------------------------------------- the matrix side (per wave)
  constexpr int HM = MT / 2;  // row blocks per half (two-wave form)
  static_assert(ADB || HM >= 1, "tile");
  bf16x8 fa[ADB ? MT : 1][NA], fa1[ADB ? MT : 1][NA], fh0[ADB ? 1 : HM][NA], fh1[ADB ? 1 : HM][NA];
  // BA (B fragments ahead): 0 = k-slab 1's fragments requested at the start of their chunk and the
  // next chunk's slab-0 fragments half a chunk ahead (one slab of lead: 6 * MT MFMAs against an L2
  // round trip); 1 = slab 1's fragments a whole chunk ahead (a second set, alternating by chunk
  // parity); 2 = both slabs' fragments a whole chunk ahead (two more sets: the 64-row forms have
  // the registers).  Single-input forms only: 1x1 layer list 2.219 -> 2.158 ms from HBM (1024 ->
  // 256: 179 -> 207 TF/s); the dual forms measured 3 % SLOWER with it (the 128-row ones spill 6-8
  // registers) and keep one slab of lead (profiles/r06_u_conv_u3_b_fragments_a_chunk_ahead.txt).
  constexpr int BA = (ADB || DUAL != 0) ? 0 : (BM == 64 ? U3_B_AHEAD_64 : U3_B_AHEAD_128);
  bf16x8 b0[NT][3], b1[NT][3];
  bf16x8 b0x[BA == 2 ? NT : 1][3], b1x[BA >= 1 ? NT : 1][3];
  f32x16 acc[MT][NT];
  const int a_off = l31 * P3_ROW + half * 16;   // + i * 32 * P3_ROW + q * 64 + s * 32
  auto loadB = [&](bf16x8 (&b)[NT][3], const int (&vb)[NT], int soff) {
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int q = 0; q < 3; ++q)
        b[j][q] = __builtin_bit_cast(
            bf16x8, __builtin_amdgcn_raw_buffer_load_b128(
                        rsrc_b, vb[j] == BUF_OOB ? BUF_OOB : vb[j] + q * 1024, soff, 0));
  };
  auto readA = [&](bf16x8 (&f)[ADB ? MT : 1][NA], const char* buf, int s) {
#pragma unroll
    for (int i = 0; i < (ADB ? MT : 1); ++i)
#pragma unroll
      for (int q = 0; q < NA; ++q)
        f[i][q] = *reinterpret_cast<const bf16x8*>(buf + a_off + i * 32 * P3_ROW + q * 64 + s * 32);
  };
  auto readH = [&](bf16x8 (&f)[ADB ? 1 : HM][NA], const char* buf, int s, int h) {
#pragma unroll
    for (int i = 0; i < (ADB ? 1 : HM); ++i)
#pragma unroll
      for (int q = 0; q < NA; ++q)
        f[i][q] = *reinterpret_cast<const bf16x8*>(buf + a_off + (h * HM + i) * 32 * P3_ROW + q * 64 +
                                                   s * 32);
  };
  constexpr int NP = PL::NP;
  auto mma = [&](const bf16x8 (&f)[ADB ? MT : 1][NA], const bf16x8 (&b)[NT][3]) {
#pragma unroll
    for (int q = 0; q < NP; ++q)
#pragma unroll
      for (int i = 0; i < (ADB ? MT : 1); ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
          acc[i][j] = plane_mfma<MATH>(f[i][PL::PA[q]], b[j][PL::PB[q]], acc[i][j]);
  };
  auto mmaH = [&](const bf16x8 (&f)[ADB ? 1 : HM][NA], const bf16x8 (&b)[NT][3], int h) {
#pragma unroll
    for (int q = 0; q < NP; ++q)
#pragma unroll
      for (int i = 0; i < (ADB ? 1 : HM); ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
          acc[h * HM + i][j] = plane_mfma<MATH>(f[i][PL::PA[q]], b[j][PL::PB[q]], acc[h * HM + i][j]);
  };
  auto vb_of = [&](int n0, int (&vb)[NT]) {
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int nb = n0 / 32 + wave * NT + j;
      vb[j] = nb * 32 < p.N ? nb * KS3 + lane * 16 : BUF_OOB;
    }
  };

  // ---------------------------------------------------------------- prologue of the stream
  // vector schedule: chunk k's vectors are channels (k % NC) * 32 ..; `cur` must hold chunk k's
  // when chunk k is transformed
  //
  // RB = 2 (round 6, single-input forms): raw rows requested TWO chunks at a time, every second
  // chunk.  A wave has ONE in-order vmcnt queue: the B fragments of the next k-slab (L2 hits,
  // needed half a chunk after they are requested) queue behind the raw rows requested just before
  // them (HBM), so every chunk waits out part of an HBM latency with one chunk of rows in flight.
  // Two chunks per burst put twice the bytes behind each such wait, for four raw sets in
  // registers instead of two.  Measured per layer from HBM (scripts/convbench.py --rotate 8,
  // profiles/r06_p_conv_u3_raw_batch_ab.txt): K >= 512 layers +7...15 % (1024 -> 512: 224 -> 258
  // TF/s), 256 -> 1024 -6 %, the 1x1 layer list 2.184 -> 2.158 ms; the dual (block-end) forms
  // measured 1-5 % SLOWER with it and keep one chunk per request; the step is unchanged within
  // its noise (5.44-5.49 ms of conv launches either way), and the file compiles 40 % longer:
  // NOT the default (-DU3_RAW_BATCH=2 builds it).
  // (Also measured and dropped: the dual forms' side_out rows stored two chunks per burst instead
  // of every chunk -- stores count in the same queue -- block ends 2.870 -> 2.946 ms, i.e. slower:
  // profiles/r06_s_conv_u3_side_out_store_bursts.txt.)
  constexpr int RB = (U3_RAW_BATCH == 2 && DUAL == 0) ? 2 : 1;
  Raw rx, ry, rz, rw;   // RB = 2: chunk j lives in set j % 4 (rx, ry, rz, rw); RB = 1: rx / ry alternate
  setup_tile(0);
  load_raw(rx);   // chunk 0
  advance_raw();
  load_raw(ry);   // chunk 1
  advance_raw();
  if constexpr (RB == 2) {
    load_raw(rz);   // chunks 2, 3
    advance_raw();
    load_raw(rw);
    advance_raw();
  }
  int m0 = 0, n0 = 0;
  tile_of(0, m0, n0);
  int vb[NT], vbn[NT];
  vb_of(n0, vb);
#pragma unroll
  for (int j = 0; j < NT; ++j) vbn[j] = BUF_OOB;
  if (my_tiles > 1) {
    int m1, n1;
    tile_of(1, m1, n1);
    vb_of(n1, vbn);
  }
  loadB(b0, vb, 0);
  if constexpr (BA >= 1) loadB(b1, vb, 3072);
  transform(rx, xsm);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  int c = 0, round = 0, ks3 = 0;
  WaveBn<NT> wbn;   // BatchNorm finished in this launch (p.bn): the wave's running column sums
  wave_bn_reset(wbn);
#ifdef P3_DBG_TIME
  long long d_blk = 0, d_epi = 0, d_bar = 0, d_book = 0;
  const long long d_t0 = clock64(), d_w0 = wall_clock64();
#endif
  // one K-chunk: MFMAs on patch (g & 1) / transform of `rn` (chunk g + 1) into patch ((g+1) & 1) /
  // raw loads of chunk g + 2 into `rf`.  Everything up to the transform is ONE basic block.
  auto chunk = [&](int g, Raw& rn, Raw& rf, Raw& rf2, auto pair_tag, bf16x8 (&b0c)[NT][3],
                   bf16x8 (&b1c)[NT][3], auto& b0n, auto& b1n) {
    constexpr bool load_pair = decltype(pair_tag)::value;   // RB = 2: this chunk ends with a request burst
    const char* const pb = xsm + (g & 1) * PBUF;
    char* const pn = xsm + ((g + 1) & 1) * PBUF;
    const bool last_of_tile = c == NC - 1;
    int vb_s0[NT];                                         // slab 0 of the next chunk
#pragma unroll
    for (int j = 0; j < NT; ++j) vb_s0[j] = last_of_tile ? vbn[j] : vb[j];
    const int so_s0 = last_of_tile ? 0 : ks3 + 6144;
#ifdef P3_DBG_TIME
    const long long d_0 = clock64();
#endif
#ifdef U3_RAW_FIRST
    if constexpr (RB == 1) load_raw(rf);
    loadB(b1c, vb, ks3 + 3072);
#else
    // k-slab 1's B fragments BEFORE the raw rows of chunk g + 2: vmcnt retires in order, and the
    // fragments (L2 hits, needed half a chunk from here) would otherwise wait out the HBM latency
    // of rows that nobody reads before the next chunk
    if constexpr (BA == 0) loadB(b1c, vb, ks3 + 3072);
    if constexpr (BA == 2) loadB(b0n, vb_s0, so_s0);              // the NEXT chunk's fragments
    if constexpr (BA >= 1) loadB(b1n, vb_s0, so_s0 + 3072);
    if constexpr (RB == 1) load_raw(rf);
#endif
    if constexpr (ADB) {
      readA(fa, pb, 0);
      readA(fa1, pb, 1);
      mma(fa, b0c);
      loadB(b0n, vb_s0, so_s0);
      mma(fa1, b1c);
    } else {
      // the MT row blocks in two halves with a fragment set each: the reads of one half land
      // under the MFMAs of the other (no spare registers for a second full set)
      readH(fh0, pb, 0, 0);
      readH(fh1, pb, 0, 1);
      mmaH(fh0, b0c, 0);
      readH(fh0, pb, 1, 0);
      mmaH(fh1, b0c, 1);
      if constexpr (BA < 2) loadB(b0n, vb_s0, so_s0);
      readH(fh1, pb, 1, 1);
      mmaH(fh0, b1c, 0);
      mmaH(fh1, b1c, 1);
    }
    transform(rn, pn);
#ifndef U3_NO_SCHED
    // Instruction order of the block (the scheduler's own choice bunches the transform behind the
    // last MFMAs and issues every fragment read right in front of its MFMA): k-slab 0's fragment
    // reads and the global loads first, a few transform instructions under their latency, then
    // per MFMA two VALU instructions of the transform and at most one LDS read (k-slab 1's
    // fragments, each as soon as the MFMAs that still read its registers have issued), one LDS
    // write, one global load.
#ifndef U3_VPM
#define U3_VPM 2   // VALU instructions of the transform per MFMA
#endif
    if constexpr (ADB) {
#pragma unroll
      for (int k = 0; k < 2 * NP * MT * NT; ++k) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, U3_VPM, 0);
      }
    } else {
      // four phases of 6 * HM * NT MFMAs (half 0 / half 1 of k-slab 0, then of k-slab 1).  Only
      // the first half's fragments are read before the first MFMA (all eight waves read at once
      // right after the barrier: every kilobyte in front of the first MFMA is exposed); the
      // other reads trickle, one per MFMA, a phase ahead of their use.
      // (Measured and dropped, profiles/r04_a_convbench_u3_loads_first.txt: pinning the chunk's raw-row
      // and slab-1 B loads in front of the first MFMA with a VMEM-read group changes no layer by
      // more than 2 %.)
      __builtin_amdgcn_sched_group_barrier(0x100, NA * HM, 0);
#pragma unroll
      for (int ph = 0; ph < 4; ++ph) {
#pragma unroll
        for (int k = 0; k < NP * HM * NT; ++k) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, MATH == MATH_F16X3 ? 2 * U3_VPM : U3_VPM, 0);
          if (ph < 3 && k < NA * HM) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
      }
    }
#endif
    // ---- bookkeeping (branches from here on)
#ifdef P3_DBG_TIME
    const long long d_1 = clock64();
    d_blk += d_1 - d_0;
#endif
    if constexpr (RB == 1) {
      advance_raw();
    } else if constexpr (load_pair) {
      load_raw(rf);    // chunks g + 3, g + 4
      advance_raw();
      load_raw(rf2);
      advance_raw();
    }
    ks3 += 6144;
#ifdef P3_DBG_TIME
    const long long d_2 = clock64();
    d_book += d_2 - d_1;
#endif
    if (last_of_tile) {
      // -------------------------------------------------------------- statistics
      if (p.bn.acc != nullptr) {
        wave_bn_tile<MT, NT>(acc, wbn, p.bn.acc, n0 + wave * NT * 32, p.N, p.M - m0, half, l31, PL::POST);
        if (round == my_tiles - 1) wave_bn_flush(wbn, p.bn.acc, p.N, half, l31);  // in front of the stores
      } else if (p.stat_partial != nullptr) {
        const int col0 = n0 + wave * NT * 32;
        if (p.stat_rows == 32 && MT > 1) {
#pragma unroll
          for (int i = 0; i < MT; ++i)
            wave_stats_block<NT>(acc[i], p.stat_partial, m0 / 32 + i, p.M - (m0 + i * 32), col0, p.N,
                                 half, l31, PL::POST);
        } else if (p.stat_rows > 0 && p.stat_rows < BM)
          wave_stats_fine<MT, NT>(acc, p.stat_partial, p.stat_rows, m0, p.M, col0, p.N, half, l31,
                                  PL::POST);
        else
          wave_stats<MT, NT>(acc, p.stat_partial, m0 / BM, p.M - m0, BM, col0, p.N, half, l31, PL::POST);
      }
      // -------------------------------------------------------------- epilogue from registers
      // (Measured alternatives, round 3, profiles/archive/r03_h_*: turning each 32x32 block around in a
      // per-wave LDS square and storing 128-byte rows with 16-byte stores -- a quarter of the
      // store instructions -- changes nothing (126 vs 122 us on the 64->256 layer): the burst
      // drains at ~5.6 TB/s either way, and what is lost is that a wave's next loads queue
      // behind its own stores in the one in-order vmcnt.  64-row tiles held to 128 VGPRs so that
      // TWO workgroups share a CU and one computes while the other drains: 2x slower, 50
      // registers spilled into the chunk loop.)
      float e_sc[NT], e_sh[NT];
      int e_voff[NT];
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int col = n0 + (wave * NT + j) * 32 + l31;
        const bool okc = col < p.N;
        e_sc[j] = ((okc && p.scale) ? p.scale[col] : 1.f) * PL::POST;
        e_sh[j] = (okc && p.shift) ? p.shift[col] : 0.f;
        e_voff[j] = okc ? (int)((((long)(m0 + 4 * half)) * p.ldc + col) * 4) : BUF_OOB;
      }
      const int rows_left = p.M - (m0 + 4 * half);
      wave_epilogue<MT, NT>(acc, e_sc, e_sh, e_voff, rows_left, p.ldc, p.act, p.residual != nullptr,
                            rsrc_c, rsrc_r, true);
      c = 0;
      ks3 = 0;
      if (++round < my_tiles) {
        tile_of(round, m0, n0);
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          vb[j] = vbn[j];
          vbn[j] = BUF_OOB;
        }
        if (round + 1 < my_tiles) {
          int m1, n1;
          tile_of(round + 1, m1, n1);
          vb_of(n1, vbn);
        }
      }
    } else {
      ++c;
    }
#ifdef P3_DBG_TIME
    const long long d_3 = clock64();
    d_epi += d_3 - d_2;
#endif
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this thread's patch writes are in LDS
    __builtin_amdgcn_s_barrier();
#ifdef P3_DBG_TIME
    d_bar += clock64() - d_3;
#endif
  };
  auto& b0alt = [&]() -> auto& { if constexpr (BA == 2) return b0x; else return b0; }();
  auto& b1alt = [&]() -> auto& { if constexpr (BA >= 1) return b1x; else return b1; }();
  if constexpr (RB == 1) {
    for (int g = 0; g < G; g += 2) {
      // (B sets by chunk parity: BA = 0 uses b0 / b1 throughout; BA = 1 alternates b1 / b1x;
      // BA = 2 alternates both)
      chunk(g, ry, rx, rx, std::false_type{}, b0, b1, b0alt, b1alt);
      if (g + 1 < G) chunk(g + 1, rx, ry, ry, std::false_type{}, b0alt, b1alt, b0, b1);
    }
  } else {
    for (int g = 0; g < G; g += 4) {             // chunk g + 1 is transformed during chunk g
      chunk(g, ry, rx, rx, std::false_type{}, b0, b1, b0alt, b1alt);
      if (g + 1 < G) chunk(g + 1, rz, rx, ry, std::true_type{}, b0alt, b1alt, b0, b1);   // requests chunks g + 4, g + 5
      if (g + 2 < G) chunk(g + 2, rw, rx, rx, std::false_type{}, b0, b1, b0alt, b1alt);
      if (g + 3 < G) chunk(g + 3, rx, rz, rw, std::true_type{}, b0alt, b1alt, b0, b1);   // requests chunks g + 6, g + 7
    }
  }
#ifdef P3_DBG_TIME
  if (blockIdx.x == 8 && (tid == 0 || tid == (WAVES - 1) * 64)) {
    const long long cy = clock64() - d_t0, w = wall_clock64() - d_w0;
    printf("u3 wave %d: tiles %d chunks/tile %d: total %lld cycles = %lld ticks of 100 MHz (%.2f GHz): "
           "MFMA+transform blocks %lld, bookkeeping %lld, epilogues %lld, barriers %lld\n",
           wave, my_tiles, NC, cy, w, (double)cy / (double)w * 0.1, d_blk, d_book, d_epi, d_bar);
  }
#endif
#endif
}

template <int BM, int DUAL, int WAVES, int LINEAR, int MATH>
int launch_u3_(const IgemmParams& p, hipStream_t stream) {
  constexpr int NV = DUAL == 2 ? 6 : 3;   // prologue vectors kept in LDS
  const int smem_bytes = 2 * BM * Planes<MATH>::ROW + NV * p.Cin * 4;
  constexpr int smem_max = 2 * BM * Planes<MATH>::ROW + NV * U3_MAX_CIN * 4;
  auto kern = conv_u3_kernel<BM, DUAL, WAVES, LINEAR, MATH>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, smem_max);
    if (e != hipSuccess) {
      vlnce_set_error("conv_u3: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
      return 2;
    }
    attr_set = true;
  }
  IgemmParams q = p;
  q.tiles_m = ceil_div(p.M, BM);
  q.tiles_n = ceil_div(p.N, 256);
  q.splitk = 1;
  const long nwg = (long)q.tiles_m * q.tiles_n;
  if (nwg <= 0 || nwg > 0x7fffffffL) {
    vlnce_set_error("conv_u3: bad grid %ld", nwg);
    return 1;
  }
  const int cus = x3_cus();
  const unsigned grid = nwg <= cus ? (unsigned)nwg : (unsigned)cus;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(WAVES * 64), smem_bytes, stream, q);
  VLNCE_CHECK_LAUNCH("conv_u3");
  return 0;
}

// conv_s3_kernel: stride-1 1x1 convolutions with SHORT K (64 or 128 input channels) and wide N
// (the 64->256 / 128->512 expansions of the bottlenecks: 268 / 134 MB of output per launch).
// These launches are bound by the HBM write of their output, and in conv_u3_kernel they reach
// 2.2-2.7 TB/s: a tile there is 2-4 K-chunks of MFMAs and a 128 KB store burst, and the loads of
// the next tile (B fragments of its second k-slab, raw A two chunks ahead) queue behind the burst
// in the wave's one in-order vmcnt, so a wave alternates between draining and computing
// (profiles/archive/r03_h_u3_phase_timers_short_k.txt).  Here nothing a tile needs is loaded less than
// a tile before its use, and a tile's stores are issued UNDER the next tile's MFMAs:
//   * the B fragments of the wave's 32 columns for ALL of K stay in registers for the whole
//     launch (K = 64: 48 VGPRs, K = 128: 96) -- a workgroup keeps its column tile;
//   * the raw A rows of tile t + 2 (K = 128: t + 1) are requested while tile t is transformed
//     (64-row tiles: one float4 per thread and chunk);
//   * TWO accumulator sets: while the MFMAs of tile t fill one, the 32 stores of tile t - 1 drain
//     the other, a few behind every k-slab (sched_group_barrier pins the interleave; the epilogue
//     activation is a select so the slab body stays one basic block).  Stores are fire-and-forget:
//     when the store queue is full the wave stalls at a store and the SIMD's other wave issues
//     its MFMAs, so per tile a CU needs max(stores, MFMAs) instead of their sum.  Round 3's
//     serial form (transform, barrier, MFMAs, then 32 stores at the CU's ~12 B/cycle = 54 % of
//     the launch) measured 90.4 / 75.0 us on 64->256 / 128->512 at num_envs 64; this form 71.4 /
//     72.1 us (profiles/r04_a_conv_s3_pipelined_epilogue.txt);
//   * the first two tiles are peeled: the compiler's s_waitcnt at a loop header is the minimum
//     over the paths into it, and entered from the preamble the wait for the raw rows would
//     drain the previous tile's stores on every trip.
// 8 waves (two per SIMD), wave w owns columns [32w, 32w + 32) of a 64 x 256 tile; one barrier
// per tile.  Arithmetic, patch rows, fragment layout and statistics are conv_u3_kernel's.
template <int NCC, int MATH>
__global__ __launch_bounds__(512) void conv_s3_kernel(IgemmParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
  typedef Planes<MATH> PL;
  constexpr int P3_ROW = PL::ROW, NA = PL::NA, NP = PL::NP;
  constexpr int BM = 64, MT = 2, KS = NCC * 2;
  constexpr int CBUF = BM * P3_ROW;            // one chunk of a tile's patch
  constexpr int PBUF = NCC * CBUF;             // one tile's patch
  constexpr int SPS = 32 / KS;                 // stores of the previous tile behind each k-slab
  extern __shared__ __attribute__((aligned(16))) char xsm[];  // [2][PBUF] + prologue vectors

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5;
  const int l31 = lane & 31;
  const int trow = tid >> 3;
  const int lk4 = (tid & 7) * 4;

  const int n0 = ((int)blockIdx.x % p.tiles_n) * 256;
  const int wg = (int)blockIdx.x / p.tiles_n, nwg = (int)gridDim.x / p.tiles_n;
  const int my_tiles = (p.tiles_m - wg + nwg - 1) / nwg;

  const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<char*>(reinterpret_cast<const char*>(p.A)), 0, (int)p.a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_b = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<char*>(reinterpret_cast<const char*>(p.Bfrag)), 0, (int)((long)p.N * p.K * 6),
      0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_c = __builtin_amdgcn_make_buffer_rsrc(
      reinterpret_cast<char*>(p.C), 0, (int)p.c_bytes, 0x00020000);
  const float relu_floor = p.in_relu ? 0.f : -__builtin_huge_valf();
  const bool relu_out = p.act == VLNCE_ACT_RELU;  // the launcher admits VLNCE_ACT_NONE / _RELU only

  bf16x8 bres[KS][3];
  {
    const int vb = (n0 / 32 + wave) * KS * 3072 + lane * 16;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int q = 0; q < 3; ++q)
        bres[ks][q] = __builtin_bit_cast(
            bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rsrc_b, vb + q * 1024, ks * 3072, 0));
  }
  // prologue vectors always in LDS here: the second accumulator set takes their registers
  float* const vlds = reinterpret_cast<float*>(xsm + 2 * PBUF);  // [3][NCC * 32]
  if (tid < 8) {
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f}, one4 = {1.f, 1.f, 1.f, 1.f};
#pragma unroll
    for (int c = 0; c < NCC; ++c) {
      f32x4 s_ = one4, t_ = zero4, c_ = zero4;
      if (p.in_scale != nullptr) {
        s_ = ldg4(p.in_scale + c * 32 + lk4);
        t_ = ldg4(p.in_shift + c * 32 + lk4);
        if (p.in_center) c_ = ldg4(p.in_center + c * 32 + lk4);
      }
      *reinterpret_cast<f32x4*>(vlds + c * 32 + lk4) = s_;
      *reinterpret_cast<f32x4*>(vlds + NCC * 32 + c * 32 + lk4) = t_;
      *reinterpret_cast<f32x4*>(vlds + 2 * NCC * 32 + c * 32 + lk4) = c_;
    }
  }
  __syncthreads();
  const int col = n0 + wave * 32 + l31;
  const float e_sc = (p.scale ? p.scale[col] : 1.f) * PL::POST;
  const float e_sh = p.shift ? p.shift[col] : 0.f;

  // raw A ring: two tiles ahead for K = 64; ONE for K = 128, where the second
  // accumulator set leaves no registers for it (with the stores spread over the MFMA phase the
  // request of tile t + 1 sits behind the stores of tile t - 2 only, a whole tile old)
  constexpr int RING = NCC > 2 ? 1 : 2;
  f32x4 raw[RING][NCC];
  auto load_raw = [&](f32x4 (&r)[NCC], int round) {
    const int m = (wg + round * nwg) * BM + trow;
    const int vo = (round < my_tiles && m < p.M) ? (m * p.lda + lk4) * 4 : BUF_OOB;
#pragma unroll
    for (int c = 0; c < NCC; ++c)
      r[c] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, vo, c * 128, 0));
  };
#pragma unroll
  for (int k = 0; k < RING; ++k) load_raw(raw[k], k);
  f32x16 acc[2][MT];  // [tile parity][row block]
#pragma unroll
  for (int b = 0; b < 2; ++b)
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[b][i][r] = 0.f;
  const int a_off = l31 * P3_ROW + half * 16;

  WaveBn<1> wbn;   // BatchNorm finished in this launch (p.bn): the wave's running column sums
  wave_bn_reset(wbn);
  // statistics of a finished tile (raw accumulators, 32-row blocks)
  auto stats = [&](const f32x16 (&a)[MT], int m0) {
    if (p.bn.acc != nullptr) {
      wave_bn_tile<MT, 1>(reinterpret_cast<const f32x16(&)[MT][1]>(a), wbn, p.bn.acc, n0 + wave * 32,
                          p.N, p.M - m0, half, l31, PL::POST);
    } else if (p.stat_partial != nullptr) {
#pragma unroll
      for (int i = 0; i < MT; ++i)
        wave_stats_block<1>(reinterpret_cast<const f32x16(&)[1]>(a[i]), p.stat_partial,
                            m0 / 32 + i, p.M - (m0 + i * 32), n0 + wave * 32, p.N, half, l31, PL::POST);
    }
  };
  // stores [first, first + count) of the 32 of a finished tile; the registers are cleared behind
  auto stores = [&](f32x16 (&a)[MT], int m0, int first, int count) {
    const int rows_left = p.M - (m0 + 4 * half);
    const int e_voff = (int)((((long)(m0 + 4 * half)) * p.ldc + col) * 4);
#pragma unroll
    for (int k = first; k < first + count; ++k) {
      const int i = k >> 4, r2 = k & 15;
      const int rw = i * 32 + (r2 & 3) + 8 * (r2 >> 2);
      const float lin = a[i][r2] * e_sc + e_sh;
      const float v = relu_out ? (lin > 0.f ? lin : 0.f) : lin;  // (act is none or ReLU: selects, no branch)
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsrc_c,
                                            rw < rows_left ? e_voff : BUF_OOB, rw * p.ldc * 4, 0);
      a[i][r2] = 0.f;
    }
  };

  // one tile: cur = accumulator set of this tile, prv = the set of the tile before it (its
  // statistics and stores are issued here, under this tile's MFMAs)
  auto tile = [&](int round, f32x4 (&r)[NCC], f32x16 (&cur)[MT], f32x16 (&prv)[MT]) {
    const int m0 = (wg + round * nwg) * BM;
    const int m0_prev = (wg + (round - 1) * nwg) * BM;
    const bool has_prev = round > 0;
    char* const pb = xsm + (round & 1) * PBUF;
    const bool row_ok = m0 + trow < p.M;
#pragma unroll
    for (int c = 0; c < NCC; ++c) {
      f32x4 v = r[c];
      const f32x4 s_ = *reinterpret_cast<const f32x4*>(vlds + c * 32 + lk4);
      const f32x4 t_ = *reinterpret_cast<const f32x4*>(vlds + NCC * 32 + c * 32 + lk4);
      const f32x4 c_ = *reinterpret_cast<const f32x4*>(vlds + 2 * NCC * 32 + c * 32 + lk4);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[e] = fmaxf(fmaf(v[e] - c_[e], s_[e], t_[e]), relu_floor);
        v[e] = row_ok ? v[e] : 0.f;
      }
      p3_split_store<MATH>(v, pb + c * CBUF + trow * P3_ROW + lk4 * 2);
    }
    load_raw(r, round + RING);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (has_prev) stats(prv, m0_prev);
#pragma unroll
    for (int c = 0; c < NCC; ++c)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        bf16x8 f[MT][NA];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int q = 0; q < NA; ++q)
            f[i][q] = *reinterpret_cast<const bf16x8*>(pb + c * CBUF + a_off + i * 32 * P3_ROW +
                                                       q * 64 + s2 * 32);
#pragma unroll
        for (int q = 0; q < NP; ++q)
#pragma unroll
          for (int i = 0; i < MT; ++i)
            cur[i] = plane_mfma<MATH>(f[i][PL::PA[q]], bres[c * 2 + s2][PL::PB[q]], cur[i]);
        // the previous tile's next SPS stores ride behind this slab's 12 MFMAs (a tile whose
        // predecessor does not exist stores to the out-of-range offset: no branch in the body)
        stores(prv, has_prev ? m0_prev : p.M, (c * 2 + s2) * SPS, SPS);
        // schedule of the slab: its 12 MFMAs with the SPS stores spread evenly between them
        // (K = 64: 2 MFMAs, store, 1 MFMA, store, four times; K = 128: 3 MFMAs, store, four times)
        if constexpr (MATH == MATH_F16X3) {
          // 6 MFMAs per slab: K = 64: MFMA, store, store, MFMA, store (x2, then 2 MFMAs + 2 stores);
          // K = 128: 3 MFMAs, 2 stores, twice
#pragma unroll
          for (int g = 0; g < 2; ++g) {
            if constexpr (SPS == 8) {
              __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
              __builtin_amdgcn_sched_group_barrier(0x040, 1, 0);
              __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
              __builtin_amdgcn_sched_group_barrier(0x040, 2, 0);
              __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
              __builtin_amdgcn_sched_group_barrier(0x040, 1, 0);
            } else {
              __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
              __builtin_amdgcn_sched_group_barrier(0x040, 2, 0);
            }
          }
        } else {
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            if constexpr (SPS == 8) {
              __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);  // MFMA
              __builtin_amdgcn_sched_group_barrier(0x040, 1, 0);  // VMEM write
              __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
              __builtin_amdgcn_sched_group_barrier(0x040, 1, 0);
            } else {
              __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
              __builtin_amdgcn_sched_group_barrier(0x040, 1, 0);
            }
          }
        }
      }
  };
  // The first two tiles are peeled: the compiler's s_waitcnt at a loop header is the minimum
  // over the paths into it, and entered straight from the preamble the first wait for raw rows
  // would be vmcnt(3) on EVERY trip -- i.e. the stores of the tile before would be drained every
  // second tile (round 3's unpeeled form had exactly that: vmcnt(3) / vmcnt(35) alternated in its ISA).
  // Behind the peeled tiles both ways into the loop have a tile's 32 stores after the request.
  if (my_tiles <= 0) return;  // (cannot happen with launch_s3's grid; uniform per workgroup)
  tile(0, raw[0], acc[0], acc[1]);
  if (1 < my_tiles) tile(1, raw[RING - 1], acc[1], acc[0]);
  for (int round = 2; round < my_tiles; round += 2) {
    tile(round, raw[0], acc[0], acc[1]);
    if (round + 1 < my_tiles) tile(round + 1, raw[RING - 1], acc[1], acc[0]);
  }
  // the last tile's epilogue has nothing left to hide under
  {
    const int m0 = (wg + (my_tiles - 1) * nwg) * BM;
    if ((my_tiles - 1) & 1) {
      stats(acc[1], m0);
      if (p.bn.acc != nullptr) wave_bn_flush(wbn, p.bn.acc, p.N, half, l31);  // in front of the stores
      stores(acc[1], m0, 0, 32);
    } else {
      stats(acc[0], m0);
      if (p.bn.acc != nullptr) wave_bn_flush(wbn, p.bn.acc, p.N, half, l31);
      stores(acc[0], m0, 0, 32);
    }
  }
#endif
}

template <int NCC, int MATH>
int launch_s3(const IgemmParams& p, hipStream_t stream) {
  constexpr int smem_bytes = 2 * NCC * 64 * Planes<MATH>::ROW + 3 * NCC * 32 * 4;  // two tile patches + the prologue vectors
  auto kern = conv_s3_kernel<NCC, MATH>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, smem_bytes);
    if (e != hipSuccess) {
      vlnce_set_error("conv_s3: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
      return 2;
    }
    attr_set = true;
  }
  IgemmParams q = p;
  q.tiles_m = ceil_div(p.M, 64);
  q.tiles_n = p.N / 256;
  q.splitk = 1;
  const int cus = x3_cus();
  long grid = (long)q.tiles_m * q.tiles_n;
  if (grid > cus) grid = cus - cus % q.tiles_n;  // resident workgroups, a multiple of tiles_n
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), smem_bytes, stream, q);
  VLNCE_CHECK_LAUNCH("conv_s3");
  return 0;
}

template <int BM, int DUAL, int WAVES, int MATH>
int launch_u3(const IgemmParams& p, hipStream_t stream) {
  return p.stride == 1 ? launch_u3_<BM, DUAL, WAVES, 1, MATH>(p, stream)
                       : launch_u3_<BM, DUAL, WAVES, 0, MATH>(p, stream);
}

// w_ohwi [N][KH][KW][Cin] fp32 -> B fragments [N/32][K/16][3][64 lanes][8 bf16]: k-slab
// ks = ((chunk * T + tap) * 2 + s) holds input channels chunk*32 + s*16 + [0, 16) of that tap;
// lane (l31, half) holds output channel nb*32 + l31, channels half*8 + [0, 8) of the slab;
// plane q is the q-th term of the exact round-to-nearest three-way bf16 split.
template <int MATH>
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
      unsigned short o3[3];
      split_weight<MATH>(src[e], o3);
#pragma unroll
      for (int q = 0; q < 3; ++q) out[q][e] = o3[q];
    }
    unsigned short* dst = frag + ((rest * 3) * 64 + lane) * 8;
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
      for (int e = 0; e < 8; ++e) dst[(long)q * 512 + e] = out[q][e];
  }
}

template <int BM, int BN, int WM, int WN, int DUAL, int MODE, int MATH>
int launch_p3(const IgemmParams& p, int rows_alloc, hipStream_t stream) {
  // two patch buffers (+ two B stages for the 1x1 form) + 4 counters
  const int smem_bytes = 2 * rows_alloc * Planes<MATH>::ROW +
                         (MODE == P3_GATHER ? 2 * BN * 192 : 3 * p.Cin * 4) + 16;
  constexpr int threads = (WM * WN + P3_PRODUCERS) * 64;
  auto kern = conv_p3_kernel<BM, BN, WM, WN, DUAL, MODE, MATH>;
  static int attr_bytes = 0;  // per instantiation: the largest dynamic LDS size enabled so far
  if (smem_bytes > attr_bytes) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, x3_lds_max());
    if (e != hipSuccess) {
      vlnce_set_error("conv_p3: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
      return 2;
    }
    attr_bytes = x3_lds_max();
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

// Tiles: 8 matrix waves, each a (BM / WM) x 32 sub-tile.  B fragments come straight from L2 (one
// 1 KB load per plane and k-slab, used for BM / WM / 32 MFMAs), so the sub-tile is TALL: with
// 128 rows the texture path carries 16 B/clk per CU, with 32 rows it would saturate (64 B/clk).
template <int DUAL, int MODE, int MATH>
int dispatch_p3(const IgemmParams& p, int tile, int rows, hipStream_t s) {
  switch (tile) {
    case 0: return launch_p3<128, 256, 1, 8, DUAL, MODE, MATH>(p, rows, s);
    case 1: return launch_p3<64, 256, 1, 8, DUAL, MODE, MATH>(p, rows, s);
    case 2: return launch_p3<256, 128, 2, 4, DUAL, MODE, MATH>(p, rows, s);
    case 3: return launch_p3<128, 128, 2, 4, DUAL, MODE, MATH>(p, rows, s);
    case 4: return launch_p3<256, 64, 4, 2, DUAL, MODE, MATH>(p, rows, s);
    default: return launch_p3<128, 64, 4, 2, DUAL, MODE, MATH>(p, rows, s);
  }
}

template <int MATH>
int p3_try_launch_(const IgemmParams& p, hipStream_t stream) {
  constexpr int P3_ROW = Planes<MATH>::ROW;
  // option "p3": 0 = off, 1 = every layer it covers, 2 = the KxK (patch) layers only, 3 = the 1x1
  // layers only.  Default 2: measured per layer at num_envs 64 (profiles/archive/r03_*_convbench_ab.txt),
  // the patch form is 1.26-1.51x conv_x3_kernel on every stride-1 3x3 layer of the trunks, the
  // 1x1 form (4 producer waves) is within +-10 % of it and slower on most.
  const int mode_env = vlnce_opt(VLNCE_OPT_P3);
  const int force = vlnce_opt(VLNCE_OPT_P3_TILE);  // tuning
  if (!mode_env || !conv_math() || !p.Bfrag) return -1;
  if (p.Cin % 32 != 0 || p.N % 32 != 0 || p.lda % 4 != 0 || p.splitk > 1) return -1;
  if (p.accumulate || p.c_bytes >= 0x7fffffffL || p.a_bytes >= 0x7fffffffL) return -1;
  // residual (eval-mode block ends): added in the register epilogues of conv_p3 / conv_u3; it must
  // have the output's raster
  if (p.residual && (p.ldr != p.ldc || p.stat_partial || p.bn.acc)) return -1;
  if ((long)p.N * p.K * 6 >= 0x7fffffffL) return -1;
  const bool one = p.KH == 1 && p.KW == 1 && p.pad == 0;
  const bool dense = !one;
  if (dense && p.stride != 1) return -1;
  const bool dual = p.A2 != nullptr || p.side_out != nullptr;
  if (dual && !(one && p.stride == 1)) return -1;
  // 1x1 layers wide enough for 256-column tiles: conv_u3_kernel (no producer waves), where its
  // 128-row tiles fill the CUs.  option "u3": 0 = off, 1 = default, 2 / 3 = force 64- / 128-row tiles
  // for every N >= 256 1x1 layer (tests).  Measured
  // per layer at num_envs 64 (profiles/archive/r03_b_convbench_ab_u3.txt): 1.07-1.23x conv_x3_kernel on
  // every N >= 256 layer of the RGB trunk with M >= 16384.
  // short-K wide 1x1 (the bottleneck expansions): conv_s3_kernel.  option "s3": 0 = off, 1 = default
  // (where the 64-row tiles give every CU at least four), 2 = every eligible shape (tests)
  const int s3_env = vlnce_opt(VLNCE_OPT_S3);
  if (!dense && !dual && !p.residual && s3_env && p.stride == 1 && (p.Cin == 64 || p.Cin == 128) && p.K == p.Cin &&
      p.N % 256 == 0 && p.N / 256 <= 8 && (p.stat_partial == nullptr || p.stat_rows == 32) &&
      (p.act == VLNCE_ACT_NONE || p.act == VLNCE_ACT_RELU) &&
      (s3_env == 2 || (long)ceil_div(p.M, 64) * (p.N / 256) >= 4L * x3_cus()))
  {
    return p.Cin == 64 ? launch_s3<2, MATH>(p, stream) : launch_s3<4, MATH>(p, stream);
  }
  const int u3_env = vlnce_opt(VLNCE_OPT_U3);
  const int u3_waves = vlnce_opt(VLNCE_OPT_U3_WAVES);
  if (!dense && u3_env && p.N >= 256 && p.Cin <= U3_MAX_CIN) {
    const int cus = x3_cus();
    auto eff = [&](int bm) {
      const long tiles = (long)ceil_div(p.M, bm) * ceil_div(p.N, 256);
      const long rounds = (tiles + cus - 1) / cus;
      return (double)tiles / (double)(rounds * cus);
    };
    // 128-row tiles where they fill the CUs; else 64-row tiles for the long-K layers whose 128-row
    // tiles would leave half of the CUs idle (the 1024 -> 256 block ends of layer 3 at num_envs 64:
    // 70.5 us on conv_x3_kernel, 59.9 us here; profiles/r04_q_convbench_dual_u3.txt)
    int bm = u3_env == 2 ? 64 : 128;
    if (u3_env == 1 && eff(128) < 0.8 && p.K >= 512 && eff(64) >= 0.8) bm = 64;
    if (eff(bm) >= 0.8 || u3_env >= 2) {
      const int kind = !dual ? 0 : (p.in2_scale != nullptr ? 2 : 1);
      if (u3_waves == 4)   // one wave per SIMD, 64 x 256 tiles (a wave owns 64 x 64): experiment
        return kind == 2 ? launch_u3<64, 2, 4, MATH>(p, stream)
                         : kind ? launch_u3<64, 1, 4, MATH>(p, stream) : launch_u3<64, 0, 4, MATH>(p, stream);
      if (bm == 128)
        return kind == 2 ? launch_u3<128, 2, 8, MATH>(p, stream)
                         : kind ? launch_u3<128, 1, 8, MATH>(p, stream) : launch_u3<128, 0, 8, MATH>(p, stream);
      return kind == 2 ? launch_u3<64, 2, 8, MATH>(p, stream)
                       : kind ? launch_u3<64, 1, 8, MATH>(p, stream) : launch_u3<64, 0, 8, MATH>(p, stream);
    }
  }
  if (mode_env == 2 && !dense) return -1;  // VLNCE_P3=2: only the patch (KxK) layers
  if (mode_env == 3 && dense) return -1;   // VLNCE_P3=3: only the 1x1 layers

  const P3Tile cand[6] = {{128, 256}, {64, 256}, {256, 128}, {128, 128}, {256, 64}, {128, 64}};
  const int cus = x3_cus();
  const bool forced = force >= 1 && force <= 6;
  // BN: the narrowest of 64 / 128 / 256 that covers N (256 for wider layers); 1x1 layers with
  // N <= 64 stay on conv_x3_kernel (8 producer waves: the transform per MFMA is what binds there)
  const int bn = p.N <= 64 ? 64 : p.N <= 128 ? 128 : 256;
  if (!dense && bn == 64 && !forced) return -1;
  int pick = -1, pick_rows = 0;
  double best = 0.0;
  for (int ci = 0; ci < 6; ++ci) {
    const P3Tile& c = cand[ci];
    if (forced ? ci != force - 1 : c.bn > bn) continue;  // (narrower tiles only to fill the CUs)
    const int rows = p3_rows_for(p, c.bm, dense);
    if (2L * rows * P3_ROW + (dense ? 3L * p.Cin * 4 : 2L * c.bn * 192) + 16 > x3_lds_max())
      continue;
    if (dense && rows > P3_MAX_ROWS) continue;   // (the producers' twelve row groups)
    const long tiles = (long)ceil_div(p.M, c.bm) * ceil_div(p.N, c.bn);
    const long rounds = (tiles + cus - 1) / cus;
    const double eff = (double)tiles / (double)(rounds * cus);
    if (eff > best) {
      best = eff;
      pick = ci;
      pick_rows = rows;
    }
    if (eff >= 0.8) break;
  }
  if (pick < 0 || (best < 0.4 && !forced)) return -1;
  if (dual) return dispatch_p3<1, P3_GATHER, MATH>(p, pick, pick_rows, stream);
  if (dense) return dispatch_p3<0, P3_DENSE, MATH>(p, pick, pick_rows, stream);
  return dispatch_p3<0, P3_GATHER, MATH>(p, pick, pick_rows, stream);
}

}  // namespace

int p3_try_launch(const IgemmParams& p, hipStream_t stream) {
  return p.math == MATH_F16X3 ? p3_try_launch_<MATH_F16X3>(p, stream)
                              : p3_try_launch_<MATH_BF16X6>(p, stream);
}

}  // namespace vlnce_detail

extern "C" long vlnce_conv2d_pack_bytes(const vlnce_conv_desc* d) {
  if (!d || d->Cin <= 0 || d->Cout <= 0 || d->Cin % 32 != 0 || d->Cout % 32 != 0) return 0;
  return (long)d->Cout * d->KH * d->KW * d->Cin * 6;
}

extern "C" int vlnce_conv2d_pack_weights(const float* w_ohwi, void* frag, const vlnce_conv_desc* d,
                                         int format, vlnce_stream_t stream) {
  VLNCE_CHECK_ARG(w_ohwi && frag && d, "conv2d_pack_weights: null argument");
  VLNCE_CHECK_ARG(format == MATH_BF16X6 || format == MATH_F16X3,
                  "conv2d_pack_weights: format must be 1 (three bf16 planes) or 2 (fp16 planes)");
  VLNCE_CHECK_ARG(vlnce_conv2d_pack_bytes(d) > 0,
                  "conv2d_pack_weights: needs Cin %% 32 == 0 and Cout %% 32 == 0");
  const int T = d->KH * d->KW;
  const long total = (long)(d->Cout / 32) * (T * d->Cin / 16) * 64;
  const long blocks = (total + 255) / 256;
  hipLaunchKernelGGL(format == MATH_F16X3 ? pack_weights_kernel<MATH_F16X3>
                                          : pack_weights_kernel<MATH_BF16X6>,
                     dim3((unsigned)(blocks > 8192 ? 8192 : blocks)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), w_ohwi,
                     reinterpret_cast<unsigned short*>(frag), d->Cout, T, d->Cin);
  VLNCE_CHECK_LAUNCH("conv2d_pack_weights");
  return 0;
}
