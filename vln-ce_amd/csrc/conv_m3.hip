// conv_m3_kernel: the bf16-plane convolution for SMALL problems (the habitat depth trunk at
// 128 x 128 -- 54 launches of 0.1 - 2.4 GFLOP at num_envs 64 --, every layer of a policy step at a
// handful of environments).
//
// conv_p3 / conv_u3 / conv_x3 are persistent, one 12-16 wave workgroup per CU walking a tile list
// through a software pipeline (LDS patch, producer / matrix hand-over): right when a launch is tens
// of tiles per CU, 10-30 us of fixed latency when it is one tile or less -- the whole reduction of
// a tile is then ONE serial K loop (a 3x3 over 128 channels at 8 x 8: 36 chunks back to back on
// 128 of the 256 CUs, 29 us for 1.2 GFLOP; profiles/archive/r04_zn_depth_trunk_convbench_*).  Here:
//   * no LDS staging and no hand-over: lane (row l31, k-half) of a wave fetches the 8 consecutive
//     channels of its A fragment row straight from global memory (32 B), applies the pending
//     normalisation, splits them into the three bf16 planes in registers -- that IS the
//     v_mfma_f32_32x32x16_bf16 A fragment; B fragments come pre-packed in fragment order
//     (vlnce_conv2d_pack_weights, conv_p3_kernel's operand) with one 1 KB load each;
//   * KSPLIT = 4 / 8: the waves of a workgroup take every 4th / 8th k-slab of the SAME
//     32 x (NT*32) output tile and add their accumulators through LDS at the end -- the reduction
//     is that much shorter in time and the tiles four times smaller (4096 x 128 outputs: 256
//     workgroups); KSPLIT = 1 (enough tiles anyway): four 32-row blocks per workgroup, no reduction;
//   * every wave keeps THREE k-slabs of loads in flight (24 x 16 B per lane): with one tile per CU
//     there is no other wave to hide the L2 / HBM latency behind (four: slower -- what a wave spends
//     per slab is its own ~50 VALU instructions of splitting next to 12 MFMAs, not waiting);
//   * the pending normalisation's vectors (scale, shift, centre per input channel) are staged in
//     LDS once: a global load between the slabs would wait for every prefetched slab in front of
//     it (vmcnt is in order);
//   * 256-512 threads, non-persistent.  Same arithmetic, operand formats, statistics and epilogue as the other bf16-plane
//     kernels (igemm_shared.h).
#include "igemm_shared.h"

using namespace vlnce_detail;

namespace vlnce_detail {
namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// 8 consecutive k of one row -> the A fragments of the planes (round-to-nearest split)
template <int MATH>
__device__ __forceinline__ void m3_split(f32x4 lo, f32x4 hi, bf16x8 (&f)[Planes<MATH>::NA]) {
  constexpr int NA = Planes<MATH>::NA;
  float x[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  u32x4 w[NA];
#pragma unroll
  for (int pr = 0; pr < 4; ++pr) {
    unsigned wp[NA];
    split_pair<MATH>(x[2 * pr], x[2 * pr + 1], wp);
#pragma unroll
    for (int q = 0; q < NA; ++q) w[q][pr] = wp[q];
  }
#pragma unroll
  for (int q = 0; q < NA; ++q) f[q] = __builtin_bit_cast(bf16x8, w[q]);
}

// NT: 32-column blocks per wave; KSPLIT: waves sharing one output block, each taking every
// KSPLIT-th k-slab; RB: 32-row blocks per workgroup (waves of different row blocks fetch the same B
// fragments at the same time: one L2 request, the others hit in L1).  RB * KSPLIT waves.
template <int NT, int KSPLIT, int RB, int DEPTH, int MATH>
__global__ __launch_bounds__(RB * KSPLIT * 64) void conv_m3_kernel(IgemmParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
  typedef Planes<MATH> PL;
  static_assert(KSPLIT == 1 || KSPLIT == 4 || KSPLIT == 8, "k split");
  static_assert(KSPLIT == 1 || KSPLIT >= NT, "wave j of a row block finishes column block j");
  constexpr int BM = RB * 32;
  constexpr int THREADS = RB * KSPLIT * 64;
  constexpr int RED = KSPLIT > 1 ? RB * KSPLIT * NT * 16 * 64 : 0;   // floats of the k-part reduction
  extern __shared__ __attribute__((aligned(16))) float m3_lds[];  // [RED] + prologue vectors [3][Cin]
  float* const red = m3_lds;
  float* const pro = m3_lds + RED;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5;
  const int l31 = lane & 31;

  // tile of this workgroup: all column tiles of a row tile on ONE XCD (workgroup v runs on XCD
  // v % 8), so the rows of A are fetched from HBM by one L2 only
  const int xcd = (int)blockIdx.x & 7, idx = (int)blockIdx.x >> 3;
  const int tm = (idx / p.tiles_n) * 8 + xcd;
  if (tm >= p.tiles_m) return;
  const int n0 = (idx - (idx / p.tiles_n) * p.tiles_n) * (NT * 32);
  const int rb = wave / KSPLIT;          // row block of this wave
  const int kpart = wave - rb * KSPLIT;  // its share of the k-slabs
  const int m0 = tm * BM + rb * 32;      // first row of this wave's block

  const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<char*>(reinterpret_cast<const char*>(p.A)), 0, (int)p.a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_b = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<char*>(reinterpret_cast<const char*>(p.Bfrag)), 0, (int)((long)p.N * p.K * 6),
      0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_c = __builtin_amdgcn_make_buffer_rsrc(
      reinterpret_cast<char*>(p.C), 0, (int)p.c_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_r = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<char*>(reinterpret_cast<const char*>(p.residual ? p.residual : p.C)), 0,
      (int)p.c_bytes, 0x00020000);

  // this lane's A row: output pixel m -> input pixel of tap (0, 0)
  const int m = m0 + l31;
  const bool row_ok = m < p.M;
  const int HoWo = p.Ho * p.Wo;
  const int img = m / HoWo;
  const int rem = m - img * HoWo;
  const int ho = rem / p.Wo;
  const int hi0 = ho * p.stride - p.pad, wi0 = (rem - ho * p.Wo) * p.stride - p.pad;
  const int img_base = img * p.H;

  const int KS = p.K / 16;               // k-slabs of 16: chunk-major / tap / 16-channel half
  const bool has_pro = p.in_scale != nullptr;
  if (has_pro) {
    for (int c = tid; c < p.Cin; c += THREADS) {
      pro[c] = p.in_scale[c];
      pro[p.Cin + c] = p.in_shift[c];
      pro[2 * p.Cin + c] = p.in_center ? p.in_center[c] : 0.f;
    }
    __syncthreads();
  }
  const float relu_floor = p.in_relu ? 0.f : -__builtin_huge_valf();
  const int bfrag0 = (n0 / 32) * KS * 3072 + lane * 16;

  // slab cursor (wave-uniform): channel chunk, tap row / column, 16-channel half
  int s_c = 0, s_r = 0, s_q = 0, s_h = 0;
  auto advance = [&]() {
    s_h ^= 1;
    if (s_h == 0 && ++s_q == p.KW) {
      s_q = 0;
      if (++s_r == p.KH) {
        s_r = 0;
        ++s_c;
      }
    }
  };
  for (int k = 0; k < kpart; ++k) advance();

  struct Slab {
    f32x4 a0, a1;        // the lane's 8 raw channels
    bf16x8 b[NT][3];
    int ch;              // first of the lane's 8 channels (prologue vectors)
    bool ok;             // the tap lies inside the image (zero padding comes AFTER the transform)
  };
  auto fetch = [&](Slab& s, int ks) {
    const bool live = ks < KS;
    const int hi = hi0 + s_r, wi = wi0 + s_q;
    s.ok = live && row_ok && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
    s.ch = s_c * 32 + s_h * 16 + half * 8;
    const int vo = s.ok ? (((img_base + hi) * p.W + wi) * p.lda + s.ch) * 4 : BUF_OOB;
    s.a0 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, vo, 0, 0));
    s.a1 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, vo, 16, 0));
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int q = 0; q < 3; ++q)
        s.b[j][q] = __builtin_bit_cast(
            bf16x8, __builtin_amdgcn_raw_buffer_load_b128(
                        rsrc_b, live ? bfrag0 + j * KS * 3072 + q * 1024 : BUF_OOB, ks * 3072, 0));
    if (live)
      for (int k = 0; k < KSPLIT; ++k) advance();
  };

  f32x16 acc[1][NT];
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[0][j][r] = 0.f;

  auto consume = [&](const Slab& s) {
    f32x4 v0 = s.a0, v1 = s.a1;
    if (has_pro) {
      const f32x4 sc0 = ldg4(pro + s.ch), sc1 = ldg4(pro + s.ch + 4);
      const f32x4 sh0 = ldg4(pro + p.Cin + s.ch), sh1 = ldg4(pro + p.Cin + s.ch + 4);
      const f32x4 c0 = ldg4(pro + 2 * p.Cin + s.ch), c1 = ldg4(pro + 2 * p.Cin + s.ch + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v0[e] = fmaxf(fmaf(v0[e] - c0[e], sc0[e], sh0[e]), relu_floor);
        v1[e] = fmaxf(fmaf(v1[e] - c1[e], sc1[e], sh1[e]), relu_floor);
      }
      if (!s.ok) {
        v0 = f32x4{0.f, 0.f, 0.f, 0.f};
        v1 = v0;
      }
    }
    bf16x8 fa[PL::NA];
    m3_split<MATH>(v0, v1, fa);
#pragma unroll
    for (int q = 0; q < PL::NP; ++q)
#pragma unroll
      for (int j = 0; j < NT; ++j)
        acc[0][j] = plane_mfma<MATH>(fa[PL::PA[q]], s.b[j][PL::PB[q]], acc[0][j]);
  };

  // this wave's slabs kpart, kpart + KSPLIT, ...: a ring of DEPTH register sets, DEPTH - 1 slabs
  // in flight behind the one in use (fully unrolled: the ring index is a compile-time constant)
  Slab ring[DEPTH];
#pragma unroll
  for (int i = 0; i < DEPTH - 1; ++i) fetch(ring[i], kpart + i * KSPLIT);
  for (int ks = kpart; ks < KS; ks += DEPTH * KSPLIT) {
#pragma unroll
    for (int i = 0; i < DEPTH; ++i) {
      if (i == 0 || ks + i * KSPLIT < KS) {
        fetch(ring[(i + DEPTH - 1) % DEPTH], ks + (i + DEPTH - 1) * KSPLIT);
        consume(ring[i]);
      }
    }
  }

  if constexpr (KSPLIT > 1) {
    // add the k-parts: wave j < NT of a row block finishes its column block j
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) red[((wave * NT + j) * 16 + r) * 64 + lane] = acc[0][j][r];
    __syncthreads();
    if (kpart >= NT) return;
    f32x16 sum[1][1];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < KSPLIT; ++w)
        v += red[(((rb * KSPLIT + w) * NT + kpart) * 16 + r) * 64 + lane];
      sum[0][0][r] = v;
    }
    if (m0 >= p.M) return;
    const int col0 = n0 + kpart * 32;
    const int col = col0 + l31;
    if (p.bn.acc != nullptr) {
      WaveBn<1> wbn;
      wave_bn_reset(wbn);
      wave_bn_tile<1, 1>(sum, wbn, p.bn.acc, col0, p.N, p.M - m0, half, l31, PL::POST);
      wave_bn_flush(wbn, p.bn.acc, p.N, half, l31);
    } else if (p.stat_partial != nullptr) {
      if (p.stat_rows == 32)
        wave_stats_block<1>(sum[0], p.stat_partial, m0 / 32, p.M - m0, col0, p.N, half, l31, PL::POST);
      else
        wave_stats_fine<1, 1>(sum, p.stat_partial, p.stat_rows, m0, p.M, col0, p.N, half, l31, PL::POST);
    }
    const float e_sc[1] = {(p.scale ? p.scale[col] : 1.f) * PL::POST};
    const float e_sh[1] = {p.shift ? p.shift[col] : 0.f};
    const int e_voff[1] = {(int)((((long)(m0 + 4 * half)) * p.ldc + col) * 4)};
    wave_epilogue<1, 1>(sum, e_sc, e_sh, e_voff, p.M - (m0 + 4 * half), p.ldc, p.act,
                        p.residual != nullptr, rsrc_c, rsrc_r, false);
  } else {
    if (m0 >= p.M) return;
    const int col0 = n0;
    if (p.bn.acc != nullptr) {
      WaveBn<NT> wbn;
      wave_bn_reset(wbn);
      wave_bn_tile<1, NT>(acc, wbn, p.bn.acc, col0, p.N, p.M - m0, half, l31, PL::POST);
      wave_bn_flush(wbn, p.bn.acc, p.N, half, l31);
    } else if (p.stat_partial != nullptr) {
      if (p.stat_rows == 32)
        wave_stats_block<NT>(acc[0], p.stat_partial, m0 / 32, p.M - m0, col0, p.N, half, l31, PL::POST);
      else
        wave_stats_fine<1, NT>(acc, p.stat_partial, p.stat_rows, m0, p.M, col0, p.N, half, l31, PL::POST);
    }
    float e_sc[NT], e_sh[NT];
    int e_voff[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int col = col0 + j * 32 + l31;
      e_sc[j] = (p.scale ? p.scale[col] : 1.f) * PL::POST;
      e_sh[j] = p.shift ? p.shift[col] : 0.f;
      e_voff[j] = (int)((((long)(m0 + 4 * half)) * p.ldc + col) * 4);
    }
    wave_epilogue<1, NT>(acc, e_sc, e_sh, e_voff, p.M - (m0 + 4 * half), p.ldc, p.act,
                         p.residual != nullptr, rsrc_c, rsrc_r, false);
  }
#endif
}

template <int NT, int KSPLIT, int RB, int MATH, int DEPTH = 3>
int launch_m3_(const IgemmParams& p, hipStream_t stream) {
  IgemmParams q = p;
  q.tiles_m = ceil_div(p.M, RB * 32);
  q.tiles_n = p.N / (NT * 32);
  q.splitk = 1;
  const long grid = (long)ceil_div(q.tiles_m, 8) * 8 * q.tiles_n;
  const int smem =
      ((KSPLIT > 1 ? RB * KSPLIT * NT * 16 * 64 : 0) + (p.in_scale ? 3 * p.Cin : 0)) * 4;
  auto kern = conv_m3_kernel<NT, KSPLIT, RB, DEPTH, MATH>;
  if (smem > 64 * 1024) {
    static bool attr_set = false;   // (per instantiation)
    if (!attr_set) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
      if (e != hipSuccess) {
        vlnce_set_error("conv_m3: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
        return 2;
      }
      attr_set = true;
    }
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(RB * KSPLIT * 64), smem, stream, q);
  VLNCE_CHECK_LAUNCH("conv_m3");
  return 0;
}
template <int NT, int KSPLIT, int RB>
int launch_m3(const IgemmParams& p, hipStream_t stream) {
  return p.math == MATH_F16X3 ? launch_m3_<NT, KSPLIT, RB, MATH_F16X3>(p, stream)
                              : launch_m3_<NT, KSPLIT, RB, MATH_BF16X6>(p, stream);
}

}  // namespace

// >= 0: launched (0) or failed (> 0); -1: not this kernel's layer.
// option "m3": 0 = off, 1 = the small launches (default rule below), 2 = every layer it covers,
// 3 = the same with two row blocks per workgroup where the reduction is split (tests)
int m3_try_launch(const IgemmParams& p, hipStream_t stream) {
  const int mode = vlnce_opt(VLNCE_OPT_M3);
  if (!mode || !conv_math() || !p.Bfrag) return -1;
  if (p.Cin % 32 != 0 || p.N % 32 != 0 || p.lda % 4 != 0 || p.splitk > 1 || p.accumulate) return -1;
  if (p.A2 != nullptr || p.side_out != nullptr || p.Cin > 2048) return -1;
  if (p.c_bytes >= 0x7fffffffL || p.a_bytes >= 0x7fffffffL || (long)p.N * p.K * 6 >= 0x7fffffffL)
    return -1;
  if (p.residual && (p.ldr != p.ldc || p.stat_partial || p.bn.acc)) return -1;
  if ((reinterpret_cast<uintptr_t>(p.A) & 15) != 0) return -1;
  if (p.stat_partial && !(p.stat_rows == 32 || p.stat_rows == 16)) return -1;
  const int KS = p.K / 16;
  // Default rule: launches whose tile list does not fill the persistent kernels' pipelines --
  // at most ~4 M outputs and 2.5 GFLOP (every layer of the depth trunk at num_envs 64, every layer
  // of a step at a few environments); not the longest reductions (K > 4608: every 32-row tile
  // re-reads its columns' whole weight set from L2, 1.8 MB per tile at K = 9216).  Measured per
  // layer in profiles/archive/r04_zp_*.
  if (mode == 1) {
    const double flop = 2.0 * p.M * (double)p.N * p.K;
    if ((long)p.M * p.N > 4L * 1024 * 1024 + 1 || flop > 2.6e9 || p.K > 4608) return -1;
  }
  // shape: 128-row workgroups (four row blocks, no reduction) where they already give every CU a
  // workgroup; else the reduction split over 4 waves per output block of a 32-row tile,
  // 32 columns wide where 64 would leave CUs without one, 8 k-parts when the reduction is long
  const int cus = x3_cus();
  const bool wide = p.N % 64 == 0;
  const long wg1 = (long)ceil_div(p.M, 128) * (p.N / (wide ? 64 : 32));
  if (wg1 >= cus || KS < 8)
    return wide ? launch_m3<2, 1, 4>(p, stream) : launch_m3<1, 1, 4>(p, stream);
  // (two row blocks per workgroup -- 64 x 64 tiles, half the B-fragment traffic from L2 -- measured
  // SLOWER where it halves the workgroup count to 128: 3x3 128 -> 128 at 8 x 8, 20.9 vs 16.0 us;
  // option "m3" = 3 selects it for the tests)
  if (mode == 3 && wide) return launch_m3<2, 4, 2>(p, stream);
  const long wg4 = (long)ceil_div(p.M, 32) * (p.N / 64);
  const bool nt2 = wide && wg4 >= cus;
  const long wgs = (long)ceil_div(p.M, 32) * (p.N / (nt2 ? 64 : 32));
  if (KS >= 128 && !nt2 && wgs <= cus) return launch_m3<1, 8, 1>(p, stream);   // (<= 56 KB of LDS)
  // (a ring of four slabs measured slower: 0.763 -> 0.780 ms over the depth trunk's layers)
  return nt2 ? launch_m3<2, 4, 1>(p, stream) : launch_m3<1, 4, 1>(p, stream);
}

}  // namespace vlnce_detail
