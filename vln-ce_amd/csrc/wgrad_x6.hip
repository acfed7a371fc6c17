// wgrad_x6_kernel: the weight gradient of a convolution on the 16-bit matrix pipe (round 6).
//
//   dW[co, (r, q, ci)] = sum over output pixels m of  dY[m, co] * X[n, ho*s - p + r, wo*s - p + q, ci]
//
// (autograd of nn.Conv2d when MODEL.{RGB,DEPTH}_ENCODER.trainable is set, resnet_encoders.py:45-46,
// 141-143).  Rounds 1-5 ran it on v_mfma_f32_32x32x2_f32 (igemm_kernel<A_TRANS, B_IM2COL>, the top
// kernel of a trainable-encoder step: 11 of 41 ms).  Here both operands are split exactly into three
// bf16 planes and a product is the six plane products of conv_x3_kernel -- plane format 1: the
// operands are GRADIENTS (and activations), far below fp16's normal range (igemm_shared.h).
//
// The reduction runs over pixels, so both MFMA operands are TRANSPOSED images of what lies in
// memory: a lane's fragment is 8 consecutive PIXELS of one output channel (A, from dY[m, co]) or of
// one (tap, input channel) (B, from the im2col of X).  A thread therefore fetches the same 4
// channels of two consecutive pixels (two 16-byte loads), splits each channel's pixel PAIR into
// the three planes' 32-bit words (split_pair: the pair is what sits side by side in a fragment)
// and writes twelve ds_write_b32 into the [channel][pixel] images -- rows of 80 bytes as in
// conv_x3_kernel, so the fragment reads are that kernel's.  Lane map of a store: 4 channel groups x
// 8 pixel pairs per 32 lanes -- rows step by four (80-byte rows: 16 banks apart, alternating) and
// the pixel pairs fill the banks in between: 2-way conflicts only, which a ds_write_b32 hides
// (first version, 8 channel groups x 4 pixel pairs: 4-way, twice the LDS-array time).
//
// Workgroup: 8 waves, tile TM x 128 of dW (TM = 128: waves 2 x 4, a wave owns 64 x 32; TM = 64:
// 32 x 32), ONE LDS stage of 32 pixels (61 KB) and TWO workgroups per CU: a workgroup alternates
// [fragment reads + 24 MFMAs per wave] and [split + transposed stores of the next chunk, whose raw
// rows were requested a chunk earlier], and the CU's two workgroups drift into opposite phases --
// the split is ~180 VALU instructions per thread and chunk, as long as the chunk's MFMAs.
// blockIdx.y owns a contiguous range of chunks (split over pixels: the output is small, the
// reduction long) and adds its partial sums with fp32 atomics into a zeroed dW -- the arrangement
// of the fp32-MFMA kernel it replaces.
//
// MATH_F16X3 form (`dy_pow2` given: the power of two 2^k at which dY tops out near 2^14, from the
// normalisation backward that wrote dY -- vlnce_bn_bwd / vlnce_gn_bwd): THREE plane products per
// multiply.  X takes the two-plane side of format 2 (|x| < 65504: activations), dY * 2^(k-10) the
// three-plane side (|.| <= 16 < 32: that side's range), the accumulators leave through
// 2^-11 * 2^(10-k), all exact.  Five plane images instead of six (51 KB), 12 MFMAs per wave and
// chunk instead of 24, a shorter split.
#include "igemm_shared.h"

using namespace vlnce_detail;

namespace vlnce_detail {
namespace {

constexpr int W6_PITCH = 80;   // bytes per LDS row of one plane: 32 pixels x 2 B + 16 pad
constexpr int W6_TN = 128;

struct WgradParams {
  const float* x;
  const float* dy;
  float* dw;
  int Cout, K;                 // dW is [Cout][K], K = KH * KW * Cin ordered (r, q, ci)
  int M;                       // output pixels N * Ho * Wo
  int H, W, Cin, KW, stride, pad, Ho, Wo;
  int ldx, ldy;
  int chunks_per_slice;        // 32-pixel chunks per blockIdx.y
  int tiles_n;
  long x_bytes, dy_bytes;
  const float* dy_up;          // MATH_F16X3: device scalars 2^k, 2^-k
  const float* dy_down;
  int accumulate;              // dW += (the caller zeroed it, or holds a sum to add to)
};

// the three-plane side of format 2 for a pixel pair: {h * 2^11, (v - h) * 2^11, h}, h = fp16(v)
// (split_weight<MATH_F16X3>'s planes, two values per word)
__device__ __forceinline__ void split_pair_b_f16(float v0, float v1, unsigned (&w)[3]) {
  typedef float f32x2_ __attribute__((ext_vector_type(2)));
  typedef _Float16 f16x2_ __attribute__((ext_vector_type(2)));
  const f32x2_ v = {v0, v1};
  const f16x2_ h = __builtin_convertvector(v, f16x2_);
  const float h0 = (float)h[0], h1 = (float)h[1];
  const f32x2_ hi = {h0 * 2048.f, h1 * 2048.f}, lo = {(v0 - h0) * 2048.f, (v1 - h1) * 2048.f};
  w[0] = __builtin_bit_cast(unsigned, __builtin_convertvector(hi, f16x2_));
  w[1] = __builtin_bit_cast(unsigned, __builtin_convertvector(lo, f16x2_));
  w[2] = __builtin_bit_cast(unsigned, h);
}

typedef unsigned u32x2_ __attribute__((ext_vector_type(2)));

template <int TM, int MATH>
__global__ __launch_bounds__(512) void wgrad_x6_kernel(WgradParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
  typedef Planes<MATH> PL;
  constexpr bool F16 = MATH == MATH_F16X3;
  constexpr int NXP = F16 ? 2 : 3;                  // plane images of X (dY: three in both forms)
  constexpr int MT = TM / 64;                       // 32-row blocks per wave (waves 2 x 4)
  constexpr int A_PLANE = TM * W6_PITCH, B_PLANE = W6_TN * W6_PITCH;
  extern __shared__ __attribute__((aligned(16))) char w6_lds[];   // [STAGE]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  const int wm = wave >> 2, wn = wave & 3;
  const int tile = blockIdx.x;
  const int co0 = (tile / p.tiles_n) * TM, k0 = (tile % p.tiles_n) * W6_TN;
  const int c_first = blockIdx.y * p.chunks_per_slice;
  const int n_chunks_all = (p.M + 31) / 32;
  const int c_end = min(c_first + p.chunks_per_slice, n_chunks_all);
  if (c_first >= c_end) return;

  const __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<char*>(reinterpret_cast<const char*>(p.x)), 0, (int)p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_dy = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<char*>(reinterpret_cast<const char*>(p.dy)), 0, (int)p.dy_bytes, 0x00020000);

  // ---- this thread's item of a chunk: 4 channels (c4) of the pixel pair pp, for A and for B.
  // lane bits: c4 & 3 | pp & 7 | c4 >> 2 | pp >> 3 -- 4 lanes cover 64 contiguous bytes of a pixel
  // row, and the 32 lanes of an LDS store cover 16 banks twice (see the header)
  const int c4 = (tid & 3) | (((tid >> 5) & 7) << 2);      // 0..31: channels 4*c4 .. +4 of the tile
  const int pp = ((tid >> 2) & 7) | ((tid >> 8) << 3);     // 0..15: pixels 2*pp, 2*pp + 1 of the chunk
  const bool a_item = c4 * 4 < TM;                         // (TM = 64: half of the threads)
  const int a_co = co0 + c4 * 4;
  const bool a_co_ok = a_item && a_co < p.Cout;            // (Cout % 32 == 0: whole float4s)
  // B: k = k0 + 4*c4 .. +4 -> (tap, ci): 4 consecutive input channels of one tap (Cin % 32 == 0)
  const int kb = k0 + c4 * 4;
  const bool b_k_ok = kb < p.K;
  const int tap = kb / p.Cin, ci = kb - tap * p.Cin;
  const int tr = tap / p.KW, tq = tap - tr * p.KW;
  const int HoWo = p.Ho * p.Wo;

  struct Raw {
    f32x4 a0, a1, b0, b1;
  };
  // dY * 2^(k-10): at most 16, the three-plane side's range (exact: a power of two)
  const float dy_scale = F16 ? *p.dy_up * (1.f / 1024.f) : 1.f;
  // (image, row, column) of this thread's two output pixels, walked 32 pixels per chunk in mixed
  // radix -- two integer divisions per pixel and chunk were a third of the thread's VALU work
  int q_img[2], q_ho[2], q_wo[2];
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const int mm = c_first * 32 + pp * 2 + e;
    q_img[e] = mm / HoWo;
    const int rem = mm - q_img[e] * HoWo;
    q_ho[e] = rem / p.Wo;
    q_wo[e] = rem - q_ho[e] * p.Wo;
  }
  const int step_img = 32 / HoWo, step_ho = (32 - step_img * HoWo) / p.Wo,
            step_wo = 32 - step_img * HoWo - step_ho * p.Wo;
  int next_fetch = c_first;   // the chunk q_* stand at
  auto fetch = [&](Raw& r, int chunk) {
    // (chunks are fetched in order: c_first, c_first + 1, ...)
    if (chunk != next_fetch) __builtin_trap();
    const int m = chunk * 32 + pp * 2;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int mm = m + e;
      const bool live = chunk < c_end && mm < p.M;
      const int va = (live && a_co_ok) ? (mm * p.ldy + a_co) * 4 : BUF_OOB;
      int vb = BUF_OOB;
      if (live && b_k_ok) {
        const int hi = q_ho[e] * p.stride - p.pad + tr, wi = q_wo[e] * p.stride - p.pad + tq;
        if ((unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W)
          vb = (((q_img[e] * p.H + hi) * p.W + wi) * p.ldx + ci) * 4;
      }
      q_wo[e] += step_wo;
      if (q_wo[e] >= p.Wo) {
        q_wo[e] -= p.Wo;
        ++q_ho[e];
      }
      q_ho[e] += step_ho;
      if (q_ho[e] >= p.Ho) {
        q_ho[e] -= p.Ho;
        ++q_img[e];
      }
      q_img[e] += step_img;
      const f32x4 av = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_dy, va, 0, 0));
      const f32x4 bv = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_x, vb, 0, 0));
      if (e == 0) {
        r.a0 = av;
        r.b0 = bv;
      } else {
        r.a1 = av;
        r.b1 = bv;
      }
    }
    ++next_fetch;
  };
  // the pixel pair of each of the 4 channels -> one 32-bit word per plane at [channel][pixel pair]
  auto stash = [&](const Raw& r, char* stage) {
    if (a_item) {
      char* dst = stage + (c4 * 4) * W6_PITCH + pp * 4;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        unsigned w[3];
        if constexpr (F16) split_pair_b_f16(r.a0[e] * dy_scale, r.a1[e] * dy_scale, w);
        else split_pair<MATH_BF16X6>(r.a0[e], r.a1[e], w);
#pragma unroll
        for (int q = 0; q < 3; ++q)
          *reinterpret_cast<unsigned*>(dst + q * A_PLANE + e * W6_PITCH) = w[q];
      }
    }
    {
      char* dst = stage + 3 * A_PLANE + (c4 * 4) * W6_PITCH + pp * 4;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        unsigned w[NXP];
        split_pair<MATH>(r.b0[e], r.b1[e], w);
#pragma unroll
        for (int q = 0; q < NXP; ++q)
          *reinterpret_cast<unsigned*>(dst + q * B_PLANE + e * W6_PITCH) = w[q];
      }
    }
  };

  f32x16 acc[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  const int a_off = (wm * (MT * 32) + l31) * W6_PITCH + half * 16;
  const int b_off = 3 * A_PLANE + (wn * 32 + l31) * W6_PITCH + half * 16;

  Raw raw;
  fetch(raw, c_first);
  stash(raw, w6_lds);
  fetch(raw, c_first + 1);
  __syncthreads();
  for (int c = c_first; c < c_end; ++c) {
    const char* st = w6_lds;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8 fa[MT][3], fb[NXP];
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        if (q < NXP) fb[q] = *reinterpret_cast<const bf16x8*>(st + b_off + q * B_PLANE + s * 32);
#pragma unroll
        for (int i = 0; i < MT; ++i)
          fa[i][q] = *reinterpret_cast<const bf16x8*>(st + a_off + q * A_PLANE + i * 32 * W6_PITCH + s * 32);
      }
      // (F16: dY holds format 2's three-plane side, X its two-plane side)
#pragma unroll
      for (int q = 0; q < PL::NP; ++q)
#pragma unroll
        for (int i = 0; i < MT; ++i)
          acc[i] = plane_mfma<MATH>(fa[i][F16 ? PL::PB[q] : PL::PA[q]],
                                    fb[F16 ? PL::PA[q] : PL::PB[q]], acc[i]);
    }
    // chunk c + 1 (in registers since the previous iteration) into the stage once every wave has
    // read chunk c out of it; chunk c + 2 requested
    if (c + 1 < c_end) {
      __syncthreads();
      stash(raw, w6_lds);
      fetch(raw, c + 2);
      __syncthreads();
    }
  }

  // ---- epilogue: accumulator register r of block i = row (r & 3) + 8 (r >> 2) + 4 half, column l31
  const int col = k0 + wn * 32 + l31;
  const float post = F16 ? *p.dy_down * 0.5f : 1.f;   // 2^-11 * 2^(10 - k)
  if (col < p.K) {
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = co0 + wm * (MT * 32) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (row < p.Cout) {
          float* dst = p.dw + (long)row * p.K + col;
          const float v = F16 ? acc[i][r] * post : acc[i][r];
          if (gridDim.y > 1 || p.accumulate) unsafeAtomicAdd(dst, v);
          else *dst = v;
        }
      }
  }
#endif
}

template <int TM, int MATH>
int launch_w6(const WgradParams& p0, hipStream_t stream) {
  WgradParams p = p0;
  // 61 KB (TM = 128; 51 KB with fp16 planes): two workgroups per CU
  constexpr int smem = (3 * TM + (MATH == MATH_F16X3 ? 2 : 3) * W6_TN) * W6_PITCH;
  auto kern = wgrad_x6_kernel<TM, MATH>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e != hipSuccess) {
      vlnce_set_error("conv2d_wgrad: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
      return 2;
    }
    attr_set = true;
  }
  const int tiles_m = ceil_div(p.Cout, TM);
  p.tiles_n = ceil_div(p.K, W6_TN);
  const long tiles = (long)tiles_m * p.tiles_n;
  const int chunks = ceil_div(p.M, 32);
  // slices over the pixels: the grid a whole number of rounds of the 2 x CUs resident workgroups
  // (a trailing fifth of a round cost 20 % of the first version), at least 8 chunks per slice
  const long resident = 2L * x3_cus();
  long rounds = (tiles * 8 + resident - 1) / resident;          // aim at ~8 slices per tile ...
  if (rounds < 1) rounds = 1;
  long sk = rounds * resident / tiles;                          // ... rounded to whole rounds
  if (sk > chunks / 8) sk = chunks / 8;
  if (sk < 1) sk = 1;
  if (sk > 65535) sk = 65535;
  p.chunks_per_slice = ceil_div(chunks, sk);
  sk = ceil_div(chunks, p.chunks_per_slice);
  if (sk > 1 && !p.accumulate)
    vlnce_zero(p.dw, 1, (int)((long)p.Cout * p.K), (long)p.Cout * p.K, stream);
  hipLaunchKernelGGL(kern, dim3((unsigned)tiles, (unsigned)sk), dim3(512), smem, stream, p);
  VLNCE_CHECK_LAUNCH("conv2d_wgrad (plane kernel)");
  return 0;
}

}  // namespace

// >= 0: launched (0) or failed (> 0); -1: not a problem this kernel covers (the caller's fp32-MFMA
// kernel takes it).  Covered: Cin % 32 == 0, Cout % 32 == 0, 16-byte aligned operands, byte
// offsets that fit 31 bits, option "conv_math" != 0.
int wgrad_x6_try_launch(const float* x, const float* dy, float* dw, const vlnce_conv_desc* d,
                        const float* dy_up, const float* dy_down, int accumulate,
                        hipStream_t stream) {
  if (!conv_math()) return -1;
  if (d->Cin % 32 != 0 || d->Cout % 32 != 0) return -1;
  const int ldx = d->ldx ? d->ldx : d->Cin, ldy = d->ldy ? d->ldy : d->Cout;
  if (ldx % 4 != 0 || ldy % 4 != 0) return -1;
  const long M = (long)d->N * d->Ho * d->Wo;
  const long x_bytes = (((long)d->N * d->H * d->W - 1) * ldx + d->Cin) * 4;
  const long dy_bytes = ((M - 1) * ldy + d->Cout) * 4;
  if (x_bytes >= 0x7fffffffL || dy_bytes >= 0x7fffffffL || (long)d->Cout * d->KH * d->KW * d->Cin >= 0x7fffffffL)
    return -1;
  if (M < 256) return -1;   // (a handful of pixels: nothing to split)
  WgradParams p{};
  p.x = x;
  p.dy = dy;
  p.dw = dw;
  p.Cout = d->Cout;
  p.K = d->KH * d->KW * d->Cin;
  p.M = (int)M;
  p.H = d->H;
  p.W = d->W;
  p.Cin = d->Cin;
  p.KW = d->KW;
  p.stride = d->stride;
  p.pad = d->pad;
  p.Ho = d->Ho;
  p.Wo = d->Wo;
  p.ldx = ldx;
  p.ldy = ldy;
  p.x_bytes = x_bytes;
  p.dy_bytes = dy_bytes;
  p.dy_up = dy_up;
  p.dy_down = dy_down;
  p.accumulate = accumulate;
  if (dy_up && dy_down)
    return d->Cout >= 128 ? launch_w6<128, MATH_F16X3>(p, stream) : launch_w6<64, MATH_F16X3>(p, stream);
  return d->Cout >= 128 ? launch_w6<128, MATH_BF16X6>(p, stream) : launch_w6<64, MATH_BF16X6>(p, stream);
}

}  // namespace vlnce_detail
