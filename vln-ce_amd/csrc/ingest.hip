// Observation ingest: the step between the simulator's frames and the first convolution.
//
// The reference does it in four separate passes over fp32 frames: habitat's batch_obs casts the
// uint8 RGB frames to fp32 on the host and ships 4 bytes per value, CenterCropperPerSensor /
// ObsStack (habitat_extensions/obs_transformers.py:21-145) slice and torch.stack them, the
// waypoint net concatenates 12 panorama frames with the done-masked history frame
// (waypoint_predictors.py:330-375), and the encoders divide by 255 / average-pool
// (resnet_encoders.py:95,198-199).  Here the frames stay in their storage type (uint8 RGB: a
// quarter of the H2D and HBM bytes) and ONE kernel per encoder reads them through a frame
// descriptor -- centre-crop window, frame stack, optional extra masked frame -- and writes what
// the stem convolution consumes: the 2x2 space-to-depth regrouping with the per-channel input
// scale/shift (RGB), or the 2x2 average pool (depth).  vlnce_frames_gather is the eager form of
// the two observation transforms for callers that want the stacked / cropped tensor itself.
// All three are HBM-bound copies; one thread per output vector, grid-stride.
#include "igemm_shared.h"

using namespace vlnce_detail;

namespace {

struct FramesArg {
  const unsigned char* x;
  const unsigned char* x2;
  const unsigned char* mask2;
  int is_u8;
  int N, F, Ft, Hs, Ws, C, y0, x0, H, W;
};

__device__ __forceinline__ float load_px(const FramesArg& f, int n, int fr, int ih, int iw, int c) {
  // value of channel c of pixel (ih, iw) of the CROPPED frame fr of env n, as fp32; the extra
  // frame (fr == F) is multiplied by its not-done mask
  const unsigned char* base;
  float m = 1.f;
  long img;
  if (fr < f.F) {
    base = f.x;
    img = (long)n * f.F + fr;
  } else {
    base = f.x2;
    img = n;
    if (f.mask2) m = (float)f.mask2[n];
  }
  const long idx = ((img * f.Hs + (f.y0 + ih)) * f.Ws + (f.x0 + iw)) * f.C + c;
  const float v = f.is_u8 ? (float)base[idx] : reinterpret_cast<const float*>(base)[idx];
  return v * m;
}

// the three channels of one RGB pixel (C == 3): one index computation, three adjacent loads
__device__ __forceinline__ void load_px3(const FramesArg& f, int n, int fr, int ih, int iw,
                                         float (&v)[3]) {
  const unsigned char* base;
  float m = 1.f;
  long img;
  if (fr < f.F) {
    base = f.x;
    img = (long)n * f.F + fr;
  } else {
    base = f.x2;
    img = n;
    if (f.mask2) m = (float)f.mask2[n];
  }
  const long idx = ((img * f.Hs + (f.y0 + ih)) * f.Ws + (f.x0 + iw)) * 3;
  if (f.is_u8) {
    const unsigned char* q = base + idx;
    v[0] = (float)q[0] * m;
    v[1] = (float)q[1] * m;
    v[2] = (float)q[2] * m;
  } else {
    const float* q = reinterpret_cast<const float*>(base) + idx;
    v[0] = q[0] * m;
    v[1] = q[1] * m;
    v[2] = q[2] * m;
  }
}

// y[img, pbh, pbw, (dy*2+dx)*C + c] = frame(img)[2*(pbh-pad_lo)+dy, 2*(pbw-pad_lo)+dx, c]*scale[c]+shift[c]
__global__ __launch_bounds__(256) void frames_s2d_kernel(FramesArg f, float* __restrict__ y,
                                                         int pad_lo, int Hb, int Wb,
                                                         const float* __restrict__ scale,
                                                         const float* __restrict__ shift) {
  const int C4 = 4 * f.C, C2 = 2 * f.C;
  const long total = (long)f.N * f.Ft * Hb * Wb * C4;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int q = (int)(i % C4);
    long t = i / C4;
    const int pbw = (int)(t % Wb);
    t /= Wb;
    const int pbh = (int)(t % Hb);
    const long img = t / Hb;
    const int n = (int)(img / f.Ft), fr = (int)(img - (long)n * f.Ft);
    const int dy = q / C2, r = q - dy * C2, dx = r / f.C, c = r - dx * f.C;
    const int ih = 2 * (pbh - pad_lo) + dy, iw = 2 * (pbw - pad_lo) + dx;
    float v = 0.f;
    if (ih >= 0 && ih < f.H && iw >= 0 && iw < f.W) {
      v = load_px(f, n, fr, ih, iw, c);
      if (scale) v = v * scale[c] + shift[c];
    }
    y[i] = v;
  }
}

// y[img, ho, wo, c] = mean of the 2x2 block, at::avg_pool2d's summation order
__global__ __launch_bounds__(256) void frames_avgpool2_kernel(FramesArg f, float* __restrict__ y) {
  const int Ho = f.H / 2, Wo = f.W / 2;
  const long total = (long)f.N * f.Ft * Ho * Wo * f.C;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = (int)(i % f.C);
    long t = i / f.C;
    const int wo = (int)(t % Wo);
    t /= Wo;
    const int ho = (int)(t % Ho);
    const long img = t / Ho;
    const int n = (int)(img / f.Ft), fr = (int)(img - (long)n * f.Ft);
    float s = load_px(f, n, fr, 2 * ho, 2 * wo, c);
    s += load_px(f, n, fr, 2 * ho, 2 * wo + 1, c);
    s += load_px(f, n, fr, 2 * ho + 1, 2 * wo, c);
    s += load_px(f, n, fr, 2 * ho + 1, 2 * wo + 1, c);
    y[i] = s / 4.0f;
  }
}

// plain fp32 frames [N*Ft, H, W, C] (stems that cannot take the space-to-depth form)
__global__ __launch_bounds__(256) void frames_f32_kernel(FramesArg f, float* __restrict__ y,
                                                         const float* __restrict__ scale,
                                                         const float* __restrict__ shift) {
  const long total = (long)f.N * f.Ft * f.H * f.W * f.C;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = (int)(i % f.C);
    long t = i / f.C;
    const int iw = (int)(t % f.W);
    t /= f.W;
    const int ih = (int)(t % f.H);
    const long img = t / f.H;
    const int n = (int)(img / f.Ft), fr = (int)(img - (long)n * f.Ft);
    float v = load_px(f, n, fr, ih, iw, c);
    if (scale) v = v * scale[c] + shift[c];
    y[i] = v;
  }
}

// out[n, f, h, w, :] = srcs[f][n, y0+h, x0+w, :], elements of `eb` bytes moved as bytes
struct GatherArg {
  const unsigned char* src[16];
};
__global__ __launch_bounds__(256) void frames_gather_kernel(GatherArg g, int F, int N, int Hs,
                                                            int Ws, int rowb_src, int y0, int xb0,
                                                            int H, int rowb, int vec,
                                                            unsigned char* __restrict__ out) {
  // one thread per 4 output bytes where rows and bases are 4-byte granular (vec == 4), else per byte
  const long per_row = rowb / vec;
  const long total = (long)N * F * H * per_row;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long b = (i % per_row) * vec;
    long t = i / per_row;
    const int h = (int)(t % H);
    t /= H;
    const int fr = (int)(t % F);
    const long n = t / F;
    const unsigned char* s = g.src[fr] + ((n * Hs + (y0 + h)) * (long)rowb_src) + xb0 + b;
    unsigned char* d = out + (((n * F + fr) * H + h) * (long)rowb) + b;
    if (vec == 4) *reinterpret_cast<unsigned*>(d) = *reinterpret_cast<const unsigned*>(s);
    else *d = *s;
  }
}

// habitat's ResizeShortestEdge = F.interpolate(mode="area") = adaptive average pooling to
// (OH, OW), cast back to the sensor's dtype -- restricted to the crop window [y0, y0+H) x
// [x0, x0+W) of the resized image (the CenterCropperPerSensor that follows it in every RxR
// config).  Arithmetic of at::adaptive_avg_pool2d on the CPU: window [floor(o*in/out),
// ceil((o+1)*in/out)), row-major fp32 sum, sum / kh / kw; uint8 results truncate like
// Tensor.to(uint8).
template <typename T>
__global__ __launch_bounds__(256) void frames_resize_area_kernel(const T* __restrict__ x, int NF,
                                                                 int Hs, int Ws, int C, int OH, int OW,
                                                                 int y0, int x0, int H, int W,
                                                                 T* __restrict__ out) {
  const long total = (long)NF * H * W * C;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = (int)(i % C);
    long t = i / C;
    const int w = (int)(t % W);
    t /= W;
    const int h = (int)(t % H);
    const long img = t / H;
    const int oh = y0 + h, ow = x0 + w;
    const int h0 = (int)(((long)oh * Hs) / OH), h1 = (int)((((long)oh + 1) * Hs + OH - 1) / OH);
    const int w0 = (int)(((long)ow * Ws) / OW), w1 = (int)((((long)ow + 1) * Ws + OW - 1) / OW);
    float s = 0.f;
    for (int a = h0; a < h1; ++a) {
      const T* row = x + ((img * Hs + a) * (long)Ws + w0) * C + c;
      for (int b = 0; b < w1 - w0; ++b) s += (float)row[(long)b * C];
    }
    const float m = s / (float)(h1 - h0) / (float)(w1 - w0);
    if constexpr (sizeof(T) == 1) out[i] = (T)m;  // float -> uint8: truncation, value <= 255
    else out[i] = m;
  }
}

// ====================================================================================
// stem7_kernel: the RGB stem -- conv 7x7 / stride 2 / pad 3, 3 -> 32 | 64 channels
// (torchvision ResNet.conv1, resnet_encoders.py:131-139) -- straight from the frames, on the bf16
// matrix pipe.  It replaces frames_s2d (a pass that wrote the regrouped frames) + a 4x4
// convolution on the fp32-MFMA kernel (Cin = 12 is no multiple of 32: 335 us at num_envs 64, 22 %
// of that pipe).  Arithmetic as in conv_p3_kernel: operands split exactly into three bf16 planes
// (round to nearest), six plane products per multiply, fp32 accumulation.
//   * K: the 21 values (7 taps x 3 channels) of one filter ROW are contiguous in an input row of
//     the frame; each row is padded to 24, K' = 7 x 24 = 168 -> 11 k-slabs of 16 (the pad weights
//     are zero).  A lane's 8 consecutive k of a slab never straddle a filter row (8 | 24), so an
//     A fragment is 16 contiguous bytes of the LDS patch: four ds_read_b32 (4-byte aligned).
//   * a workgroup owns tiles of 4 x 16 output pixels: its input patch (13 x 37 pixels, /255 or the
//     ImageNet transform applied, zero outside the frame) is split once into an LDS patch of three
//     planes (rows of 288 B: the two output rows a wave reads fall into disjoint banks), two
//     buffers, one barrier per tile;
//   * a workgroup's waves = groups of 32 pixels (2 output rows x 16) x blocks of 32 output
//     channels (4 waves, Cout = 64: a 4 x 16 tile; Cout = 32: 8 x 16); a wave keeps the B fragments of its 32
//     channels for all of K in registers (132 VGPRs) for the whole launch.
// Epilogue: raw output + BatchNorm column sums (vlnce_bn_sums) or act(y * scale + shift).
constexpr int S7_TW = 16, S7_ROWB = 288, S7_KS = 11;
#ifndef S7_WAVES
#define S7_WAVES 4
#endif

struct Stem7Params {
  FramesArg f;
  const float* in_scale;   // [3] or null
  const float* in_shift;
  const void* wfrag;       // [Cout/32][11][3][64 lanes][8 bf16]
  float* y;                // [images, Ho, Wo, Cout]
  int Cout, Ho, Wo;
  const float* scale;      // epilogue (eval BatchNorm folded) or null
  const float* shift;
  int act;
  double* bn_acc;          // vlnce_bn_sums.acc or null
  long y_bytes;
};

// WAVES: waves of a workgroup = 32-pixel groups x channel blocks.  With 231 registers a CU holds 8
// waves either way; as TWO workgroups of 4 waves their phases (patch build / matrix
// instructions / output stores, which a workgroup runs one after the other: 79 + 100 + 77 us of
// the 258 us launch at num_envs 64, profiles/archive/r04_zh_*) drift apart and overlap.
template <int NB, int WAVES, int MATH>   // NB = Cout / 32 (1 or 2); MATH: Planes<> of igemm_shared.h
__global__ __launch_bounds__(WAVES * 64, 8 / WAVES) void stem7_kernel(Stem7Params p) {
#if defined(__HIP_DEVICE_COMPILE__)
  typedef Planes<MATH> PL;
  constexpr int NA = PL::NA, NP = PL::NP;
  constexpr int NTHR = WAVES * 64;
  constexpr int TH = 2 * WAVES / NB;              // output rows of a tile
  constexpr int PH = 2 * TH + 5, PW = 2 * S7_TW + 5;
  constexpr int PLANE = PH * S7_ROWB, PBUF = NA * PLANE;
  extern __shared__ __attribute__((aligned(16))) char xsm[];   // [2][PBUF]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  const int wn = NB == 2 ? (wave & 1) : 0;          // block of 32 output channels
  const int wm = NB == 2 ? (wave >> 1) : wave;      // group of 32 pixels: output rows 2 wm, 2 wm + 1
  const FramesArg& f = p.f;
  const int tiles_x = (p.Wo + S7_TW - 1) / S7_TW, tiles_y = (p.Ho + TH - 1) / TH;
  const int per_img = tiles_x * tiles_y;
  const long ntiles = (long)f.N * f.Ft * per_img;

  // resident B fragments of this wave's 32 output channels
  bf16x8 bres[S7_KS][3];
  {
    const char* wb = static_cast<const char*>(p.wfrag) + (long)wn * S7_KS * 3072 + lane * 16;
#pragma unroll
    for (int ks = 0; ks < S7_KS; ++ks)
#pragma unroll
      for (int q = 0; q < 3; ++q)
        bres[ks][q] = *reinterpret_cast<const bf16x8*>(wb + ks * 3072 + q * 1024);
  }
  float isc[3] = {1.f, 1.f, 1.f}, ish[3] = {0.f, 0.f, 0.f};
  if (p.in_scale) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      isc[c] = p.in_scale[c];
      ish[c] = p.in_shift[c];
    }
  }
  const int col = wn * 32 + l31;
  const float e_sc = (p.scale ? p.scale[col] : 1.f) * PL::POST, e_sh = p.shift ? p.shift[col] : 0.f;
  double bn_s = 0.0, bn_q = 0.0;
  const bool relu_out = p.act == VLNCE_ACT_RELU;   // (the launcher admits none / ReLU)
  const __amdgpu_buffer_rsrc_t rsrc_y = __builtin_amdgcn_make_buffer_rsrc(
      reinterpret_cast<char*>(p.y), 0, (int)p.y_bytes, 0x00020000);

  // the patch rows hold PW * 3 values + 8 pad elements that only zero weights ever meet: they
  // must be finite, so both buffers start zeroed
  for (int i = tid; i < 2 * PBUF / 4; i += NTHR) reinterpret_cast<unsigned*>(xsm)[i] = 0u;
  __syncthreads();

  // The patch of tile t + 1 is FETCHED (raw frame values into registers) in front of tile t's
  // matrix instructions and converted / written to the other LDS buffer behind them: a tile is
  // only ~2 us of MFMAs, and a build whose loads are waited for one pixel at a time (first
  // version: 250 us per launch at num_envs 64) is all latency.
  constexpr int NPX = (PH * PW + NTHR - 1) / NTHR;   // pixels of a patch per thread
  float pv[NPX][3];
  auto fetch = [&](long tile) {
    const int img = (int)(tile / per_img), rem = (int)(tile - (long)img * per_img);
    const int ty = rem / tiles_x, tx = rem - ty * tiles_x;
    const int n = img / f.Ft, fr = img - n * f.Ft;
    const int iy0 = 2 * ty * TH - 3, ix0 = 2 * tx * S7_TW - 3;
#pragma unroll
    for (int k = 0; k < NPX; ++k) {
      const int e = tid + k * NTHR;
      const int row = e / PW, px = e - row * PW;
      const int ih = iy0 + row, iw = ix0 + px;
      const bool in = e < PH * PW && ih >= 0 && ih < f.H && iw >= 0 && iw < f.W;
      float raw3[3] = {0.f, 0.f, 0.f};
      if (in) load_px3(f, n, fr, ih, iw, raw3);
#pragma unroll
      for (int c = 0; c < 3; ++c) pv[k][c] = in ? raw3[c] * isc[c] + ish[c] : 0.f;
    }
  };
  auto stash = [&](char* buf) {
#pragma unroll
    for (int k = 0; k < NPX; ++k) {
      const int e = tid + k * NTHR;
      if (e < PH * PW) {
        const int row = e / PW, px = e - row * PW;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          float v = pv[k][c];
          char* dst = buf + row * S7_ROWB + (px * 3 + c) * 2;
          if constexpr (MATH == MATH_F16X3) {
            const _Float16 h = (_Float16)v;   // round to nearest even
            *reinterpret_cast<unsigned short*>(dst) = __builtin_bit_cast(unsigned short, h);
            *reinterpret_cast<unsigned short*>(dst + PLANE) =
                __builtin_bit_cast(unsigned short, (_Float16)((v - (float)h) * 2048.f));
          } else {
#pragma unroll
            for (int q = 0; q < 3; ++q) {
              const __bf16 hb = (__bf16)v;   // round to nearest even
              *reinterpret_cast<unsigned short*>(dst + q * PLANE) = __builtin_bit_cast(unsigned short, hb);
              v -= (float)hb;
            }
          }
        }
      }
    }
  };

  long tile = blockIdx.x;
  if (tile < ntiles) {
    fetch(tile);
    stash(xsm);
  }
  __syncthreads();
  if (tile + gridDim.x < ntiles) fetch(tile + gridDim.x);
  int par = 0;
  // The 16 stores of a finished tile are issued UNDER the next tile's matrix instructions
  // (conv_s3_kernel's recipe: a second accumulator set, one or two stores behind each k-slab's six
  // MFMAs, fire-and-forget): a tile used to be [patch build | 66 MFMAs | 16 stores] back to back,
  // with the CU's two workgroups in phase.  prv = the previous tile's raw accumulators, p_* where
  // its stores go (p_rows < 0: there is no previous tile, every store takes the out-of-range offset).
  f32x16 prv;
#pragma unroll
  for (int r = 0; r < 16; ++r) prv[r] = 0.f;
  long p_base = 0;
  int p_oy0 = 0x40000000, p_cols_left = 0;
  const long row_b = (long)p.Wo * p.Cout * 4;
  const int lane_off = (4 * half * p.Cout + col) * 4;
  auto store_prev = [&](int first, int count) {
#pragma unroll
    for (int r = first; r < first + count; ++r) {
      const int row = r >> 3, cpart = (r & 3) + 8 * ((r >> 2) & 1);
      const float lin = prv[r] * e_sc + e_sh;
      const float v = relu_out ? (lin > 0.f ? lin : 0.f) : lin;
      const bool ok = p_oy0 + row < p.Ho && cpart < p_cols_left;
#ifndef S7_DBG_NOSTORE
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsrc_y,
                                            ok ? lane_off + cpart * p.Cout * 4 : BUF_OOB,
                                            (int)(p_base + row * row_b), 0);
#endif
    }
  };
  for (; tile < ntiles; tile += gridDim.x, par ^= 1) {
    const char* buf = xsm + par * PBUF;
    // the next tile's patch (fetched a tile ago) goes into the other buffer FIRST -- its LDS
    // writes and the fetch of the tile after it then run under this tile's matrix instructions
#ifndef S7_DBG_NOBUILD
    if (tile + gridDim.x < ntiles) stash(xsm + (par ^ 1) * PBUF);
    if (tile + 2L * gridDim.x < ntiles) fetch(tile + 2L * gridDim.x);
#endif
    // this lane's pixel of the wave's 32: (dy, dx) inside the tile
    const int dy = 2 * wm + (l31 >> 4), dx = l31 & 15;
    const char* abase = buf + (2 * dy) * S7_ROWB + dx * 12;
    f32x16 acc;   // (one accumulation chain: the second set of 16 registers holds the previous
                  // tile's results for the deferred stores; two waves per SIMD interleave their chains)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    // A fragments one k-slab ahead (two register sets, the loop is fully unrolled): left to the
    // compiler every slab was [6 LDS reads, s_waitcnt lgkmcnt(0), 6 MFMAs]
    auto readA = [&](u32x4 (&fa)[NA], int ks) {
      // k' = 16 ks + 8 half + [0, 8): filter row kh = k' / 24 (clamped for the all-zero tail
      // slab), offset k' % 24 inside the row's 24
      const int k0 = 16 * ks;
      const int kh0 = k0 / 24, off0 = k0 - kh0 * 24;    // for half 0
      const int kh1 = (k0 + 8) / 24, off1 = (k0 + 8) - kh1 * 24;  // for half 1
      const int o0 = (kh0 > 6 ? 6 : kh0) * S7_ROWB + off0 * 2, o1 = (kh1 > 6 ? 6 : kh1) * S7_ROWB + off1 * 2;
      const char* a = abase + (half ? o1 : o0);
#pragma unroll
      for (int q = 0; q < NA; ++q) {
#ifdef S7_DBG_NOA   // bisection build (results are garbage)
        fa[q] = u32x4{(unsigned)ks, (unsigned)q, 1u, 2u};
        (void)a;
#else
        const unsigned* w4 = reinterpret_cast<const unsigned*>(a + q * PLANE);
        fa[q] = u32x4{w4[0], w4[1], w4[2], w4[3]};
#endif
      }
    };
    u32x4 f0[NA], f1[NA];
    readA(f0, 0);
#pragma unroll
    for (int ks = 0; ks < S7_KS; ++ks) {
      if (ks + 1 < S7_KS) {
        if (ks & 1) readA(f0, ks + 1);
        else readA(f1, ks + 1);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < NP; ++q) {
        if (ks & 1)
          acc = plane_mfma<MATH>(__builtin_bit_cast(bf16x8, f1[PL::PA[q]]), bres[ks][PL::PB[q]], acc);
        else
          acc = plane_mfma<MATH>(__builtin_bit_cast(bf16x8, f0[PL::PA[q]]), bres[ks][PL::PB[q]], acc);
      }
      // the previous tile's stores: two behind each of the first five slabs, one behind the rest
      if (ks < 5) {
        store_prev(2 * ks, 2);
        __builtin_amdgcn_sched_group_barrier(0x008, NP / 2, 0);  // MFMA
        __builtin_amdgcn_sched_group_barrier(0x040, 1, 0);  // VMEM write
        __builtin_amdgcn_sched_group_barrier(0x008, NP - NP / 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x040, 1, 0);
      } else {
        store_prev(10 + (ks - 5), 1);
        __builtin_amdgcn_sched_group_barrier(0x008, NP, 0);
        __builtin_amdgcn_sched_group_barrier(0x040, 1, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    // ---- epilogue: MFMA row index pi = (r & 3) + 8 (r >> 2) + 4 half -> pixel (2 wm + pi / 16, pi % 16)
    const int img = (int)(tile / per_img), rem = (int)(tile - (long)img * per_img);
    const int ty = rem / tiles_x, tx = rem - ty * tiles_x;
    const int oy0 = ty * TH + 2 * wm, ox0 = tx * S7_TW;
    if (p.bn_acc != nullptr) {
      int nvalid = 0;
      float s = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int pi = (r & 3) + 8 * (r >> 2) + 4 * half;
        const bool ok = oy0 + (pi >> 4) < p.Ho && ox0 + (pi & 15) < p.Wo;
        s += ok ? acc[r] : 0.f;
        nvalid += ok ? 1 : 0;
      }
      s += __shfl_xor(s, 32, 64);
      nvalid += __shfl_xor(nvalid, 32, 64);
      if (nvalid > 0) {   // (wave-uniform: the validity pattern does not depend on the column)
        const float mean = s / (float)nvalid;
        float m2 = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int pi = (r & 3) + 8 * (r >> 2) + 4 * half;
          const bool ok = oy0 + (pi >> 4) < p.Ho && ox0 + (pi & 15) < p.Wo;
          const float d = acc[r] - mean;
          m2 += ok ? d * d : 0.f;
        }
        m2 += __shfl_xor(m2, 32, 64);
        bn_s += (double)s * (double)PL::POST;
        bn_q += ((double)m2 + (double)s * (double)s / (double)nvalid) * ((double)PL::POST * (double)PL::POST);
      }
    }
    // stores: pixel pi -> output row pi >> 4 = r >> 3 (wave-uniform per register), column
    // (r & 3) + 8 ((r >> 2) & 1) + 4 half: one buffer store per register with the lane part in
    // the vector offset, the tile / row part in the scalar offset and the rest an immediate --
    // issued under the NEXT tile's matrix instructions (store_prev)
    prv = acc;
    p_base = (((long)img * p.Ho + oy0) * p.Wo + ox0) * p.Cout * 4;   // < 2^31 (launcher)
    p_oy0 = oy0;
    p_cols_left = p.Wo - ox0 - 4 * half;   // columns this lane's first pixel may still use
    __syncthreads();   // the next tile's patch is complete; this tile's buffer is free again
  }
  store_prev(0, 16);   // the last tile's stores have nothing left to hide under
  if (p.bn_acc != nullptr && half == 0) {
    double* q = p.bn_acc + ((long)(blockIdx.x % VLNCE_BN_SHARDS) * p.Cout + col) * 2;
    unsafeAtomicAdd(q, bn_s);
    unsafeAtomicAdd(q + 1, bn_q);
  }
#endif
}

inline int grid_for(long work) {
  long g = (work + 255) / 256;
  if (g < 1) g = 1;
  if (g > 16384) g = 16384;
  return (int)g;
}

int fill(const vlnce_frames* d, FramesArg* f, const char* who) {
  VLNCE_CHECK_ARG(d && d->x && d->N > 0 && d->F > 0 && d->C > 0, "%s: bad frames descriptor", who);
  VLNCE_CHECK_ARG(d->dtype == VLNCE_DT_F32 || d->dtype == VLNCE_DT_U8, "%s: dtype %d", who, d->dtype);
  VLNCE_CHECK_ARG(d->H > 0 && d->W > 0 && d->y0 >= 0 && d->x0 >= 0 && d->y0 + d->H <= d->Hs &&
                      d->x0 + d->W <= d->Ws,
                  "%s: crop window (%d,%d)+(%d,%d) outside the %dx%d frame", who, d->y0, d->x0,
                  d->H, d->W, d->Hs, d->Ws);
  VLNCE_CHECK_ARG(!d->mask2 || d->x2, "%s: mask2 without x2", who);
  f->x = static_cast<const unsigned char*>(d->x);
  f->x2 = static_cast<const unsigned char*>(d->x2);
  f->mask2 = d->mask2;
  f->is_u8 = d->dtype == VLNCE_DT_U8;
  f->N = d->N;
  f->F = d->F;
  f->Ft = d->F + (d->x2 ? 1 : 0);
  f->Hs = d->Hs;
  f->Ws = d->Ws;
  f->C = d->C;
  f->y0 = d->y0;
  f->x0 = d->x0;
  f->H = d->H;
  f->W = d->W;
  return 0;
}

}  // namespace

extern "C" int vlnce_frames_s2d(const vlnce_frames* frames, float* y, int pad_lo, int pad_hi,
                                const float* scale, const float* shift, vlnce_stream_t stream) {
  FramesArg f;
  if (int rc = fill(frames, &f, "frames_s2d")) return rc;
  VLNCE_CHECK_ARG(y && (f.H % 2) == 0 && (f.W % 2) == 0 && pad_lo >= 0 && pad_hi >= 0 &&
                      (!scale == !shift),
                  "frames_s2d: bad argument");
  const int Hb = f.H / 2 + pad_lo + pad_hi, Wb = f.W / 2 + pad_lo + pad_hi;
  hipLaunchKernelGGL(frames_s2d_kernel, dim3(grid_for((long)f.N * f.Ft * Hb * Wb * 4 * f.C)),
                     dim3(256), 0, reinterpret_cast<hipStream_t>(stream), f, y, pad_lo, Hb, Wb, scale,
                     shift);
  VLNCE_CHECK_LAUNCH("frames_s2d");
  return 0;
}

extern "C" int vlnce_stem7_fwd(const vlnce_frames* frames, const float* in_scale,
                               const float* in_shift, const void* w_frag, int w_format, float* y,
                               int Cout, const vlnce_epilogue* epi, vlnce_stream_t stream) {
  Stem7Params p{};
  if (int rc = fill(frames, &p.f, "stem7_fwd")) return rc;
  VLNCE_CHECK_ARG(w_format == MATH_BF16X6 || w_format == MATH_F16X3,
                  "stem7_fwd: w_format must be 1 (three bf16 planes) or 2 (fp16 planes)");
  VLNCE_CHECK_ARG(w_frag && y && p.f.C == 3 && (Cout == 32 || Cout == 64) && (!in_scale == !in_shift),
                  "stem7_fwd: 3-channel frames, 32 or 64 output channels");
  VLNCE_CHECK_ARG(!epi || (!epi->residual && !epi->accumulate && !epi->stat_partial),
                  "stem7_fwd: epilogue = scale / shift / act or bn");
  VLNCE_CHECK_ARG(!epi || !epi->bn || (!epi->scale && !epi->shift && !epi->act && epi->bn->acc),
                  "stem7_fwd: bn excludes scale / shift / act");
  p.in_scale = in_scale;
  p.in_shift = in_shift;
  p.wfrag = w_frag;
  p.y = y;
  p.Cout = Cout;
  p.Ho = (p.f.H + 6 - 7) / 2 + 1;
  p.Wo = (p.f.W + 6 - 7) / 2 + 1;
  p.scale = epi ? epi->scale : nullptr;
  p.shift = epi ? epi->shift : nullptr;
  p.act = epi ? epi->act : 0;
  p.bn_acc = (epi && epi->bn) ? epi->bn->acc : nullptr;
  p.y_bytes = (long)p.f.N * p.f.Ft * p.Ho * p.Wo * Cout * 4;
  VLNCE_CHECK_ARG(p.y_bytes < 0x7fffffffL, "stem7_fwd: output of %ld bytes (>= 2 GiB)", p.y_bytes);
  VLNCE_CHECK_ARG(p.act == VLNCE_ACT_NONE || p.act == VLNCE_ACT_RELU, "stem7_fwd: act = none | relu");
  constexpr int WAVES = S7_WAVES;
  const int TH = 2 * WAVES / (Cout / 32);
  const long ntiles = (long)p.f.N * p.f.Ft * ceil_div(p.Ho, TH) * ceil_div(p.Wo, S7_TW);
  const long resident = (long)x3_cus() * (8 / WAVES);   // 8 waves of 231 registers per CU
  const unsigned grid = (unsigned)(ntiles < resident ? ntiles : resident);
  const int NA = w_format == MATH_F16X3 ? 2 : 3;
  const int smem = 2 * NA * (2 * TH + 5) * S7_ROWB;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (w_format == MATH_F16X3) {
    if (Cout == 64) hipLaunchKernelGGL((stem7_kernel<2, WAVES, MATH_F16X3>), dim3(grid), dim3(WAVES * 64), smem, s, p);
    else hipLaunchKernelGGL((stem7_kernel<1, WAVES, MATH_F16X3>), dim3(grid), dim3(WAVES * 64), smem, s, p);
  } else {
    if (Cout == 64) hipLaunchKernelGGL((stem7_kernel<2, WAVES, MATH_BF16X6>), dim3(grid), dim3(WAVES * 64), smem, s, p);
    else hipLaunchKernelGGL((stem7_kernel<1, WAVES, MATH_BF16X6>), dim3(grid), dim3(WAVES * 64), smem, s, p);
  }
  VLNCE_CHECK_LAUNCH("stem7_fwd");
  return 0;
}

extern "C" int vlnce_frames_avgpool2(const vlnce_frames* frames, float* y, vlnce_stream_t stream) {
  FramesArg f;
  if (int rc = fill(frames, &f, "frames_avgpool2")) return rc;
  VLNCE_CHECK_ARG(y && f.H >= 2 && f.W >= 2, "frames_avgpool2: bad argument");
  hipLaunchKernelGGL(frames_avgpool2_kernel,
                     dim3(grid_for((long)f.N * f.Ft * (f.H / 2) * (f.W / 2) * f.C)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), f, y);
  VLNCE_CHECK_LAUNCH("frames_avgpool2");
  return 0;
}

extern "C" int vlnce_frames_f32(const vlnce_frames* frames, float* y, const float* scale,
                                const float* shift, vlnce_stream_t stream) {
  FramesArg f;
  if (int rc = fill(frames, &f, "frames_f32")) return rc;
  VLNCE_CHECK_ARG(y && (!scale == !shift), "frames_f32: bad argument");
  hipLaunchKernelGGL(frames_f32_kernel, dim3(grid_for((long)f.N * f.Ft * f.H * f.W * f.C)), dim3(256),
                     0, reinterpret_cast<hipStream_t>(stream), f, y, scale, shift);
  VLNCE_CHECK_LAUNCH("frames_f32");
  return 0;
}

extern "C" int vlnce_frames_gather(const void* const* srcs, int F, int elem_bytes, int N, int Hs,
                                   int Ws, int C, int y0, int x0, int H, int W, void* out,
                                   vlnce_stream_t stream) {
  VLNCE_CHECK_ARG(srcs && out && F > 0 && F <= 16, "frames_gather: 1..16 source frames");
  VLNCE_CHECK_ARG(elem_bytes > 0 && N > 0 && C > 0 && H > 0 && W > 0 && y0 >= 0 && x0 >= 0 &&
                      y0 + H <= Hs && x0 + W <= Ws,
                  "frames_gather: bad geometry");
  GatherArg g;
  bool al = (reinterpret_cast<uintptr_t>(out) & 3) == 0;
  for (int i = 0; i < 16; ++i) {
    g.src[i] = static_cast<const unsigned char*>(srcs[i < F ? i : 0]);
    VLNCE_CHECK_ARG(g.src[i] != nullptr, "frames_gather: null source %d", i);
    al = al && (reinterpret_cast<uintptr_t>(g.src[i]) & 3) == 0;
  }
  const int px = C * elem_bytes;
  const int rowb = W * px, rowb_src = Ws * px, xb0 = x0 * px;
  const int vec = (al && ((rowb | xb0 | rowb_src) & 3) == 0) ? 4 : 1;
  hipLaunchKernelGGL(frames_gather_kernel, dim3(grid_for((long)N * F * H * (rowb / vec))), dim3(256),
                     0, reinterpret_cast<hipStream_t>(stream), g, F, N, Hs, Ws, rowb_src, y0, xb0, H,
                     rowb, vec, static_cast<unsigned char*>(out));
  VLNCE_CHECK_LAUNCH("frames_gather");
  return 0;
}

extern "C" int vlnce_frames_resize_area(const void* x, int dtype, int NF, int Hs, int Ws, int C,
                                        int OH, int OW, int y0, int x0, int H, int W, void* out,
                                        vlnce_stream_t stream) {
  VLNCE_CHECK_ARG(x && out && NF > 0 && Hs > 0 && Ws > 0 && C > 0 && OH > 0 && OW > 0,
                  "frames_resize_area: bad argument");
  VLNCE_CHECK_ARG(dtype == VLNCE_DT_F32 || dtype == VLNCE_DT_U8, "frames_resize_area: dtype %d", dtype);
  VLNCE_CHECK_ARG(H > 0 && W > 0 && y0 >= 0 && x0 >= 0 && y0 + H <= OH && x0 + W <= OW,
                  "frames_resize_area: window (%d,%d)+(%d,%d) outside the %dx%d resized frame", y0, x0,
                  H, W, OH, OW);
  const dim3 grid(grid_for((long)NF * H * W * C));
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (dtype == VLNCE_DT_U8)
    hipLaunchKernelGGL(frames_resize_area_kernel<unsigned char>, grid, dim3(256), 0, s,
                       static_cast<const unsigned char*>(x), NF, Hs, Ws, C, OH, OW, y0, x0, H, W,
                       static_cast<unsigned char*>(out));
  else
    hipLaunchKernelGGL(frames_resize_area_kernel<float>, grid, dim3(256), 0, s,
                       static_cast<const float*>(x), NF, Hs, Ws, C, OH, OW, y0, x0, H, W,
                       static_cast<float*>(out));
  VLNCE_CHECK_LAUNCH("frames_resize_area");
  return 0;
}
