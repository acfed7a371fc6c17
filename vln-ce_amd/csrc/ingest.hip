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
#include "common.h"

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
