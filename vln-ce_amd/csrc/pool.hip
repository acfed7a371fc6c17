// Pooling kernels, channels-last, one thread per output vector (HBM-bound).
#include "common.h"

namespace {

template <bool VEC>
__global__ __launch_bounds__(256) void maxpool3x3s2_kernel(const float* __restrict__ x,
                                                           float* __restrict__ y, int N, int H,
                                                           int W, int C, int Ho, int Wo,
                                                           const float* __restrict__ in_scale,
                                                           const float* __restrict__ in_shift,
                                                           const float* __restrict__ in_center,
                                                           int in_relu) {
  constexpr int V = VEC ? 4 : 1;
  const int Cv = C / V;
  const long total = (long)N * Ho * Wo * Cv;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long)gridDim.x * blockDim.x) {
    const int cv = (int)(i % Cv);
    long t = i / Cv;
    const int wo = (int)(t % Wo);
    t /= Wo;
    const int ho = (int)(t % Ho);
    const int n = (int)(t / Ho);
    float m[V], sc[V], sh[V], ce[V];
    for (int e = 0; e < V; ++e) {
      m[e] = -INFINITY;
      sc[e] = in_scale ? in_scale[cv * V + e] : 1.f;
      sh[e] = in_scale ? in_shift[cv * V + e] : 0.f;
      ce[e] = in_center ? in_center[cv * V + e] : 0.f;
    }
    for (int r = 0; r < 3; ++r) {
      const int hi = ho * 2 - 1 + r;
      if ((unsigned)hi >= (unsigned)H) continue;
      for (int q = 0; q < 3; ++q) {
        const int wi = wo * 2 - 1 + q;
        if ((unsigned)wi >= (unsigned)W) continue;
        const float* src = x + (((long)n * H + hi) * W + wi) * C + cv * V;
        if constexpr (VEC) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(src);
          for (int e = 0; e < 4; ++e) {
            float t = (v[e] - ce[e]) * sc[e] + sh[e];
            if (in_relu) t = fmaxf(t, 0.f);
            m[e] = fmaxf(m[e], t);
          }
        } else {
          float t = (src[0] - ce[0]) * sc[0] + sh[0];
          if (in_relu) t = fmaxf(t, 0.f);
          m[0] = fmaxf(m[0], t);
        }
      }
    }
    float* dst = y + i * V;
    for (int e = 0; e < V; ++e) dst[e] = m[e];
  }
}

__global__ __launch_bounds__(256) void avgpool2x2_kernel(const float* __restrict__ x,
                                                         float* __restrict__ y, int N, int H, int W,
                                                         int C) {
  const int Ho = H / 2, Wo = W / 2;
  const long total = (long)N * Ho * Wo * C;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    long t = i / C;
    const int wo = (int)(t % Wo);
    t /= Wo;
    const int ho = (int)(t % Ho);
    const int n = (int)(t / Ho);
    const float* p = x + (((long)n * H + ho * 2) * W + wo * 2) * C + c;
    // same summation order as at::avg_pool2d: row-major over the window
    float s = p[0];
    s += p[C];
    s += p[(long)W * C];
    s += p[(long)W * C + C];
    y[i] = s / 4.0f;
  }
}

// adaptive average pool: out cell (oh, ow) averages rows [floor(oh*H/OH), ceil((oh+1)*H/OH)) etc.
__global__ __launch_bounds__(256) void adaptive_avgpool_kernel(const float* __restrict__ x,
                                                               float* __restrict__ y, int N, int H,
                                                               int W, int C, int OH, int OW,
                                                               int ldy) {
  const long total = (long)N * OH * OW * C;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    long t = i / C;
    const int ow = (int)(t % OW);
    t /= OW;
    const int oh = (int)(t % OH);
    const int n = (int)(t / OH);
    const int hs = (oh * H) / OH, he = ((oh + 1) * H + OH - 1) / OH;
    const int ws = (ow * W) / OW, we = ((ow + 1) * W + OW - 1) / OW;
    float s = 0.f;
    for (int h = hs; h < he; ++h)
      for (int w = ws; w < we; ++w) s += x[(((long)n * H + h) * W + w) * C + c];
    y[(((long)n * OH + oh) * OW + ow) * ldy + c] = s / (float)((he - hs) * (we - ws));
  }
}

// y[b, c] = mean_p x[b, p, c]
__global__ __launch_bounds__(256) void mean_rows_kernel(const float* __restrict__ x,
                                                        float* __restrict__ y, int B, int P,
                                                        int C) {
  const long total = (long)B * C;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const long b = i / C;
    float s = 0.f;
    for (int p = 0; p < P; ++p) s += x[(b * P + p) * C + c];
    y[i] = s / (float)P;
  }
}

inline int grid_for(long work) {
  long g = (work + 255) / 256;
  if (g < 1) g = 1;
  if (g > 8192) g = 8192;
  return (int)g;
}

}  // namespace

extern "C" int vlnce_maxpool3x3s2(const float* x, float* y, int N, int H, int W, int C, int Ho,
                                  int Wo, const float* in_scale, const float* in_shift,
                                  const float* in_center, int in_relu, vlnce_stream_t stream) {
  VLNCE_CHECK_ARG(x && y, "maxpool: null argument");
  VLNCE_CHECK_ARG((in_scale == nullptr) == (in_shift == nullptr),
                  "maxpool: in_scale and in_shift come together");
  VLNCE_CHECK_ARG(Ho == (H + 2 - 3) / 2 + 1 && Wo == (W + 2 - 3) / 2 + 1, "maxpool: bad Ho/Wo");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const bool vec = (C % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) == 0;
  if (vec)
    hipLaunchKernelGGL(maxpool3x3s2_kernel<true>, dim3(grid_for((long)N * Ho * Wo * C / 4)),
                       dim3(256), 0, s, x, y, N, H, W, C, Ho, Wo, in_scale, in_shift, in_center, in_relu);
  else
    hipLaunchKernelGGL(maxpool3x3s2_kernel<false>, dim3(grid_for((long)N * Ho * Wo * C)), dim3(256),
                       0, s, x, y, N, H, W, C, Ho, Wo, in_scale, in_shift, in_center, in_relu);
  VLNCE_CHECK_LAUNCH("maxpool3x3s2");
  return 0;
}

// y[n, pb_h, pb_w, (dy*2+dx)*C + c] = x[n, 2*(pb_h-pad_lo)+dy, 2*(pb_w-pad_lo)+dx, c] * scale[c] + shift[c]
// and 0 in the pad_lo leading / pad_hi trailing border blocks.
__global__ __launch_bounds__(256) void space_to_depth2_kernel(
    const float* __restrict__ x, float* __restrict__ y, int N, int H, int W, int C, int pad_lo,
    int Hb, int Wb, const float* __restrict__ scale, const float* __restrict__ shift) {
  const int C4 = 4 * C, C2 = 2 * C;
  const long total = (long)N * Hb * Wb * C4;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int q = (int)(i % C4);
    long t = i / C4;
    const int pbw = (int)(t % Wb);
    t /= Wb;
    const int pbh = (int)(t % Hb);
    const int n = (int)(t / Hb);
    const int dy = q / C2, r = q - dy * C2, dx = r / C, c = r - dx * C;
    const int ih = 2 * (pbh - pad_lo) + dy, iw = 2 * (pbw - pad_lo) + dx;
    float v = 0.f;
    if (ih >= 0 && ih < H && iw >= 0 && iw < W) {
      v = x[(((long)n * H + ih) * W + iw) * C + c];
      if (scale) v = v * scale[c] + shift[c];
    }
    y[i] = v;
  }
}

extern "C" int vlnce_space_to_depth2(const float* x, float* y, int N, int H, int W, int C,
                                     int pad_lo, int pad_hi, const float* scale,
                                     const float* shift, vlnce_stream_t stream) {
  VLNCE_CHECK_ARG(x && y && N > 0 && C > 0 && H >= 2 && W >= 2 && (H % 2) == 0 && (W % 2) == 0 &&
                      pad_lo >= 0 && pad_hi >= 0 && (!scale == !shift),
                  "space_to_depth2: bad argument");
  const int Hb = H / 2 + pad_lo + pad_hi, Wb = W / 2 + pad_lo + pad_hi;
  hipLaunchKernelGGL(space_to_depth2_kernel, dim3(grid_for((long)N * Hb * Wb * 4 * C)), dim3(256),
                     0, reinterpret_cast<hipStream_t>(stream), x, y, N, H, W, C, pad_lo, Hb, Wb,
                     scale, shift);
  VLNCE_CHECK_LAUNCH("space_to_depth2");
  return 0;
}

extern "C" int vlnce_avgpool2x2(const float* x, float* y, int N, int H, int W, int C,
                                vlnce_stream_t stream) {
  VLNCE_CHECK_ARG(x && y && H >= 2 && W >= 2, "avgpool2x2: bad argument");
  hipLaunchKernelGGL(avgpool2x2_kernel, dim3(grid_for((long)N * (H / 2) * (W / 2) * C)), dim3(256),
                     0, reinterpret_cast<hipStream_t>(stream), x, y, N, H, W, C);
  VLNCE_CHECK_LAUNCH("avgpool2x2");
  return 0;
}

extern "C" int vlnce_adaptive_avgpool(const float* x, float* y, int N, int H, int W, int C, int OH,
                                      int OW, int ldy, vlnce_stream_t stream) {
  VLNCE_CHECK_ARG(x && y && OH > 0 && OW > 0 && ldy >= C, "adaptive_avgpool: bad argument");
  hipLaunchKernelGGL(adaptive_avgpool_kernel, dim3(grid_for((long)N * OH * OW * C)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), x, y, N, H, W, C, OH, OW, ldy);
  VLNCE_CHECK_LAUNCH("adaptive_avgpool");
  return 0;
}

extern "C" int vlnce_mean_rows(const float* x, float* y, int B, int P, int C,
                               vlnce_stream_t stream) {
  VLNCE_CHECK_ARG(x && y && B > 0 && P > 0 && C > 0, "mean_rows: bad argument");
  hipLaunchKernelGGL(mean_rows_kernel, dim3(grid_for((long)B * C)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), x, y, B, P, C);
  VLNCE_CHECK_LAUNCH("mean_rows");
  return 0;
}
