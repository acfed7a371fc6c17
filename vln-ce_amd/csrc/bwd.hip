// Backward kernels of the normalisation / pooling layers of the visual trunks (needed only
// when MODEL.*_ENCODER.trainable=True).  All HBM-bound: float4 channel vectors, per-block
// partial reductions combined with one atomic per (block, channel).
#include "common.h"

namespace {

inline int grid_for(long work, int cap = 8192) {
  long g = (work + 255) / 256;
  if (g < 1) g = 1;
  if (g > cap) g = cap;
  return (int)g;
}

// ---------------------------------------------------------------- BatchNorm backward
// y = act(x*scale[c] + shift[c] (+ residual)),  scale = gamma*rstd, xhat = (x-mean)*rstd
// g = dy * [y > 0] (ReLU) ;  dbeta = sum g ; dgamma = sum g*xhat
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(
    const float* __restrict__ dy, const float* __restrict__ y, const float* __restrict__ x,
    const float* __restrict__ mean, const float* __restrict__ rstd, long M, int C, int relu,
    int rows_per_block, float* __restrict__ dgamma, float* __restrict__ dbeta) {
  __shared__ float red[4][64][2];
  const int cl = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  const long m0 = (long)blockIdx.y * rows_per_block;
  const long m1 = m0 + rows_per_block < M ? m0 + rows_per_block : M;
  float sg = 0.f, sgx = 0.f;
  if (c < C) {
    const float mu = mean[c], rs = rstd[c];
    for (long m = m0 + sl; m < m1; m += 4) {
      const long i = m * C + c;
      float g = dy[i];
      if (relu && !(y[i] > 0.f)) g = 0.f;
      sg += g;
      sgx += g * (x[i] - mu) * rs;
    }
  }
  red[sl][cl][0] = sg;
  red[sl][cl][1] = sgx;
  __syncthreads();
  if (sl == 0 && c < C) {
    atomicAdd(dbeta + c, red[0][cl][0] + red[1][cl][0] + red[2][cl][0] + red[3][cl][0]);
    atomicAdd(dgamma + c, red[0][cl][1] + red[1][cl][1] + red[2][cl][1] + red[3][cl][1]);
  }
}

// dx = gamma*rstd * (g - dbeta/M - xhat*dgamma/M)   (batch statistics)
// dx = gamma*rstd * g                               (use_batch_stats == 0: running statistics)
// g is optionally written out (gradient of the residual input)
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(
    const float* __restrict__ dy, const float* __restrict__ y, const float* __restrict__ x,
    const float* __restrict__ mean, const float* __restrict__ rstd,
    const float* __restrict__ gamma, const float* __restrict__ dgamma,
    const float* __restrict__ dbeta, long M, int C, int relu, int use_batch_stats,
    float* __restrict__ dx, float* __restrict__ dres) {
  const long total = M * C;
  const float invM = 1.f / (float)M;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    float g = dy[i];
    if (relu && !(y[i] > 0.f)) g = 0.f;
    if (dres) dres[i] = g;
    const float rs = rstd[c];
    const float k = (gamma ? gamma[c] : 1.f) * rs;
    float v = g;
    if (use_batch_stats) v = g - dbeta[c] * invM - (x[i] - mean[c]) * rs * dgamma[c] * invM;
    dx[i] = k * v;
  }
}

// ---------------------------------------------------------------- GroupNorm backward
// per (sample, chunk of 128 pixels, channel): {sum g, sum g*xhat}
constexpr int GN_CHUNK = 128;
__global__ __launch_bounds__(256) void gn_bwd_partial_kernel(
    const float* __restrict__ dy, const float* __restrict__ y, const float* __restrict__ x,
    const float* __restrict__ mean, const float* __restrict__ rstd, int HW, int C, int groups,
    int chunks, int relu, float* __restrict__ partial) {
  const int n = blockIdx.x / chunks;
  const int ch = blockIdx.x - n * chunks;
  const int p0 = ch * GN_CHUNK;
  const int p1 = min(HW, p0 + GN_CHUNK);
  const int cpg = C / groups;
  for (int c = threadIdx.x; c < C; c += 256) {
    const int g_ = c / cpg;
    const float mu = mean[n * groups + g_], rs = rstd[n * groups + g_];
    float sg = 0.f, sgx = 0.f;
    for (int p = p0; p < p1; ++p) {
      const long i = ((long)n * HW + p) * C + c;
      float g = dy[i];
      if (relu && !(y[i] > 0.f)) g = 0.f;
      sg += g;
      sgx += g * (x[i] - mu) * rs;
    }
    float* out = partial + (((long)n * chunks + ch) * C + c) * 2;
    out[0] = sg;
    out[1] = sgx;
  }
}

// one block per sample: s1[g] = sum_c gamma_c * SG[c], s2[g] = sum_c gamma_c * SGX[c];
// dgamma[c] += SGX[c], dbeta[c] += SG[c]  (atomics across samples)
__global__ __launch_bounds__(256) void gn_bwd_finalize_kernel(
    const float* __restrict__ partial, int C, int groups, int chunks,
    const float* __restrict__ gamma, float* __restrict__ s12 /* [N,groups,2] */,
    float* __restrict__ dgamma, float* __restrict__ dbeta) {
  extern __shared__ float sh[];  // [C][2]
  const int n = blockIdx.x;
  const int cpg = C / groups;
  for (int c = threadIdx.x; c < C; c += 256) {
    float sg = 0.f, sgx = 0.f;
    for (int ch = 0; ch < chunks; ++ch) {
      const float* q = partial + (((long)n * chunks + ch) * C + c) * 2;
      sg += q[0];
      sgx += q[1];
    }
    sh[c * 2] = sg;
    sh[c * 2 + 1] = sgx;
    atomicAdd(dbeta + c, sg);
    atomicAdd(dgamma + c, sgx);
  }
  __syncthreads();
  for (int g_ = threadIdx.x; g_ < groups; g_ += 256) {
    float s1 = 0.f, s2 = 0.f;
    for (int c = g_ * cpg; c < (g_ + 1) * cpg; ++c) {
      const float ga = gamma ? gamma[c] : 1.f;
      s1 += ga * sh[c * 2];
      s2 += ga * sh[c * 2 + 1];
    }
    s12[((long)n * groups + g_) * 2] = s1;
    s12[((long)n * groups + g_) * 2 + 1] = s2;
  }
}

// dx = rstd * (g*gamma - s1/cnt - xhat*s2/cnt)
__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(
    const float* __restrict__ dy, const float* __restrict__ y, const float* __restrict__ x,
    const float* __restrict__ mean, const float* __restrict__ rstd,
    const float* __restrict__ gamma, const float* __restrict__ s12, long total, int HW, int C,
    int groups, int relu, float* __restrict__ dx, float* __restrict__ dres) {
  const int cpg = C / groups;
  const float inv = 1.f / ((float)HW * (float)cpg);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const long n = i / ((long)HW * C);
    const int g_ = c / cpg;
    float g = dy[i];
    if (relu && !(y[i] > 0.f)) g = 0.f;
    if (dres) dres[i] = g;
    const long sg = n * groups + g_;
    const float rs = rstd[sg];
    const float xh = (x[i] - mean[sg]) * rs;
    dx[i] = rs * (g * (gamma ? gamma[c] : 1.f) - s12[sg * 2] * inv - xh * s12[sg * 2 + 1] * inv);
  }
}

// ---------------------------------------------------------------- pooling backward
// max-pool 3x3/s2/p1 with the argmax tap (0..8, first maximum in scan order) saved by the forward
__global__ __launch_bounds__(256) void maxpool_argmax_kernel(const float* __restrict__ x,
                                                             float* __restrict__ y,
                                                             uint8_t* __restrict__ arg, int N,
                                                             int H, int W, int C, int Ho, int Wo) {
  const long total = (long)N * Ho * Wo * C;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    long t = i / C;
    const int wo = (int)(t % Wo);
    t /= Wo;
    const int ho = (int)(t % Ho);
    const int n = (int)(t / Ho);
    float m = -INFINITY;
    int best = -1;  // first valid tap, then every strictly larger value (at::max_pool2d order)
    for (int r = 0; r < 3; ++r) {
      const int hi = ho * 2 - 1 + r;
      if ((unsigned)hi >= (unsigned)H) continue;
      for (int q = 0; q < 3; ++q) {
        const int wi = wo * 2 - 1 + q;
        if ((unsigned)wi >= (unsigned)W) continue;
        const float v = x[(((long)n * H + hi) * W + wi) * C + c];
        if (best < 0 || v > m) {
          m = v;
          best = r * 3 + q;
        }
      }
    }
    y[i] = m;
    arg[i] = (uint8_t)best;
  }
}

__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const float* __restrict__ dy,
                                                          const uint8_t* __restrict__ arg,
                                                          float* __restrict__ dx, int N, int H,
                                                          int W, int C, int Ho, int Wo) {
  const long total = (long)N * H * W * C;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    long t = i / C;
    const int wi = (int)(t % W);
    t /= W;
    const int hi = (int)(t % H);
    const int n = (int)(t / H);
    float s = 0.f;
    // windows (ho, wo) with ho*2-1+r == hi  =>  r = hi + 1 - 2*ho in [0,2]
    for (int ho = (hi + 1 - 2 + 1) / 2; ho <= (hi + 1) / 2; ++ho) {
      if (ho < 0 || ho >= Ho) continue;
      const int r = hi + 1 - 2 * ho;
      if (r < 0 || r > 2) continue;
      for (int wo = (wi + 1 - 2 + 1) / 2; wo <= (wi + 1) / 2; ++wo) {
        if (wo < 0 || wo >= Wo) continue;
        const int q = wi + 1 - 2 * wo;
        if (q < 0 || q > 2) continue;
        const long o = (((long)n * Ho + ho) * Wo + wo) * C + c;
        if (arg[o] == r * 3 + q) s += dy[o];
      }
    }
    dx[i] = s;
  }
}

__global__ __launch_bounds__(256) void adaptive_avgpool_bwd_kernel(const float* __restrict__ dy,
                                                                   float* __restrict__ dx, int N,
                                                                   int H, int W, int C, int OH,
                                                                   int OW) {
  const long total = (long)N * H * W * C;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    long t = i / C;
    const int w = (int)(t % W);
    t /= W;
    const int h = (int)(t % H);
    const int n = (int)(t / H);
    float s = 0.f;
    for (int oh = 0; oh < OH; ++oh) {
      const int hs = (oh * H) / OH, he = ((oh + 1) * H + OH - 1) / OH;
      if (h < hs || h >= he) continue;
      for (int ow = 0; ow < OW; ++ow) {
        const int ws = (ow * W) / OW, we = ((ow + 1) * W + OW - 1) / OW;
        if (w < ws || w >= we) continue;
        s += dy[(((long)n * OH + oh) * OW + ow) * C + c] / (float)((he - hs) * (we - ws));
      }
    }
    dx[i] = s;
  }
}

}  // namespace

extern "C" int vlnce_bn_bwd(const float* dy, const float* y, const float* x, const float* mean,
                            const float* rstd, const float* gamma, long M, int C, int relu,
                            int use_batch_stats, float* dx, float* dres, float* dgamma,
                            float* dbeta, vlnce_stream_t stream) {
  VLNCE_CHECK_ARG(dy && x && mean && rstd && dx && dgamma && dbeta, "bn_bwd: null argument");
  VLNCE_CHECK_ARG(!relu || y, "bn_bwd: ReLU backward needs the forward output y");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  vlnce_zero(dgamma, 1, C, C, s);
  vlnce_zero(dbeta, 1, C, C, s);
  const int col_blocks = ceil_div(C, 64);
  long slices = (1024 + col_blocks - 1) / col_blocks;
  if (slices > (M + 63) / 64) slices = (M + 63) / 64;
  if (slices < 1) slices = 1;
  const int rows_per_block = (int)((M + slices - 1) / slices);
  hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(col_blocks, ceil_div(M, rows_per_block)), dim3(256),
                     0, s, dy, y, x, mean, rstd, M, C, relu, rows_per_block, dgamma, dbeta);
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(grid_for(M * C)), dim3(256), 0, s, dy, y, x, mean,
                     rstd, gamma, dgamma, dbeta, M, C, relu, use_batch_stats, dx, dres);
  VLNCE_CHECK_LAUNCH("bn_bwd");
  return 0;
}

extern "C" int vlnce_gn_bwd(const float* dy, const float* y, const float* x, const float* mean,
                            const float* rstd, const float* gamma, int Nimg, int HW, int C,
                            int groups, int relu, float* dx, float* dres, float* dgamma,
                            float* dbeta, float* workspace /* [N,chunks,C,2] + [N,groups,2] */,
                            vlnce_stream_t stream) {
  VLNCE_CHECK_ARG(dy && x && mean && rstd && dx && dgamma && dbeta && workspace,
                  "gn_bwd: null argument");
  VLNCE_CHECK_ARG(!relu || y, "gn_bwd: ReLU backward needs the forward output y");
  VLNCE_CHECK_ARG(groups > 0 && C % groups == 0, "gn_bwd: C %% groups != 0");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  vlnce_zero(dgamma, 1, C, C, s);
  vlnce_zero(dbeta, 1, C, C, s);
  const int chunks = ceil_div(HW, GN_CHUNK);
  float* partial = workspace;
  float* s12 = workspace + (size_t)Nimg * chunks * C * 2;
  hipLaunchKernelGGL(gn_bwd_partial_kernel, dim3(Nimg * chunks), dim3(256), 0, s, dy, y, x, mean,
                     rstd, HW, C, groups, chunks, relu, partial);
  hipLaunchKernelGGL(gn_bwd_finalize_kernel, dim3(Nimg), dim3(256), (size_t)C * 2 * sizeof(float),
                     s, partial, C, groups, chunks, gamma, s12, dgamma, dbeta);
  const long total = (long)Nimg * HW * C;
  hipLaunchKernelGGL(gn_bwd_apply_kernel, dim3(grid_for(total)), dim3(256), 0, s, dy, y, x, mean,
                     rstd, gamma, s12, total, HW, C, groups, relu, dx, dres);
  VLNCE_CHECK_LAUNCH("gn_bwd");
  return 0;
}

extern "C" size_t vlnce_gn_bwd_workspace_floats(int Nimg, int HW, int C, int groups) {
  return (size_t)Nimg * ceil_div(HW, GN_CHUNK) * C * 2 + (size_t)Nimg * groups * 2;
}

extern "C" int vlnce_maxpool3x3s2_argmax(const float* x, float* y, uint8_t* argmax, int N, int H,
                                         int W, int C, int Ho, int Wo, vlnce_stream_t stream) {
  VLNCE_CHECK_ARG(x && y && argmax, "maxpool_argmax: null argument");
  hipLaunchKernelGGL(maxpool_argmax_kernel, dim3(grid_for((long)N * Ho * Wo * C)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), x, y, argmax, N, H, W, C, Ho, Wo);
  VLNCE_CHECK_LAUNCH("maxpool_argmax");
  return 0;
}

extern "C" int vlnce_maxpool3x3s2_bwd(const float* dy, const uint8_t* argmax, float* dx, int N,
                                      int H, int W, int C, int Ho, int Wo, vlnce_stream_t stream) {
  VLNCE_CHECK_ARG(dy && argmax && dx, "maxpool_bwd: null argument");
  hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(grid_for((long)N * H * W * C)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), dy, argmax, dx, N, H, W, C, Ho, Wo);
  VLNCE_CHECK_LAUNCH("maxpool_bwd");
  return 0;
}

extern "C" int vlnce_adaptive_avgpool_bwd(const float* dy, float* dx, int N, int H, int W, int C,
                                          int OH, int OW, vlnce_stream_t stream) {
  VLNCE_CHECK_ARG(dy && dx, "adaptive_avgpool_bwd: null argument");
  hipLaunchKernelGGL(adaptive_avgpool_bwd_kernel, dim3(grid_for((long)N * H * W * C)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), dy, dx, N, H, W, C, OH, OW);
  VLNCE_CHECK_LAUNCH("adaptive_avgpool_bwd");
  return 0;
}
