// Backward kernels of the normalisation / pooling layers of the visual trunks (needed only
// when MODEL.*_ENCODER.trainable=True).  All HBM-bound.  Channel counts that are a multiple of
// four take the *4 kernels: a thread owns FOUR consecutive channels (one 16-byte request per
// tensor and row) of a column strip of up to 64 quads and walks rows, so the per-channel vectors
// are loaded once per thread; the 256/strip row lanes of a block are combined through LDS and
// one atomic per (block, channel).  Other channel counts keep the scalar kernels.
#include "common.h"
#include <cstdint>
#include <initializer_list>

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

// ---- strip indexing shared by the *4 kernels: quads per strip = 2^qwl (<= 64), lanes = 256 >> qwl
inline bool aligned16(std::initializer_list<const void*> ps) {
  uintptr_t bits = 0;
  for (const void* q : ps) bits |= reinterpret_cast<uintptr_t>(q);
  return (bits & 15) == 0;
}
inline int strip_log2(int cq) {
  int l = 0;
  while ((1 << l) < cq && l < 6) ++l;
  return l;
}

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ float4 relu_mask(float4 g, float4 y) {
  return make_float4(y.x > 0.f ? g.x : 0.f, y.y > 0.f ? g.y : 0.f, y.z > 0.f ? g.z : 0.f,
                     y.w > 0.f ? g.w : 0.f);
}

__global__ __launch_bounds__(256) void zero2_kernel(float* __restrict__ a, float* __restrict__ b,
                                                    int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) a[i] = 0.f, b[i] = 0.f;
}

// combine the row lanes of a block: acc[8] = {sum g (4 channels), sum g*xhat (4 channels)}, or
// the running maxima (MAX)
template <bool MAX = false>
__device__ __forceinline__ void strip_reduce(float (&acc)[8], float (*red)[9], int qwl, int ql,
                                             int lane_row) {
  const int lanes = 256 >> qwl;
#pragma unroll
  for (int j = 0; j < 8; ++j) red[threadIdx.x][j] = acc[j];
  __syncthreads();
  if (lane_row == 0) {
    for (int l = 1; l < lanes; ++l)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float v = red[(l << qwl) + ql][j];
        acc[j] = MAX ? fmaxf(acc[j], v) : acc[j] + v;
      }
  }
}

// 2^k with bound * 2^k in (2^13, 2^14]: the scale at which a gradient tensor whose magnitude is at
// most `bound` enters the fp16 planes (format 2) with 2^28 of normal range below its largest value
__device__ __forceinline__ float pow2_for_fp16_planes(float bound) {
  if (!(bound > 0.f) || !isfinite(bound)) return 1.f;
  int e;
  frexpf(bound, &e);  // bound = m * 2^e, m in [0.5, 1)
  int k = 14 - e;
  k = k > 100 ? 100 : (k < -100 ? -100 : k);
  return ldexpf(1.f, k);
}

template <int RELU>
__global__ __launch_bounds__(256) void bn_bwd_reduce4_kernel(
    const float* __restrict__ dy, const float* __restrict__ y, const float* __restrict__ x,
    const float* __restrict__ mean, const float* __restrict__ rstd, long M, int C, int qwl,
    int rows_per_block, float* __restrict__ partial /* [slices][4][C] */) {
  __shared__ float red[256][9];
  const int ql = threadIdx.x & ((1 << qwl) - 1), rl = threadIdx.x >> qwl, lanes = 256 >> qwl;
  const int c = ((blockIdx.x << qwl) + ql) * 4;
  const long m0 = (long)blockIdx.y * rows_per_block;
  const long m1 = m0 + rows_per_block < M ? m0 + rows_per_block : M;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  float top[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};  // max |g|, max |x - mean| (4 channels)
  if (c < C) {
    const float4 mu = ld4(mean + c);
    auto one = [&](float4 g, float4 yv, float4 xv) {
      if (RELU) g = relu_mask(g, yv);
      const float4 xc = make_float4(xv.x - mu.x, xv.y - mu.y, xv.z - mu.z, xv.w - mu.w);
      acc[0] += g.x, acc[1] += g.y, acc[2] += g.z, acc[3] += g.w;
      acc[4] += g.x * xc.x, acc[5] += g.y * xc.y, acc[6] += g.z * xc.z, acc[7] += g.w * xc.w;
      top[0] = fmaxf(top[0], fabsf(g.x)), top[1] = fmaxf(top[1], fabsf(g.y));
      top[2] = fmaxf(top[2], fabsf(g.z)), top[3] = fmaxf(top[3], fabsf(g.w));
      top[4] = fmaxf(top[4], fabsf(xc.x)), top[5] = fmaxf(top[5], fabsf(xc.y));
      top[6] = fmaxf(top[6], fabsf(xc.z)), top[7] = fmaxf(top[7], fabsf(xc.w));
    };
    long m = m0 + rl;
    const long step = (long)lanes * C;
    const float *pd = dy + m * C + c, *px = x + m * C + c, *py = RELU ? y + m * C + c : nullptr;
    for (; m + 3L * lanes < m1; m += 4L * lanes) {  // four independent rows in flight
      float4 g[4], xv[4], yv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        g[u] = ld4(pd + u * step);
        xv[u] = ld4(px + u * step);
        yv[u] = RELU ? ld4(py + u * step) : g[u];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) one(g[u], yv[u], xv[u]);
      pd += 4 * step, px += 4 * step;
      if (RELU) py += 4 * step;
    }
    for (; m < m1; m += lanes) {
      one(ld4(pd), RELU ? ld4(py) : make_float4(0, 0, 0, 0), ld4(px));
      pd += step, px += step;
      if (RELU) py += step;
    }
    const float4 rs = ld4(rstd + c);  // xhat = (x - mean) * rstd: the factor once per thread
    acc[4] *= rs.x, acc[5] *= rs.y, acc[6] *= rs.z, acc[7] *= rs.w;
  }
  strip_reduce(acc, red, qwl, ql, rl);
  __syncthreads();
  strip_reduce<true>(top, red, qwl, ql, rl);
  if (rl == 0 && c < C) {
    float* out = partial + (long)blockIdx.y * 4 * C + c;
    st4(out, make_float4(acc[0], acc[1], acc[2], acc[3]));
    st4(out + C, make_float4(acc[4], acc[5], acc[6], acc[7]));
    st4(out + 2 * C, make_float4(top[0], top[1], top[2], top[3]));
    st4(out + 3 * C, make_float4(top[4], top[5], top[6], top[7]));
  }
}

// dbeta | dgamma = column sums of partial[slices][2C].  Same-address atomics from ~2000 blocks were
// the whole cost of the small layers (~40 ns each, serial: 65 us for a 50 MB layer), hence partials
// and this pass: one wave per four columns, lane = slice, then a wave sum.
// columns [0,C) -> dbeta, [C,2C) -> dgamma (sums); [2C,4C) -> tops[2][C] (maxima of |g|, |x - mean|)
__global__ __launch_bounds__(256) void bn_bwd_finalize4_kernel(const float* __restrict__ partial,
                                                               int slices, int C,
                                                               float* __restrict__ dgamma,
                                                               float* __restrict__ dbeta,
                                                               float* __restrict__ tops) {
  const int col = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 4, lane = threadIdx.x & 63;
  if (col >= 4 * C) return;
  const bool mx = col >= 2 * C;
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int sl = lane; sl < slices; sl += 64) {
    const float4 v = ld4(partial + (long)sl * 4 * C + col);
    if (mx) a.x = fmaxf(a.x, v.x), a.y = fmaxf(a.y, v.y), a.z = fmaxf(a.z, v.z), a.w = fmaxf(a.w, v.w);
    else a.x += v.x, a.y += v.y, a.z += v.z, a.w += v.w;
  }
  if (mx) a.x = wave_max(a.x), a.y = wave_max(a.y), a.z = wave_max(a.z), a.w = wave_max(a.w);
  else a.x = wave_sum(a.x), a.y = wave_sum(a.y), a.z = wave_sum(a.z), a.w = wave_sum(a.w);
  if (lane == 0) st4(col < C ? dbeta + col : col < 2 * C ? dgamma + (col - C) : tops + (col - 2 * C), a);
}

// |dx| <= max_c |k_c| (max|g|_c + |a_c| + max|x - mean|_c |b_c|): an upper bound from the vectors
// alone, within a small factor of the true maximum (a and b are the batch-statistics corrections);
// -> pow2[0..P) = 2^k, pow2[P..2P) = 2^-k (per-channel vectors for a convolution's prologue /
// epilogue, all equal).  One workgroup (block 0 of the apply kernel), ~3 us beside the others.
__device__ void bn_bwd_pow2(const float* rstd, const float* gamma, const float* dgamma,
                            const float* dbeta, const float* tops, int C, float invM, int batch,
                            float* pow2, int P) {
  __shared__ float wmax[4];
  float m = 0.f;
  for (int c = threadIdx.x; c < C; c += 256) {
    const float rs = rstd[c], k = fabsf((gamma ? gamma[c] : 1.f) * rs);
    float b = tops[c];
    if (batch) b += fabsf(dbeta[c] * invM) + tops[C + c] * fabsf(rs * dgamma[c] * invM);
    m = fmaxf(m, k * b);
  }
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
  __syncthreads();
  m = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3])) * 1.0001f;
  const float up = pow2_for_fp16_planes(m), down = 1.f / up;
  for (int i = threadIdx.x; i < P; i += 256) pow2[i] = up, pow2[P + i] = down;
}

template <int RELU, int BATCH, int RES>
__global__ __launch_bounds__(256) void bn_bwd_apply4_kernel(
    const float* __restrict__ dy, const float* __restrict__ y, const float* __restrict__ x,
    const float* __restrict__ mean, const float* __restrict__ rstd,
    const float* __restrict__ gamma, const float* __restrict__ dgamma,
    const float* __restrict__ dbeta, long M, int C, int qwl, int rows_per_block,
    float* __restrict__ dx, float* __restrict__ dres, const float* __restrict__ tops,
    float* __restrict__ pow2, int P) {
  if (pow2 && blockIdx.x == 0 && blockIdx.y == 0)
    bn_bwd_pow2(rstd, gamma, dgamma, dbeta, tops, C, 1.f / (float)M, BATCH, pow2, P);
  const int ql = threadIdx.x & ((1 << qwl) - 1), rl = threadIdx.x >> qwl, lanes = 256 >> qwl;
  const int c = ((blockIdx.x << qwl) + ql) * 4;
  if (c >= C) return;
  const long m0 = (long)blockIdx.y * rows_per_block;
  const long m1 = m0 + rows_per_block < M ? m0 + rows_per_block : M;
  const float invM = 1.f / (float)M;
  const float4 rs = ld4(rstd + c);
  const float4 ga = gamma ? ld4(gamma + c) : make_float4(1.f, 1.f, 1.f, 1.f);
  const float4 k = make_float4(ga.x * rs.x, ga.y * rs.y, ga.z * rs.z, ga.w * rs.w);
  float4 a = make_float4(0, 0, 0, 0), b = a, mu = a;
  if (BATCH) {
    const float4 db = ld4(dbeta + c), dg = ld4(dgamma + c);
    mu = ld4(mean + c);
    a = make_float4(db.x * invM, db.y * invM, db.z * invM, db.w * invM);
    b = make_float4(rs.x * dg.x * invM, rs.y * dg.y * invM, rs.z * dg.z * invM, rs.w * dg.w * invM);
  }
  auto one = [&](float4 g, float4 yv, float4 xv, long off) {
    if (RELU) g = relu_mask(g, yv);
    if (RES) st4(dres + off, g);
    float4 v = g;
    if (BATCH)
      v = make_float4(g.x - a.x - (xv.x - mu.x) * b.x, g.y - a.y - (xv.y - mu.y) * b.y,
                      g.z - a.z - (xv.z - mu.z) * b.z, g.w - a.w - (xv.w - mu.w) * b.w);
    st4(dx + off, make_float4(k.x * v.x, k.y * v.y, k.z * v.z, k.w * v.w));
  };
  long m = m0 + rl;
  const long step = (long)lanes * C;
  long off = m * C + c;
  for (; m + 3L * lanes < m1; m += 4L * lanes, off += 4 * step) {
    float4 g[4], xv[4], yv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      g[u] = ld4(dy + off + u * step);
      xv[u] = BATCH ? ld4(x + off + u * step) : g[u];
      yv[u] = RELU ? ld4(y + off + u * step) : g[u];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) one(g[u], yv[u], xv[u], off + u * step);
  }
  for (; m < m1; m += lanes, off += step) {
    const float4 g = ld4(dy + off);
    one(g, RELU ? ld4(y + off) : g, BATCH ? ld4(x + off) : g, off);
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
    float* __restrict__ dgamma, float* __restrict__ dbeta,
    const float* __restrict__ tops_partial, const float* __restrict__ rstd, float inv,
    float* __restrict__ bound /* [N] or null */) {
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
  if (!bound) return;
  // |dx| <= |rs| (max|g| |gamma| + |s1| inv + max|x - mean| |rs s2| inv) over the sample's channels
  __syncthreads();  // (s12 of this sample: written by this block, read back below)
  __shared__ float wmax[4];
  float m = 0.f;
  for (int c = threadIdx.x; c < C; c += 256) {
    float tg = 0.f, tx = 0.f;
    for (int ch = 0; ch < chunks; ++ch) {
      const float* q = tops_partial + (((long)n * chunks + ch) * C + c) * 2;
      tg = fmaxf(tg, q[0]);
      tx = fmaxf(tx, q[1]);
    }
    const long sg = (long)n * groups + c / cpg;
    const float rs = rstd[sg];
    m = fmaxf(m, fabsf(rs) * (tg * fabsf(gamma ? gamma[c] : 1.f) + fabsf(s12[sg * 2]) * inv +
                              tx * fabsf(rs * s12[sg * 2 + 1]) * inv));
  }
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) bound[n] = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
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

// the *4 forms of the two kernels above: thread = 4 channels x one pixel lane
template <int RELU>
__global__ __launch_bounds__(256) void gn_bwd_partial4_kernel(
    const float* __restrict__ dy, const float* __restrict__ y, const float* __restrict__ x,
    const float* __restrict__ mean, const float* __restrict__ rstd, int HW, int C, int groups,
    int chunks, int qwl, float* __restrict__ partial, float* __restrict__ tops_partial) {
  __shared__ float red[256][9];
  const int ql = threadIdx.x & ((1 << qwl) - 1), pl = threadIdx.x >> qwl, lanes = 256 >> qwl;
  float top[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const int n = blockIdx.x / chunks;
  const int ch = blockIdx.x - n * chunks;
  const int p0 = ch * GN_CHUNK;
  const int p1 = min(HW, p0 + GN_CHUNK);
  const int c = ((blockIdx.y << qwl) + ql) * 4;
  const int cpg = C / groups;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (c < C) {
    float mu[4], rs[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      mu[j] = mean[n * groups + (c + j) / cpg];
      rs[j] = rstd[n * groups + (c + j) / cpg];
    }
    const long base = (long)n * HW * C + c;
    for (int p = p0 + pl; p < p1; p += lanes) {
      const long i = base + (long)p * C;
      float4 g = ld4(dy + i);
      if (RELU) g = relu_mask(g, ld4(y + i));
      const float4 xv = ld4(x + i);
      const float4 xc = make_float4(xv.x - mu[0], xv.y - mu[1], xv.z - mu[2], xv.w - mu[3]);
      acc[0] += g.x, acc[1] += g.y, acc[2] += g.z, acc[3] += g.w;
      acc[4] += g.x * xc.x, acc[5] += g.y * xc.y, acc[6] += g.z * xc.z, acc[7] += g.w * xc.w;
      top[0] = fmaxf(top[0], fabsf(g.x)), top[1] = fmaxf(top[1], fabsf(g.y));
      top[2] = fmaxf(top[2], fabsf(g.z)), top[3] = fmaxf(top[3], fabsf(g.w));
      top[4] = fmaxf(top[4], fabsf(xc.x)), top[5] = fmaxf(top[5], fabsf(xc.y));
      top[6] = fmaxf(top[6], fabsf(xc.z)), top[7] = fmaxf(top[7], fabsf(xc.w));
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[4 + j] *= rs[j];
  }
  strip_reduce(acc, red, qwl, ql, pl);
  if (tops_partial) {
    __syncthreads();
    strip_reduce<true>(top, red, qwl, ql, pl);
  }
  if (pl == 0 && c < C) {
    float* out = partial + (((long)n * chunks + ch) * C + c) * 2;
    st4(out, make_float4(acc[0], acc[4], acc[1], acc[5]));
    st4(out + 4, make_float4(acc[2], acc[6], acc[3], acc[7]));
    if (tops_partial) {
      out = tops_partial + (((long)n * chunks + ch) * C + c) * 2;
      st4(out, make_float4(top[0], top[4], top[1], top[5]));
      st4(out + 4, make_float4(top[2], top[6], top[3], top[7]));
    }
  }
}

template <int RELU, int RES>
__global__ __launch_bounds__(256) void gn_bwd_apply4_kernel(
    const float* __restrict__ dy, const float* __restrict__ y, const float* __restrict__ x,
    const float* __restrict__ mean, const float* __restrict__ rstd,
    const float* __restrict__ gamma, const float* __restrict__ s12, int HW, int C, int groups,
    int qwl, int pix_per_block, float* __restrict__ dx, float* __restrict__ dres,
    const float* __restrict__ bound, int Nimg, float* __restrict__ pow2, int P) {
  if (pow2 && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0) {
    __shared__ float wmax[4];
    float m = 0.f;
    for (int i = threadIdx.x; i < Nimg; i += 256) m = fmaxf(m, bound[i]);
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
    __syncthreads();
    m = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3])) * 1.0001f;
    const float up = pow2_for_fp16_planes(m), down = 1.f / up;
    for (int i = threadIdx.x; i < P; i += 256) pow2[i] = up, pow2[P + i] = down;
  }
  const int ql = threadIdx.x & ((1 << qwl) - 1), pl = threadIdx.x >> qwl, lanes = 256 >> qwl;
  const int c = ((blockIdx.x << qwl) + ql) * 4;
  if (c >= C) return;
  const int n = blockIdx.z;
  const int cpg = C / groups;
  const float inv = 1.f / ((float)HW * (float)cpg);
  float mu[4], rs[4], kg[4], a[4], b[4];  // dx = rs*gamma*g - rs*s1*inv - (x-mu) * rs*rs*s2*inv
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const long sg = (long)n * groups + (c + j) / cpg;
    mu[j] = mean[sg];
    rs[j] = rstd[sg];
    kg[j] = gamma ? gamma[c + j] : 1.f;
    a[j] = s12[sg * 2] * inv;
    b[j] = rs[j] * s12[sg * 2 + 1] * inv;
  }
  const int p0 = blockIdx.y * pix_per_block;
  const int p1 = min(HW, p0 + pix_per_block);
  const long base = (long)n * HW * C + c;
  auto one = [&](float4 g, float4 yv, float4 xv, long i) {
    if (RELU) g = relu_mask(g, yv);
    if (RES) st4(dres + i, g);
    st4(dx + i, make_float4(rs[0] * (g.x * kg[0] - a[0] - (xv.x - mu[0]) * b[0]),
                            rs[1] * (g.y * kg[1] - a[1] - (xv.y - mu[1]) * b[1]),
                            rs[2] * (g.z * kg[2] - a[2] - (xv.z - mu[2]) * b[2]),
                            rs[3] * (g.w * kg[3] - a[3] - (xv.w - mu[3]) * b[3])));
  };
  int p = p0 + pl;
  for (; p + 3 * lanes < p1; p += 4 * lanes) {
    float4 g[4], xv[4], yv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long i = base + (long)(p + u * lanes) * C;
      g[u] = ld4(dy + i);
      xv[u] = ld4(x + i);
      yv[u] = RELU ? ld4(y + i) : g[u];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) one(g[u], yv[u], xv[u], base + (long)(p + u * lanes) * C);
  }
  for (; p < p1; p += lanes) {
    const long i = base + (long)p * C;
    const float4 g = ld4(dy + i);
    one(g, RELU ? ld4(y + i) : g, ld4(x + i), i);
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

// four channels per thread (C % 4 == 0): 16-byte dy / dx requests, one 4-byte arg-max word
__global__ __launch_bounds__(256) void maxpool_bwd4_kernel(const float* __restrict__ dy,
                                                           const uint8_t* __restrict__ arg,
                                                           float* __restrict__ dx, int N, int H,
                                                           int W, int C, int Ho, int Wo) {
  const int cq = C / 4;
  const long total = (long)N * H * W * cq;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % cq) * 4;
    long t = i / cq;
    const int wi = (int)(t % W);
    t /= W;
    const int hi = (int)(t % H);
    const int n = (int)(t / H);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int ho = hi / 2; ho <= (hi + 1) / 2; ++ho) {
      if (ho >= Ho) continue;
      const int r = hi + 1 - 2 * ho;
      for (int wo = wi / 2; wo <= (wi + 1) / 2; ++wo) {
        if (wo >= Wo) continue;
        const unsigned tap = (unsigned)(r * 3 + (wi + 1 - 2 * wo));
        const long o = (((long)n * Ho + ho) * Wo + wo) * C + c;
        const unsigned a = *reinterpret_cast<const unsigned*>(arg + o);
        const float4 g = ld4(dy + o);
        if ((a & 255u) == tap) s.x += g.x;
        if (((a >> 8) & 255u) == tap) s.y += g.y;
        if (((a >> 16) & 255u) == tap) s.z += g.z;
        if ((a >> 24) == tap) s.w += g.w;
      }
    }
    st4(dx + (((long)n * H + hi) * W + wi) * C + c, s);
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

namespace {
struct BnBwdPlan {
  int qwl, strips, rpb, slices;
};
// ~2048 blocks (8 per CU), each lane at least 8 rows
inline BnBwdPlan bn_bwd_plan(long M, int C) {
  BnBwdPlan p;
  p.qwl = strip_log2(C / 4);
  p.strips = ceil_div(C / 4, 1 << p.qwl);
  const int lanes = 256 >> p.qwl;
  long slices = (2048 + p.strips - 1) / p.strips;
  const long most = (M + 8L * lanes - 1) / (8L * lanes);
  if (slices > most) slices = most;
  if (slices < 1) slices = 1;
  p.rpb = (int)((M + slices - 1) / slices);
  p.slices = ceil_div(M, p.rpb);
  return p;
}
}  // namespace

extern "C" size_t vlnce_bn_bwd_workspace_floats(long M, int C) {
  if (C <= 0 || C % 4 != 0 || M <= 0) return 0;
  return (size_t)bn_bwd_plan(M, C).slices * 4 * C + 2 * (size_t)C;
}

extern "C" int vlnce_bn_bwd(const float* dy, const float* y, const float* x, const float* mean,
                            const float* rstd, const float* gamma, long M, int C, int relu,
                            int use_batch_stats, float* dx, float* dres, float* dgamma,
                            float* dbeta, float* workspace, float* pow2, int P,
                            vlnce_stream_t stream) {
  VLNCE_CHECK_ARG(dy && x && mean && rstd && dx && dgamma && dbeta, "bn_bwd: null argument");
  VLNCE_CHECK_ARG(!pow2 || (P > 0 && C % 4 == 0), "bn_bwd: pow2 needs P > 0 and C %% 4 == 0");
  VLNCE_CHECK_ARG(!relu || y, "bn_bwd: ReLU backward needs the forward output y");
  VLNCE_CHECK_ARG(C % 4 != 0 || workspace, "bn_bwd: C %% 4 == 0 needs the workspace");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (C % 4 == 0 && aligned16({dy, y, x, mean, rstd, gamma, dx, dres, dgamma, dbeta, workspace})) {
    const BnBwdPlan pl = bn_bwd_plan(M, C);
    const int qwl = pl.qwl, strips = pl.strips, rpb = pl.rpb;
    float* tops = workspace + (size_t)pl.slices * 4 * C;
    const dim3 grid(strips, ceil_div(M, rpb));
#define BN_R(R) hipLaunchKernelGGL(bn_bwd_reduce4_kernel<R>, grid, dim3(256), 0, s, dy, y, x, mean, \
                                   rstd, M, C, qwl, rpb, workspace)
    if (relu) BN_R(1); else BN_R(0);
#undef BN_R
    hipLaunchKernelGGL(bn_bwd_finalize4_kernel, dim3(ceil_div(C, 4)), dim3(256), 0, s, workspace,
                       pl.slices, C, dgamma, dbeta, tops);
#define BN_A(R, B, D) hipLaunchKernelGGL((bn_bwd_apply4_kernel<R, B, D>), grid, dim3(256), 0, s, dy, y, \
                                         x, mean, rstd, gamma, dgamma, dbeta, M, C, qwl, rpb, dx, dres, \
                                         tops, pow2, P)
    const int sel = (relu ? 4 : 0) | (use_batch_stats ? 2 : 0) | (dres ? 1 : 0);
    switch (sel) {
      case 0: BN_A(0, 0, 0); break;
      case 1: BN_A(0, 0, 1); break;
      case 2: BN_A(0, 1, 0); break;
      case 3: BN_A(0, 1, 1); break;
      case 4: BN_A(1, 0, 0); break;
      case 5: BN_A(1, 0, 1); break;
      case 6: BN_A(1, 1, 0); break;
      default: BN_A(1, 1, 1); break;
    }
#undef BN_A
    VLNCE_CHECK_LAUNCH("bn_bwd");
    return 0;
  }
  VLNCE_CHECK_ARG(!pow2, "bn_bwd: pow2 needs 16-byte aligned tensors");
  hipLaunchKernelGGL(zero2_kernel, dim3(ceil_div(C, 256)), dim3(256), 0, s, dgamma, dbeta, C);
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
                            float* dbeta,
                            float* workspace /* [N,chunks,C,2] + [N,groups,2] + [N,chunks,C,2] + [N] */,
                            float* pow2, int P, vlnce_stream_t stream) {
  VLNCE_CHECK_ARG(dy && x && mean && rstd && dx && dgamma && dbeta && workspace,
                  "gn_bwd: null argument");
  VLNCE_CHECK_ARG(!relu || y, "gn_bwd: ReLU backward needs the forward output y");
  VLNCE_CHECK_ARG(groups > 0 && C % groups == 0, "gn_bwd: C %% groups != 0");
  VLNCE_CHECK_ARG(!pow2 || P > 0, "gn_bwd: pow2 needs P > 0");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(zero2_kernel, dim3(ceil_div(C, 256)), dim3(256), 0, s, dgamma, dbeta, C);
  const int chunks = ceil_div(HW, GN_CHUNK);
  float* partial = workspace;
  float* s12 = workspace + (size_t)Nimg * chunks * C * 2;
  float* tops_partial = s12 + (size_t)Nimg * groups * 2;
  float* bound = tops_partial + (size_t)Nimg * chunks * C * 2;
  const bool quads = C % 4 == 0 && Nimg <= 65535 &&
                     aligned16({dy, y, x, dx, dres, workspace});
  const int qwl = quads ? strip_log2(C / 4) : 0, strips = quads ? ceil_div(C / 4, 1 << qwl) : 0;
  if (quads) {
    const dim3 grid(Nimg * chunks, strips);
    if (relu)
      hipLaunchKernelGGL(gn_bwd_partial4_kernel<1>, grid, dim3(256), 0, s, dy, y, x, mean, rstd, HW,
                         C, groups, chunks, qwl, partial, pow2 ? tops_partial : nullptr);
    else
      hipLaunchKernelGGL(gn_bwd_partial4_kernel<0>, grid, dim3(256), 0, s, dy, y, x, mean, rstd, HW,
                         C, groups, chunks, qwl, partial, pow2 ? tops_partial : nullptr);
  } else {
    VLNCE_CHECK_ARG(!pow2, "gn_bwd: pow2 needs C %% 4 == 0 and 16-byte aligned tensors");
    hipLaunchKernelGGL(gn_bwd_partial_kernel, dim3(Nimg * chunks), dim3(256), 0, s, dy, y, x, mean,
                       rstd, HW, C, groups, chunks, relu, partial);
  }
  hipLaunchKernelGGL(gn_bwd_finalize_kernel, dim3(Nimg), dim3(256), (size_t)C * 2 * sizeof(float),
                     s, partial, C, groups, chunks, gamma, s12, dgamma, dbeta, tops_partial, rstd,
                     1.f / ((float)HW * (float)(C / groups)), pow2 ? bound : nullptr);
  const long total = (long)Nimg * HW * C;
  if (quads) {
    const int lanes = 256 >> qwl;
    long slices = (4096 + (long)strips * Nimg - 1) / ((long)strips * Nimg);
    const long most = (HW + 4L * lanes - 1) / (4L * lanes);
    if (slices > most) slices = most;
    if (slices < 1) slices = 1;
    const int ppb = (int)((HW + slices - 1) / slices);
    const dim3 grid(strips, ceil_div(HW, ppb), Nimg);
#define GN_A(R, D) hipLaunchKernelGGL((gn_bwd_apply4_kernel<R, D>), grid, dim3(256), 0, s, dy, y, x, mean, \
                                      rstd, gamma, s12, HW, C, groups, qwl, ppb, dx, dres, bound, Nimg, \
                                      pow2, P)
    if (relu) { if (dres) GN_A(1, 1); else GN_A(1, 0); }
    else      { if (dres) GN_A(0, 1); else GN_A(0, 0); }
#undef GN_A
  } else {
    hipLaunchKernelGGL(gn_bwd_apply_kernel, dim3(grid_for(total)), dim3(256), 0, s, dy, y, x, mean,
                       rstd, gamma, s12, total, HW, C, groups, relu, dx, dres);
  }
  VLNCE_CHECK_LAUNCH("gn_bwd");
  return 0;
}

extern "C" size_t vlnce_gn_bwd_workspace_floats(int Nimg, int HW, int C, int groups) {
  return (size_t)Nimg * ceil_div(HW, GN_CHUNK) * C * 4 + (size_t)Nimg * groups * 2 + (size_t)Nimg;
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
  if (C % 4 == 0 && aligned16({dy, dx}) && (reinterpret_cast<uintptr_t>(argmax) & 3) == 0)
    hipLaunchKernelGGL(maxpool_bwd4_kernel, dim3(grid_for((long)N * H * W * (C / 4), 16384)),
                       dim3(256), 0, reinterpret_cast<hipStream_t>(stream), dy, argmax, dx, N, H, W,
                       C, Ho, Wo);
  else
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
