// Normalisation kernels (HBM-bound): BatchNorm finalize, GroupNorm statistics,
// and the shared scale/shift/residual/activation apply pass.  float4 accesses,
// grid-stride, fp64 only in the tiny finalize reductions.
#include "common.h"

namespace {

// ---------------------------------------------------------------- BatchNorm finalize
// partial: [tiles_m][C][2] = {sum, M2 about the tile mean} written by the conv epilogue.
// One launch, division-free, fp64: a workgroup owns 16 channels and spreads the tiles over
// 64 lanes each.  Pass 1 adds the tile sums -> batch mean; pass 2 adds
// M2_t + n_t * (mean_t - mean)^2 (Chan's pairwise update written for a known grand mean).
// The partials of even the largest layer (4096 tiles x 256 channels) are L2-resident.
// 1024 threads, lane = slice * BNF_CH + channel; <4, 256> for many tiles, <16, 64> for <= 256 tiles

// sum over the 256 slices of each channel: xor-shuffles across the 16 slices a wave holds,
// then 16 wave results through LDS
template <int BNF_CH, int BNF_SL>
__device__ __forceinline__ double slice_sum(double v, double (*red)[BNF_CH], int cl, int tid) {
#pragma unroll
  for (int off = BNF_CH; off < 64; off <<= 1) v += __shfl_xor(v, off, 64);
  const int wave = tid >> 6;
  if ((tid & 63) < BNF_CH) red[wave][cl] = v;
  __syncthreads();
  double r = 0.0;
#pragma unroll
  for (int w = 0; w < (BNF_CH * BNF_SL) / 64; ++w) r += red[w][cl];
  __syncthreads();
  return r;
}

template <int BNF_CH, int BNF_SL>
__global__ __launch_bounds__(BNF_CH* BNF_SL) void bn_finalize_kernel(
    const float* __restrict__ partial, int tiles_m, int tile_rows, int M, int C,
    const float* __restrict__ gamma, const float* __restrict__ beta, float eps, float momentum,
    float* running_mean, float* running_var, float* scale_out, float* shift_out, float* mean_out,
    float* rstd_out) {
  __shared__ double red[(BNF_CH * BNF_SL) / 64][BNF_CH];
  const int tid = threadIdx.x;
  const int cl = tid % BNF_CH, sl = tid / BNF_CH;
  const int c = blockIdx.x * BNF_CH + cl;
  const bool live = c < C;
  const long stride = (long)BNF_SL * C * 2;
  constexpr int HOLD = BNF_SL >= 256 ? 16 : 4;  // tiles a thread keeps in registers: one trip to memory
  const bool held = tiles_m <= HOLD * BNF_SL;
  float2 v[HOLD];
  double s = 0.0;
  if (live) {
    const float* q = partial + ((long)sl * C + c) * 2;
    if (held) {
#pragma unroll
      for (int i = 0; i < HOLD; ++i) {
        v[i] = float2{0.f, 0.f};
        if (sl + i * BNF_SL < tiles_m) v[i] = *reinterpret_cast<const float2*>(q + i * stride);
      }
#pragma unroll
      for (int i = 0; i < HOLD; ++i) s += (double)v[i].x;
    } else {
      for (int t = sl; t < tiles_m; t += BNF_SL, q += stride) s += (double)q[0];
    }
  }
  const double mean = slice_sum<BNF_CH, BNF_SL>(s, red, cl, tid) / (double)M;
  double m2 = 0.0;
  if (live) {
    const double inv_full = 1.0 / (double)tile_rows;
    auto term = [&](float2 p, int t) {
      const int nt = min(tile_rows, M - t * tile_rows);
      const double mt = (double)p.x * (nt == tile_rows ? inv_full : 1.0 / (double)nt);
      const double d = mt - mean;
      return (double)p.y + (double)nt * d * d;
    };
    if (held) {
#pragma unroll
      for (int i = 0; i < HOLD; ++i)
        if (sl + i * BNF_SL < tiles_m) m2 += term(v[i], sl + i * BNF_SL);
    } else {
      const float* q = partial + ((long)sl * C + c) * 2;
      for (int t = sl; t < tiles_m; t += BNF_SL, q += stride)
        m2 += term(*reinterpret_cast<const float2*>(q), t);
    }
  }
  m2 = slice_sum<BNF_CH, BNF_SL>(m2, red, cl, tid);
  if (sl == 0 && live) {
    const double var = m2 / (double)M;  // biased, used for normalisation
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    const float g = gamma ? gamma[c] : 1.f;
    const float b = beta ? beta[c] : 0.f;
    const float sc = g * rstd;
    scale_out[c] = sc;
    shift_out[c] = b - (float)mean * sc;
    if (mean_out) mean_out[c] = (float)mean;
    if (rstd_out) rstd_out[c] = rstd;
    if (running_mean) {
      const double unbiased = M > 1 ? m2 / (double)(M - 1) : var;
      running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
      running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
    }
  }
}

// Layers with very many statistics partials (the 7x7 stem at num_envs=64: 16384 tiles of 64
// rows) first merge groups of BN_COARSEN consecutive tiles -- Chan's update, one thread per (coarse
// tile, channel), all loads independent -- so that the finalize keeps its tiles in registers.
constexpr int BN_COARSEN = 16;
__global__ __launch_bounds__(256) void bn_coarsen_kernel(const float* __restrict__ partial,
                                                         int tiles_m, int tile_rows, int M, int C,
                                                         float* __restrict__ coarse) {
  const int ctiles = (tiles_m + BN_COARSEN - 1) / BN_COARSEN;
  const long total = (long)ctiles * C;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int ct = (int)(i / C), c = (int)(i - (long)ct * C);
    const int t0 = ct * BN_COARSEN;
    float2 v[BN_COARSEN];
#pragma unroll
    for (int k = 0; k < BN_COARSEN; ++k) {
      v[k] = float2{0.f, 0.f};
      if (t0 + k < tiles_m) v[k] = *reinterpret_cast<const float2*>(partial + ((long)(t0 + k) * C + c) * 2);
    }
    double s = 0.0;
    long rows = 0;
#pragma unroll
    for (int k = 0; k < BN_COARSEN; ++k)
      if (t0 + k < tiles_m) {
        s += (double)v[k].x;
        rows += min(tile_rows, M - (t0 + k) * tile_rows);
      }
    const double mean = s / (double)rows;
    double m2 = 0.0;
#pragma unroll
    for (int k = 0; k < BN_COARSEN; ++k)
      if (t0 + k < tiles_m) {
        const int nt = min(tile_rows, M - (t0 + k) * tile_rows);
        const double d = (double)v[k].x / (double)nt - mean;
        m2 += (double)v[k].y + (double)nt * d * d;
      }
    coarse[i * 2] = (float)s;
    coarse[i * 2 + 1] = (float)m2;
  }
}

// ---------------------------------------------------------------- scale/shift/act apply
template <bool VEC>
__global__ __launch_bounds__(256) void scale_shift_act_kernel(
    const float* __restrict__ x, const float* __restrict__ scale, const float* __restrict__ shift,
    const float* __restrict__ center, int rows_per_sample, const float* __restrict__ residual,
    float* __restrict__ y, long M, int C, int act) {
  if constexpr (VEC) {
    const int C4 = C >> 2;
    const long total = M * C4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long)gridDim.x * blockDim.x) {
      const long row = i / C4;
      const int c = (int)(i - row * C4) * 4;
      const long so = rows_per_sample > 0 ? (row / rows_per_sample) * C + c : c;
      f32x4 v = *reinterpret_cast<const f32x4*>(x + i * 4);
      const f32x4 sc = *reinterpret_cast<const f32x4*>(scale + so);
      const f32x4 sh = *reinterpret_cast<const f32x4*>(shift + so);
      if (center) v -= *reinterpret_cast<const f32x4*>(center + so);
      v = v * sc + sh;
      if (residual) v += *reinterpret_cast<const f32x4*>(residual + i * 4);
      v.x = apply_act(v.x, act);
      v.y = apply_act(v.y, act);
      v.z = apply_act(v.z, act);
      v.w = apply_act(v.w, act);
      *reinterpret_cast<f32x4*>(y + i * 4) = v;
    }
  } else {
    const long total = M * C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long)gridDim.x * blockDim.x) {
      const long row = i / C;
      const int c = (int)(i - row * C);
      const long so = rows_per_sample > 0 ? (row / rows_per_sample) * C + c : c;
      float v = (center ? x[i] - center[so] : x[i]) * scale[so] + shift[so];
      if (residual) v += residual[i];
      y[i] = apply_act(v, act);
    }
  }
}

// y = act(x1*s1[c]+t1[c] + x2*s2[c]+t2[c]): end of a residual block whose skip path is a
// (conv + BatchNorm) downsample -- both raw conv outputs are normalised in one pass
__global__ __launch_bounds__(256) void scale_shift_add_act_kernel(
    const float* __restrict__ x1, const float* __restrict__ s1, const float* __restrict__ t1,
    const float* __restrict__ c1, const float* __restrict__ x2, const float* __restrict__ s2,
    const float* __restrict__ t2, const float* __restrict__ c2, float* __restrict__ y, long M,
    int C, int act) {
  const int C4 = C >> 2;
  const long total = M * C4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C4) * 4;
    f32x4 a = *reinterpret_cast<const f32x4*>(x1 + i * 4);
    f32x4 b = *reinterpret_cast<const f32x4*>(x2 + i * 4);
    if (c1) a -= *reinterpret_cast<const f32x4*>(c1 + c);
    if (c2) b -= *reinterpret_cast<const f32x4*>(c2 + c);
    f32x4 v = a * *reinterpret_cast<const f32x4*>(s1 + c) + *reinterpret_cast<const f32x4*>(t1 + c);
    v += b * *reinterpret_cast<const f32x4*>(s2 + c) + *reinterpret_cast<const f32x4*>(t2 + c);
    v.x = apply_act(v.x, act);
    v.y = apply_act(v.y, act);
    v.z = apply_act(v.z, act);
    v.w = apply_act(v.w, act);
    *reinterpret_cast<f32x4*>(y + i * 4) = v;
  }
}

// ---------------------------------------------------------------- GroupNorm statistics
constexpr int GN_CHUNK = 128;  // pixels per block

// x: [N, HW, C]; partial: [N, chunks, C, 2] = {sum, sumsq} over the chunk's pixels
__global__ __launch_bounds__(256) void gn_partial_kernel(const float* __restrict__ x, int HW, int C,
                                                         int chunks, float* __restrict__ partial) {
  __shared__ float red[256 * 8];
  const int n = blockIdx.x / chunks;
  const int ch = blockIdx.x - n * chunks;
  const int p0 = ch * GN_CHUNK;
  const int p1 = min(HW, p0 + GN_CHUNK);
  const int C4 = C >> 2;
  const int tid = threadIdx.x;
  const float* base = x + (long)n * HW * C;
  float* out = partial + ((long)n * chunks + ch) * C * 2;
  if (C4 >= 256) {
    for (int c4 = tid; c4 < C4; c4 += 256) {
      f32x4 s = {0, 0, 0, 0}, q = {0, 0, 0, 0};
      for (int p = p0; p < p1; ++p) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(base + (long)p * C + c4 * 4);
        s += v;
        q += v * v;
      }
      for (int e = 0; e < 4; ++e) {
        out[(c4 * 4 + e) * 2 + 0] = s[e];
        out[(c4 * 4 + e) * 2 + 1] = q[e];
      }
    }
    return;
  }
  const int PL = 256 / C4;  // pixel lanes
  const int pl = tid / C4;
  const int c4 = tid - pl * C4;
  f32x4 s = {0, 0, 0, 0}, q = {0, 0, 0, 0};
  if (pl < PL)
    for (int p = p0 + pl; p < p1; p += PL) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(base + (long)p * C + c4 * 4);
      s += v;
      q += v * v;
    }
  for (int e = 0; e < 4; ++e) {
    red[tid * 8 + e] = s[e];
    red[tid * 8 + 4 + e] = q[e];
  }
  __syncthreads();
  if (tid < C4) {
    for (int l = 1; l < PL; ++l)
      for (int e = 0; e < 4; ++e) {
        s[e] += red[(l * C4 + tid) * 8 + e];
        q[e] += red[(l * C4 + tid) * 8 + 4 + e];
      }
    for (int e = 0; e < 4; ++e) {
      out[(tid * 4 + e) * 2 + 0] = s[e];
      out[(tid * 4 + e) * 2 + 1] = q[e];
    }
  }
}

// one wave per (sample, group)
// tile_rows == 0: partial holds {sum, sum of squares} per (sample, pixel chunk, channel)
// (gn_partial_kernel); tile_rows > 0: it holds the convolution epilogue's {sum, M2 about the tile
// mean} per (M-tile of tile_rows pixels, channel), `chunks` tiles per sample.
__global__ __launch_bounds__(256) void gn_finalize_kernel(
    const float* __restrict__ partial, int Nimg, int HW, int C, int groups, int chunks,
    int tile_rows,
    const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
    float* __restrict__ scale_out, float* __restrict__ shift_out, float* center_out,
    float* mean_out, float* rstd_out) {
  const int lane = threadIdx.x & 63;
  const int pair = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (pair >= Nimg * groups) return;
  const int n = pair / groups;
  const int g = pair - n * groups;
  const int cpg = C / groups;
  const int items = chunks * cpg;
  double s = 0.0, q = 0.0;
  for (int i = lane; i < items; i += 64) {
    const int ch = i / cpg;
    const int c = g * cpg + (i - ch * cpg);
    const float* src = partial + (((long)n * chunks + ch) * C + c) * 2;
    const double si = (double)src[0];
    s += si;
    q += tile_rows > 0 ? (double)src[1] + si * si / (double)tile_rows : (double)src[1];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    s += __shfl_xor(s, o, 64);
    q += __shfl_xor(q, o, 64);
  }
  const double cnt = (double)HW * (double)cpg;
  const double mean = s / cnt;
  double var = q / cnt - mean * mean;
  if (var < 0.0) var = 0.0;
  const float rstd = (float)(1.0 / sqrt(var + (double)eps));
  for (int i = lane; i < cpg; i += 64) {
    const int c = g * cpg + i;
    const float sc = (gamma ? gamma[c] : 1.f) * rstd;
    scale_out[(long)n * C + c] = sc;
    if (center_out) {  // y = (x - center) * scale + shift: the reference's own arithmetic
      center_out[(long)n * C + c] = (float)mean;
      shift_out[(long)n * C + c] = beta ? beta[c] : 0.f;
    } else {
      shift_out[(long)n * C + c] = (beta ? beta[c] : 0.f) - (float)mean * sc;
    }
  }
  if (lane == 0) {
    if (mean_out) mean_out[pair] = (float)mean;
    if (rstd_out) rstd_out[pair] = rstd;
  }
}

}  // namespace

extern "C" size_t vlnce_bn_finalize_workspace_bytes(int tiles_m, int C) {
  // more tiles than the finalize holds in registers: room for the coarsened partials
  if (tiles_m <= 4096) return 0;
  return (size_t)ceil_div(tiles_m, BN_COARSEN) * C * 2 * sizeof(float);
}

extern "C" int vlnce_bn_finalize(const float* stat_partial, int tiles_m, int tile_rows, int M,
                                 int C, const float* gamma, const float* beta, float eps,
                                 float momentum, float* running_mean, float* running_var,
                                 float* scale_out, float* shift_out, float* mean_out,
                                 float* rstd_out, void* workspace, size_t workspace_bytes,
                                 vlnce_stream_t stream) {
  VLNCE_CHECK_ARG(stat_partial && scale_out && shift_out, "bn_finalize: null argument");
  VLNCE_CHECK_ARG(tiles_m > 0 && tile_rows > 0 && M > 0 && C > 0, "bn_finalize: bad shape");
  VLNCE_CHECK_ARG((long)(tiles_m - 1) * tile_rows < M && (long)tiles_m * tile_rows >= M,
                  "bn_finalize: %d tiles of %d rows do not cover M=%d", tiles_m, tile_rows, M);
  VLNCE_CHECK_ARG((running_mean == nullptr) == (running_var == nullptr),
                  "bn_finalize: running stats must come together");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const size_t need = vlnce_bn_finalize_workspace_bytes(tiles_m, C);
  if (need > 0 && workspace != nullptr && workspace_bytes >= need) {
    float* coarse = static_cast<float*>(workspace);
    const int ctiles = ceil_div(tiles_m, BN_COARSEN);
    long g = ((long)ctiles * C + 255) / 256;
    hipLaunchKernelGGL(bn_coarsen_kernel, dim3((unsigned)(g > 4096 ? 4096 : g)), dim3(256), 0, s,
                       stat_partial, tiles_m, tile_rows, M, C, coarse);
    VLNCE_CHECK_LAUNCH("bn_coarsen");
    stat_partial = coarse;
    tiles_m = ctiles;
    tile_rows *= BN_COARSEN;
  }
  if (tiles_m > 256)
    hipLaunchKernelGGL((bn_finalize_kernel<4, 256>), dim3(ceil_div(C, 4)), dim3(1024), 0, s,
                       stat_partial, tiles_m, tile_rows, M, C, gamma, beta, eps, momentum,
                       running_mean, running_var, scale_out, shift_out, mean_out, rstd_out);
  else
    hipLaunchKernelGGL((bn_finalize_kernel<16, 64>), dim3(ceil_div(C, 16)), dim3(1024), 0, s,
                       stat_partial, tiles_m, tile_rows, M, C, gamma, beta, eps, momentum,
                       running_mean, running_var, scale_out, shift_out, mean_out, rstd_out);
  VLNCE_CHECK_LAUNCH("bn_finalize");
  return 0;
}

extern "C" int vlnce_scale_shift_act(const float* x, const float* scale, const float* shift,
                                     const float* center, int rows_per_sample,
                                     const float* residual, float* y, long M, int C, int act,
                                     vlnce_stream_t stream) {
  VLNCE_CHECK_ARG(x && scale && shift && y, "scale_shift_act: null argument");
  VLNCE_CHECK_ARG(M > 0 && C > 0, "scale_shift_act: bad shape");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  auto al = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  const bool vec = (C % 4 == 0) && al(x) && al(y) && al(scale) && al(shift) &&
                   (!residual || al(residual)) && (!center || al(center));
  const long work = vec ? M * (C / 4) : M * (long)C;
  const int grid = (int)(work / 256 + 1 < 8192 ? work / 256 + 1 : 8192);
  if (vec)
    hipLaunchKernelGGL(scale_shift_act_kernel<true>, dim3(grid), dim3(256), 0, s, x, scale, shift,
                       center, rows_per_sample, residual, y, M, C, act);
  else
    hipLaunchKernelGGL(scale_shift_act_kernel<false>, dim3(grid), dim3(256), 0, s, x, scale, shift,
                       center, rows_per_sample, residual, y, M, C, act);
  VLNCE_CHECK_LAUNCH("scale_shift_act");
  return 0;
}

extern "C" int vlnce_scale_shift_add_act(const float* x1, const float* scale1, const float* shift1,
                                         const float* center1, const float* x2,
                                         const float* scale2, const float* shift2,
                                         const float* center2, float* y, long M, int C, int act,
                                         vlnce_stream_t stream) {
  VLNCE_CHECK_ARG(x1 && scale1 && shift1 && x2 && scale2 && shift2 && y,
                  "scale_shift_add_act: null argument");
  VLNCE_CHECK_ARG(M > 0 && C > 0 && C % 4 == 0, "scale_shift_add_act: C must be a multiple of 4");
  const long work = M * (C / 4);
  const int grid = (int)(work / 256 + 1 < 8192 ? work / 256 + 1 : 8192);
  hipLaunchKernelGGL(scale_shift_add_act_kernel, dim3(grid), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), x1, scale1, shift1, center1, x2, scale2,
                     shift2, center2, y, M, C, act);
  VLNCE_CHECK_LAUNCH("scale_shift_add_act");
  return 0;
}

extern "C" int vlnce_gn_chunks(int HW) { return ceil_div(HW, GN_CHUNK); }

extern "C" int vlnce_gn_partial(const float* x, int Nimg, int HW, int C, float* partial,
                                vlnce_stream_t stream) {
  VLNCE_CHECK_ARG(x && partial, "gn_partial: null argument");
  VLNCE_CHECK_ARG(Nimg > 0 && HW > 0 && C > 0 && C % 4 == 0, "gn_partial: C must be a multiple of 4");
  const int chunks = ceil_div(HW, GN_CHUNK);
  hipLaunchKernelGGL(gn_partial_kernel, dim3(Nimg * chunks), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), x, HW, C, chunks, partial);
  VLNCE_CHECK_LAUNCH("gn_partial");
  return 0;
}

extern "C" int vlnce_gn_finalize(const float* partial, int Nimg, int HW, int C, int groups,
                                 const float* gamma, const float* beta, float eps,
                                 float* scale_out, float* shift_out, float* center_out,
                                 float* mean_out, float* rstd_out, vlnce_stream_t stream) {
  VLNCE_CHECK_ARG(partial && scale_out && shift_out, "gn_finalize: null argument");
  VLNCE_CHECK_ARG(groups > 0 && C % groups == 0, "gn_finalize: C %% groups != 0");
  const int chunks = ceil_div(HW, GN_CHUNK);
  hipLaunchKernelGGL(gn_finalize_kernel, dim3(ceil_div((long)Nimg * groups, 4)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), partial, Nimg, HW, C, groups, chunks, 0,
                     gamma, beta, eps, scale_out, shift_out, center_out, mean_out, rstd_out);
  VLNCE_CHECK_LAUNCH("gn_finalize");
  return 0;
}

// GroupNorm (+residual)(+act) of a SMALL activation in one launch: one workgroup per (sample,
// group, up to 1024 threads) reads its [HW, C/groups] slab twice out of L2 (shifted sum and sum
// of squares; apply) -- at one to a few environments a GroupNorm layer is otherwise statistics + finalize +
// apply, three launches of ~5 us each on the critical path of act() (the depth trunk is 54 such
// layers).  The arithmetic of the apply is scale_shift_act's: (x - mean) * (gamma * rstd) + beta.
template <int VEC>
__global__ __launch_bounds__(1024) void gn_small_kernel(const float* __restrict__ x, int HW, int C,
                                                        int groups, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float eps,
                                                        const float* __restrict__ residual,
                                                        float* __restrict__ y, int act) {
  __shared__ double red[2][16];
  const int n = blockIdx.x / groups, g = blockIdx.x - n * groups;
  const int cpg = C / groups;
  const long base = (long)n * HW * C + (long)g * cpg;
  const int cv = cpg / VEC;            // vectors per pixel of this group's channels
  const int total = HW * cv;
  typedef float vec_t __attribute__((ext_vector_type(VEC)));
  // pass 1: sum and sum of squares about the slab's first element (the shift keeps the
  // cancellation of E[x^2] - mean^2 harmless when |mean| >> std)
  const float x0 = x[base];
  float s = 0.f, q = 0.f;
  for (int e = threadIdx.x; e < total; e += blockDim.x) {
    const int px = e / cv, c = (e - px * cv) * VEC;
    const vec_t v = *reinterpret_cast<const vec_t*>(x + base + (long)px * C + c);
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
      const float d = v[k] - x0;
      s += d;
      q = fmaf(d, d, q);
    }
  }
  double ds = s, dq = q;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    ds += __shfl_xor(ds, o, 64);
    dq += __shfl_xor(dq, o, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    red[0][threadIdx.x >> 6] = ds;
    red[1][threadIdx.x >> 6] = dq;
  }
  __syncthreads();
  double ts = 0.0, tq = 0.0;
  for (int w = 0; w < (int)(blockDim.x >> 6); ++w) {
    ts += red[0][w];
    tq += red[1][w];
  }
  const double cnt = (double)HW * (double)cpg;
  const double md = ts / cnt;                       // mean - x0
  const float mean = (float)(md + (double)x0);
  const float rstd = (float)(1.0 / sqrt(fmax(tq / cnt - md * md, 0.0) + (double)eps));
  // pass 2: apply
  for (int e = threadIdx.x; e < total; e += blockDim.x) {
    const int px = e / cv, c = (e - px * cv) * VEC;
    const long o = base + (long)px * C + c;
    const vec_t v = *reinterpret_cast<const vec_t*>(x + o);
    vec_t r;
    if (residual) r = *reinterpret_cast<const vec_t*>(residual + o);
    vec_t out;
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
      const int ch = g * cpg + c + k;
      const float sc = (gamma ? gamma[ch] : 1.f) * rstd;
      float t = (v[k] - mean) * sc + (beta ? beta[ch] : 0.f);
      if (residual) t += r[k];
      out[k] = apply_act(t, act);
    }
    *reinterpret_cast<vec_t*>(y + o) = out;
  }
}

extern "C" int vlnce_group_norm_small(const float* x, int Nimg, int HW, int C, int groups,
                                      const float* gamma, const float* beta, float eps,
                                      const float* residual, int act, float* y,
                                      vlnce_stream_t stream) {
  VLNCE_CHECK_ARG(x && y, "group_norm_small: null argument");
  VLNCE_CHECK_ARG(Nimg > 0 && HW > 0 && groups > 0 && C % groups == 0,
                  "group_norm_small: bad shape (N %d, HW %d, C %d, groups %d)", Nimg, HW, C, groups);
  VLNCE_CHECK_ARG((long)HW * (C / groups) < (1L << 30), "group_norm_small: slab too large");
  const int cpg = C / groups;
  const long slab = (long)HW * cpg;
  const int threads = slab >= 4096 ? 1024 : slab >= 1024 ? 512 : 256;
  const bool vec4 = cpg % 4 == 0 && C % 4 == 0 && ((reinterpret_cast<uintptr_t>(x) |
                    reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(residual)) & 15) == 0;
  if (vec4)
    hipLaunchKernelGGL(gn_small_kernel<4>, dim3((unsigned)(Nimg * groups)), dim3(threads), 0,
                       reinterpret_cast<hipStream_t>(stream), x, HW, C, groups, gamma, beta, eps,
                       residual, y, act);
  else
    hipLaunchKernelGGL(gn_small_kernel<1>, dim3((unsigned)(Nimg * groups)), dim3(threads), 0,
                       reinterpret_cast<hipStream_t>(stream), x, HW, C, groups, gamma, beta, eps,
                       residual, y, act);
  VLNCE_CHECK_LAUNCH("group_norm_small");
  return 0;
}

extern "C" int vlnce_gn_finalize_tiles(const float* stat_partial, int tile_rows, int Nimg, int HW,
                                       int C, int groups, const float* gamma, const float* beta,
                                       float eps, float* scale_out, float* shift_out,
                                       float* center_out, float* mean_out, float* rstd_out,
                                       vlnce_stream_t stream) {
  VLNCE_CHECK_ARG(stat_partial && scale_out && shift_out, "gn_finalize_tiles: null argument");
  VLNCE_CHECK_ARG(groups > 0 && C % groups == 0, "gn_finalize_tiles: C %% groups != 0");
  VLNCE_CHECK_ARG(tile_rows > 0 && HW % tile_rows == 0,
                  "gn_finalize_tiles: the %d-row tiles must not straddle samples of %d pixels",
                  tile_rows, HW);
  hipLaunchKernelGGL(gn_finalize_kernel, dim3(ceil_div((long)Nimg * groups, 4)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), stat_partial, Nimg, HW, C, groups,
                     HW / tile_rows, tile_rows, gamma, beta, eps, scale_out, shift_out, center_out,
                     mean_out, rstd_out);
  VLNCE_CHECK_LAUNCH("gn_finalize_tiles");
  return 0;
}
