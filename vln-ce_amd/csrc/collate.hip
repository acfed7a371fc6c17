// Cached-feature DAgger data path: ragged trajectories -> the padded,
// time-major batch `_update_agent` consumes, built on the device.  The host ships each
// trajectory's rows once, compact (optionally fp16, as the LMDB feature cache stores them); the
// padding, the [T, B] interleave, the fp16 -> fp32 widening, the inflection weights and the
// not-done masks of dagger_trainer.py:39-114,196-208 happen here at HBM speed.
#include <hip/hip_fp16.h>

#include "common.h"

namespace {

// dst[(t * B + b) * D + d] = t < len_b ? src[(off_b + t) * D + d] : fill,  len_b = off[b+1] - off[b]
template <typename TIN, bool VEC>
__global__ __launch_bounds__(256) void ragged_pad_rows_kernel(const TIN* __restrict__ src,
                                                              const int* __restrict__ off, int B,
                                                              int Tmax, long D, float fill,
                                                              float* __restrict__ dst) {
  const long per = VEC ? D / 4 : D;
  const long total = (long)Tmax * B * per;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long row = i / per;
    const long d = (i - row * per) * (VEC ? 4 : 1);
    const int t = (int)(row / B), b = (int)(row - (long)t * B);
    const int o = off[b], len = off[b + 1] - o;
    if constexpr (VEC) {
      f32x4 v = {fill, fill, fill, fill};
      if (t < len) {
        const TIN* p = src + (long)(o + t) * D + d;
        if constexpr (sizeof(TIN) == 8) {
          v = f32x4{0.f, 0.f, 0.f, 0.f};  // (never instantiated with VEC)
        } else if constexpr (sizeof(TIN) == 2) {
          const __half2 a = *reinterpret_cast<const __half2*>(p);
          const __half2 c = *reinterpret_cast<const __half2*>(p + 2);
          v = f32x4{__low2float(a), __high2float(a), __low2float(c), __high2float(c)};
        } else {
          v = *reinterpret_cast<const f32x4*>(p);
        }
      }
      *reinterpret_cast<f32x4*>(dst + row * D + d) = v;
    } else {
      float v = fill;
      if (t < len) v = (float)src[(long)(o + t) * D + d];
      dst[row * D + d] = v;
    }
  }
}

__global__ __launch_bounds__(256) void ragged_pad_rows_i64_kernel(const long* __restrict__ src,
                                                                  const int* __restrict__ off, int B,
                                                                  int Tmax, long D, long fill,
                                                                  long* __restrict__ dst) {
  const long total = (long)Tmax * B * D;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long row = i / D;
    const long d = i - row * D;
    const int t = (int)(row / B), b = (int)(row - (long)t * B);
    const int o = off[b], len = off[b + 1] - o;
    dst[i] = t < len ? src[(long)(o + t) * D + d] : fill;
  }
}

// per (t, b): corrected action (0 past the end), inflection weight (1 at t = 0, `coef` where the
// oracle action changes, 1 elsewhere, 0 past the end) and the not-done mask (0 on the first row)
__global__ __launch_bounds__(256) void dagger_targets_kernel(const long* __restrict__ oracle,
                                                             const int* __restrict__ off, int B,
                                                             int Tmax, float coef,
                                                             long* __restrict__ corrected,
                                                             float* __restrict__ weights,
                                                             unsigned char* __restrict__ masks) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= Tmax * B) return;
  const int t = i / B, b = i - t * B;
  const int o = off[b], len = off[b + 1] - o;
  long a = 0;
  float w = 0.f;
  if (t < len) {
    a = oracle[o + t];
    w = (t == 0 || oracle[o + t - 1] == a) ? 1.f : coef;
    if (t == 0) w = coef;  // the reference marks the first step as an inflection
  }
  corrected[i] = a;
  weights[i] = w;
  masks[i] = t == 0 ? 0 : 1;
}

inline int grid_for(long work) {
  long g = (work + 255) / 256;
  if (g < 1) g = 1;
  if (g > 8192) g = 8192;
  return (int)g;
}

}  // namespace

extern "C" int vlnce_ragged_pad_rows(const void* src, int src_dtype, const int* offsets, int B,
                                     int Tmax, long D, float fill, float* dst,
                                     vlnce_stream_t stream) {
  VLNCE_CHECK_ARG(src && offsets && dst && B > 0 && Tmax > 0 && D > 0 && src_dtype >= 0 &&
                      src_dtype <= 2,
                  "ragged_pad_rows: bad argument");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const bool src_is_half = src_dtype == VLNCE_SRC_F16;
  const bool vec = src_dtype != VLNCE_SRC_I64 && (D % 4 == 0) &&
                   ((reinterpret_cast<uintptr_t>(dst) & 15) == 0) &&
                   ((reinterpret_cast<uintptr_t>(src) & (src_is_half ? 7 : 15)) == 0);
  const long work = (long)Tmax * B * (vec ? D / 4 : D);
  const dim3 g(grid_for(work)), blk(256);
  if (src_dtype == VLNCE_SRC_I64) {  // token / index sensors: upstream casts every sensor to fp32
    hipLaunchKernelGGL((ragged_pad_rows_kernel<long, false>), g, blk, 0, s,
                       reinterpret_cast<const long*>(src), offsets, B, Tmax, D, fill, dst);
  } else if (src_is_half) {
    const __half* p = reinterpret_cast<const __half*>(src);
    if (vec)
      hipLaunchKernelGGL((ragged_pad_rows_kernel<__half, true>), g, blk, 0, s, p, offsets, B, Tmax,
                         D, fill, dst);
    else
      hipLaunchKernelGGL((ragged_pad_rows_kernel<__half, false>), g, blk, 0, s, p, offsets, B,
                         Tmax, D, fill, dst);
  } else {
    const float* p = reinterpret_cast<const float*>(src);
    if (vec)
      hipLaunchKernelGGL((ragged_pad_rows_kernel<float, true>), g, blk, 0, s, p, offsets, B, Tmax,
                         D, fill, dst);
    else
      hipLaunchKernelGGL((ragged_pad_rows_kernel<float, false>), g, blk, 0, s, p, offsets, B, Tmax,
                         D, fill, dst);
  }
  VLNCE_CHECK_LAUNCH("ragged_pad_rows");
  return 0;
}

extern "C" int vlnce_ragged_pad_rows_i64(const int64_t* src, const int* offsets, int B, int Tmax,
                                         long D, int64_t fill, int64_t* dst,
                                         vlnce_stream_t stream) {
  VLNCE_CHECK_ARG(src && offsets && dst && B > 0 && Tmax > 0 && D > 0,
                  "ragged_pad_rows_i64: bad argument");
  static_assert(sizeof(long) == sizeof(int64_t), "LP64");
  hipLaunchKernelGGL(ragged_pad_rows_i64_kernel, dim3(grid_for((long)Tmax * B * D)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), reinterpret_cast<const long*>(src),
                     offsets, B, Tmax, D, (long)fill, reinterpret_cast<long*>(dst));
  VLNCE_CHECK_LAUNCH("ragged_pad_rows_i64");
  return 0;
}

extern "C" int vlnce_dagger_targets(const int64_t* oracle_actions, const int* offsets, int B,
                                    int Tmax, float inflection_coef, int64_t* corrected_out,
                                    float* weights_out, uint8_t* masks_out,
                                    vlnce_stream_t stream) {
  VLNCE_CHECK_ARG(oracle_actions && offsets && corrected_out && weights_out && masks_out && B > 0 &&
                      Tmax > 0,
                  "dagger_targets: bad argument");
  hipLaunchKernelGGL(dagger_targets_kernel, dim3(ceil_div(Tmax * B, 256)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream),
                     reinterpret_cast<const long*>(oracle_actions), offsets, B, Tmax,
                     inflection_coef, reinterpret_cast<long*>(corrected_out), weights_out,
                     masks_out);
  VLNCE_CHECK_LAUNCH("dagger_targets");
  return 0;
}

// ---- DD-PPO returns: RolloutStorage.compute_returns, rollout_storage.py:127-152.
// One thread per environment walks its T steps backwards; rewards [T,N], value_preds / masks /
// returns [T+1,N] (value_preds[T] = the bootstrap value, written here as upstream does).
namespace {
__global__ __launch_bounds__(256) void ppo_returns_kernel(const float* __restrict__ rewards,
                                                          float* __restrict__ value_preds,
                                                          const float* __restrict__ masks,
                                                          const float* __restrict__ next_value,
                                                          float* __restrict__ returns, int T, int N,
                                                          float gamma, float gamma_tau, int use_gae) {
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n >= N) return;
  if (use_gae) {
    value_preds[(long)T * N + n] = next_value[n];
    float gae = 0.f;
    for (int s = T - 1; s >= 0; --s) {
      const float v1 = value_preds[(long)(s + 1) * N + n], m1 = masks[(long)(s + 1) * N + n];
      const float v0 = value_preds[(long)s * N + n];
      const float delta = rewards[(long)s * N + n] + gamma * v1 * m1 - v0;
      gae = delta + gamma_tau * m1 * gae;
      returns[(long)s * N + n] = gae + v0;
    }
  } else {
    float ret = next_value[n];
    returns[(long)T * N + n] = ret;
    for (int s = T - 1; s >= 0; --s) {
      ret = ret * gamma * masks[(long)(s + 1) * N + n] + rewards[(long)s * N + n];
      returns[(long)s * N + n] = ret;
    }
  }
}

// WDDPPO minibatch loss (ddppo_alg.py:78-121) and ALL its gradients in one launch of one
// workgroup: clipped surrogate, (clipped) value loss, the three entropy terms, the offset L1
// regulariser (a constant: the sampled offsets carry no gradient).  stats[8] = {loss, value_loss,
// action_loss, entropy_loss, mean pano / offset / distance entropy, offset_loss}; grads [5][B] =
// d loss / d {values, log-probs, pano / offset / distance entropy} for a unit upstream gradient.
// Ties follow torch: max / min send half the gradient to each side, clamp passes it on [lo, hi].
struct PpoLossParams {
  const float *values, *returns, *value_preds, *logp, *old_logp, *adv, *ent[3], *radians;
  float* stats;
  float* grads;
  int B;
  float clip, value_coef, entropy_coef, ent_coef[3], reg_coef;
  int use_clipped;
};
__global__ __launch_bounds__(256) void ppo_loss_kernel(PpoLossParams p) {
  __shared__ double red[4][8];
  const int tid = threadIdx.x, B = p.B;
  const double inv = 1.0 / (double)B;
  double acc[7] = {0, 0, 0, 0, 0, 0, 0};   // value, surrogate, weighted entropy, 3 entropies, |radians|
  for (int i = tid; i < B; i += 256) {
    const float v = p.values[i], R = p.returns[i];
    const float a = (v - R) * (v - R);
    float vl = a, dv = 2.f * (v - R);
    if (p.use_clipped) {
      const float vp = p.value_preds[i];
      const float diff = v - vp;
      const float cl = fminf(fmaxf(diff, -p.clip), p.clip);
      const bool pass = diff >= -p.clip && diff <= p.clip;   // clamp backward
      const float vpc = vp + cl;
      const float b = (vpc - R) * (vpc - R);
      const float db = pass ? 2.f * (vpc - R) : 0.f;
      if (a > b) { vl = a; }
      else if (a < b) { vl = b; dv = db; }
      else { vl = a; dv = 0.5f * dv + 0.5f * db; }
    }
    acc[0] += (double)vl;
    p.grads[i] = (float)(0.5 * (double)p.value_coef * inv) * dv;
    const float A = p.adv[i];
    const float r = __expf(p.logp[i] - p.old_logp[i]);
    const float lo = 1.f - p.clip, hi = 1.f + p.clip;
    const float rc = fminf(fmaxf(r, lo), hi);
    const float s1 = r * A, s2 = rc * A;
    const float d2 = (r >= lo && r <= hi) ? A : 0.f;
    float dr;
    if (s1 < s2) dr = A;
    else if (s1 > s2) dr = d2;
    else dr = 0.5f * A + 0.5f * d2;
    acc[1] += (double)fminf(s1, s2);
    p.grads[B + i] = (float)(-inv) * dr * r;
    double we = 0.0;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const float e = p.ent[k][i];
      acc[3 + k] += (double)e;
      we += (double)p.ent_coef[k] * (double)e;
      p.grads[(2 + k) * B + i] = (float)(-(double)p.entropy_coef * (double)p.ent_coef[k] * inv);
    }
    acc[2] += we;
    if (p.radians) acc[6] += (double)fabsf(p.radians[i]);
  }
#pragma unroll
  for (int k = 0; k < 7; ++k) {
    double v = acc[k];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if ((tid & 63) == 0) red[tid >> 6][k] = v;
  }
  __syncthreads();
  if (tid == 0) {
    double t[7];
    for (int k = 0; k < 7; ++k) t[k] = (red[0][k] + red[1][k] + red[2][k] + red[3][k]) * inv;
    const double value_loss = 0.5 * t[0] * (double)p.value_coef;
    const double action_loss = -t[1];
    const double entropy_loss = t[2] * (double)p.entropy_coef;
    const double offset_loss = p.radians ? (double)p.reg_coef * t[6] : 0.0;
    p.stats[0] = (float)(value_loss + action_loss + offset_loss - entropy_loss);
    p.stats[1] = (float)value_loss;
    p.stats[2] = (float)action_loss;
    p.stats[3] = (float)entropy_loss;
    p.stats[4] = (float)t[3];
    p.stats[5] = (float)t[4];
    p.stats[6] = (float)t[5];
    p.stats[7] = (float)offset_loss;
  }
}

}  // namespace

extern "C" int vlnce_ppo_returns(const float* rewards, float* value_preds, const float* masks,
                                 const float* next_value, float* returns, int T, int N,
                                 float gamma, float tau, int use_gae, vlnce_stream_t stream) {
  VLNCE_CHECK_ARG(rewards && value_preds && masks && next_value && returns && T > 0 && N > 0,
                  "ppo_returns: bad argument");
  hipLaunchKernelGGL(ppo_returns_kernel, dim3(ceil_div(N, 256)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), rewards, value_preds, masks, next_value,
                     returns, T, N, gamma, (float)((double)gamma * (double)tau), use_gae);
  VLNCE_CHECK_LAUNCH("ppo_returns");
  return 0;
}

extern "C" int vlnce_ppo_loss(const float* values, const float* returns, const float* value_preds,
                              const float* logp, const float* old_logp, const float* adv,
                              const float* ent_pano, const float* ent_offset, const float* ent_distance,
                              const float* radians, int B, float clip, float value_coef,
                              float entropy_coef, float pano_coef, float offset_coef,
                              float distance_coef, float reg_coef, int use_clipped, float* stats,
                              float* grads, vlnce_stream_t stream) {
  VLNCE_CHECK_ARG(values && returns && logp && old_logp && adv && ent_pano && ent_offset && ent_distance &&
                      stats && grads && (value_preds || !use_clipped),
                  "ppo_loss: null argument");
  VLNCE_CHECK_ARG(B > 0, "ppo_loss: empty minibatch");
  PpoLossParams p{};
  p.values = values; p.returns = returns; p.value_preds = value_preds; p.logp = logp;
  p.old_logp = old_logp; p.adv = adv; p.ent[0] = ent_pano; p.ent[1] = ent_offset; p.ent[2] = ent_distance;
  p.radians = radians; p.stats = stats; p.grads = grads; p.B = B; p.clip = clip;
  p.value_coef = value_coef; p.entropy_coef = entropy_coef; p.ent_coef[0] = pano_coef;
  p.ent_coef[1] = offset_coef; p.ent_coef[2] = distance_coef; p.reg_coef = reg_coef;
  p.use_clipped = use_clipped;
  hipLaunchKernelGGL(ppo_loss_kernel, dim3(1), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), p);
  VLNCE_CHECK_LAUNCH("ppo_loss");
  return 0;
}
