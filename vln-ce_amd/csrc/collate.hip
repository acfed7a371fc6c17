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
