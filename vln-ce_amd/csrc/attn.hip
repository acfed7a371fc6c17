// Single-query dot-product attention (one workgroup per batch row).
//   logits[i] = <q, K[i]>  -> mask -> * scale -> softmax -> out = sum_i attn[i] V[i]
// K and V rows are each read exactly once, so they are streamed straight from
// HBM with float4 loads (no LDS staging: nothing is reused); the P logits live
// in LDS and the row reductions are wavefront shuffles.
#include "common.h"

namespace {

constexpr int MAXP = 1024;

__device__ __forceinline__ float block_reduce(float v, float* sbuf, bool is_max) {
  v = is_max ? wave_max(v) : wave_sum(v);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sbuf[w] = v;
  __syncthreads();
  float r = sbuf[0];
  for (int i = 1; i < 4; ++i) r = is_max ? fmaxf(r, sbuf[i]) : r + sbuf[i];
  return r;
}

__global__ __launch_bounds__(256) void attn_fwd_kernel(
    const float* __restrict__ q, const float* __restrict__ K, int ldk, const float* __restrict__ V,
    int ldv, const uint8_t* __restrict__ mask, int mask_mode, float scale, float* __restrict__ out,
    float* __restrict__ attn_out, int P, int Dk, int Dv, const long long* __restrict__ kv_index) {
  __shared__ float logits[MAXP];
  __shared__ float sbuf[4];
  const int b = blockIdx.x;
  const long kb = kv_index ? (long)kv_index[b] : (long)b;   // the K / V / mask block of this query
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* qb = q + (long)b * Dk;
  const float* Kb = K + kb * P * ldk;
  const float* Vb = V + kb * P * ldv;
  const bool vec = ((Dk & 3) == 0) && ((ldk & 3) == 0) &&
                   ((reinterpret_cast<uintptr_t>(K) | reinterpret_cast<uintptr_t>(q)) & 15) == 0;
  for (int i = wave; i < P; i += 4) {
    const float* kr = Kb + (long)i * ldk;
    float s = 0.f;
    if (vec) {
      for (int c = lane * 4; c < Dk; c += 256) {
        const f32x4 kv = *reinterpret_cast<const f32x4*>(kr + c);
        const f32x4 qv = *reinterpret_cast<const f32x4*>(qb + c);
        s += kv.x * qv.x + kv.y * qv.y + kv.z * qv.z + kv.w * qv.w;
      }
    } else {
      for (int c = lane; c < Dk; c += 64) s += kr[c] * qb[c];
    }
    s = wave_sum(s);
    if (lane == 0) {
      if (mask != nullptr) {
        const float mk = (float)mask[kb * P + i];
        if (mask_mode == 1) s = s - mk * 1e8f;
        if (mask_mode == 2) s = s * mk;
      }
      logits[i] = s * scale;
    }
  }
  __syncthreads();
  float mx = -INFINITY;
  for (int i = tid; i < P; i += 256) mx = fmaxf(mx, logits[i]);
  mx = block_reduce(mx, sbuf, true);
  float sum = 0.f;
  for (int i = tid; i < P; i += 256) {
    const float e = expf(logits[i] - mx);
    logits[i] = e;
    sum += e;
  }
  sum = block_reduce(sum, sbuf, false);
  const float inv = 1.f / sum;
  __syncthreads();
  for (int i = tid; i < P; i += 256) {
    const float a = logits[i] * inv;
    logits[i] = a;
    if (attn_out) attn_out[(long)b * P + i] = a;
  }
  __syncthreads();
  for (int c = tid; c < Dv; c += 256) {
    float acc = 0.f;
    for (int i = 0; i < P; ++i) acc += logits[i] * Vb[(long)i * ldv + c];
    out[(long)b * Dv + c] = acc;
  }
}

__global__ __launch_bounds__(256) void attn_bwd_kernel(
    const float* __restrict__ dout, const float* __restrict__ q, const float* __restrict__ K,
    int ldk, const float* __restrict__ V, int ldv, const uint8_t* __restrict__ mask, int mask_mode,
    float scale, const float* __restrict__ attn, float* __restrict__ dq, float* __restrict__ dK,
    int lddk, float* __restrict__ dV, int lddv, int P, int Dk, int Dv,
    const long long* __restrict__ kv_index) {
  __shared__ float dl[MAXP];  // d(attn) then d(logit)
  __shared__ float sbuf[4];
  const int b = blockIdx.x;
  const long kb = kv_index ? (long)kv_index[b] : (long)b;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* dob = dout + (long)b * Dv;
  const float* ab = attn + (long)b * P;
  const float* Vb = V + kb * P * ldv;
  const float* Kb = K + kb * P * ldk;
  const float* qb = q + (long)b * Dk;
  // d attn[i] = <dout, V[i]>,  dV[i] = attn[i] * dout   (a wave per key row; 16-byte accesses where
  // the rows allow it: the kernel is a chain of P row reads and writes, 4-byte ones were 47 us for
  // the text attention of a 64-environment step)
  const bool vec_v = ((Dv & 3) == 0) && ((ldv & 3) == 0) && (dV == nullptr || (lddv & 3) == 0) &&
                     ((reinterpret_cast<uintptr_t>(V) | reinterpret_cast<uintptr_t>(dout) |
                       reinterpret_cast<uintptr_t>(dV)) & 15) == 0;
  for (int i = wave; i < P; i += 4) {
    const float a = ab[i];
    float s = 0.f;
    if (vec_v) {
      for (int c = lane * 4; c < Dv; c += 256) {
        const f32x4 g = *reinterpret_cast<const f32x4*>(dob + c);
        const f32x4 v = *reinterpret_cast<const f32x4*>(Vb + (long)i * ldv + c);
        s += g.x * v.x + g.y * v.y + g.z * v.z + g.w * v.w;
        if (dV) *reinterpret_cast<f32x4*>(dV + ((long)b * P + i) * lddv + c) = f32x4{a * g.x, a * g.y, a * g.z, a * g.w};
      }
    } else {
      for (int c = lane; c < Dv; c += 64) {
        const float g = dob[c];
        s += g * Vb[(long)i * ldv + c];
        if (dV) dV[((long)b * P + i) * lddv + c] = a * g;
      }
    }
    s = wave_sum(s);
    if (lane == 0) dl[i] = s;
  }
  __syncthreads();
  float dot = 0.f;
  for (int i = tid; i < P; i += 256) dot += ab[i] * dl[i];
  dot = block_reduce(dot, sbuf, false);
  __syncthreads();
  for (int i = tid; i < P; i += 256) {
    float g = ab[i] * (dl[i] - dot) * scale;
    if (mask != nullptr && mask_mode == 2) g *= (float)mask[kb * P + i];
    dl[i] = g;
  }
  __syncthreads();
  // dK[i] = dl[i] * q ; dq = sum_i dl[i] * K[i]
  const bool vec_k = ((Dk & 3) == 0) && ((ldk & 3) == 0) && (dK == nullptr || (lddk & 3) == 0) &&
                     Dk <= 1024 &&
                     ((reinterpret_cast<uintptr_t>(K) | reinterpret_cast<uintptr_t>(q) |
                       reinterpret_cast<uintptr_t>(dK) | reinterpret_cast<uintptr_t>(dq)) & 15) == 0;
  if (vec_k) {
    // a wave per key row, 16 bytes per lane; the waves' partial dq meet in LDS (the logits' array
    // is free: dl has been read into registers per row)
    __shared__ f32x4 dq_part[4][256];   // [wave][Dk / 4 <= 256]
    f32x4 acc[4];                       // Dk <= 1024: up to 4 strips of 256 columns per wave
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int i = wave; i < P; i += 4) {
      const float g = dl[i];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int c = lane * 4 + t * 256;
        if (c < Dk) {
          const f32x4 kv = *reinterpret_cast<const f32x4*>(Kb + (long)i * ldk + c);
          acc[t].x += g * kv.x;
          acc[t].y += g * kv.y;
          acc[t].z += g * kv.z;
          acc[t].w += g * kv.w;
          if (dK) {
            const f32x4 qv = *reinterpret_cast<const f32x4*>(qb + c);
            *reinterpret_cast<f32x4*>(dK + ((long)b * P + i) * lddk + c) =
                f32x4{g * qv.x, g * qv.y, g * qv.z, g * qv.w};
          }
        }
      }
    }
#pragma unroll
    for (int t = 0; t < 4; ++t)
      if (lane * 4 + t * 256 < Dk) dq_part[wave][lane + t * 64] = acc[t];
    __syncthreads();
    if (dq)
      for (int c4 = tid; c4 * 4 < Dk; c4 += 256) {
        const f32x4 a0 = dq_part[0][c4], a1 = dq_part[1][c4], a2 = dq_part[2][c4], a3 = dq_part[3][c4];
        *reinterpret_cast<f32x4*>(dq + (long)b * Dk + c4 * 4) =
            f32x4{(a0.x + a1.x) + (a2.x + a3.x), (a0.y + a1.y) + (a2.y + a3.y),
                  (a0.z + a1.z) + (a2.z + a3.z), (a0.w + a1.w) + (a2.w + a3.w)};
      }
    return;
  }
  for (int c = tid; c < Dk; c += 256) {
    const float qc = qb[c];
    float acc = 0.f;
    for (int i = 0; i < P; ++i) {
      const float g = dl[i];
      acc += g * Kb[(long)i * ldk + c];
      if (dK) dK[((long)b * P + i) * lddk + c] = g * qc;
    }
    if (dq) dq[(long)b * Dk + c] = acc;
  }
}

// out[u, :] = sum over the rows b with index[b] == u of x[b, :], in row order (deterministic).
// One workgroup per (1024-float strip, u): the index test is wave-uniform, only matching rows are read.
__global__ __launch_bounds__(256) void segment_sum_kernel(const float* __restrict__ x,
                                                          const long long* __restrict__ index, int B,
                                                          long row_elems, float* __restrict__ out) {
  const long u = blockIdx.y;
  const long c = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (c >= row_elems) return;
  const bool v4 = c + 4 <= row_elems;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int b = 0; b < B; ++b) {
    if (index[b] != u) continue;
    const float* r = x + (long)b * row_elems + c;
    if (v4) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(r);
      acc.x += v.x;
      acc.y += v.y;
      acc.z += v.z;
      acc.w += v.w;
    } else {
      for (int e = 0; c + e < row_elems; ++e) acc[e] += r[e];
    }
  }
  float* o = out + u * row_elems + c;
  if (v4)
    *reinterpret_cast<f32x4*>(o) = acc;
  else
    for (int e = 0; c + e < row_elems; ++e) o[e] = acc[e];
}

__global__ __launch_bounds__(256) void rowzero_mask_kernel(const float* __restrict__ x, int ld,
                                                           long rows, int C,
                                                           uint8_t* __restrict__ mask) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* r = x + row * ld;
  int nz = 0;
  for (int c = lane; c < C; c += 64) nz |= (r[c] != 0.0f) ? 1 : 0;
  const unsigned long long any = __ballot(nz);
  if (lane == 0) mask[row] = any ? 0 : 1;
}

}  // namespace

extern "C" int vlnce_attn_fwd(const float* q, const float* K, int ldk, const float* V, int ldv,
                              const uint8_t* mask, int mask_mode, float scale, float* out,
                              float* attn_out, int B, int P, int Dk, int Dv,
                              vlnce_stream_t stream) {
  VLNCE_CHECK_ARG(q && K && V && out, "attn_fwd: null argument");
  VLNCE_CHECK_ARG(B > 0 && P > 0 && P <= MAXP && Dk > 0 && Dv > 0, "attn_fwd: bad shape (P<=%d)",
                  MAXP);
  VLNCE_CHECK_ARG(mask_mode >= 0 && mask_mode <= 2, "attn_fwd: bad mask_mode");
  hipLaunchKernelGGL(attn_fwd_kernel, dim3(B), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     q, K, ldk, V, ldv, mask, mask_mode, scale, out, attn_out, P, Dk, Dv,
                     (const long long*)nullptr);
  VLNCE_CHECK_LAUNCH("attn_fwd");
  return 0;
}

extern "C" int vlnce_attn_fwd_shared(const float* q, const float* K, int ldk, const float* V,
                                     int ldv, const uint8_t* mask, int mask_mode, float scale,
                                     const int64_t* kv_index, float* out, float* attn_out, int B,
                                     int P, int Dk, int Dv, vlnce_stream_t stream) {
  VLNCE_CHECK_ARG(q && K && V && out && kv_index, "attn_fwd_shared: null argument");
  VLNCE_CHECK_ARG(B > 0 && P > 0 && P <= MAXP && Dk > 0 && Dv > 0,
                  "attn_fwd_shared: bad shape (P<=%d)", MAXP);
  VLNCE_CHECK_ARG(mask_mode >= 0 && mask_mode <= 2, "attn_fwd_shared: bad mask_mode");
  hipLaunchKernelGGL(attn_fwd_kernel, dim3(B), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     q, K, ldk, V, ldv, mask, mask_mode, scale, out, attn_out, P, Dk, Dv,
                     reinterpret_cast<const long long*>(kv_index));
  VLNCE_CHECK_LAUNCH("attn_fwd_shared");
  return 0;
}

extern "C" int vlnce_attn_bwd(const float* dout, const float* q, const float* K, int ldk,
                              const float* V, int ldv, const uint8_t* mask, int mask_mode,
                              float scale, const float* attn, float* dq, float* dK, int lddk,
                              float* dV, int lddv, int B, int P, int Dk, int Dv,
                              vlnce_stream_t stream) {
  VLNCE_CHECK_ARG(dout && q && K && V && attn, "attn_bwd: null argument");
  VLNCE_CHECK_ARG(B > 0 && P > 0 && P <= MAXP, "attn_bwd: bad shape");
  hipLaunchKernelGGL(attn_bwd_kernel, dim3(B), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     dout, q, K, ldk, V, ldv, mask, mask_mode, scale, attn, dq, dK, lddk, dV, lddv,
                     P, Dk, Dv, (const long long*)nullptr);
  VLNCE_CHECK_LAUNCH("attn_bwd");
  return 0;
}

extern "C" int vlnce_attn_bwd_shared(const float* dout, const float* q, const float* K, int ldk,
                                     const float* V, int ldv, const uint8_t* mask, int mask_mode,
                                     float scale, const int64_t* kv_index, const float* attn,
                                     float* dq, float* dK, int lddk, float* dV, int lddv, int B,
                                     int P, int Dk, int Dv, vlnce_stream_t stream) {
  VLNCE_CHECK_ARG(dout && q && K && V && attn && kv_index, "attn_bwd_shared: null argument");
  VLNCE_CHECK_ARG(B > 0 && P > 0 && P <= MAXP, "attn_bwd_shared: bad shape");
  hipLaunchKernelGGL(attn_bwd_kernel, dim3(B), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     dout, q, K, ldk, V, ldv, mask, mask_mode, scale, attn, dq, dK, lddk, dV, lddv,
                     P, Dk, Dv, reinterpret_cast<const long long*>(kv_index));
  VLNCE_CHECK_LAUNCH("attn_bwd_shared");
  return 0;
}

extern "C" int vlnce_segment_sum(const float* x, const int64_t* index, int B, int U, long row_elems,
                                 float* out, vlnce_stream_t stream) {
  VLNCE_CHECK_ARG(x && index && out, "segment_sum: null argument");
  VLNCE_CHECK_ARG(B > 0 && U > 0 && U <= 65535 && row_elems > 0, "segment_sum: bad shape");
  VLNCE_CHECK_ARG(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out)) & 15) == 0 &&
                      row_elems % 4 == 0,
                  "segment_sum: rows must be 16-byte aligned multiples of 4 floats");
  hipLaunchKernelGGL(segment_sum_kernel, dim3(ceil_div(row_elems, 1024), U), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), x,
                     reinterpret_cast<const long long*>(index), B, row_elems, out);
  VLNCE_CHECK_LAUNCH("segment_sum");
  return 0;
}

extern "C" int vlnce_rowzero_mask(const float* x, int ld, long rows, int C, uint8_t* mask,
                                  vlnce_stream_t stream) {
  VLNCE_CHECK_ARG(x && mask && rows > 0 && C > 0, "rowzero_mask: bad argument");
  hipLaunchKernelGGL(rowzero_mask_kernel, dim3(ceil_div(rows, 4)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), x, ld, rows, C, mask);
  VLNCE_CHECK_LAUNCH("rowzero_mask");
  return 0;
}
