// Recurrent-cell pointwise stages (GRU / LSTM, torch gate order) and small
// row utilities.  The GEMM halves (x W_ih^T, h W_hh^T) run on the MFMA kernel.
#include "common.h"

namespace {

__device__ __forceinline__ float sigm(float x) { return 1.f / (1.f + expf(-x)); }

__global__ __launch_bounds__(256) void gru_gates_fwd_kernel(
    const float* __restrict__ gi, const float* __restrict__ gh, const float* __restrict__ h_prev,
    const uint8_t* __restrict__ mask, float* __restrict__ h_out, float* __restrict__ gates_out,
    float* __restrict__ hn_out, int B, int H) {
  const long total = (long)B * H;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long)gridDim.x * blockDim.x) {
    const long b = i / H;
    const int j = (int)(i - b * H);
    const float* gib = gi + b * 3 * H;
    const float* ghb = gh + b * 3 * H;
    float hp = h_prev[i];
    if (mask) hp *= (float)mask[b];
    const float r = sigm(gib[j] + ghb[j]);
    const float z = sigm(gib[H + j] + ghb[H + j]);
    const float hn = ghb[2 * H + j];
    const float n = tanhf(gib[2 * H + j] + r * hn);
    h_out[i] = (1.f - z) * n + z * hp;
    if (gates_out) {
      float* g = gates_out + b * 3 * H;
      g[j] = r;
      g[H + j] = z;
      g[2 * H + j] = n;
    }
    if (hn_out) hn_out[i] = hn;
  }
}

__global__ __launch_bounds__(256) void gru_gates_bwd_kernel(
    const float* __restrict__ dh_out, const float* __restrict__ gates, const float* __restrict__ hn,
    const float* __restrict__ h_prev, const uint8_t* __restrict__ mask, float* __restrict__ dgi,
    float* __restrict__ dgh, float* __restrict__ dh_prev, int B, int H) {
  const long total = (long)B * H;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long)gridDim.x * blockDim.x) {
    const long b = i / H;
    const int j = (int)(i - b * H);
    const float* g = gates + b * 3 * H;
    const float r = g[j], z = g[H + j], n = g[2 * H + j];
    const float mk = mask ? (float)mask[b] : 1.f;
    const float hp = h_prev[i] * mk;
    const float d = dh_out[i];
    const float dn = d * (1.f - z);
    const float dz = d * (hp - n);
    const float dnp = dn * (1.f - n * n);
    const float dr = dnp * hn[i];
    const float drp = dr * r * (1.f - r);
    const float dzp = dz * z * (1.f - z);
    float* a = dgi + b * 3 * H;
    float* c = dgh + b * 3 * H;
    a[j] = drp;
    a[H + j] = dzp;
    a[2 * H + j] = dnp;
    c[j] = drp;
    c[H + j] = dzp;
    c[2 * H + j] = dnp * r;
    dh_prev[i] = d * z * mk;
  }
}

__global__ __launch_bounds__(256) void lstm_gates_fwd_kernel(
    const float* __restrict__ gi, const float* __restrict__ gh, const float* __restrict__ c_prev,
    const uint8_t* __restrict__ mask, float* __restrict__ h_out, float* __restrict__ c_out,
    float* __restrict__ gates_out, int B, int H) {
  const long total = (long)B * H;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long)gridDim.x * blockDim.x) {
    const long b = i / H;
    const int j = (int)(i - b * H);
    const float* a = gi + b * 4 * H;
    const float* c = gh + b * 4 * H;
    float cp = c_prev[i];
    if (mask) cp *= (float)mask[b];
    const float ig = sigm(a[j] + c[j]);
    const float fg = sigm(a[H + j] + c[H + j]);
    const float gg = tanhf(a[2 * H + j] + c[2 * H + j]);
    const float og = sigm(a[3 * H + j] + c[3 * H + j]);
    const float cn = fg * cp + ig * gg;
    c_out[i] = cn;
    h_out[i] = og * tanhf(cn);
    if (gates_out) {
      float* g = gates_out + b * 4 * H;
      g[j] = ig;
      g[H + j] = fg;
      g[2 * H + j] = gg;
      g[3 * H + j] = og;
    }
  }
}

__global__ __launch_bounds__(256) void lstm_gates_bwd_kernel(
    const float* __restrict__ dh_out, const float* __restrict__ dc_out,
    const float* __restrict__ gates, const float* __restrict__ c_prev,
    const float* __restrict__ c_out, const uint8_t* __restrict__ mask, float* __restrict__ dgates,
    float* __restrict__ dc_prev, int B, int H) {
  const long total = (long)B * H;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long)gridDim.x * blockDim.x) {
    const long b = i / H;
    const int j = (int)(i - b * H);
    const float* g = gates + b * 4 * H;
    const float ig = g[j], fg = g[H + j], gg = g[2 * H + j], og = g[3 * H + j];
    const float mk = mask ? (float)mask[b] : 1.f;
    const float cp = c_prev[i] * mk;
    const float tc = tanhf(c_out[i]);
    const float dh = dh_out ? dh_out[i] : 0.f;
    float dc = (dc_out ? dc_out[i] : 0.f) + dh * og * (1.f - tc * tc);
    float* d = dgates + b * 4 * H;
    d[j] = dc * gg * ig * (1.f - ig);
    d[H + j] = dc * cp * fg * (1.f - fg);
    d[2 * H + j] = dc * ig * (1.f - gg * gg);
    d[3 * H + j] = dh * tc * og * (1.f - og);
    dc_prev[i] = dc * fg * mk;
  }
}

__global__ __launch_bounds__(256) void mask_rows_kernel(const float* __restrict__ x,
                                                        const uint8_t* __restrict__ mask,
                                                        float* __restrict__ out, int B, int H) {
  const long total = (long)B * H;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long)gridDim.x * blockDim.x)
    out[i] = x[i] * (float)mask[i / H];
}

// out[b, :] = mask[b] ? a[b, :] : b_[b, :]   (a or b_ may be NULL => zeros)
__global__ __launch_bounds__(256) void select_rows_kernel(const uint8_t* __restrict__ mask,
                                                          const float* __restrict__ a,
                                                          const float* __restrict__ b_,
                                                          float* __restrict__ out, int B, int H) {
  const long total = (long)B * H;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long)gridDim.x * blockDim.x) {
    const bool m = mask[i / H] != 0;
    out[i] = m ? (a ? a[i] : 0.f) : (b_ ? b_[i] : 0.f);
  }
}

// dz = dy * act'(y)  expressed through the activation OUTPUT y
__global__ __launch_bounds__(256) void act_bwd_kernel(const float* __restrict__ dy,
                                                      const float* __restrict__ y,
                                                      float* __restrict__ dz, long n, int act) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long)gridDim.x * blockDim.x) {
    const float g = dy[i], v = y[i];
    float r = g;
    if (act == VLNCE_ACT_RELU) r = v > 0.f ? g : 0.f;
    if (act == VLNCE_ACT_SIGMOID) r = g * v * (1.f - v);
    if (act == VLNCE_ACT_TANH) r = g * (1.f - v * v);
    dz[i] = r;
  }
}

// out[n] += sum_m x[m, n]: 64 columns x a slice of rows per block; slices combine with
// atomics into an output the host entry has zeroed (unless accumulating).
// DIRECT: one row slice covers all of M, the block owns its 64 outputs (no atomics, no memset).
template <bool DIRECT>
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ x, int ldx, int M,
                                                     int N, float* __restrict__ out,
                                                     int rows_per_block, int accumulate) {
  __shared__ float red[4][64];
  const int cl = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const int n = blockIdx.x * 64 + cl;
  const int m0 = blockIdx.y * rows_per_block;
  const int m1 = min(M, m0 + rows_per_block);
  float s = 0.f;
  if (n < N)
    for (int m = m0 + sl; m < m1; m += 4) s += x[(long)m * ldx + n];
  red[sl][cl] = s;
  __syncthreads();
  if (sl == 0 && n < N) {
    const float t = red[0][cl] + red[1][cl] + red[2][cl] + red[3][cl];
    if (DIRECT)
      out[n] = accumulate ? out[n] + t : t;
    else
      atomicAdd(out + n, t);
  }
}

// ------------------------------------------------------------------ fused rollout steps
// One step of a masked GRU / LSTM state encoder over N <= 16 episodes is a GEMV-sized product
// (h W_hh^T: N x H x G*H) between two pointwise stages; as three launches forward and six backward
// it is pure launch latency (a cached-feature DAgger update is ~2700 dependent launches).
// Fused: forward = ONE launch per step, backward = TWO.
//
// rows_dot: res[r * MAXN + n] = sum_k X[n][k] * Wrows[r][k] for the R weight rows of this
// workgroup.  Both operands sit in LDS (row pitch K + 4 floats: neighbouring rows land on
// different banks): the weight rows are fetched with coalesced, independent float4 loads -- one
// memory latency for the whole panel -- and every (row, episode) pair is a private dot product
// of a K-slice in one thread, so there is no cross-lane reduction chain (a first version reduced
// 64-lane partials with six dependent ds_bpermute per value: 19 us per step).
__device__ __forceinline__ void stage_rows(float* __restrict__ dst, const float* __restrict__ src,
                                           long ld, int r0, int R, int K) {
  // 8 independent 16-byte loads in flight per thread before the first LDS write: the panel costs
  // ~2 memory latencies, not one per element
  const int Kp = K + 4, k4 = K >> 2, total = R * k4;
  for (int base = threadIdx.x; base < total; base += 256 * 8) {
    f32x4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = base + u * 256;
      if (i < total) {
        const int r = i / k4, c = (i - r * k4) * 4;
        v[u] = *reinterpret_cast<const f32x4*>(src + (long)(r0 + r) * ld + c);
      }
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = base + u * 256;
      if (i < total) {
        const int r = i / k4, c = (i - r * k4) * 4;
        *reinterpret_cast<f32x4*>(dst + r * Kp + c) = v[u];
      }
    }
  }
}

template <int MAXN>
__device__ __forceinline__ void rows_dot(const float* __restrict__ Xs, const float* __restrict__ Ws,
                                         int N, int K, int R, float* __restrict__ part,
                                         float* __restrict__ res) {
  // pairs (r, n) x K-slices over the 256 threads; slice partials meet in `part`
  const int Kp = K + 4;
  const int P = R * N;
  int KS = 256 / P;
  KS = KS >= 4 ? 4 : (KS >= 2 ? 2 : 1);
  const int len = K / KS;  // multiple of 4 (K % 16 == 0)
  for (int t = threadIdx.x; t < P * KS; t += 256) {
    const int pair = t % P, ks = t / P;
    const int r = pair / N, n = pair - r * N;
    const float* w = Ws + r * Kp + ks * len;
    const float* x = Xs + n * Kp + ks * len;
    float a0 = 0.f, a1 = 0.f;
    for (int k = 0; k < len; k += 8) {
      const f32x4 w0 = *reinterpret_cast<const f32x4*>(w + k);
      const f32x4 x0 = *reinterpret_cast<const f32x4*>(x + k);
      const f32x4 w1 = *reinterpret_cast<const f32x4*>(w + k + 4);
      const f32x4 x1 = *reinterpret_cast<const f32x4*>(x + k + 4);
      a0 += w0.x * x0.x + w0.y * x0.y + w0.z * x0.z + w0.w * x0.w;
      a1 += w1.x * x1.x + w1.y * x1.y + w1.z * x1.z + w1.w * x1.w;
    }
    part[ks * P + pair] = a0 + a1;
  }
  __syncthreads();
  for (int pair = threadIdx.x; pair < P; pair += 256) {
    float v = part[pair];
    for (int ks = 1; ks < KS; ++ks) v += part[ks * P + pair];
    const int r = pair / N, n = pair - r * N;
    res[r * MAXN + n] = v;
  }
  __syncthreads();
}

constexpr int STEP_MAXN = 16;   // episodes per fused step
constexpr int STEP_UNITS = 8;   // hidden units per workgroup (forward)
constexpr int STEP_COLS = 8;    // carry columns per workgroup (backward)

// forward step: hp = mask * h_prev (stored for backward); gh = hp W_hh^T + b_hh for this
// workgroup's hidden units; gates; h (and c).  LSTM: c_prev masked too.
template <bool LSTM>
__global__ __launch_bounds__(256) void rnn_step_fwd_kernel(
    const float* __restrict__ gi, const float* __restrict__ h_prev, const float* __restrict__ c_prev,
    const uint8_t* __restrict__ mask, const float* __restrict__ w_hh,
    const float* __restrict__ b_hh, float* __restrict__ hp_out, float* __restrict__ h_out,
    float* __restrict__ aux_out, float* __restrict__ gates_out, int N, int H) {
  constexpr int G = LSTM ? 4 : 3;
  constexpr int R = G * STEP_UNITS;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int Hp = H + 4;
  float* Xs = sm;                          // [N][Hp] masked previous state
  float* Ws = Xs + N * Hp;                 // [R][Hp] weight rows of this workgroup's units
  float* part = Ws + R * Hp;               // [4][R*N]
  float* res = part + 4 * R * STEP_MAXN;   // [R][MAXN]
  const int j0 = blockIdx.x * STEP_UNITS;
  // weight rows of unit j: j, H+j, 2H+j(, 3H+j), staged as G groups of UNITS consecutive rows
  for (int g = 0; g < G; ++g) stage_rows(Ws + g * STEP_UNITS * Hp, w_hh, H, g * H + j0, STEP_UNITS, H);
  for (int i = threadIdx.x; i < N * (H >> 2); i += 256) {
    const int n = i / (H >> 2), k = (i - n * (H >> 2)) * 4;
    const f32x4 v = *reinterpret_cast<const f32x4*>(h_prev + (long)n * H + k) *
                    (mask ? (float)mask[n] : 1.f);
    *reinterpret_cast<f32x4*>(Xs + n * Hp + k) = v;
    if (blockIdx.x == 0 && hp_out) *reinterpret_cast<f32x4*>(hp_out + (long)n * H + k) = v;
  }
  __syncthreads();
  rows_dot<STEP_MAXN>(Xs, Ws, N, H, R, part, res);
  for (int i = threadIdx.x; i < N * STEP_UNITS; i += 256) {
    const int n = i / STEP_UNITS, u = i - n * STEP_UNITS, j = j0 + u;
    if (j >= H) continue;
    const float* gib = gi + (long)n * G * H;
    float gh[G];
#pragma unroll
    for (int g = 0; g < G; ++g) gh[g] = res[(g * STEP_UNITS + u) * STEP_MAXN + n] + b_hh[g * H + j];
    const long o = (long)n * H + j;
    float* gs = gates_out + (long)n * G * H;
    if constexpr (LSTM) {
      const float cp = c_prev[o] * (mask ? (float)mask[n] : 1.f);
      const float ig = sigm(gib[j] + gh[0]);
      const float fg = sigm(gib[H + j] + gh[1]);
      const float gg = tanhf(gib[2 * H + j] + gh[2]);
      const float og = sigm(gib[3 * H + j] + gh[3]);
      const float cn = fg * cp + ig * gg;
      aux_out[o] = cn;
      h_out[o] = og * tanhf(cn);
      gs[j] = ig;
      gs[H + j] = fg;
      gs[2 * H + j] = gg;
      gs[3 * H + j] = og;
    } else {
      const float hp = Xs[n * Hp + j];
      const float r = sigm(gib[j] + gh[0]);
      const float z = sigm(gib[H + j] + gh[1]);
      const float n_ = tanhf(gib[2 * H + j] + r * gh[2]);
      h_out[o] = (1.f - z) * n_ + z * hp;
      aux_out[o] = gh[2];
      gs[j] = r;
      gs[H + j] = z;
      gs[2 * H + j] = n_;
    }
  }
}

// backward step, part 1 (pointwise): dh = dout + carry; gate gradients; acc0 = dh * z (GRU) |
// dc_prev (LSTM).  Part 2 below turns dgh into the recurrent data gradient.
template <bool LSTM>
__global__ __launch_bounds__(256) void rnn_step_bwd_gates_kernel(
    const float* __restrict__ dout, const float* __restrict__ carry, const float* __restrict__ dc,
    const float* __restrict__ gates, const float* __restrict__ aux, const float* __restrict__ hp,
    const float* __restrict__ c_prev, const uint8_t* __restrict__ mask, float* __restrict__ dgi,
    float* __restrict__ dgh, float* __restrict__ acc0, float* __restrict__ dc_prev, int N, int H) {
  constexpr int G = LSTM ? 4 : 3;
  const long total = (long)N * H;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long n = i / H;
    const int j = (int)(i - n * H);
    const float* g = gates + n * G * H;
    const float d = dout[i] + carry[i];
    float* a = dgi + n * G * H;
    if constexpr (LSTM) {
      const float ig = g[j], fg = g[H + j], gg = g[2 * H + j], og = g[3 * H + j];
      const float mk = mask ? (float)mask[n] : 1.f;
      const float cp = c_prev[i] * mk;
      const float tc = tanhf(aux[i]);
      const float dcc = dc[i] + d * og * (1.f - tc * tc);
      a[j] = dcc * gg * ig * (1.f - ig);
      a[H + j] = dcc * cp * fg * (1.f - fg);
      a[2 * H + j] = dcc * ig * (1.f - gg * gg);
      a[3 * H + j] = d * tc * og * (1.f - og);
      dc_prev[i] = dcc * fg * mk;
      acc0[i] = 0.f;
    } else {
      const float r = g[j], z = g[H + j], nn = g[2 * H + j];
      const float dn = d * (1.f - z);
      const float dz = d * (hp[i] - nn);
      const float dnp = dn * (1.f - nn * nn);
      const float dr = dnp * aux[i];
      const float drp = dr * r * (1.f - r);
      const float dzp = dz * z * (1.f - z);
      a[j] = drp;
      a[H + j] = dzp;
      a[2 * H + j] = dnp;
      float* c = dgh + n * G * H;
      c[j] = drp;
      c[H + j] = dzp;
      c[2 * H + j] = dnp * r;
      acc0[i] = d * z;
    }
  }
}

// backward step, part 2: carry[n, k] = mask[n] * (acc0[n, k] + sum_w dgh[n, w] * W_hh[w, k]),
// with W_hh^T rows (wt [H][G*H]) so the reduction runs along contiguous memory
__global__ __launch_bounds__(256) void rnn_step_bwd_carry_kernel(
    const float* __restrict__ dgh, const float* __restrict__ wt, const float* __restrict__ acc0,
    const uint8_t* __restrict__ mask, float* __restrict__ carry, int N, int H, int GH) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int Kp = GH + 4;
  float* Xs = sm;                           // [N][Kp] gate gradients of this step
  float* Ws = Xs + N * Kp;                  // [COLS][Kp] rows of W_hh^T
  float* part = Ws + STEP_COLS * Kp;        // [4][COLS*N]
  float* res = part + 4 * STEP_COLS * STEP_MAXN;
  const int k0 = blockIdx.x * STEP_COLS;
  const int R = min(STEP_COLS, H - k0);
  stage_rows(Ws, wt, GH, k0, R, GH);
  for (int i = threadIdx.x; i < N * (GH >> 2); i += 256) {
    const int n = i / (GH >> 2), c = (i - n * (GH >> 2)) * 4;
    *reinterpret_cast<f32x4*>(Xs + n * Kp + c) = *reinterpret_cast<const f32x4*>(dgh + (long)n * GH + c);
  }
  __syncthreads();
  rows_dot<STEP_MAXN>(Xs, Ws, N, GH, R, part, res);
  for (int i = threadIdx.x; i < N * STEP_COLS; i += 256) {
    const int n = i / STEP_COLS, u = i - n * STEP_COLS, k = k0 + u;
    if (k >= H) continue;
    const long o = (long)n * H + k;
    carry[o] = (acc0[o] + res[u * STEP_MAXN + n]) * (mask ? (float)mask[n] : 1.f);
  }
}

inline int grid_for(long work) {
  long g = (work + 255) / 256;
  if (g < 1) g = 1;
  if (g > 4096) g = 4096;
  return (int)g;
}

}  // namespace

extern "C" int vlnce_gru_gates_fwd(const float* gi, const float* gh, const float* h_prev,
                                   const uint8_t* mask, float* h_out, float* gates_out,
                                   float* hn_out, int B, int H, vlnce_stream_t stream) {
  VLNCE_CHECK_ARG(gi && gh && h_prev && h_out && B > 0 && H > 0, "gru_gates_fwd: bad argument");
  hipLaunchKernelGGL(gru_gates_fwd_kernel, dim3(grid_for((long)B * H)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), gi, gh, h_prev, mask, h_out, gates_out,
                     hn_out, B, H);
  VLNCE_CHECK_LAUNCH("gru_gates_fwd");
  return 0;
}

extern "C" int vlnce_gru_gates_bwd(const float* dh_out, const float* gates, const float* hn,
                                   const float* h_prev, const uint8_t* mask, float* dgi, float* dgh,
                                   float* dh_prev, int B, int H, vlnce_stream_t stream) {
  VLNCE_CHECK_ARG(dh_out && gates && hn && h_prev && dgi && dgh && dh_prev,
                  "gru_gates_bwd: null argument");
  hipLaunchKernelGGL(gru_gates_bwd_kernel, dim3(grid_for((long)B * H)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), dh_out, gates, hn, h_prev, mask, dgi,
                     dgh, dh_prev, B, H);
  VLNCE_CHECK_LAUNCH("gru_gates_bwd");
  return 0;
}

extern "C" int vlnce_lstm_gates_fwd(const float* gi, const float* gh, const float* c_prev,
                                    const uint8_t* mask, float* h_out, float* c_out,
                                    float* gates_out, int B, int H, vlnce_stream_t stream) {
  VLNCE_CHECK_ARG(gi && gh && c_prev && h_out && c_out, "lstm_gates_fwd: null argument");
  hipLaunchKernelGGL(lstm_gates_fwd_kernel, dim3(grid_for((long)B * H)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), gi, gh, c_prev, mask, h_out, c_out,
                     gates_out, B, H);
  VLNCE_CHECK_LAUNCH("lstm_gates_fwd");
  return 0;
}

extern "C" int vlnce_lstm_gates_bwd(const float* dh_out, const float* dc_out, const float* gates,
                                    const float* c_prev, const float* c_out, const uint8_t* mask,
                                    float* dgates, float* dc_prev, int B, int H,
                                    vlnce_stream_t stream) {
  VLNCE_CHECK_ARG(gates && c_prev && c_out && dgates && dc_prev, "lstm_gates_bwd: null argument");
  hipLaunchKernelGGL(lstm_gates_bwd_kernel, dim3(grid_for((long)B * H)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), dh_out, dc_out, gates, c_prev, c_out,
                     mask, dgates, dc_prev, B, H);
  VLNCE_CHECK_LAUNCH("lstm_gates_bwd");
  return 0;
}

extern "C" int vlnce_mask_rows(const float* x, const uint8_t* mask, float* out, int B, int H,
                               vlnce_stream_t stream) {
  VLNCE_CHECK_ARG(x && mask && out && B > 0 && H > 0, "mask_rows: bad argument");
  hipLaunchKernelGGL(mask_rows_kernel, dim3(grid_for((long)B * H)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), x, mask, out, B, H);
  VLNCE_CHECK_LAUNCH("mask_rows");
  return 0;
}

extern "C" int vlnce_colsum(const float* x, int ldx, int M, int N, float* out, int accumulate,
                            vlnce_stream_t stream) {
  VLNCE_CHECK_ARG(x && out && M > 0 && N > 0, "colsum: bad argument");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int col_blocks = ceil_div(N, 64);
  if (M <= 256) {  // the tail's bias gradients (M = num_envs rows): one launch, nothing else
    hipLaunchKernelGGL(colsum_kernel<true>, dim3(col_blocks, 1), dim3(256), 0, s, x, ldx, M, N,
                       out, M, accumulate);
    VLNCE_CHECK_LAUNCH("colsum");
    return 0;
  }
  if (!accumulate) {
    vlnce_zero(out, 1, N, N, s);
  }
  int slices = ceil_div(256, col_blocks);
  if (slices > ceil_div(M, 32)) slices = ceil_div(M, 32);
  const int rows_per_block = ceil_div(M, slices);
  hipLaunchKernelGGL(colsum_kernel<false>, dim3(col_blocks, ceil_div(M, rows_per_block)),
                     dim3(256), 0, s, x, ldx, M, N, out, rows_per_block, accumulate);
  VLNCE_CHECK_LAUNCH("colsum");
  return 0;
}

// dW[tok[r], :] += g[r, :] for every row r with tok[r] != padding_idx (dW zeroed by the caller)
__global__ __launch_bounds__(256) void embedding_bwd_kernel(const long* __restrict__ tok,
                                                            const float* __restrict__ g,
                                                            float* __restrict__ dw, long rows, int E,
                                                            long padding_idx, long vocab) {
  const long total = rows * E;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long r = i / E;
    const long t = tok[r];
    if (t != padding_idx && t >= 0 && t < vocab) atomicAdd(dw + t * E + (i - r * E), g[i]);
  }
}

extern "C" int vlnce_embedding_bwd(const long* tokens, const float* grad_rows, float* grad_weight,
                                   long rows, int E, long padding_idx, long vocab,
                                   vlnce_stream_t stream) {
  VLNCE_CHECK_ARG(tokens && grad_rows && grad_weight && rows > 0 && E > 0 && vocab > 0,
                  "embedding_bwd: bad argument");
  hipLaunchKernelGGL(embedding_bwd_kernel, dim3(grid_for(rows * E)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), tokens, grad_rows, grad_weight, rows, E,
                     padding_idx, vocab);
  VLNCE_CHECK_LAUNCH("embedding_bwd");
  return 0;
}

extern "C" int vlnce_select_rows(const uint8_t* mask, const float* a, const float* b, float* out,
                                 int B, int H, vlnce_stream_t stream) {
  VLNCE_CHECK_ARG(mask && out && B > 0 && H > 0, "select_rows: bad argument");
  hipLaunchKernelGGL(select_rows_kernel, dim3(grid_for((long)B * H)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), mask, a, b, out, B, H);
  VLNCE_CHECK_LAUNCH("select_rows");
  return 0;
}

extern "C" int vlnce_act_bwd(const float* dy, const float* y, float* dz, long n, int act,
                             vlnce_stream_t stream) {
  VLNCE_CHECK_ARG(dy && y && dz && n > 0, "act_bwd: bad argument");
  hipLaunchKernelGGL(act_bwd_kernel, dim3(grid_for(n)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), dy, y, dz, n, act);
  VLNCE_CHECK_LAUNCH("act_bwd");
  return 0;
}

extern "C" int vlnce_rnn_step_supported(int N, int H, int lstm) {
  const int G = lstm ? 4 : 3;
  return N > 0 && N <= STEP_MAXN && H % 32 == 0 &&
         ((long)(N + STEP_COLS) * (G * H + 4) + 5 * STEP_COLS * STEP_MAXN) * 4 <= 152 * 1024 &&
         ((long)(N + G * STEP_UNITS) * (H + 4) + 5 * G * STEP_UNITS * STEP_MAXN) * 4 <= 152 * 1024;
}

extern "C" int vlnce_rnn_step_fwd(int lstm, const float* gi, const float* h_prev,
                                  const float* c_prev, const uint8_t* mask, const float* w_hh,
                                  const float* b_hh, float* hp_out, float* h_out, float* aux_out,
                                  float* gates_out, int N, int H, vlnce_stream_t stream) {
  VLNCE_CHECK_ARG(gi && h_prev && w_hh && b_hh && h_out && aux_out && gates_out && (!lstm || c_prev),
                  "rnn_step_fwd: null argument");
  VLNCE_CHECK_ARG(vlnce_rnn_step_supported(N, H, lstm), "rnn_step_fwd: N=%d H=%d not supported", N, H);
  const int G = lstm ? 4 : 3;
  const int smem = ((N + G * STEP_UNITS) * (H + 4) + 5 * G * STEP_UNITS * STEP_MAXN) * 4;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  static bool attr[2] = {false, false};
  auto k0 = rnn_step_fwd_kernel<false>;
  auto k1 = rnn_step_fwd_kernel<true>;
  if (!attr[lstm ? 1 : 0]) {
    (void)hipFuncSetAttribute(lstm ? reinterpret_cast<const void*>(k1) : reinterpret_cast<const void*>(k0),
                        hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024);
    attr[lstm ? 1 : 0] = true;
  }
  if (lstm)
    hipLaunchKernelGGL(k1, dim3(H / STEP_UNITS), dim3(256), smem, s, gi, h_prev, c_prev, mask, w_hh,
                       b_hh, hp_out, h_out, aux_out, gates_out, N, H);
  else
    hipLaunchKernelGGL(k0, dim3(H / STEP_UNITS), dim3(256), smem, s, gi, h_prev, c_prev, mask, w_hh,
                       b_hh, hp_out, h_out, aux_out, gates_out, N, H);
  VLNCE_CHECK_LAUNCH("rnn_step_fwd");
  return 0;
}

extern "C" int vlnce_rnn_step_bwd(int lstm, const float* dout, float* carry, const float* dc,
                                  const float* gates, const float* aux, const float* hp,
                                  const float* c_prev, const uint8_t* mask, const float* w_hh_t,
                                  float* dgi, float* dgh, float* acc0, float* dc_prev, int N, int H,
                                  vlnce_stream_t stream) {
  VLNCE_CHECK_ARG(dout && carry && gates && aux && w_hh_t && dgi && dgh && acc0,
                  "rnn_step_bwd: null argument");
  VLNCE_CHECK_ARG(!lstm || (dc && c_prev && dc_prev), "rnn_step_bwd: LSTM needs dc / c_prev / dc_prev");
  VLNCE_CHECK_ARG(lstm || hp, "rnn_step_bwd: GRU needs the masked previous state");
  VLNCE_CHECK_ARG(vlnce_rnn_step_supported(N, H, lstm), "rnn_step_bwd: N=%d H=%d not supported", N, H);
  const int G = lstm ? 4 : 3, GH = G * H;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (lstm)
    hipLaunchKernelGGL(rnn_step_bwd_gates_kernel<true>, dim3(grid_for((long)N * H)), dim3(256), 0, s,
                       dout, carry, dc, gates, aux, hp, c_prev, mask, dgi, dgh, acc0, dc_prev, N, H);
  else
    hipLaunchKernelGGL(rnn_step_bwd_gates_kernel<false>, dim3(grid_for((long)N * H)), dim3(256), 0, s,
                       dout, carry, dc, gates, aux, hp, c_prev, mask, dgi, dgh, acc0, dc_prev, N, H);
  VLNCE_CHECK_LAUNCH("rnn_step_bwd (gates)");
  const int smem = ((N + STEP_COLS) * (GH + 4) + 5 * STEP_COLS * STEP_MAXN) * 4;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(rnn_step_bwd_carry_kernel),
                        hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024);
    attr = true;
  }
  // the LSTM's recurrent pre-activation gradient is dgi itself (dgh aliases it at the call site)
  hipLaunchKernelGGL(rnn_step_bwd_carry_kernel, dim3(ceil_div(H, STEP_COLS)), dim3(256), smem, s,
                     lstm ? dgi : dgh, w_hh_t, acc0, mask, carry, N, H, GH);
  VLNCE_CHECK_LAUNCH("rnn_step_bwd (carry)");
  return 0;
}
