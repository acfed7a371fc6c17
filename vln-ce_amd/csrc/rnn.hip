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
