// Skinny linear layers of the policy tail: y = act(x W^T + b) over M <= 128 rows (M = num_envs),
// forward in ONE launch, backward (dx, dW, db, activation derivative) in ONE launch.
//
// Reference: the nn.Linear / nn.GRU projections of CMANet / Seq2SeqNet / WaypointPredictionNet at
// one row per environment (cma_policy.py:103-131,140-177; seq2seq_policy.py:109-121;
// waypoint_predictors.py:76-180) and their autograd.
//
// Why: at 64 rows a layer is 0.05-0.4 GFLOP.  Through the general GEMM it was split over K to get
// enough workgroups, i.e. zero-fill + split-K GEMM (atomics) + bias/activation pass forward and
// activation-backward + zero-fill + dx GEMM + dW GEMM + column sum backward: 3 and 5 launches of
// 4-8 us each.  The tail's graphs replay at the rate of their node count (~4 us per dependent
// node, profiles/r05_d_*), so nine such layers were ~70 of the ~200 nodes of a step's tail.
//
// Arithmetic: v_mfma_f32_16x16x4_f32 (exact fp32 products, fp32 accumulate), reduction split over
// the 16 (8) waves of a workgroup and added through LDS: no atomics, no pre-zeroed outputs, results
// independent of the launch geometry.
//
// Forward, workgroup = 16 output columns x all rows, 16 waves splitting K:  lane (l15, g) of a wave loads 16 bytes of
// its x row (rows l15 + 16 mb) and of its weight row (n0 + l15) at k = k0 + 4 g .. + 3; MFMA c
// multiplies the k-slots {k0 + 4 g' + c : g' = 0..3}, four MFMAs cover 16 k-values.
// Backward: three kinds of workgroup in one grid --
//   dx  [M, K] = dz W          16 columns of K per workgroup, reduction over N split over the waves
//   dW  [N, K] = dz^T x        one 16-row strip of N per workgroup, each wave 16x16 tiles along K,
//                              reduction over the M rows (4 MFMAs per 16 rows)
//   db  [N]    = colsum dz     by the dW workgroups that own a strip's first tiles
// with dz = dy * act'(y) evaluated where dy is loaded.
#include "common.h"

namespace {

struct LinRows {
  const float* x;   // [M, K], row stride ldx
  const float* w;   // [N, K], row stride ldw
  const float* bias;
  float* y;         // fwd out [M, N] (row stride ldy) / bwd: the activation OUTPUT (read)
  const float* dy;  // bwd [M, N], row stride lddy
  float* dx;        // bwd [M, K] contiguous, or null
  float* dw;        // bwd [N, K] contiguous, or null
  float* db;        // bwd [N], or null
  int M, N, K, ldx, ldw, ldy, lddy, act;
  int dx_groups;    // bwd: workgroups [0, dx_groups) compute dx, the rest dW / db
  int dw_tiles_per_wg;
};

__device__ __forceinline__ f32x4 load4(const float* p, bool ok) {
  return ok ? *reinterpret_cast<const f32x4*>(p) : f32x4{0.f, 0.f, 0.f, 0.f};
}

__device__ __forceinline__ float act_grad(float g, float v, int act) {
  if (act == VLNCE_ACT_RELU) return v > 0.f ? g : 0.f;
  if (act == VLNCE_ACT_SIGMOID) return g * v * (1.f - v);
  if (act == VLNCE_ACT_TANH) return g * (1.f - v * v);
  return g;
}

// ------------------------------------------------------------------------------ forward
// Workgroup = 16 rows x 16 output columns (an fp32 MFMA tile is 512 FLOP per 32 cycles of ONE SIMD:
// all rows of a strip in one workgroup left the launch bound by the matrix pipe of N / 16 CUs);
// NW waves split the reduction, and a wave requests the operands of U steps of 16 k-values at once
// (a step is 4 MFMAs, ~0.05 us: with one step in flight the loop ran at the L2 latency).
template <int NW, int U>
__global__ __launch_bounds__(NW * 64) void linear_rows_fwd_kernel(LinRows p) {
  __shared__ float red[NW][256];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, g = lane >> 4;
  const int n0 = blockIdx.x * 16, m0 = blockIdx.y * 16;
  const int K = p.K;
  const int k16 = (K + 15) / 16;
  const int per = (k16 + NW - 1) / NW;
  const int kbeg = wave * per * 16;
  const int kend = min(K, kbeg + per * 16);
  const bool n_ok = n0 + l15 < p.N, m_ok = m0 + l15 < p.M;
  const float* wrow = p.w + (long)(n0 + l15) * p.ldw + 4 * g;
  const float* xrow = p.x + (long)(m0 + l15) * p.ldx + 4 * g;
  f32x4 acc0 = f32x4{0.f, 0.f, 0.f, 0.f}, acc1 = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int k0 = kbeg; k0 < kend; k0 += 16 * U) {
    f32x4 b4[U], a4[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int kk = k0 + 16 * u;
      const bool in = kk + 4 * g < kend;
      b4[u] = load4(wrow + kk, n_ok && in);
      a4[u] = load4(xrow + kk, m_ok && in);
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int c = 0; c < 4; c += 2) {   // two accumulator chains (dependent latency 40 cycles)
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[u][c], b4[u][c], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[u][c + 1], b4[u][c + 1], acc1, 0, 0, 0);
      }
  }
  // D: lane holds rows 4 g + r (r = 0..3) of column l15
#pragma unroll
  for (int r = 0; r < 4; ++r) red[wave][(4 * g + r) * 16 + l15] = acc0[r] + acc1[r];
  __syncthreads();
  if (tid < 256) {
    const int row = m0 + (tid >> 4), col = n0 + (tid & 15);
    if (row < p.M && col < p.N) {
      float v = 0.f;
#pragma unroll
      for (int wv = 0; wv < NW; ++wv) v += red[wv][tid];
      if (p.bias) v += p.bias[col];
      p.y[(long)row * p.ldy + col] = apply_act(v, p.act);
    }
  }
}

// ------------------------------------------------------------------------------ backward
// MB = 16-row blocks of the batch (the dW workgroups reduce over all of them; a dx workgroup owns one)
template <int MB, int NW, int U>
__global__ __launch_bounds__(NW * 64) void linear_rows_bwd_kernel(LinRows p) {
  __shared__ float red[NW][256];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, g = lane >> 4;
  const int M = p.M, N = p.N, K = p.K;
  if ((int)blockIdx.x < p.dx_groups) {
    // ---- dx[m0 .. m0+16, kc0 .. kc0+16) = dz W: A = dz rows (16 bytes along N), B[k = n][j] = W[n][kc0 + j]
    const int kc0 = ((int)blockIdx.x / MB) * 16, m0 = ((int)blockIdx.x % MB) * 16;
    const int n16 = (N + 15) / 16;
    const int per = (n16 + NW - 1) / NW;
    const int nbeg = wave * per * 16;
    const int nend = min(N, nbeg + per * 16);
    const bool k_ok = kc0 + l15 < K, m_ok = m0 + l15 < M;
    const float* dyrow = p.dy + (long)(m0 + l15) * p.lddy;
    const float* yrow = p.y + (long)(m0 + l15) * p.ldy;
    f32x4 acc0 = f32x4{0.f, 0.f, 0.f, 0.f}, acc1 = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int nn = nbeg; nn < nend; nn += 16 * U) {
      float b[U][4];
      f32x4 a[U], yv[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int nq = nn + 16 * u + 4 * g;
        const bool in = nq < nend;   // (N % 4 == 0: a quad is entirely inside or outside)
#pragma unroll
        for (int c = 0; c < 4; ++c) b[u][c] = (in && k_ok) ? p.w[(long)(nq + c) * p.ldw + kc0 + l15] : 0.f;
        a[u] = load4(dyrow + nq, in && m_ok);
        if (p.act != 0) yv[u] = load4(yrow + nq, in && m_ok);
      }
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int c = 0; c < 4; c += 2) {
          const float a0 = p.act != 0 ? act_grad(a[u][c], yv[u][c], p.act) : a[u][c];
          const float a1 = p.act != 0 ? act_grad(a[u][c + 1], yv[u][c + 1], p.act) : a[u][c + 1];
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b[u][c], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b[u][c + 1], acc1, 0, 0, 0);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave][(4 * g + r) * 16 + l15] = acc0[r] + acc1[r];
    __syncthreads();
    if (tid < 256) {
      const int row = m0 + (tid >> 4), col = kc0 + (tid & 15);
      if (row < M && col < K) {
        float v = 0.f;
#pragma unroll
        for (int wv = 0; wv < NW; ++wv) v += red[wv][tid];
        p.dx[(long)row * K + col] = v;
      }
    }
    return;
  }
  // ---- dW[n0 .. n0+16, :] = dz^T x and db: A[i = n][k = m] = dz[m][n0 + i], B[k = m][j] = x[m][kc0 + j]
  // one 16 x 16 tile per wave and round; the x values of a tile are requested before the dz values
  // are turned into the A operand, both before the MB*4 MFMAs
  const int k16 = (K + 15) / 16;
  const int groups = (k16 + p.dw_tiles_per_wg - 1) / p.dw_tiles_per_wg;   // workgroups per strip
  const int wg = blockIdx.x - p.dx_groups;
  const int n0 = (wg / groups) * 16;
  const int t_beg = (wg % groups) * p.dw_tiles_per_wg;
  const int t_end = min(k16, t_beg + p.dw_tiles_per_wg);
  const bool n_ok = n0 + l15 < N;
  const bool has_tile = p.dw && t_beg + wave < t_end;
  float b[MB][4];
  {
    const int kc0 = (t_beg + wave) * 16;
    const bool k_ok = has_tile && kc0 + l15 < K;
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int m = mb * 16 + 4 * g + c;
        b[mb][c] = (k_ok && m < M) ? p.x[(long)m * p.ldx + kc0 + l15] : 0.f;
      }
  }
  float a[MB][4];
  float colsum = 0.f;
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int m = mb * 16 + 4 * g + c;
      const bool ok = n_ok && m < M;
      float v = ok ? p.dy[(long)m * p.lddy + n0 + l15] : 0.f;
      if (p.act != 0 && ok) v = act_grad(v, p.y[(long)m * p.ldy + n0 + l15], p.act);
      a[mb][c] = v;
      colsum += v;
    }
  if (p.db && t_beg == 0 && wave == 0) {
    colsum += __shfl_xor(colsum, 16, 64);
    colsum += __shfl_xor(colsum, 32, 64);
    if (g == 0 && n_ok) p.db[n0 + l15] = colsum;
  }
  if (!p.dw) return;
  for (int t = t_beg + wave; t < t_end; t += NW) {
    const int kc0 = t * 16;
    const bool k_ok = kc0 + l15 < K;
    // next round's x values (if any) while this tile's MFMAs run
    float bn[MB][4];
    const int kn0 = (t + NW) * 16;
    const bool nxt = t + NW < t_end && kn0 + l15 < K;
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int m = mb * 16 + 4 * g + c;
        bn[mb][c] = (nxt && m < M) ? p.x[(long)m * p.ldx + kn0 + l15] : 0.f;
      }
    f32x4 acc0 = f32x4{0.f, 0.f, 0.f, 0.f}, acc1 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int c = 0; c < 4; c += 2) {   // two accumulator chains (dependent latency 40 cycles)
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mb][c], b[mb][c], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mb][c + 1], b[mb][c + 1], acc1, 0, 0, 0);
      }
    // D: rows n0 + 4 g + r, column kc0 + l15
    if (k_ok)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n0 + 4 * g + r;
        if (n < N) p.dw[(long)n * K + kc0 + l15] = acc0[r] + acc1[r];
      }
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int c = 0; c < 4; ++c) b[mb][c] = bn[mb][c];
  }
}

int launch_rows_fwd(const LinRows& p, hipStream_t s) {
  const dim3 grid(ceil_div(p.N, 16), ceil_div(p.M, 16));
  // long reductions: 16 waves x 4 steps of 16 k-values in flight; short ones: fewer, smaller waves' worth
  if (p.K > 1024) hipLaunchKernelGGL((linear_rows_fwd_kernel<16, 4>), grid, dim3(1024), 0, s, p);
  else if (p.K > 256) hipLaunchKernelGGL((linear_rows_fwd_kernel<8, 4>), grid, dim3(512), 0, s, p);
  else hipLaunchKernelGGL((linear_rows_fwd_kernel<4, 4>), grid, dim3(256), 0, s, p);
  return 0;
}

template <int MB>
int launch_rows_bwd(const LinRows& p, unsigned grid, hipStream_t s) {
  if constexpr (MB <= 4)
    hipLaunchKernelGGL((linear_rows_bwd_kernel<MB, 16, 2>), dim3(grid), dim3(1024), 0, s, p);
  else
    hipLaunchKernelGGL((linear_rows_bwd_kernel<MB, 8, 2>), dim3(grid), dim3(512), 0, s, p);
  return 0;
}

bool aligned16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }

}  // namespace

extern "C" int vlnce_linear_rows_supported(int M, int N, int K) {
  return M >= 1 && M <= 128 && N >= 1 && K >= 4 && K % 4 == 0 && N % 4 == 0;
}

extern "C" int vlnce_linear_rows_fwd(const float* x, int ldx, const float* w, int ldw,
                                     const float* bias, int act, float* y, int ldy, int M, int N,
                                     int K, vlnce_stream_t stream) {
  VLNCE_CHECK_ARG(x && w && y, "linear_rows_fwd: null argument");
  VLNCE_CHECK_ARG(vlnce_linear_rows_supported(M, N, K), "linear_rows_fwd: M=%d N=%d K=%d not supported", M, N, K);
  VLNCE_CHECK_ARG(ldx >= K && ldw >= K && ldy >= N && ldx % 4 == 0 && ldw % 4 == 0 && aligned16(x) && aligned16(w),
                  "linear_rows_fwd: rows of x and w must be 16-byte aligned");
  LinRows p{};
  p.x = x; p.w = w; p.bias = bias; p.y = y;
  p.M = M; p.N = N; p.K = K; p.ldx = ldx; p.ldw = ldw; p.ldy = ldy; p.act = act;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  launch_rows_fwd(p, s);
  VLNCE_CHECK_LAUNCH("linear_rows_fwd");
  return 0;
}

// dy [M,N] (row stride lddy); y = the forward's activation output (read only when act != 0);
// dx [M,K], dw [N,K], db [N]: contiguous, each may be NULL.
extern "C" int vlnce_linear_rows_bwd(const float* x, int ldx, const float* w, int ldw,
                                     const float* dy, int lddy, const float* y, int ldy, int act,
                                     float* dx, float* dw, float* db, int M, int N, int K,
                                     vlnce_stream_t stream) {
  VLNCE_CHECK_ARG(x && w && dy, "linear_rows_bwd: null argument");
  VLNCE_CHECK_ARG(act == 0 || y, "linear_rows_bwd: the activation output is needed for its derivative");
  VLNCE_CHECK_ARG(vlnce_linear_rows_supported(M, N, K), "linear_rows_bwd: M=%d N=%d K=%d not supported", M, N, K);
  VLNCE_CHECK_ARG(lddy >= N && lddy % 4 == 0 && aligned16(dy) && (act == 0 || (ldy % 4 == 0 && aligned16(y))),
                  "linear_rows_bwd: rows of dy (and y) must be 16-byte aligned");
  if (!dx && !dw && !db) return 0;
  LinRows p{};
  p.x = x; p.w = w; p.y = const_cast<float*>(y); p.dy = dy; p.dx = dx; p.dw = dw; p.db = db;
  p.M = M; p.N = N; p.K = K; p.ldx = ldx; p.ldw = ldw; p.ldy = ldy; p.lddy = lddy; p.act = act;
  const int k16 = ceil_div(K, 16);
  const int mb = ceil_div(M, 16);
  const int MBt = mb <= 1 ? 1 : mb <= 2 ? 2 : mb <= 4 ? 4 : 8;   // the kernel's row-block template
  p.dx_groups = dx ? k16 * MBt : 0;
  // dW: a wave computes 16 x 16 tiles along K, one per round; a workgroup takes NW (16 / 8) tiles
  // of a strip, two rounds of them where that still leaves >= 256 workgroups
  const int strips = ceil_div(N, 16);
  const int nw = M <= 64 ? 16 : 8;
  int per = 2 * nw;
  if ((long)strips * ceil_div(k16, per) < 256) per = nw;
  p.dw_tiles_per_wg = per;
  const int dw_groups = (dw || db) ? strips * ceil_div(k16, dw ? per : k16) : 0;
  if (!dw) p.dw_tiles_per_wg = k16;               // db only: one workgroup per strip
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const unsigned grid = (unsigned)(p.dx_groups + dw_groups);
  if (MBt == 1) launch_rows_bwd<1>(p, grid, s);
  else if (MBt == 2) launch_rows_bwd<2>(p, grid, s);
  else if (MBt == 4) launch_rows_bwd<4>(p, grid, s);
  else launch_rows_bwd<8>(p, grid, s);
  VLNCE_CHECK_LAUNCH("linear_rows_bwd");
  return 0;
}
