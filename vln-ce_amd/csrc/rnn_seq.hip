// Packed-sequence LSTM / GRU over the instruction (instruction_encoder.py:27-32,
// 80-94): the WHOLE time loop (and its BPTT) runs inside one launch.
//
// One workgroup per (direction, 16-sample tile); 4 waves, each owning H/4 hidden
// units for every gate, so the gate math is wave-local.  Per step the wave
// computes h_{t-1} W_hh^T for its units with v_mfma_f32_16x16x4_f32 (exact fp32):
//   A = h tile [16 x H] in LDS (double-buffered, one barrier per step),
//   B = W_hh fragments, loaded ONCE and kept in registers for all steps (256 VGPRs
//       of the 512-entry unified file; one wave per SIMD).
// The input projection x W_ih^T (+b_ih) for all steps is a single big MFMA GEMM
// done beforehand (time-major [L,B,G*H]).  Packed semantics: steps >= len[b]
// leave the state untouched and emit zeros; the reverse direction walks
// t = len[b]-1 .. 0.  The backward kernel mirrors this with the transposed
// recurrent weights and writes the pre-activation gate gradients for the big
// dW / dX GEMMs that follow.
#include "common.h"

namespace {

struct RnnSeqParams {
  const float* gi[2];     // [L,B,G*H] per direction (x W_ih^T + b_ih)
  const float* w_hh[2];   // fwd: [G*H,H]; bwd: transposed [H,G*H]
  const float* b_hh[2];   // [G*H]
  float* out[2];          // [L,B,H] time-major hidden outputs (pre-zeroed)
  float* h_final[2];      // [B,H]
  float* gates[2];        // [L,B,G*H] activated gates (saved)
  float* aux[2];          // LSTM: c_t [L,B,H]; GRU: hn_t = W_hn h + b_hn [L,B,H]
  // backward only
  const float* dout[2];     // [L,B,H] or null
  const float* dh_final[2]; // [B,H] or null
  float* dgi[2];            // [L,B,G*H] grads wrt gi (pre-zeroed)
  float* dgh[2];            // GRU only: [L,B,G*H] grads wrt (h W_hh^T + b_hh) (pre-zeroed)
  const int* lengths;       // [B]
  int B, L;
};

__device__ __forceinline__ float sigm(float x) { return 1.f / (1.f + expf(-x)); }

template <int KIND, int NTW>
__global__ __launch_bounds__(256) void rnn_seq_fwd_kernel(RnnSeqParams p) {
  constexpr int G = KIND == 0 ? 4 : 3;
  constexpr int H = NTW * 64;
  constexpr int KG = H / 16;
  constexpr int LDH = H + 4;
  __shared__ __attribute__((aligned(16))) float h_lds[2][16][LDH];
  const int d = blockIdx.y;
  const int b0 = blockIdx.x * 16;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, quad = lane >> 4;
  const bool reverse = d == 1;
  const float* __restrict__ gi = p.gi[d];
  const float* __restrict__ W = p.w_hh[d];
  const int B = p.B, L = p.L;

  int len[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int b = b0 + quad * 4 + r;
    len[r] = b < B ? p.lengths[b] : 0;
  }
  // recurrent weights -> registers (B operand: lane holds W[n = unit][k = 16g + 4*quad + e])
  f32x4 wf[G][NTW][KG];
  float bias[G][NTW];
#pragma unroll
  for (int gt = 0; gt < G; ++gt)
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
      const int n = gt * H + wave * (H / 4) + nt * 16 + l15;
      bias[gt][nt] = p.b_hh[d][n];
#pragma unroll
      for (int g = 0; g < KG; ++g)
        wf[gt][nt][g] = *reinterpret_cast<const f32x4*>(W + (long)n * H + 16 * g + 4 * quad);
    }
  float hreg[NTW][4], creg[NTW][4];
#pragma unroll
  for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
    for (int r = 0; r < 4; ++r) hreg[nt][r] = creg[nt][r] = 0.f;
  for (int i = tid; i < 16 * LDH; i += 256) (&h_lds[0][0][0])[i] = 0.f;
  __syncthreads();

  for (int s = 0; s < L; ++s) {
    const int cur = s & 1;
    f32x4 acc[G][NTW];
    float xn[NTW][4];  // GRU: input part of the n gate
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const bool active = s < len[r];
      const int tt = reverse ? len[r] - 1 - s : s;
      const int b = b0 + quad * 4 + r;
      const float* row = gi + ((long)tt * B + b) * (G * H);
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt) {
        const int u = wave * (H / 4) + nt * 16 + l15;
#pragma unroll
        for (int gt = 0; gt < G; ++gt) {
          const float x = active ? row[gt * H + u] : 0.f;
          if (KIND == 1 && gt == 2) {
            xn[nt][r] = x;
            acc[gt][nt][r] = bias[gt][nt];
          } else {
            acc[gt][nt][r] = x + bias[gt][nt];
          }
        }
      }
    }
#pragma unroll
    for (int g = 0; g < KG; ++g) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(&h_lds[cur][l15][16 * g + 4 * quad]);
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int gt = 0; gt < G; ++gt)
#pragma unroll
          for (int nt = 0; nt < NTW; ++nt)
            acc[gt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], wf[gt][nt][g][e], acc[gt][nt],
                                                               0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const bool active = s < len[r];
      const int tt = reverse ? len[r] - 1 - s : s;
      const int b = b0 + quad * 4 + r;
      const int row_i = quad * 4 + r;
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt) {
        const int u = wave * (H / 4) + nt * 16 + l15;
        float hnew;
        if (KIND == 0) {
          const float ig = sigm(acc[0][nt][r]), fg = sigm(acc[1][nt][r]);
          const float gg = tanhf(acc[2][nt][r]), og = sigm(acc[3][nt][r]);
          const float cn = fg * creg[nt][r] + ig * gg;
          hnew = og * tanhf(cn);
          if (active) {
            creg[nt][r] = cn;
            if (p.gates[d]) {
              float* gs = p.gates[d] + ((long)tt * B + b) * (G * H);
              gs[u] = ig;
              gs[H + u] = fg;
              gs[2 * H + u] = gg;
              gs[3 * H + u] = og;
              p.aux[d][((long)tt * B + b) * H + u] = cn;
            }
          }
        } else {
          const float rg = sigm(acc[0][nt][r]), zg = sigm(acc[1][nt][r]);
          const float hn = acc[2][nt][r];
          const float ng = tanhf(xn[nt][r] + rg * hn);
          hnew = (1.f - zg) * ng + zg * hreg[nt][r];
          if (active && p.gates[d]) {
            float* gs = p.gates[d] + ((long)tt * B + b) * (G * H);
            gs[u] = rg;
            gs[H + u] = zg;
            gs[2 * H + u] = ng;
            p.aux[d][((long)tt * B + b) * H + u] = hn;
          }
        }
        if (active) {
          hreg[nt][r] = hnew;
          p.out[d][((long)tt * B + b) * H + u] = hnew;
        }
        h_lds[cur ^ 1][row_i][u] = hreg[nt][r];
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int b = b0 + quad * 4 + r;
    if (b < B)
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt)
        p.h_final[d][(long)b * H + wave * (H / 4) + nt * 16 + l15] = hreg[nt][r];
  }
}

template <int KIND, int NTW>
__global__ __launch_bounds__(256) void rnn_seq_bwd_kernel(RnnSeqParams p) {
  constexpr int G = KIND == 0 ? 4 : 3;
  constexpr int H = NTW * 64;
  constexpr int GH = G * H;
  constexpr int KG = GH / 16;
  constexpr int LDG = GH + 4;
  __shared__ __attribute__((aligned(16))) float dg_lds[16][LDG];
  const int d = blockIdx.y;
  const int b0 = blockIdx.x * 16;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, quad = lane >> 4;
  const bool reverse = d == 1;
  const float* __restrict__ WT = p.w_hh[d];  // [H, G*H]: WT[n][k] = W_hh[k][n]
  const int B = p.B, L = p.L;

  int len[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int b = b0 + quad * 4 + r;
    len[r] = b < B ? p.lengths[b] : 0;
  }
  f32x4 wt[NTW][KG];
#pragma unroll
  for (int nt = 0; nt < NTW; ++nt) {
    const int n = wave * (H / 4) + nt * 16 + l15;
#pragma unroll
    for (int g = 0; g < KG; ++g)
      wt[nt][g] = *reinterpret_cast<const f32x4*>(WT + (long)n * GH + 16 * g + 4 * quad);
  }
  float dh[NTW][4], dc[NTW][4];
#pragma unroll
  for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int b = b0 + quad * 4 + r;
      const int u = wave * (H / 4) + nt * 16 + l15;
      dh[nt][r] = (p.dh_final[d] && b < B) ? p.dh_final[d][(long)b * H + u] : 0.f;
      dc[nt][r] = 0.f;
    }

  for (int s = L - 1; s >= 0; --s) {
    float keep_z[NTW][4];  // GRU: dh * z carried straight to h_prev
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const bool active = s < len[r];
      const int tt = reverse ? len[r] - 1 - s : s;
      const int tp = reverse ? tt + 1 : tt - 1;  // time index of the previous step's state
      const bool has_prev = s > 0;
      const int b = b0 + quad * 4 + r;
      const int row_i = quad * 4 + r;
      const long base = ((long)tt * B + b);
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt) {
        const int u = wave * (H / 4) + nt * 16 + l15;
        float dpre[G];
#pragma unroll
        for (int gt = 0; gt < G; ++gt) dpre[gt] = 0.f;
        float dgh_n = 0.f;
        keep_z[nt][r] = 0.f;
        if (active) {
          const float* gs = p.gates[d] + base * GH;
          float dht = dh[nt][r];
          if (p.dout[d]) dht += p.dout[d][base * H + u];
          if (KIND == 0) {
            const float ig = gs[u], fg = gs[H + u], gg = gs[2 * H + u], og = gs[3 * H + u];
            const float c = p.aux[d][base * H + u];
            const float cp = has_prev ? p.aux[d][((long)tp * B + b) * H + u] : 0.f;
            const float tc = tanhf(c);
            const float dct = dc[nt][r] + dht * og * (1.f - tc * tc);
            dpre[0] = dct * gg * ig * (1.f - ig);
            dpre[1] = dct * cp * fg * (1.f - fg);
            dpre[2] = dct * ig * (1.f - gg * gg);
            dpre[3] = dht * tc * og * (1.f - og);
            dc[nt][r] = dct * fg;
          } else {
            const float rg = gs[u], zg = gs[H + u], ng = gs[2 * H + u];
            const float hn = p.aux[d][base * H + u];
            const float hp = has_prev ? p.out[d][((long)tp * B + b) * H + u] : 0.f;
            const float dn = dht * (1.f - zg);
            const float dz = dht * (hp - ng);
            const float dnp = dn * (1.f - ng * ng);
            dpre[0] = dnp * hn * rg * (1.f - rg);
            dpre[1] = dz * zg * (1.f - zg);
            dpre[2] = dnp;
            dgh_n = dnp * rg;
            keep_z[nt][r] = dht * zg;
          }
          float* dgi = p.dgi[d] + base * GH;
#pragma unroll
          for (int gt = 0; gt < G; ++gt) dgi[gt * H + u] = dpre[gt];
          if (KIND == 1) {
            float* dgh = p.dgh[d] + base * GH;
            dgh[u] = dpre[0];
            dgh[H + u] = dpre[1];
            dgh[2 * H + u] = dgh_n;
          }
        }
        // A operand of dh_{t-1} = dgates_h * W_hh  (zeros for finished / padded rows)
#pragma unroll
        for (int gt = 0; gt < G; ++gt)
          dg_lds[row_i][gt * H + u] = (KIND == 1 && gt == 2) ? dgh_n : dpre[gt];
      }
    }
    __syncthreads();
    f32x4 acc[NTW];
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int g = 0; g < KG; ++g) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(&dg_lds[l15][16 * g + 4 * quad]);
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt)
          acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], wt[nt][g][e], acc[nt], 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const bool active = s < len[r];
      if (active)
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) dh[nt][r] = acc[nt][r] + keep_z[nt][r];
    }
    __syncthreads();
  }
}

template <int KIND>
int launch_fwd(const RnnSeqParams& p, int H, int dirs, hipStream_t s) {
  dim3 grid(ceil_div(p.B, 16), dirs);
  if (H == 64)
    hipLaunchKernelGGL((rnn_seq_fwd_kernel<KIND, 1>), grid, dim3(256), 0, s, p);
  else if (H == 128)
    hipLaunchKernelGGL((rnn_seq_fwd_kernel<KIND, 2>), grid, dim3(256), 0, s, p);
  else
    return 1;
  return 0;
}
template <int KIND>
int launch_bwd(const RnnSeqParams& p, int H, int dirs, hipStream_t s) {
  dim3 grid(ceil_div(p.B, 16), dirs);
  if (H == 64)
    hipLaunchKernelGGL((rnn_seq_bwd_kernel<KIND, 1>), grid, dim3(256), 0, s, p);
  else if (H == 128)
    hipLaunchKernelGGL((rnn_seq_bwd_kernel<KIND, 2>), grid, dim3(256), 0, s, p);
  else
    return 1;
  return 0;
}

}  // namespace

extern "C" int vlnce_rnn_seq_supported(int kind, int H) {
  return (kind == 0 || kind == 1) && (H == 64 || H == 128);
}

extern "C" int vlnce_rnn_seq_fwd(int kind, int dirs, const float* const* gi,
                                 const float* const* w_hh, const float* const* b_hh,
                                 const int* lengths, float* const* out, float* const* h_final,
                                 float* const* gates_save, float* const* aux_save, int B, int L,
                                 int H, vlnce_stream_t stream) {
  VLNCE_CHECK_ARG(gi && w_hh && b_hh && lengths && out && h_final, "rnn_seq_fwd: null argument");
  VLNCE_CHECK_ARG(dirs == 1 || dirs == 2, "rnn_seq_fwd: dirs must be 1 or 2");
  VLNCE_CHECK_ARG(vlnce_rnn_seq_supported(kind, H), "rnn_seq_fwd: unsupported kind/H (%d,%d)", kind, H);
  VLNCE_CHECK_ARG(B > 0 && L > 0, "rnn_seq_fwd: bad shape");
  RnnSeqParams p{};
  for (int d = 0; d < dirs; ++d) {
    p.gi[d] = gi[d];
    p.w_hh[d] = w_hh[d];
    p.b_hh[d] = b_hh[d];
    p.out[d] = out[d];
    p.h_final[d] = h_final[d];
    p.gates[d] = gates_save ? gates_save[d] : nullptr;
    p.aux[d] = aux_save ? aux_save[d] : nullptr;
    VLNCE_CHECK_ARG((p.gates[d] == nullptr) == (p.aux[d] == nullptr),
                    "rnn_seq_fwd: gates_save and aux_save come together");
  }
  p.lengths = lengths;
  p.B = B;
  p.L = L;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int rc = kind == 0 ? launch_fwd<0>(p, H, dirs, s) : launch_fwd<1>(p, H, dirs, s);
  VLNCE_CHECK_ARG(rc == 0, "rnn_seq_fwd: no kernel for H=%d", H);
  VLNCE_CHECK_LAUNCH("rnn_seq_fwd");
  return 0;
}

extern "C" int vlnce_rnn_seq_bwd(int kind, int dirs, const float* const* w_hh_t,
                                 const int* lengths, const float* const* out,
                                 const float* const* gates_save, const float* const* aux_save,
                                 const float* const* dout, const float* const* dh_final,
                                 float* const* dgi, float* const* dgh, int B, int L, int H,
                                 vlnce_stream_t stream) {
  VLNCE_CHECK_ARG(w_hh_t && lengths && out && gates_save && aux_save && dgi,
                  "rnn_seq_bwd: null argument");
  VLNCE_CHECK_ARG(dirs == 1 || dirs == 2, "rnn_seq_bwd: dirs must be 1 or 2");
  VLNCE_CHECK_ARG(vlnce_rnn_seq_supported(kind, H), "rnn_seq_bwd: unsupported kind/H (%d,%d)", kind, H);
  VLNCE_CHECK_ARG(kind == 0 || dgh, "rnn_seq_bwd: GRU needs dgh");
  RnnSeqParams p{};
  for (int d = 0; d < dirs; ++d) {
    p.w_hh[d] = w_hh_t[d];
    p.out[d] = const_cast<float*>(out[d]);
    p.gates[d] = const_cast<float*>(gates_save[d]);
    p.aux[d] = const_cast<float*>(aux_save[d]);
    p.dout[d] = dout ? dout[d] : nullptr;
    p.dh_final[d] = dh_final ? dh_final[d] : nullptr;
    p.dgi[d] = dgi[d];
    p.dgh[d] = dgh ? dgh[d] : nullptr;
  }
  p.lengths = lengths;
  p.B = B;
  p.L = L;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int rc = kind == 0 ? launch_bwd<0>(p, H, dirs, s) : launch_bwd<1>(p, H, dirs, s);
  VLNCE_CHECK_ARG(rc == 0, "rnn_seq_bwd: no kernel for H=%d", H);
  VLNCE_CHECK_LAUNCH("rnn_seq_bwd");
  return 0;
}
