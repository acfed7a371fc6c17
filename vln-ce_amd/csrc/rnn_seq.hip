// Packed-sequence LSTM / GRU over the instruction (instruction_encoder.py:27-32,
// 80-94): the WHOLE time loop (and its BPTT) runs inside one launch.
//
// One workgroup per (direction, 16-sample tile); 4 waves (8 at H = 128), each owning H/NW
// hidden units for every gate, so the gate math is wave-local.  Per step the wave computes
// h_{t-1} W_hh^T for its units on the bf16 matrix pipe as six plane products of exactly split
// fp32 operands (v_mfma_f32_16x16x32_bf16, fp32 accumulate; see split_planes):
//   A = h tile [16 x H] as three bf16 planes in LDS (double-buffered, one barrier per step),
//   B = W_hh planes, split ONCE and kept in registers for all steps.
// What a step reads from global memory (input projection; in the backward pass the saved gates,
// states and output gradients) is fetched one step ahead.
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
  // round 5 (vlnce_rnn_seq_fwd2 / _bwd2); all zero for the first-generation entry points
  float* seq[2];            // fwd: optional second copy of the outputs in the CONSUMER's layout,
                            // element (t, b, u) of direction d at seq[d][t * seq_st + b * seq_sb + u]
  long seq_st, seq_sb;
  int self_zero;            // 1: the launch itself writes the zeros past each row's length
                            //    (out / seq in the forward, dgi / dgh in the backward)
  int w_plain;              // bwd: w_hh[d] is W_hh [G*H, H] as the module stores it, not its transpose
};

// dout_tm[d][t][b][u] = dseq[t * st + b * sb + d * H + u]: the output gradient from the consumer's
// row layout into the time-major layout BPTT reads.  (Reading it strided inside rnn_seq_bwd_kernel
// costs four loop-invariant offset registers, which spills the LSTM / H = 128 instance: 254 VGPRs.)
__global__ __launch_bounds__(256) void dseq_to_time_major_kernel(const float* __restrict__ dseq, long st,
                                                                 long sb, float* __restrict__ dst, int dirs,
                                                                 int L, int B, int H) {
  const int h4 = H / 4;
  const long n = (long)dirs * L * B * h4;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int u4 = (int)(i % h4);
    long r = i / h4;
    const int b = (int)(r % B);
    r /= B;
    const int t = (int)(r % L);
    const int d = (int)(r / L);
    const float* src = dseq + (long)t * st + (long)b * sb + d * H + 4 * u4;
    f32x4 v = {src[0], src[1], src[2], src[3]};
    *reinterpret_cast<f32x4*>(dst + i * 4) = v;
  }
}

// zeros of the rows [len[b], L) of a [L, B, width]-addressed array for the 16 samples of a tile
template <int NT>
__device__ __forceinline__ void zero_tails(float* __restrict__ base, long st, long sb, int width,
                                           const int* __restrict__ lengths, int b0, int B, int L,
                                           int tid) {
  for (int r = 0; r < 16; ++r) {
    const int b = b0 + r;
    if (b >= B) break;
    const int lb = min(max(lengths[b], 0), L);
    const int n = (L - lb) * width;
    for (int i = tid; i < n; i += NT) {
      const int q = i / width;
      base[(long)(lb + q) * st + (long)b * sb + (i - q * width)] = 0.f;
    }
  }
}

// Gate non-linearities on the hardware exp2 / rcp (v_exp_f32, v_rcp_f32; absolute error < 3e-7,
// tests at 1e-5): libm's expf / tanhf are ~25 / ~50 VALU instructions each and a step evaluates
// five per (row, unit) -- 2/3 of the forward step's instruction stream, which the two waves of a
// SIMD execute one after the other (measured: 5.0 -> see profiles/archive/r03_f_seqbench.txt us per step).
__device__ __forceinline__ float sigm(float x) {
  return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
__device__ __forceinline__ float tanh_fast(float x) {
  return 1.f - 2.f * __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(2.8853900817779268f * x) + 1.f);
}

// fp32 recurrent product on the bf16 matrix pipe: both operands are split exactly into three
// bf16 planes (truncation: 8 + 8 + 8 mantissa bits) and multiplied as the six plane products of
// order <= 2^-16, fp32 accumulate -- the arithmetic of conv_x3_kernel (igemm.hip).  One
// v_mfma_f32_16x16x32_bf16 (16 cycles) covers 32 k-values where v_mfma_f32_16x16x4_f32
// (32 cycles) covers 4: 2.7x less matrix-pipe time per step, on the step's critical path.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4r __attribute__((ext_vector_type(4)));
struct Planes3 {
  bf16x8 p[3];
};
__device__ __forceinline__ Planes3 split_planes(f32x4 lo4, f32x4 hi4) {
  const float x[8] = {lo4[0], lo4[1], lo4[2], lo4[3], hi4[0], hi4[1], hi4[2], hi4[3]};
  u32x4r h, m, l;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float x0 = x[2 * q], x1 = x[2 * q + 1];
    const float r0 = x0 - __uint_as_float(__float_as_uint(x0) & 0xffff0000u);
    const float r1 = x1 - __uint_as_float(__float_as_uint(x1) & 0xffff0000u);
    const float s0 = r0 - __uint_as_float(__float_as_uint(r0) & 0xffff0000u);
    const float s1 = r1 - __uint_as_float(__float_as_uint(r1) & 0xffff0000u);
    h[q] = __builtin_amdgcn_perm(__float_as_uint(x1), __float_as_uint(x0), 0x07060302u);
    m[q] = __builtin_amdgcn_perm(__float_as_uint(r1), __float_as_uint(r0), 0x07060302u);
    l[q] = __builtin_amdgcn_perm(__float_as_uint(s1), __float_as_uint(s0), 0x07060302u);
  }
  Planes3 o;
  o.p[0] = __builtin_bit_cast(bf16x8, h);
  o.p[1] = __builtin_bit_cast(bf16x8, m);
  o.p[2] = __builtin_bit_cast(bf16x8, l);
  return o;
}
// one value -> its three plane words
__device__ __forceinline__ void split_scalar(float x, unsigned short (&o)[3]) {
  const unsigned xb = __float_as_uint(x);
  const float r = x - __uint_as_float(xb & 0xffff0000u);
  const unsigned rb = __float_as_uint(r);
  const float t = r - __uint_as_float(rb & 0xffff0000u);
  o[0] = (unsigned short)(xb >> 16);
  o[1] = (unsigned short)(rb >> 16);
  o[2] = (unsigned short)(__float_as_uint(t) >> 16);
}
constexpr int X3_PA[6] = {2, 0, 1, 1, 0, 0};  // plane pairs, smallest products first
constexpr int X3_PB[6] = {0, 2, 1, 0, 1, 0};

// NW waves per workgroup, each owning H / NW hidden units for every gate (NW = 8 at H = 128: two
// waves per SIMD, so that a wave's three weight planes fit its 256 registers).
// Lane (l15, quad) owns unit l15 of the wave's 16-unit block for the four rows 4 quad .. 4 quad + 3:
// consecutive lanes are consecutive units, so every 4-byte access of a wave coalesces into four
// 64-byte segments.  (Tried: the transposed product W_slice x h^T, which gives a lane four
// consecutive units of ONE row and 16-byte accesses -- a quarter of the memory instructions, but
// 64 separate 16-byte requests per instruction: 4.4 vs 3.4 us per forward step at 64 rows, better
// only for nearly empty tiles; profiles/archive/r03_g_seqbench_transposed_assignment.txt.)
template <int KIND, int H, int NW>
__global__ __launch_bounds__(NW * 64) void rnn_seq_fwd_kernel(RnnSeqParams p) {
  constexpr int G = KIND == 0 ? 4 : 3;
  constexpr int NTW = H / (NW * 16);
  static_assert(NTW >= 1, "units per wave");
  constexpr int KS = H / 32;     // k-steps of 32 per recurrent product
  constexpr int LDHB = H + 16;   // bf16 row pitch: conflict-free ds_read_b128 of 8 k-values
  __shared__ __attribute__((aligned(16))) unsigned short h_pl[2][3][16][LDHB];
  // The LSTM at H = 128 needs 16 weight fragments of 12 registers per wave: with everything else
  // that is past 256 VGPRs, and what the compiler spills it reloads from scratch inside the MFMA
  // loop.  The last gate's fragments of k-steps 1..3 live in LDS instead (one 16-byte slot per
  // thread and plane: conflict-free) and are read right before their k-step.
  constexpr int KL = (KIND == 0 && H == 128) ? 3 : 0;
  __shared__ __attribute__((aligned(16))) bf16x8 w_lds[KL > 0 ? KL : 1][3][NW * 64];
  const int d = blockIdx.y;
  const int b0 = blockIdx.x * 16;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, quad = lane >> 4;
  const bool reverse = d == 1;
  const float* __restrict__ gi = p.gi[d];
  const float* __restrict__ W = p.w_hh[d];
  const int B = p.B, L = p.L;
  if (p.self_zero) {  // (stores only: they drain under the weight split below)
    zero_tails<NW * 64>(p.out[d], (long)B * H, H, H, p.lengths, b0, B, L, tid);
    if (p.seq[d]) zero_tails<NW * 64>(p.seq[d], p.seq_st, p.seq_sb, H, p.lengths, b0, B, L, tid);
  }

  int len[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int b = b0 + quad * 4 + r;
    len[r] = b < B ? p.lengths[b] : 0;
  }
  // steps past the longest sequence of this tile change nothing (state kept, zeros emitted into
  // the pre-zeroed outputs): the loop ends there, not at the padded length (80 of 200 tokens
  // in the R2R batches)
  int Lt = max(max(len[0], len[1]), max(len[2], len[3]));
  Lt = max(Lt, __shfl_xor(Lt, 16, 64));
  Lt = max(Lt, __shfl_xor(Lt, 32, 64));
  Lt = min(Lt, L);
  // recurrent weights -> registers, split once (B operand: lane holds the three planes of
  // W[n = unit][k = 32 ks + 8 quad + 0..7])
  Planes3 wp[G][NTW][KS];
  float bias[G][NTW];
#pragma unroll
  for (int gt = 0; gt < G; ++gt)
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
      const int n = gt * H + wave * (H / NW) + nt * 16 + l15;
      bias[gt][nt] = p.b_hh[d][n];
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const float* w8 = W + (long)n * H + 32 * ks + 8 * quad;
        wp[gt][nt][ks] = split_planes(*reinterpret_cast<const f32x4*>(w8),
                                      *reinterpret_cast<const f32x4*>(w8 + 4));
        if (KL > 0 && gt == G - 1 && nt == 0 && ks >= KS - KL) {
#pragma unroll
          for (int q = 0; q < 3; ++q) w_lds[ks - (KS - KL)][q][tid] = wp[gt][nt][ks].p[q];
        }
      }
    }
  float hreg[NTW][4], creg[NTW][4];
#pragma unroll
  for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
    for (int r = 0; r < 4; ++r) hreg[nt][r] = creg[nt][r] = 0.f;
  for (int i = tid; i < 3 * 16 * LDHB; i += NW * 64) (&h_pl[0][0][0][0])[i] = 0;
  __syncthreads();

  // The input projection of a step is fetched ONE STEP AHEAD (gx): read at the top of the step
  // it would put a global-load round trip in front of every step's MFMAs.
  float gx[G][NTW][4];
  auto fetch_into = [&](int s, float (&dst)[G][NTW][4]) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const bool active = s < len[r];
      const int tt = reverse ? len[r] - 1 - s : s;
      const int b = b0 + quad * 4 + r;
      const float* row = gi + ((long)tt * B + b) * (G * H);
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt) {
        const int u = wave * (H / NW) + nt * 16 + l15;
#pragma unroll
        for (int gt = 0; gt < G; ++gt) dst[gt][nt][r] = active ? row[gt * H + u] : 0.f;
      }
    }
  };
  auto fetch = [&](int s) { fetch_into(s, gx); };
  fetch(0);

#ifdef RNN_SEQ_DBG_TIME
  long long f_m = 0, f_g = 0, f_b = 0;
#endif
  for (int s = 0; s < Lt; ++s) {
#ifdef RNN_SEQ_DBG_TIME
    const long long f_t0 = clock64();
#endif
    const int cur = s & 1;
    f32x4 acc[G][NTW];
    float xn[NTW][4];  // GRU: input part of the n gate
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
        for (int gt = 0; gt < G; ++gt) {
          const float x = gx[gt][nt][r];
          if (KIND == 1 && gt == 2) {
            xn[nt][r] = x;
            acc[gt][nt][r] = bias[gt][nt];
          } else {
            acc[gt][nt][r] = x + bias[gt][nt];
          }
        }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      bf16x8 a[3];
#pragma unroll
      for (int q = 0; q < 3; ++q)
        a[q] = *reinterpret_cast<const bf16x8*>(&h_pl[cur][q][l15][32 * ks + 8 * quad]);
      Planes3 wl;  // the LDS-resident fragment of this k-step, if any
      const bool from_lds = KL > 0 && ks >= KS - KL;
      if (from_lds) {
#pragma unroll
        for (int q = 0; q < 3; ++q) wl.p[q] = w_lds[KL > 0 ? ks - (KS - KL) : 0][q][tid];
      }
#pragma unroll
      for (int q = 0; q < 6; ++q)
#pragma unroll
        for (int gt = 0; gt < G; ++gt)
#pragma unroll
          for (int nt = 0; nt < NTW; ++nt) {
            const bf16x8 wb = (from_lds && gt == G - 1 && nt == 0) ? wl.p[X3_PB[q]]
                                                                   : wp[gt][nt][ks].p[X3_PB[q]];
            acc[gt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[X3_PA[q]], wb, acc[gt][nt], 0, 0, 0);
          }
    }
    // The next step's input projection is requested HERE, behind the MFMAs (it has the gate
    // phase and the barrier to land): requested at the top of the step its 16 registers were live
    // through the MFMA loop, the kernel went past 256 VGPRs and the compiler reloaded spilled
    // weight planes from scratch inside the loop.
    fetch(s + 1);  // (rows with s + 1 >= len load nothing)
#ifdef RNN_SEQ_DBG_TIME
    const long long f_t1 = clock64();
#endif
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const bool active = s < len[r];
      const int tt = reverse ? len[r] - 1 - s : s;
      const int b = b0 + quad * 4 + r;
      const int row_i = quad * 4 + r;
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt) {
        const int u = wave * (H / NW) + nt * 16 + l15;
        float hnew;
        if (KIND == 0) {
          const float ig = sigm(acc[0][nt][r]), fg = sigm(acc[1][nt][r]);
          const float gg = tanh_fast(acc[2][nt][r]), og = sigm(acc[3][nt][r]);
          const float cn = fg * creg[nt][r] + ig * gg;
          hnew = og * tanh_fast(cn);
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
          const float ng = tanh_fast(xn[nt][r] + rg * hn);
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
          if (p.seq[d]) p.seq[d][(long)tt * p.seq_st + (long)b * p.seq_sb + u] = hnew;
        }
        unsigned short hw[3];
        split_scalar(hreg[nt][r], hw);
#pragma unroll
        for (int q = 0; q < 3; ++q) h_pl[cur ^ 1][q][row_i][u] = hw[q];
      }
    }
#ifdef RNN_SEQ_DBG_TIME
    const long long f_t2 = clock64();
#endif
    __syncthreads();
#ifdef RNN_SEQ_DBG_TIME
    const long long f_t3 = clock64();
    f_m += f_t1 - f_t0;
    f_g += f_t2 - f_t1;
    f_b += f_t3 - f_t2;
#endif
  }
#ifdef RNN_SEQ_DBG_TIME
  if (blockIdx.x == 0 && blockIdx.y == 0 && (tid == 0 || tid == 64 * (NW - 1)))
    printf("rnn_seq_fwd wave %d: %d steps; per step: init+fetch+reads+MFMA %lld, gates+stores+LDS write %lld, "
           "barrier %lld cycles\n", wave, Lt, f_m / Lt, f_g / Lt, f_b / Lt);
#endif
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int b = b0 + quad * 4 + r;
    if (b < B)
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt)
        p.h_final[d][(long)b * H + wave * (H / NW) + nt * 16 + l15] = hreg[nt][r];
  }
}

template <int KIND, int H, int NW>
__global__ __launch_bounds__(NW * 64) void rnn_seq_bwd_kernel(RnnSeqParams p) {
  constexpr int G = KIND == 0 ? 4 : 3;
  constexpr int NTW = H / (NW * 16);
  constexpr int GH = G * H;
  constexpr int KS = GH / 32;    // k-steps of 32 of dh_{t-1} = dgates * W_hh
  constexpr int LDGB = GH + 16;  // bf16 row pitch (conflict-free ds_read_b128)
  static_assert(NTW >= 1, "units per wave");
  __shared__ __attribute__((aligned(16))) unsigned short dg_pl[3][16][LDGB];
  constexpr int KL = (KIND == 0 && H == 128) ? 4 : 0;  // weight fragments kept in LDS (see the forward kernel)
  __shared__ __attribute__((aligned(16))) bf16x8 w_lds[KL > 0 ? KL : 1][3][NW * 64];
  const int d = blockIdx.y;
  const int b0 = blockIdx.x * 16;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, quad = lane >> 4;
  const bool reverse = d == 1;
  const float* __restrict__ WT = p.w_hh[d];  // [H, G*H]: WT[n][k] = W_hh[k][n]  (w_plain: W_hh itself)
  const int B = p.B, L = p.L;
  if (p.self_zero) {
    zero_tails<NW * 64>(p.dgi[d], (long)B * GH, GH, GH, p.lengths, b0, B, L, tid);
    if (KIND == 1) zero_tails<NW * 64>(p.dgh[d], (long)B * GH, GH, GH, p.lengths, b0, B, L, tid);
  }

  int len[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int b = b0 + quad * 4 + r;
    len[r] = b < B ? p.lengths[b] : 0;
  }
  int Lt = max(max(len[0], len[1]), max(len[2], len[3]));  // (see the forward kernel)
  Lt = max(Lt, __shfl_xor(Lt, 16, 64));
  Lt = max(Lt, __shfl_xor(Lt, 32, 64));
  Lt = min(Lt, L);
  Planes3 wt[NTW][KS];  // the three planes of WT[n][32 ks + 8 quad + 0..7]
#pragma unroll
  for (int nt = 0; nt < NTW; ++nt) {
    const int n = wave * (H / NW) + nt * 16 + l15;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      f32x4 lo4, hi4;
      if (p.w_plain) {  // W_hh [G*H, H] as stored: WT[n][k] = W[k][n], eight strided 4-byte loads
        const float* wk = WT + (long)(32 * ks + 8 * quad) * H + n;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          lo4[i] = wk[(long)i * H];
          hi4[i] = wk[(long)(i + 4) * H];
        }
      } else {
        const float* w8 = WT + (long)n * GH + 32 * ks + 8 * quad;
        lo4 = *reinterpret_cast<const f32x4*>(w8);
        hi4 = *reinterpret_cast<const f32x4*>(w8 + 4);
      }
      wt[nt][ks] = split_planes(lo4, hi4);
      if (KL > 0 && nt == 0 && ks >= KS - KL) {
#pragma unroll
        for (int q = 0; q < 3; ++q) w_lds[ks - (KS - KL)][q][tid] = wt[nt][ks].p[q];
      }
    }
  }
  float dh[NTW][4], dc[NTW][4];
#pragma unroll
  for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int b = b0 + quad * 4 + r;
      const int u = wave * (H / NW) + nt * 16 + l15;
      dh[nt][r] = (p.dh_final[d] && b < B) ? p.dh_final[d][(long)b * H + u] : 0.f;
      dc[nt][r] = 0.f;
    }

  // What a step reads from memory (saved gates, cell / candidate state, the output gradient,
  // the previous step's state) is fetched one step ahead -- two for the previous state, which is
  // the next step's own state -- so no step starts with a global-load round trip.
  float pg[G][NTW][4], pa[NTW][4], pa_prev[NTW][4], pd[NTW][4];
  auto fetch_row = [&](int s, int r, float (&g_)[G][NTW][4], float (&a_)[NTW][4], float (&d_)[NTW][4]) {
    {
      const bool active = s >= 0 && s < len[r];
      const int tt = reverse ? len[r] - 1 - s : s;
      const int b = b0 + quad * 4 + r;
      const long base = ((long)tt * B + b);
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt) {
        const int u = wave * (H / NW) + nt * 16 + l15;
#pragma unroll
        for (int gt = 0; gt < G; ++gt) g_[gt][nt][r] = active ? p.gates[d][base * GH + gt * H + u] : 0.f;
        a_[nt][r] = active ? p.aux[d][base * H + u] : 0.f;
        d_[nt][r] = (active && p.dout[d]) ? p.dout[d][base * H + u] : 0.f;
      }
    }
  };
  // previous state of step s: LSTM c_{s-1} = aux of step s-1; GRU h_{s-1} = out of step s-1
  auto fetch = [&](int s, float (&g_)[G][NTW][4], float (&a_)[NTW][4], float (&d_)[NTW][4]) {
#pragma unroll
    for (int r = 0; r < 4; ++r) fetch_row(s, r, g_, a_, d_);
  };
  auto fetch_prev_row = [&](int s, int r, float (&a_)[NTW][4]) {
    {
      const bool active = s >= 0 && s < len[r];
      const int tt = reverse ? len[r] - 1 - s : s;
      const int b = b0 + quad * 4 + r;
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt) {
        const int u = wave * (H / NW) + nt * 16 + l15;
        const float* src = KIND == 0 ? p.aux[d] : p.out[d];
        a_[nt][r] = active ? src[((long)tt * B + b) * H + u] : 0.f;
      }
    }
  };
  auto fetch_prev = [&](int s, float (&a_)[NTW][4]) {
#pragma unroll
    for (int r = 0; r < 4; ++r) fetch_prev_row(s, r, a_);
  };
  fetch(Lt - 1, pg, pa, pd);
  fetch_prev(Lt - 2, pa_prev);

#ifdef RNN_SEQ_DBG_TIME
  long long d_a = 0, d_b1 = 0, d_m = 0, d_b2 = 0;
#endif
  for (int s = Lt - 1; s >= 0; --s) {
#ifdef RNN_SEQ_DBG_TIME
    const long long d_t0 = clock64();
#endif
    float keep_z[NTW][4];  // GRU: dh * z carried straight to h_prev
    // (fetched at the top, consumed from copies: requesting the next step's values only after
    // this step's last use of the registers -- no second set -- measured 5.2 -> 6.6 us per step:
    // the loads then have too little time to land)
    float cg[G][NTW][4], ca[NTW][4], cprev[NTW][4], cd[NTW][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt) {
#pragma unroll
        for (int gt = 0; gt < G; ++gt) cg[gt][nt][r] = pg[gt][nt][r];
        ca[nt][r] = pa[nt][r];
        cprev[nt][r] = pa_prev[nt][r];
        cd[nt][r] = pd[nt][r];
      }
    fetch(s - 1, pg, pa, pd);
    fetch_prev(s - 2, pa_prev);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const bool active = s < len[r];
      const int tt = reverse ? len[r] - 1 - s : s;
      const int b = b0 + quad * 4 + r;
      const int row_i = quad * 4 + r;
      const long base = ((long)tt * B + b);
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt) {
        const int u = wave * (H / NW) + nt * 16 + l15;
        float dpre[G];
#pragma unroll
        for (int gt = 0; gt < G; ++gt) dpre[gt] = 0.f;
        float dgh_n = 0.f;
        keep_z[nt][r] = 0.f;
        if (active) {
          const float dht = dh[nt][r] + cd[nt][r];
          if (KIND == 0) {
            const float ig = cg[0][nt][r], fg = cg[1][nt][r], gg = cg[2][nt][r], og = cg[3][nt][r];
            const float c = ca[nt][r];
            const float cp = cprev[nt][r];  // (0 at the sequence's first step)
            const float tc = tanh_fast(c);
            const float dct = dc[nt][r] + dht * og * (1.f - tc * tc);
            dpre[0] = dct * gg * ig * (1.f - ig);
            dpre[1] = dct * cp * fg * (1.f - fg);
            dpre[2] = dct * ig * (1.f - gg * gg);
            dpre[3] = dht * tc * og * (1.f - og);
            dc[nt][r] = dct * fg;
          } else {
            const float rg = cg[0][nt][r], zg = cg[1][nt][r], ng = cg[2][nt][r];
            const float hn = ca[nt][r];
            const float hp = cprev[nt][r];
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
        for (int gt = 0; gt < G; ++gt) {
          unsigned short dw[3];
          split_scalar((KIND == 1 && gt == 2) ? dgh_n : dpre[gt], dw);
#pragma unroll
          for (int q = 0; q < 3; ++q) dg_pl[q][row_i][gt * H + u] = dw[q];
        }
      }
    }
#ifdef RNN_SEQ_DBG_TIME
    const long long d_t1 = clock64();
#endif
    __syncthreads();
#ifdef RNN_SEQ_DBG_TIME
    const long long d_t2 = clock64();
#endif
    f32x4 acc[NTW];
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      bf16x8 a[3];
#pragma unroll
      for (int q = 0; q < 3; ++q)
        a[q] = *reinterpret_cast<const bf16x8*>(&dg_pl[q][l15][32 * ks + 8 * quad]);
      Planes3 wl;
      const bool from_lds = KL > 0 && ks >= KS - KL;
      if (from_lds) {
#pragma unroll
        for (int q = 0; q < 3; ++q) wl.p[q] = w_lds[KL > 0 ? ks - (KS - KL) : 0][q][tid];
      }
#pragma unroll
      for (int q = 0; q < 6; ++q)
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
          const bf16x8 wb = (from_lds && nt == 0) ? wl.p[X3_PB[q]] : wt[nt][ks].p[X3_PB[q]];
          acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[X3_PA[q]], wb, acc[nt], 0, 0, 0);
        }
    }
    // (two accumulator chains instead of one change nothing: 5.21 us per step either way)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const bool active = s < len[r];
      if (active)
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) dh[nt][r] = acc[nt][r] + keep_z[nt][r];
    }
#ifdef RNN_SEQ_DBG_TIME
    const long long d_t3 = clock64();
#endif
    __syncthreads();
#ifdef RNN_SEQ_DBG_TIME
    const long long d_t4 = clock64();
    d_a += d_t1 - d_t0;
    d_b1 += d_t2 - d_t1;
    d_m += d_t3 - d_t2;
    d_b2 += d_t4 - d_t3;
#endif
  }
#ifdef RNN_SEQ_DBG_TIME
  if (blockIdx.x == 0 && blockIdx.y == 0 && (tid == 0 || tid == 64 * (NW - 1)))
    printf("rnn_seq_bwd wave %d: %d steps; per step: gates+LDS write %lld, barrier %lld, reads+MFMA %lld, "
           "barrier %lld cycles\n", wave, Lt, d_a / Lt, d_b1 / Lt, d_m / Lt, d_b2 / Lt);
#endif
}

template <int KIND>
int launch_fwd(const RnnSeqParams& p, int H, int dirs, hipStream_t s) {
  dim3 grid(ceil_div(p.B, 16), dirs);
  if (H == 64)
    hipLaunchKernelGGL((rnn_seq_fwd_kernel<KIND, 64, 4>), grid, dim3(256), 0, s, p);
  else if (H == 128)
    hipLaunchKernelGGL((rnn_seq_fwd_kernel<KIND, 128, 8>), grid, dim3(512), 0, s, p);
  else
    return 1;
  return 0;
}
template <int KIND>
int launch_bwd(const RnnSeqParams& p, int H, int dirs, hipStream_t s) {
  dim3 grid(ceil_div(p.B, 16), dirs);
  if (H == 64)
    hipLaunchKernelGGL((rnn_seq_bwd_kernel<KIND, 64, 4>), grid, dim3(256), 0, s, p);
  else if (H == 128)
    hipLaunchKernelGGL((rnn_seq_bwd_kernel<KIND, 128, 8>), grid, dim3(512), 0, s, p);
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


// ---- round 5: the same two kernels behind self-contained entry points (no zero-fills, no
// transposed weight copies, no layout copies around them) and the parameter gradients of the whole
// recurrent layer behind ONE call.  The instruction encoder's backward runs eagerly on a side
// stream behind the tail's backward graph; what it cost was the host issuing ~45 launches
// (fills, transposes, contiguous copies, 6 GEMMs with their split-K zero-fills, 4 column sums).
extern "C" int vlnce_rnn_seq_fwd2(int kind, int dirs, const float* const* gi,
                                  const float* const* w_hh, const float* const* b_hh,
                                  const int* lengths, float* const* out_tm, float* seq,
                                  long seq_st, long seq_sb, float* const* h_final,
                                  float* const* gates_save, float* const* aux_save, int B, int L,
                                  int H, vlnce_stream_t stream) {
  VLNCE_CHECK_ARG(gi && w_hh && b_hh && lengths && out_tm && h_final, "rnn_seq_fwd2: null argument");
  VLNCE_CHECK_ARG(dirs == 1 || dirs == 2, "rnn_seq_fwd2: dirs must be 1 or 2");
  VLNCE_CHECK_ARG(vlnce_rnn_seq_supported(kind, H), "rnn_seq_fwd2: unsupported kind/H (%d,%d)", kind, H);
  VLNCE_CHECK_ARG(B > 0 && L > 0, "rnn_seq_fwd2: bad shape");
  VLNCE_CHECK_ARG(!seq || (seq_st > 0 && seq_sb > 0), "rnn_seq_fwd2: seq needs its strides");
  RnnSeqParams p{};
  for (int d = 0; d < dirs; ++d) {
    p.gi[d] = gi[d];
    p.w_hh[d] = w_hh[d];
    p.b_hh[d] = b_hh[d];
    p.out[d] = out_tm[d];
    p.seq[d] = seq ? seq + (long)d * H : nullptr;
    p.h_final[d] = h_final[d];
    p.gates[d] = gates_save ? gates_save[d] : nullptr;
    p.aux[d] = aux_save ? aux_save[d] : nullptr;
    VLNCE_CHECK_ARG((p.gates[d] == nullptr) == (p.aux[d] == nullptr),
                    "rnn_seq_fwd2: gates_save and aux_save come together");
  }
  p.seq_st = seq_st;
  p.seq_sb = seq_sb;
  p.self_zero = 1;
  p.lengths = lengths;
  p.B = B;
  p.L = L;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int rc = kind == 0 ? launch_fwd<0>(p, H, dirs, s) : launch_fwd<1>(p, H, dirs, s);
  VLNCE_CHECK_ARG(rc == 0, "rnn_seq_fwd2: no kernel for H=%d", H);
  VLNCE_CHECK_LAUNCH("rnn_seq_fwd2");
  return 0;
}

extern "C" int vlnce_rnn_seq_bwd2(int kind, int dirs, const float* const* w_hh, const int* lengths,
                                  const float* const* out_tm, const float* const* gates_save,
                                  const float* const* aux_save, const float* dseq, long dseq_st,
                                  long dseq_sb, float* dout_ws, const float* const* dh_final,
                                  float* const* dgi, float* const* dgh, int B, int L, int H,
                                  vlnce_stream_t stream) {
  VLNCE_CHECK_ARG(w_hh && lengths && out_tm && gates_save && aux_save && dgi,
                  "rnn_seq_bwd2: null argument");
  VLNCE_CHECK_ARG(dirs == 1 || dirs == 2, "rnn_seq_bwd2: dirs must be 1 or 2");
  VLNCE_CHECK_ARG(vlnce_rnn_seq_supported(kind, H), "rnn_seq_bwd2: unsupported kind/H (%d,%d)", kind, H);
  VLNCE_CHECK_ARG(kind == 0 || dgh, "rnn_seq_bwd2: GRU needs dgh");
  VLNCE_CHECK_ARG(!dseq || (dseq_st > 0 && dseq_sb > 0), "rnn_seq_bwd2: dseq needs its strides");
  VLNCE_CHECK_ARG(!dseq || dout_ws, "rnn_seq_bwd2: dseq needs the dirs*L*B*H workspace");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (dseq) {
    const long n4 = (long)dirs * L * B * (H / 4);
    const int grid = (int)((n4 + 255) / 256 > 2048 ? 2048 : (n4 + 255) / 256);
    hipLaunchKernelGGL(dseq_to_time_major_kernel, dim3(grid), dim3(256), 0, s, dseq, dseq_st, dseq_sb,
                       dout_ws, dirs, L, B, H);
    VLNCE_CHECK_LAUNCH("rnn_seq_bwd2 (output gradient to time-major)");
  }
  RnnSeqParams p{};
  for (int d = 0; d < dirs; ++d) {
    p.w_hh[d] = w_hh[d];
    p.out[d] = const_cast<float*>(out_tm[d]);
    p.gates[d] = const_cast<float*>(gates_save[d]);
    p.aux[d] = const_cast<float*>(aux_save[d]);
    p.dout[d] = dseq ? dout_ws + (long)d * L * B * H : nullptr;
    p.dh_final[d] = dh_final ? dh_final[d] : nullptr;
    p.dgi[d] = dgi[d];
    p.dgh[d] = dgh ? dgh[d] : nullptr;
  }
  p.self_zero = 1;
  p.w_plain = 1;
  p.lengths = lengths;
  p.B = B;
  p.L = L;
  const int rc = kind == 0 ? launch_bwd<0>(p, H, dirs, s) : launch_bwd<1>(p, H, dirs, s);
  VLNCE_CHECK_ARG(rc == 0, "rnn_seq_bwd2: no kernel for H=%d", H);
  VLNCE_CHECK_LAUNCH("rnn_seq_bwd2");
  return 0;
}

// Parameter gradients of the recurrent layer (and the gradient of its input rows) from what BPTT
// left in dgi / dgh:  dW_hh[d] = dGh^T Hprev (Hprev = the time-major outputs shifted by one step in
// processing order: outputs past a row's length are zeros, so the shift is two views),
// db_hh[d] = colsum dGh, dW_ih[d] = dgi^T X, db_ih[d] = colsum dgi (LSTM: dGh = dgi, so db_ih ==
// db_hh and the caller may pass the same pointer), dX = sum_d dgi[d] W_ih[d].  `first_dir`: which
// direction array element 0 is (0 = forward; 1 = a call for the reverse direction alone -- the
// host runs the two directions' calls on two streams).
extern "C" int vlnce_rnn_seq_wgrad(int kind, int dirs, int first_dir, const float* const* dgi,
                                   const float* const* dgh, const float* const* out_tm,
                                   const float* x_tm, int ldx, int E, const float* const* w_ih,
                                   float* const* dw_ih, float* const* dw_hh, float* const* db_ih,
                                   float* const* db_hh, float* dx_tm, int B, int L, int H,
                                   vlnce_stream_t stream) {
  VLNCE_CHECK_ARG(dgi && out_tm && x_tm && dw_ih && dw_hh && db_ih && db_hh,
                  "rnn_seq_wgrad: null argument");
  VLNCE_CHECK_ARG(dirs == 1 || dirs == 2, "rnn_seq_wgrad: dirs must be 1 or 2");
  VLNCE_CHECK_ARG((first_dir == 0 || first_dir == 1) && first_dir + dirs <= 2,
                  "rnn_seq_wgrad: first_dir + dirs must stay within the two directions");
  VLNCE_CHECK_ARG(kind == 0 || dgh, "rnn_seq_wgrad: GRU needs dgh");
  VLNCE_CHECK_ARG(!dx_tm || w_ih, "rnn_seq_wgrad: dx_tm needs w_ih");
  const int GH = (kind == 0 ? 4 : 3) * H;
  const long rows = (long)L * B;
  VLNCE_CHECK_ARG(B > 0 && L > 0 && E > 0 && ldx >= E && rows < 0x7fffffffL, "rnn_seq_wgrad: bad shape");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  for (int d = 0; d < dirs; ++d) {
    const float* dGh = kind == 1 ? dgh[d] : dgi[d];
    if (L > 1) {
      // forward direction: step t's previous state is the output of t - 1; reverse: of t + 1
      const bool reverse = first_dir + d == 1;
      const float* dG_s = reverse ? dGh : dGh + (long)B * GH;
      const float* h_s = reverse ? out_tm[d] + (long)B * H : out_tm[d];
      int rc = vlnce_gemm(dG_s, GH, 1, h_s, H, 1, dw_hh[d], H, GH, H, (int)((long)(L - 1) * B), nullptr, stream);
      if (rc != 0) return rc;
    } else {
      vlnce_zero(dw_hh[d], GH, H, H, s);
    }
    int rc = vlnce_colsum(dGh, GH, (int)rows, GH, db_hh[d], 0, stream);
    if (rc != 0) return rc;
    rc = vlnce_gemm(dgi[d], GH, 1, x_tm, ldx, 1, dw_ih[d], E, GH, E, (int)rows, nullptr, stream);
    if (rc != 0) return rc;
    if (db_ih[d] != db_hh[d]) {
      rc = vlnce_colsum(dgi[d], GH, (int)rows, GH, db_ih[d], 0, stream);
      if (rc != 0) return rc;
    }
    if (dx_tm) {
      vlnce_epilogue e{};
      e.accumulate = d > 0;
      rc = vlnce_gemm(dgi[d], GH, 0, w_ih[d], E, 1, dx_tm, E, (int)rows, E, GH, &e, stream);
      if (rc != 0) return rc;
    }
  }
  return 0;
}
