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
};

// Gate non-linearities on the hardware exp2 / rcp (v_exp_f32, v_rcp_f32; absolute error < 3e-7,
// tests at 1e-5): libm's expf / tanhf are ~25 / ~50 VALU instructions each and a step evaluates
// five per (row, unit) -- 2/3 of the forward step's instruction stream, which the two waves of a
// SIMD execute one after the other (measured: 5.0 -> see profiles/r03_f_seqbench.txt us per step).
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

// four values -> three words of 4 bf16 each (one 8-byte LDS store per plane)
__device__ __forceinline__ void split4(const float (&x)[4], uint2 (&o)[3]) {
  unsigned short w[4][3];
#pragma unroll
  for (int r = 0; r < 4; ++r) split_scalar(x[r], w[r]);
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    o[q].x = (unsigned)w[0][q] | ((unsigned)w[1][q] << 16);
    o[q].y = (unsigned)w[2][q] | ((unsigned)w[3][q] << 16);
  }
}

// NW waves per workgroup, each owning H / NW hidden units for every gate (NW = 8 at H = 128: two
// waves per SIMD, so that a wave's three weight planes fit its 256 registers).
// The recurrent product is issued as  W_slice (A operand: 16 units x k)  x  h^T (B operand:
// k x 16 rows), so that the accumulator of lane (l15, quad) holds row b0 + l15 and the FOUR
// CONSECUTIVE units 4 quad .. 4 quad + 3 of the wave's 16-unit block: everything a step reads and
// writes per lane (input projection, saved gates / states, the new state's planes in LDS) is then
// a 16-byte access instead of four 4-byte ones -- a quarter of the vector-memory and LDS
// instructions of the row-major assignment (profiles/r03_g_seqbench.txt).
template <int KIND, int H, int NW>
__global__ __launch_bounds__(NW * 64) void rnn_seq_fwd_kernel(RnnSeqParams p) {
  constexpr int G = KIND == 0 ? 4 : 3;
  constexpr int NTW = H / (NW * 16);
  static_assert(NTW >= 1, "units per wave");
  constexpr int KS = H / 32;     // k-steps of 32 per recurrent product
  constexpr int LDHB = H + 16;   // bf16 row pitch: conflict-free ds_read_b128 of 8 k-values
  __shared__ __attribute__((aligned(16))) unsigned short h_pl[2][3][16][LDHB];
  const int d = blockIdx.y;
  const int b0 = blockIdx.x * 16;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, quad = lane >> 4;
  const bool reverse = d == 1;
  const float* __restrict__ gi = p.gi[d];
  const float* __restrict__ W = p.w_hh[d];
  const int B = p.B, L = p.L;
  const int b = b0 + l15;                      // this lane's row
  const int len = b < B ? p.lengths[b] : 0;
  // steps past the longest sequence of this tile change nothing (state kept, zeros emitted into
  // the pre-zeroed outputs): the loop ends there, not at the padded length
  int Lt = len;
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) Lt = max(Lt, __shfl_xor(Lt, o, 64));
  Lt = min(Lt, L);
  // recurrent weights -> registers, split once (A operand: lane holds the three planes of
  // W[unit = block + l15][k = 32 ks + 8 quad + 0..7])
  Planes3 wp[G][NTW][KS];
  f32x4 bias[G][NTW];
#pragma unroll
  for (int gt = 0; gt < G; ++gt)
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
      const int blk = gt * H + wave * (H / NW) + nt * 16;
      bias[gt][nt] = *reinterpret_cast<const f32x4*>(p.b_hh[d] + blk + 4 * quad);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const float* w8 = W + (long)(blk + l15) * H + 32 * ks + 8 * quad;
        wp[gt][nt][ks] = split_planes(*reinterpret_cast<const f32x4*>(w8),
                                      *reinterpret_cast<const f32x4*>(w8 + 4));
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
  f32x4 gx[G][NTW];
  auto fetch = [&](int s) {
    const bool active = s < len;
    const int tt = reverse ? len - 1 - s : s;
    const float* row = gi + ((long)tt * B + b) * (G * H) + wave * (H / NW) + 4 * quad;
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
      for (int gt = 0; gt < G; ++gt)
        gx[gt][nt] = active ? *reinterpret_cast<const f32x4*>(row + gt * H + nt * 16)
                            : f32x4{0.f, 0.f, 0.f, 0.f};
  };
  fetch(0);

  for (int s = 0; s < Lt; ++s) {
    const int cur = s & 1;
    f32x4 acc[G][NTW];
    f32x4 xn[NTW];  // GRU: input part of the n gate
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
      for (int gt = 0; gt < G; ++gt) {
        if (KIND == 1 && gt == 2) {
          xn[nt] = gx[gt][nt];
          acc[gt][nt] = bias[gt][nt];
        } else {
          acc[gt][nt] = gx[gt][nt] + bias[gt][nt];
        }
      }
    fetch(s + 1);  // (rows with s + 1 >= len load nothing)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      bf16x8 a[3];
#pragma unroll
      for (int q = 0; q < 3; ++q)
        a[q] = *reinterpret_cast<const bf16x8*>(&h_pl[cur][q][l15][32 * ks + 8 * quad]);
#pragma unroll
      for (int q = 0; q < 6; ++q)
#pragma unroll
        for (int gt = 0; gt < G; ++gt)
#pragma unroll
          for (int nt = 0; nt < NTW; ++nt)
            acc[gt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                wp[gt][nt][ks].p[X3_PB[q]], a[X3_PA[q]], acc[gt][nt], 0, 0, 0);
    }
    const bool active = s < len;
    const int tt = reverse ? len - 1 - s : s;
    const long rowi = (long)tt * B + b;
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
      const int u0 = wave * (H / NW) + nt * 16 + 4 * quad;
      f32x4 g0, g1, g2, g3, ax, hn4;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float hnew;
        if (KIND == 0) {
          const float ig = sigm(acc[0][nt][r]), fg = sigm(acc[1][nt][r]);
          const float gg = tanh_fast(acc[2][nt][r]), og = sigm(acc[3][nt][r]);
          const float cn = fg * creg[nt][r] + ig * gg;
          hnew = og * tanh_fast(cn);
          if (active) creg[nt][r] = cn;
          g0[r] = ig;
          g1[r] = fg;
          g2[r] = gg;
          g3[r] = og;
          ax[r] = cn;
        } else {
          const float rg = sigm(acc[0][nt][r]), zg = sigm(acc[1][nt][r]);
          const float hn = acc[2][nt][r];
          const float ng = tanh_fast(xn[nt][r] + rg * hn);
          hnew = (1.f - zg) * ng + zg * hreg[nt][r];
          g0[r] = rg;
          g1[r] = zg;
          g2[r] = ng;
          ax[r] = hn;
        }
        if (active) hreg[nt][r] = hnew;
        hn4[r] = hreg[nt][r];
      }
      if (active) {
        if (p.gates[d]) {
          float* gs = p.gates[d] + rowi * (G * H) + u0;
          *reinterpret_cast<f32x4*>(gs) = g0;
          *reinterpret_cast<f32x4*>(gs + H) = g1;
          *reinterpret_cast<f32x4*>(gs + 2 * H) = g2;
          if (KIND == 0) *reinterpret_cast<f32x4*>(gs + 3 * H) = g3;
          *reinterpret_cast<f32x4*>(p.aux[d] + rowi * H + u0) = ax;
        }
        *reinterpret_cast<f32x4*>(p.out[d] + rowi * H + u0) = hn4;
      }
      const float hv[4] = {hn4[0], hn4[1], hn4[2], hn4[3]};
      uint2 hw[3];
      split4(hv, hw);
#pragma unroll
      for (int q = 0; q < 3; ++q)
        *reinterpret_cast<uint2*>(&h_pl[cur ^ 1][q][l15][u0]) = hw[q];
    }
    __syncthreads();
  }
  if (b < B)
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt)
      *reinterpret_cast<f32x4*>(p.h_final[d] + (long)b * H + wave * (H / NW) + nt * 16 + 4 * quad) =
          f32x4{hreg[nt][0], hreg[nt][1], hreg[nt][2], hreg[nt][3]};
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
  const int d = blockIdx.y;
  const int b0 = blockIdx.x * 16;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, quad = lane >> 4;
  const bool reverse = d == 1;
  const float* __restrict__ WT = p.w_hh[d];  // [H, G*H]: WT[n][k] = W_hh[k][n]
  const int B = p.B, L = p.L;
  const int b = b0 + l15;  // this lane's row; its units: block + 4 quad .. + 3 (see the forward kernel)
  const int len = b < B ? p.lengths[b] : 0;
  int Lt = len;
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) Lt = max(Lt, __shfl_xor(Lt, o, 64));
  Lt = min(Lt, L);
  Planes3 wt[NTW][KS];  // the three planes of WT[unit = block + l15][32 ks + 8 quad + 0..7]
#pragma unroll
  for (int nt = 0; nt < NTW; ++nt) {
    const int n = wave * (H / NW) + nt * 16 + l15;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const float* w8 = WT + (long)n * GH + 32 * ks + 8 * quad;
      wt[nt][ks] = split_planes(*reinterpret_cast<const f32x4*>(w8),
                                *reinterpret_cast<const f32x4*>(w8 + 4));
    }
  }
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  f32x4 dh[NTW], dc[NTW];
#pragma unroll
  for (int nt = 0; nt < NTW; ++nt) {
    const int u0 = wave * (H / NW) + nt * 16 + 4 * quad;
    dh[nt] = (p.dh_final[d] && b < B)
                 ? *reinterpret_cast<const f32x4*>(p.dh_final[d] + (long)b * H + u0)
                 : zero4;
    dc[nt] = zero4;
  }

  // What a step reads from memory (saved gates, cell / candidate state, the output gradient,
  // the previous step's state) is fetched one step ahead -- two for the previous state, which is
  // the next step's own state -- so no step starts with a global-load round trip.
  f32x4 pg[G][NTW], pa[NTW], pa_prev[NTW], pd[NTW];
  auto fetch = [&](int s, f32x4 (&g_)[G][NTW], f32x4 (&a_)[NTW], f32x4 (&d_)[NTW]) {
    const bool active = s >= 0 && s < len;
    const int tt = reverse ? len - 1 - s : s;
    const long base = (long)tt * B + b;
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
      const int u0 = wave * (H / NW) + nt * 16 + 4 * quad;
#pragma unroll
      for (int gt = 0; gt < G; ++gt)
        g_[gt][nt] = active ? *reinterpret_cast<const f32x4*>(p.gates[d] + base * GH + gt * H + u0)
                            : zero4;
      a_[nt] = active ? *reinterpret_cast<const f32x4*>(p.aux[d] + base * H + u0) : zero4;
      d_[nt] = (active && p.dout[d]) ? *reinterpret_cast<const f32x4*>(p.dout[d] + base * H + u0)
                                     : zero4;
    }
  };
  // previous state of step s: LSTM c_{s-1} = aux of step s-1; GRU h_{s-1} = out of step s-1
  auto fetch_prev = [&](int s, f32x4 (&a_)[NTW]) {
    const bool active = s >= 0 && s < len;
    const int tt = reverse ? len - 1 - s : s;
    const float* src = KIND == 0 ? p.aux[d] : p.out[d];
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
      const int u0 = wave * (H / NW) + nt * 16 + 4 * quad;
      a_[nt] = active ? *reinterpret_cast<const f32x4*>(src + ((long)tt * B + b) * H + u0) : zero4;
    }
  };
  fetch(Lt - 1, pg, pa, pd);
  fetch_prev(Lt - 2, pa_prev);

  for (int s = Lt - 1; s >= 0; --s) {
    f32x4 keep_z[NTW];  // GRU: dh * z carried straight to h_prev
    f32x4 cg[G][NTW], ca[NTW], cprev[NTW], cd[NTW];
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
#pragma unroll
      for (int gt = 0; gt < G; ++gt) cg[gt][nt] = pg[gt][nt];
      ca[nt] = pa[nt];
      cprev[nt] = pa_prev[nt];
      cd[nt] = pd[nt];
    }
    fetch(s - 1, pg, pa, pd);
    fetch_prev(s - 2, pa_prev);
    const bool active = s < len;
    const int tt = reverse ? len - 1 - s : s;
    const long base = (long)tt * B + b;
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
      const int u0 = wave * (H / NW) + nt * 16 + 4 * quad;
      float dpre[G][4], dgh_n[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int gt = 0; gt < G; ++gt) dpre[gt][r] = 0.f;
        dgh_n[r] = 0.f;
        keep_z[nt][r] = 0.f;
        if (active) {
          const float dht = dh[nt][r] + cd[nt][r];
          if (KIND == 0) {
            const float ig = cg[0][nt][r], fg = cg[1][nt][r], gg = cg[2][nt][r], og = cg[3][nt][r];
            const float c = ca[nt][r];
            const float cp = cprev[nt][r];  // (0 at the sequence's first step)
            const float tc = tanh_fast(c);
            const float dct = dc[nt][r] + dht * og * (1.f - tc * tc);
            dpre[0][r] = dct * gg * ig * (1.f - ig);
            dpre[1][r] = dct * cp * fg * (1.f - fg);
            dpre[2][r] = dct * ig * (1.f - gg * gg);
            dpre[3][r] = dht * tc * og * (1.f - og);
            dc[nt][r] = dct * fg;
          } else {
            const float rg = cg[0][nt][r], zg = cg[1][nt][r], ng = cg[2][nt][r];
            const float hn = ca[nt][r];
            const float hp = cprev[nt][r];
            const float dn = dht * (1.f - zg);
            const float dz = dht * (hp - ng);
            const float dnp = dn * (1.f - ng * ng);
            dpre[0][r] = dnp * hn * rg * (1.f - rg);
            dpre[1][r] = dz * zg * (1.f - zg);
            dpre[2][r] = dnp;
            dgh_n[r] = dnp * rg;
            keep_z[nt][r] = dht * zg;
          }
        }
      }
      if (active) {
        float* dgi = p.dgi[d] + base * GH + u0;
#pragma unroll
        for (int gt = 0; gt < G; ++gt)
          *reinterpret_cast<f32x4*>(dgi + gt * H) =
              f32x4{dpre[gt][0], dpre[gt][1], dpre[gt][2], dpre[gt][3]};
        if (KIND == 1) {
          float* dgh = p.dgh[d] + base * GH + u0;
          *reinterpret_cast<f32x4*>(dgh) = f32x4{dpre[0][0], dpre[0][1], dpre[0][2], dpre[0][3]};
          *reinterpret_cast<f32x4*>(dgh + H) = f32x4{dpre[1][0], dpre[1][1], dpre[1][2], dpre[1][3]};
          *reinterpret_cast<f32x4*>(dgh + 2 * H) = f32x4{dgh_n[0], dgh_n[1], dgh_n[2], dgh_n[3]};
        }
      }
      // B operand of dh_{t-1}^T = W_hh^T dgates_h^T  (zeros for finished / padded rows)
#pragma unroll
      for (int gt = 0; gt < G; ++gt) {
        uint2 dw[3];
        if (KIND == 1 && gt == 2)
          split4(dgh_n, dw);
        else
          split4(dpre[gt], dw);
#pragma unroll
        for (int q = 0; q < 3; ++q)
          *reinterpret_cast<uint2*>(&dg_pl[q][l15][gt * H + u0]) = dw[q];
      }
    }
    __syncthreads();
    f32x4 acc[NTW];
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) acc[nt] = zero4;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      bf16x8 a[3];
#pragma unroll
      for (int q = 0; q < 3; ++q)
        a[q] = *reinterpret_cast<const bf16x8*>(&dg_pl[q][l15][32 * ks + 8 * quad]);
#pragma unroll
      for (int q = 0; q < 6; ++q)
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt)
          acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wt[nt][ks].p[X3_PB[q]], a[X3_PA[q]],
                                                            acc[nt], 0, 0, 0);
    }
    if (active)
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt) dh[nt] = acc[nt] + keep_z[nt];
    __syncthreads();
  }
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
