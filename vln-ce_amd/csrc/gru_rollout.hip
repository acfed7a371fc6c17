// A T-step rollout of the masked GRU state encoder as ONE launch forward and ONE backward
// (habitat RNNStateEncoder.seq_forward semantics; call sites cma_policy.py:249-256,287-294,
// seq2seq_policy.py:128-136; the batches are the cached-feature DAgger batch,
// dagger_trainer.py:39-114, and a DD-PPO minibatch, rollout_storage.py:154-276).
//
// The recurrence h_t = GRU(gi_t, mask_t * h_{t-1}) is a chain of T dependent products with a
// [3H, H] matrix that is far too large for one workgroup's registers at H = 512 (3 MB) and far
// too small to fill the chip: as T launches the step costs a launch round trip each (measured:
// the update of a 5-episode x 100-step batch spends most of its time in ~600 such launches).
// Here H/16 workgroups stay resident for the whole rollout:
//   * workgroup b owns hidden units [16b, 16b+16): their 3 x 16 rows of W_hh (forward) or their
//     16 columns (backward, rows of W_hh^T) live in REGISTERS for all T steps -- thread
//     (unit u = tid/16, slice s = tid%16) holds the k-slice [s K/16, (s+1) K/16) of its rows;
//   * per step the workgroups exchange the new state (backward: the gate gradients) through a
//     small double-buffered area of (value, step tag) PAIRS written with one 64-bit device-scope
//     atomic store each and polled with 64-bit atomic loads until the tag is the step's: data
//     and flag travel together, so a step costs one store -> load latency across the chip and
//     needs no fence, no counter and no second round trip (a first version with an arrival
//     counter + thread fences, the cooperative-groups grid-sync pattern, measured 6.2 us per
//     step; profiles/archive/r03_f_*);
//   * the [N, K] operand of the step (N <= 16 episodes) is staged in LDS, each thread multiplies
//     its register slice with all N rows (fp32 FMA: this is the literal fp32 arithmetic of the
//     step kernels in rnn.hip, only the summation order differs), slice partials meet in LDS and
//     thread (u, n) finishes unit u of episode n: gates, new state, what backward needs.
// What a step reads that does not depend on the recurrence (gi; in the backward pass the saved
// gates / states / output gradient) is fetched one step ahead.
#include <stdlib.h>

#include "common.h"

namespace {

constexpr int RO_UNITS = 16;   // hidden units (fwd) / carry columns (bwd) per workgroup
constexpr int RO_SLICES = 16;  // k-slices per unit = lanes that share a unit
constexpr int RO_MAXN = 16;    // episodes per rollout step
constexpr int RO_PP = 20;      // pitch of a unit's 16 slice partials (conflict-free b128 reads)

struct GruRolloutParams {
  const float* gi;      // [T,N,3H]  x W_ih^T + b_ih
  const float* h0;      // [N,H]
  const uint8_t* mask;  // [T,N] not-done masks
  const float* w;       // fwd: W_hh [3H,H]; bwd: W_hh^T [H,3H]
  const float* b_hh;    // [3H]
  float* hp;            // [T,N,H]  mask_t * h_{t-1}
  float* out;           // [T,N,H]
  float* gates;         // [T,N,3H] r, z, n (activated)
  float* aux;           // [T,N,H]  hn = W_hn hp + b_hn
  const float* dout;    // [T,N,H]
  const float* dh_fin;  // [N,H] or null
  float* dgi;           // [T,N,3H]
  float* dgh;           // [T,N,3H]
  float* dh0;           // [N,H]
  unsigned long long* ex;  // [2][N][K] (value, tag) pairs, zero at launch; K = H (fwd) | 3H (bwd)
  int T, N;
  int spread;  // 1, or 8: only every 8th workgroup works, i.e. all of them on one XCD (experiment)
};

__device__ __forceinline__ float ro_sigm(float x) { return 1.f / (1.f + expf(-x)); }

constexpr int ro_chunk(int ld) {
  int c = ld < 16 ? ld : 16;
  while (ld % c) --c;
  return c;
}

__device__ __forceinline__ void ex_store(unsigned long long* p, float v, unsigned tag) {
  const unsigned long long w = ((unsigned long long)tag << 32) | __float_as_uint(v);
  __hip_atomic_store(p, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// stage src [N, K] (K-contiguous rows) into LDS rows of 16 slices x (SL + 4) floats, scaled per row
template <int K, int NT, bool MASKED>
__device__ __forceinline__ void stage_operand(float* __restrict__ Xs, const float* __restrict__ src,
                                              const uint8_t* __restrict__ mrow, int N) {
  constexpr int SL = K / RO_SLICES, PITCH = SL + 4, ROW = RO_SLICES * PITCH;
  constexpr int K4 = K / 4;
  constexpr int LD = (NT * K4 + 255) / 256;
  const int total = N * K4;
  f32x4 v[LD];
#pragma unroll
  for (int q = 0; q < LD; ++q) {
    const int i = threadIdx.x + q * 256;
    if (i < total) v[q] = *reinterpret_cast<const f32x4*>(src + (long)i * 4);
  }
#pragma unroll
  for (int q = 0; q < LD; ++q) {
    const int i = threadIdx.x + q * 256;
    if (i < total) {
      const int n = i / K4, k = (i - n * K4) * 4;
      f32x4 x = v[q];
      if (MASKED) x *= (float)mrow[n];
      *reinterpret_cast<f32x4*>(Xs + n * ROW + (k / SL) * PITCH + (k % SL)) = x;
    }
  }
}

// the same from the exchange area: every thread polls its share of the N*K pairs until each
// carries `tag` (bounded: a lost workgroup ends in a trap, not in a hung device)
template <int K, int NT, bool MASKED>
__device__ __forceinline__ void stage_exchanged(float* __restrict__ Xs,
                                                const unsigned long long* __restrict__ ex,
                                                unsigned tag, const uint8_t* __restrict__ mrow,
                                                int N) {
  constexpr int SL = K / RO_SLICES, PITCH = SL + 4, ROW = RO_SLICES * PITCH;
  constexpr int LD = (NT * K + 255) / 256;
  constexpr int CH = ro_chunk(LD);  // pairs in flight per thread: the largest divisor <= 16
  static_assert(LD % CH == 0, "whole chunks");
  const int total = N * K;
  long spins = 0;
  for (int c0 = 0; c0 < LD; c0 += CH) {
    if (threadIdx.x + c0 * 256 >= total) break;
    unsigned long long w[CH];
    bool again;
    do {
      again = false;
#pragma unroll
      for (int q = 0; q < CH; ++q) {
        const int i = threadIdx.x + (c0 + q) * 256;
        if (i < total)
          w[q] = __hip_atomic_load(ex + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
#pragma unroll
      for (int q = 0; q < CH; ++q) {
        const int i = threadIdx.x + (c0 + q) * 256;
        if (i < total && (unsigned)(w[q] >> 32) != tag) again = true;
      }
      if (again && ++spins > (1L << 22)) __builtin_trap();
    } while (again);
#pragma unroll
    for (int q = 0; q < CH; ++q) {
      const int i = threadIdx.x + (c0 + q) * 256;
      if (i < total) {
        const int n = i / K, k = i - n * K;
        float x = __uint_as_float((unsigned)w[q]);
        if (MASKED) x *= (float)mrow[n];
        Xs[n * ROW + (k / SL) * PITCH + (k % SL)] = x;
      }
    }
  }
}

template <int H, int NT>
__global__ __launch_bounds__(256) void gru_rollout_fwd_kernel(GruRolloutParams p) {
  constexpr int SL = H / RO_SLICES, PITCH = SL + 4, ROW = RO_SLICES * PITCH;
  static_assert(SL % 4 == 0, "slice of whole float4s");
  extern __shared__ __attribute__((aligned(16))) float ro_sm[];
  float* Xs = ro_sm;               // [NT][ROW]   masked previous state
  float* part = ro_sm + NT * ROW;  // [3][NT][16 units][RO_PP]
  const int tid = threadIdx.x;
  const int u = tid >> 4, sl = tid & 15;
  const int n_own = sl;  // final stage: this thread finishes (unit u, episode sl)
  if (blockIdx.x % p.spread) return;
  const int j = (blockIdx.x / p.spread) * RO_UNITS + u;
  const int N = p.N, T = p.T;
  const bool fin = n_own < N;

  float w[3][SL];
  float bh[3];
#pragma unroll
  for (int g = 0; g < 3; ++g) {
    const float* row = p.w + (long)(g * H + j) * H + sl * SL;
#pragma unroll
    for (int i4 = 0; i4 < SL / 4; ++i4) {
      const f32x4 t4 = *reinterpret_cast<const f32x4*>(row + 4 * i4);
      w[g][4 * i4] = t4[0];
      w[g][4 * i4 + 1] = t4[1];
      w[g][4 * i4 + 2] = t4[2];
      w[g][4 * i4 + 3] = t4[3];
    }
    bh[g] = p.b_hh[g * H + j];
  }
  float gx[3] = {0.f, 0.f, 0.f}, gnext[3] = {0.f, 0.f, 0.f};
  if (fin) {
#pragma unroll
    for (int g = 0; g < 3; ++g) gx[g] = p.gi[(long)n_own * 3 * H + g * H + j];
  }

  for (int t = 0; t < T; ++t) {
    if (fin && t + 1 < T) {
#pragma unroll
      for (int g = 0; g < 3; ++g) gnext[g] = p.gi[((long)(t + 1) * N + n_own) * 3 * H + g * H + j];
    }
    if (t == 0)
      stage_operand<H, NT, true>(Xs, p.h0, p.mask, N);
    else
      stage_exchanged<H, NT, true>(Xs, p.ex + (long)((t - 1) & 1) * N * H, (unsigned)t,
                                   p.mask + (long)t * N, N);
    __syncthreads();
    float acc[3][NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      acc[0][n] = acc[1][n] = acc[2][n] = 0.f;
      if (n < N) {
        const float* xr = Xs + n * ROW + sl * PITCH;
#pragma unroll
        for (int i4 = 0; i4 < SL / 4; ++i4) {
          const f32x4 x = *reinterpret_cast<const f32x4*>(xr + 4 * i4);
#pragma unroll
          for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int g = 0; g < 3; ++g) acc[g][n] = fmaf(w[g][4 * i4 + c], x[c], acc[g][n]);
        }
#pragma unroll
        for (int g = 0; g < 3; ++g) part[((g * NT + n) * RO_UNITS + u) * RO_PP + sl] = acc[g][n];
      }
    }
    __syncthreads();
    if (fin) {
      float gh[3];
#pragma unroll
      for (int g = 0; g < 3; ++g) {
        const float* pr = part + ((g * NT + n_own) * RO_UNITS + u) * RO_PP;
        f32x4 s4 = *reinterpret_cast<const f32x4*>(pr);
#pragma unroll
        for (int q = 1; q < 4; ++q) s4 += *reinterpret_cast<const f32x4*>(pr + 4 * q);
        gh[g] = (s4[0] + s4[1]) + (s4[2] + s4[3]) + bh[g];
      }
      const float hpv = Xs[n_own * ROW + (j / SL) * PITCH + (j % SL)];
      const float r = ro_sigm(gx[0] + gh[0]);
      const float z = ro_sigm(gx[1] + gh[1]);
      const float nn = tanhf(gx[2] + r * gh[2]);
      const long o = ((long)t * N + n_own) * H + j;
      const float hnew = (1.f - z) * nn + z * hpv;
      if (t + 1 < T)  // first: the other workgroups are waiting for it
        ex_store(p.ex + (long)(t & 1) * N * H + (long)n_own * H + j, hnew, (unsigned)(t + 1));
      p.out[o] = hnew;
      p.hp[o] = hpv;
      p.aux[o] = gh[2];
      float* gs = p.gates + ((long)t * N + n_own) * 3 * H;
      gs[j] = r;
      gs[H + j] = z;
      gs[2 * H + j] = nn;
#pragma unroll
      for (int g = 0; g < 3; ++g) gx[g] = gnext[g];
    }
    __syncthreads();  // Xs / part are free for the next step
  }
}

template <int H, int NT>
__global__ __launch_bounds__(256) void gru_rollout_bwd_kernel(GruRolloutParams p) {
  constexpr int GH = 3 * H;
  constexpr int SL = GH / RO_SLICES, PITCH = SL + 4, ROW = RO_SLICES * PITCH;
  static_assert(SL % 4 == 0, "slice of whole float4s");
  extern __shared__ __attribute__((aligned(16))) float ro_sm[];
  float* Xs = ro_sm;               // [NT][ROW]  dgh of this step
  float* part = ro_sm + NT * ROW;  // [NT][16 columns][RO_PP]
  const int tid = threadIdx.x;
  const int u = tid >> 4, sl = tid & 15;
  const int n_own = sl;
  if (blockIdx.x % p.spread) return;
  const int j = (blockIdx.x / p.spread) * RO_UNITS + u;
  const int N = p.N, T = p.T;
  const bool fin = n_own < N;

  float w[SL];  // row j of W_hh^T = column j of W_hh, slice sl of the 3H gate rows
  {
    const float* row = p.w + (long)j * GH + sl * SL;
#pragma unroll
    for (int i4 = 0; i4 < SL / 4; ++i4) {
      const f32x4 t4 = *reinterpret_cast<const f32x4*>(row + 4 * i4);
      w[4 * i4] = t4[0];
      w[4 * i4 + 1] = t4[1];
      w[4 * i4 + 2] = t4[2];
      w[4 * i4 + 3] = t4[3];
    }
  }
  float cr = (fin && p.dh_fin) ? p.dh_fin[(long)n_own * H + j] : 0.f;

  struct Saved {
    float r, z, nn, hn, hpv, d, mk;
  };
  auto load = [&](int t) {
    Saved s{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (fin && t >= 0) {
      const long b = (long)t * N + n_own;
      const float* gs = p.gates + b * GH;
      s.r = gs[j];
      s.z = gs[H + j];
      s.nn = gs[2 * H + j];
      s.hn = p.aux[b * H + j];
      s.hpv = p.hp[b * H + j];
      s.d = p.dout ? p.dout[b * H + j] : 0.f;
      s.mk = (float)p.mask[b];
    }
    return s;
  };
  Saved nxt = load(T - 1);

  for (int t = T - 1; t >= 0; --t) {
    const Saved cur = nxt;
    nxt = load(t - 1);
    float acc0 = 0.f;
    if (fin) {
      const float d = cur.d + cr;
      const float dn = d * (1.f - cur.z);
      const float dz = d * (cur.hpv - cur.nn);
      const float dnp = dn * (1.f - cur.nn * cur.nn);
      const float drp = dnp * cur.hn * cur.r * (1.f - cur.r);
      const float dzp = dz * cur.z * (1.f - cur.z);
      unsigned long long* e = p.ex + (long)(t & 1) * N * GH + (long)n_own * GH;
      ex_store(e + j, drp, (unsigned)(T - t));  // first: the other workgroups are waiting for them
      ex_store(e + H + j, dzp, (unsigned)(T - t));
      ex_store(e + 2 * H + j, dnp * cur.r, (unsigned)(T - t));
      const long b = (long)t * N + n_own;
      float* a = p.dgi + b * GH;
      a[j] = drp;
      a[H + j] = dzp;
      a[2 * H + j] = dnp;
      float* c = p.dgh + b * GH;
      c[j] = drp;
      c[H + j] = dzp;
      c[2 * H + j] = dnp * cur.r;
      acc0 = d * cur.z;
    }
    stage_exchanged<GH, NT, false>(Xs, p.ex + (long)(t & 1) * N * GH, (unsigned)(T - t), nullptr, N);
    __syncthreads();
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      if (n < N) {
        float a0 = 0.f, a1 = 0.f;
        const float* xr = Xs + n * ROW + sl * PITCH;
#pragma unroll
        for (int i4 = 0; i4 < SL / 4; ++i4) {
          const f32x4 x = *reinterpret_cast<const f32x4*>(xr + 4 * i4);
          a0 = fmaf(w[4 * i4], x[0], a0);
          a1 = fmaf(w[4 * i4 + 1], x[1], a1);
          a0 = fmaf(w[4 * i4 + 2], x[2], a0);
          a1 = fmaf(w[4 * i4 + 3], x[3], a1);
        }
        part[(n * RO_UNITS + u) * RO_PP + sl] = a0 + a1;
      }
    }
    __syncthreads();
    if (fin) {
      const float* pr = part + (n_own * RO_UNITS + u) * RO_PP;
      f32x4 s4 = *reinterpret_cast<const f32x4*>(pr);
#pragma unroll
      for (int q = 1; q < 4; ++q) s4 += *reinterpret_cast<const f32x4*>(pr + 4 * q);
      cr = (acc0 + ((s4[0] + s4[1]) + (s4[2] + s4[3]))) * cur.mk;
    }
    __syncthreads();  // Xs / part are free for the next step
  }
  if (fin) p.dh0[(long)n_own * H + j] = cr;
}

template <int H, int NT>
int launch_rollout(const GruRolloutParams& p, bool bwd, hipStream_t s) {
  constexpr int K = 3 * H;
  const size_t lds_f = (size_t)(NT * RO_SLICES * (H / RO_SLICES + 4) + 3 * NT * RO_UNITS * RO_PP) * 4;
  const size_t lds_b = (size_t)(NT * RO_SLICES * (K / RO_SLICES + 4) + NT * RO_UNITS * RO_PP) * 4;
  const size_t lds = bwd ? lds_b : lds_f;
  static bool attr_f = false, attr_b = false;
  bool& done = bwd ? attr_b : attr_f;
  if (!done && lds > 48 * 1024) {
    const void* fn = bwd ? reinterpret_cast<const void*>(&gru_rollout_bwd_kernel<H, NT>)
                         : reinterpret_cast<const void*>(&gru_rollout_fwd_kernel<H, NT>);
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return 1;
    done = true;
  }
  const long words = 2L * 2 * p.N * (bwd ? K : H);  // [2][N][K] pairs as 32-bit words
  vlnce_zero(reinterpret_cast<float*>(p.ex), 1, (int)words, words, s);
  const dim3 grid(H / RO_UNITS * p.spread);
  if (bwd)
    hipLaunchKernelGGL((gru_rollout_bwd_kernel<H, NT>), grid, dim3(256), lds, s, p);
  else
    hipLaunchKernelGGL((gru_rollout_fwd_kernel<H, NT>), grid, dim3(256), lds, s, p);
  return 0;
}

int dispatch_rollout(GruRolloutParams& p, int H, bool bwd, hipStream_t s) {
  const int spread = vlnce_opt(VLNCE_OPT_ROLLOUT_ONE_XCD) ? 8 : 1;
  p.spread = spread;
  const bool small = p.N <= 8;
  switch (H) {
    case 64: return small ? launch_rollout<64, 8>(p, bwd, s) : launch_rollout<64, 16>(p, bwd, s);
    case 128: return small ? launch_rollout<128, 8>(p, bwd, s) : launch_rollout<128, 16>(p, bwd, s);
    case 256: return small ? launch_rollout<256, 8>(p, bwd, s) : launch_rollout<256, 16>(p, bwd, s);
    case 512: return small ? launch_rollout<512, 8>(p, bwd, s) : launch_rollout<512, 16>(p, bwd, s);
  }
  return 1;
}

}  // namespace

// The H / 16 workgroups of a rollout launch hand each other the state every step: all of them must
// be RESIDENT at once (up to 122 KB of LDS each, i.e. one per CU).  That is a property of the
// device, checked here once per device: a part with less LDS per workgroup or fewer CUs than
// workgroups reports "unsupported" and the host takes the per-step kernels instead of a launch
// that would spin until its poll bound traps (round-3 ADVICE).
static bool rollout_device_ok(int H) {
  static int lds_max[16], cus[16];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return false;
  if (cus[dev] == 0) {
    int l = 0, c = 0;
    if (hipDeviceGetAttribute(&l, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) != hipSuccess ||
        hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
      return false;
    lds_max[dev] = l;
    cus[dev] = c > 0 ? c : -1;
  }
  return lds_max[dev] >= 128 * 1024 && cus[dev] >= H / 16;
}

extern "C" int vlnce_gru_rollout_supported(int N, int H) {
  return N > 0 && N <= RO_MAXN && (H == 64 || H == 128 || H == 256 || H == 512) &&
         rollout_device_ok(H);
}

extern "C" long vlnce_gru_rollout_workspace_bytes(int N, int H) {
  return vlnce_gru_rollout_supported(N, H) ? 2L * N * 3 * H * 8 : 0;
}

extern "C" int vlnce_gru_rollout_fwd(const float* gi, const float* h0, const uint8_t* mask,
                                     const float* w_hh, const float* b_hh, float* hp, float* out,
                                     float* gates, float* aux, void* workspace, int T, int N,
                                     int H, vlnce_stream_t stream) {
  VLNCE_CHECK_ARG(gi && h0 && mask && w_hh && b_hh && hp && out && gates && aux && workspace,
                  "gru_rollout_fwd: null argument");
  VLNCE_CHECK_ARG(T > 0, "gru_rollout_fwd: T must be positive");
  VLNCE_CHECK_ARG(vlnce_gru_rollout_supported(N, H), "gru_rollout_fwd: unsupported N/H (%d,%d)", N, H);
  GruRolloutParams p{};
  p.gi = gi;
  p.h0 = h0;
  p.mask = mask;
  p.w = w_hh;
  p.b_hh = b_hh;
  p.hp = hp;
  p.out = out;
  p.gates = gates;
  p.aux = aux;
  p.ex = static_cast<unsigned long long*>(workspace);
  p.T = T;
  p.N = N;
  VLNCE_CHECK_ARG(dispatch_rollout(p, H, false, reinterpret_cast<hipStream_t>(stream)) == 0,
                  "gru_rollout_fwd: no kernel for H=%d", H);
  VLNCE_CHECK_LAUNCH("gru_rollout_fwd");
  return 0;
}

extern "C" int vlnce_gru_rollout_bwd(const float* dout, const float* dh_final, const float* gates,
                                     const float* aux, const float* hp, const uint8_t* mask,
                                     const float* w_hh_t, float* dgi, float* dgh, float* dh0,
                                     void* workspace, int T, int N, int H,
                                     vlnce_stream_t stream) {
  VLNCE_CHECK_ARG(gates && aux && hp && mask && w_hh_t && dgi && dgh && dh0 && workspace,
                  "gru_rollout_bwd: null argument");
  VLNCE_CHECK_ARG(T > 0, "gru_rollout_bwd: T must be positive");
  VLNCE_CHECK_ARG(vlnce_gru_rollout_supported(N, H), "gru_rollout_bwd: unsupported N/H (%d,%d)", N, H);
  GruRolloutParams p{};
  p.dout = dout;
  p.dh_fin = dh_final;
  p.gates = const_cast<float*>(gates);
  p.aux = const_cast<float*>(aux);
  p.hp = const_cast<float*>(hp);
  p.mask = mask;
  p.w = w_hh_t;
  p.dgi = dgi;
  p.dgh = dgh;
  p.dh0 = dh0;
  p.ex = static_cast<unsigned long long*>(workspace);
  p.T = T;
  p.N = N;
  VLNCE_CHECK_ARG(dispatch_rollout(p, H, true, reinterpret_cast<hipStream_t>(stream)) == 0,
                  "gru_rollout_bwd: no kernel for H=%d", H);
  VLNCE_CHECK_LAUNCH("gru_rollout_bwd");
  return 0;
}
