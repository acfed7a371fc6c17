// The categorical action head of an imitation-learning policy in one launch per direction.
//
// Reference: habitat-lab CategoricalNet (models/policy.py:19-21 builds it; utils.py:269-289 is the
// distribution): logits = Linear(features); torch.distributions.Categorical then normalises them,
// `logits - logits.logsumexp(-1, keepdim=True)` (nine elementwise / reduction launches of a
// [num_envs, 4..6] tensor), and validates the result (`value == value` -> all() -> host read-back).
// Eagerly that is ~16 launches forward and ~12 backward between the tail's HIP graph and the
// trainer's loss, every one of them host-paced.  Here: one launch computes the dot products
// (A <= 16 actions, K = the policy's output size), the log-softmax and a NaN count the host reads;
// one launch does the whole backward (dx, dW, db).
#include "common.h"

namespace {

constexpr int AH_ROWS = 256;  // rows of dz a weight-gradient workgroup keeps in LDS at a time

// One wavefront per row.  out[m, a] = z[m, a] - logsumexp_a z[m, :],  z = x W^T + b.
template <int AM>
__global__ __launch_bounds__(256) void action_head_fwd_kernel(
    const float* __restrict__ x, int ldx, const float* __restrict__ w, const float* __restrict__ b,
    int M, int K, int A, float* __restrict__ out, int* __restrict__ bad, int vec) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const float* xr = x + (long)row * ldx;
  float acc[AM];
#pragma unroll
  for (int a = 0; a < AM; ++a) acc[a] = 0.f;
  if (vec) {
    for (int c = lane * 4; c < K; c += 256) {
      const f32x4 xv = *reinterpret_cast<const f32x4*>(xr + c);
#pragma unroll
      for (int a = 0; a < AM; ++a)
        if (a < A) {
          const f32x4 wv = *reinterpret_cast<const f32x4*>(w + (long)a * K + c);
          acc[a] += xv.x * wv.x + xv.y * wv.y + xv.z * wv.z + xv.w * wv.w;
        }
    }
  } else {
    for (int c = lane; c < K; c += 64) {
      const float xv = xr[c];
#pragma unroll
      for (int a = 0; a < AM; ++a)
        if (a < A) acc[a] += xv * w[(long)a * K + c];
    }
  }
  float mx = -INFINITY;
#pragma unroll
  for (int a = 0; a < AM; ++a)
    if (a < A) {
      acc[a] = wave_sum(acc[a]) + (b ? b[a] : 0.f);
      mx = fmaxf(mx, acc[a]);
    }
  // torch.logsumexp: an infinite row maximum is taken as 0 (so +inf / all -inf rows give the same
  // inf / nan pattern as the reference's, and the validation below raises on the same inputs)
  if (fabsf(mx) == INFINITY) mx = 0.f;
  float s = 0.f;
#pragma unroll
  for (int a = 0; a < AM; ++a)
    if (a < A) s += expf(acc[a] - mx);
  const float lse = logf(s) + mx;
  float mine = 0.f;
  bool nan = false;
#pragma unroll
  for (int a = 0; a < AM; ++a)
    if (a < A) {
      const float v = acc[a] - lse;
      nan |= (v != v);
      if (lane == a) mine = v;
    }
  if (lane < A) out[(long)row * A + lane] = mine;
  if (bad != nullptr && nan && lane == 0) atomicAdd(bad, 1);
}

// dz[m, a] = dn[m, a] - exp(n[m, a]) * sum_a dn[m, :]   (n = the normalised logits saved by forward)
template <int AM>
__device__ __forceinline__ void head_dz(const float* __restrict__ n, const float* __restrict__ dn,
                                        int A, float (&dz)[AM]) {
  float g = 0.f;
#pragma unroll
  for (int a = 0; a < AM; ++a) {
    dz[a] = a < A ? dn[a] : 0.f;
    g += dz[a];
  }
#pragma unroll
  for (int a = 0; a < AM; ++a)
    if (a < A) dz[a] -= expf(n[a]) * g;
}

// Workgroups [0, row_blocks): dx rows, one wavefront per row.  The rest: a 256-column strip of dW
// (and db in the first strip) over a chunk of AH_ROWS rows at a time; `atomic` when the rows are
// split over several workgroups (dW / db zeroed by the caller).
template <int AM>
__global__ __launch_bounds__(256) void action_head_bwd_kernel(
    const float* __restrict__ x, int ldx, const float* __restrict__ w, const float* __restrict__ n,
    const float* __restrict__ dn, int M, int K, int A, float* __restrict__ dx,
    float* __restrict__ dw, float* __restrict__ db, int row_blocks, int col_blocks,
    int rows_per_block, int vec) {
  __shared__ float dzs[AH_ROWS][AM];
  const int tid = threadIdx.x, lane = tid & 63;
  if ((int)blockIdx.x < row_blocks) {
    const int row = blockIdx.x * 4 + (tid >> 6);
    if (row >= M) return;
    float dz[AM];
    head_dz<AM>(n + (long)row * A, dn + (long)row * A, A, dz);
    float* dr = dx + (long)row * K;
    if (vec) {
      for (int c = lane * 4; c < K; c += 256) {
        f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int a = 0; a < AM; ++a)
          if (a < A) {
            const f32x4 wv = *reinterpret_cast<const f32x4*>(w + (long)a * K + c);
            o.x += dz[a] * wv.x;
            o.y += dz[a] * wv.y;
            o.z += dz[a] * wv.z;
            o.w += dz[a] * wv.w;
          }
        *reinterpret_cast<f32x4*>(dr + c) = o;
      }
    } else {
      for (int c = lane; c < K; c += 64) {
        float o = 0.f;
#pragma unroll
        for (int a = 0; a < AM; ++a)
          if (a < A) o += dz[a] * w[(long)a * K + c];
        dr[c] = o;
      }
    }
    return;
  }
  const int cb = ((int)blockIdx.x - row_blocks) % col_blocks;
  const int rb = ((int)blockIdx.x - row_blocks) / col_blocks;
  const int k = cb * 256 + tid;
  const int m0 = rb * rows_per_block;
  const int m1 = m0 + rows_per_block < M ? m0 + rows_per_block : M;
  const bool atomic = rows_per_block < M;
  float accw[AM];
#pragma unroll
  for (int a = 0; a < AM; ++a) accw[a] = 0.f;
  float accb = 0.f;
  for (int base = m0; base < m1; base += AH_ROWS) {
    const int rows = m1 - base < AH_ROWS ? m1 - base : AH_ROWS;
    __syncthreads();
    if (tid < rows) {
      float dz[AM];
      head_dz<AM>(n + (long)(base + tid) * A, dn + (long)(base + tid) * A, A, dz);
#pragma unroll
      for (int a = 0; a < AM; ++a) dzs[tid][a] = dz[a];
    }
    __syncthreads();
    if (k < K) {
      for (int r = 0; r < rows; ++r) {
        const float xv = x[(long)(base + r) * ldx + k];
#pragma unroll
        for (int a = 0; a < AM; ++a) accw[a] += dzs[r][a] * xv;
      }
    }
    if (cb == 0 && tid < A && db != nullptr)
      for (int r = 0; r < rows; ++r) accb += dzs[r][tid];
  }
  if (k < K && dw != nullptr) {
#pragma unroll
    for (int a = 0; a < AM; ++a)
      if (a < A) {
        if (atomic)
          atomicAdd(dw + (long)a * K + k, accw[a]);
        else
          dw[(long)a * K + k] = accw[a];
      }
  }
  if (cb == 0 && tid < A && db != nullptr) {
    if (atomic)
      atomicAdd(db + tid, accb);
    else
      db[tid] = accb;
  }
}

inline bool head_vec(const float* x, int ldx, const float* w, int K, const float* extra) {
  return (K % 4 == 0) && (ldx % 4 == 0) &&
         ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(w) |
           reinterpret_cast<uintptr_t>(extra)) & 15) == 0;
}

}  // namespace

extern "C" int vlnce_action_head_fwd(const float* x, int ldx, const float* w, const float* b, int M,
                                     int K, int A, float* logits_out, int* nan_count,
                                     vlnce_stream_t stream) {
  VLNCE_CHECK_ARG(x && w && logits_out, "action_head_fwd: null argument");
  VLNCE_CHECK_ARG(M > 0 && K > 0 && A > 0 && A <= 16 && ldx >= K,
                  "action_head_fwd: bad shape M %d K %d A %d ldx %d (A <= 16)", M, K, A, ldx);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int vec = head_vec(x, ldx, w, K, nullptr);
  const dim3 grid(ceil_div(M, 4)), block(256);
  if (A <= 4)
    hipLaunchKernelGGL(action_head_fwd_kernel<4>, grid, block, 0, s, x, ldx, w, b, M, K, A,
                       logits_out, nan_count, vec);
  else if (A <= 8)
    hipLaunchKernelGGL(action_head_fwd_kernel<8>, grid, block, 0, s, x, ldx, w, b, M, K, A,
                       logits_out, nan_count, vec);
  else
    hipLaunchKernelGGL(action_head_fwd_kernel<16>, grid, block, 0, s, x, ldx, w, b, M, K, A,
                       logits_out, nan_count, vec);
  VLNCE_CHECK_LAUNCH("action_head_fwd");
  return 0;
}

extern "C" int vlnce_action_head_bwd(const float* x, int ldx, const float* w, const float* logits,
                                     const float* dlogits, int M, int K, int A, float* dx,
                                     float* dw, float* db, vlnce_stream_t stream) {
  VLNCE_CHECK_ARG(x && w && logits && dlogits, "action_head_bwd: null argument");
  VLNCE_CHECK_ARG(M > 0 && K > 0 && A > 0 && A <= 16 && ldx >= K,
                  "action_head_bwd: bad shape M %d K %d A %d ldx %d (A <= 16)", M, K, A, ldx);
  VLNCE_CHECK_ARG(dw != nullptr || db == nullptr, "action_head_bwd: db without dw");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int row_blocks = dx ? ceil_div(M, 4) : 0;
  const int col_blocks = dw ? ceil_div(K, 256) : 0;
  // rows of the weight gradient per workgroup: everything in one workgroup per strip up to 1024
  // rows (a policy step: M = num_envs), otherwise 512-row chunks added atomically
  const int rows_per_block = M <= 1024 ? M : 512;
  const int chunks = dw ? ceil_div(M, rows_per_block) : 0;
  if (chunks > 1) {
    vlnce_zero(dw, A, K, K, s);
    if (db) vlnce_zero(db, 1, A, A, s);
  }
  const int vec = head_vec(x, ldx, w, K, dx);
  const dim3 grid(row_blocks + col_blocks * chunks), block(256);
  if (grid.x == 0) return 0;
  if (A <= 4)
    hipLaunchKernelGGL(action_head_bwd_kernel<4>, grid, block, 0, s, x, ldx, w, logits, dlogits, M,
                       K, A, dx, dw, db, row_blocks, col_blocks, rows_per_block, vec);
  else if (A <= 8)
    hipLaunchKernelGGL(action_head_bwd_kernel<8>, grid, block, 0, s, x, ldx, w, logits, dlogits, M,
                       K, A, dx, dw, db, row_blocks, col_blocks, rows_per_block, vec);
  else
    hipLaunchKernelGGL(action_head_bwd_kernel<16>, grid, block, 0, s, x, ldx, w, logits, dlogits, M,
                       K, A, dx, dw, db, row_blocks, col_blocks, rows_per_block, vec);
  VLNCE_CHECK_LAUNCH("action_head_bwd");
  return 0;
}
