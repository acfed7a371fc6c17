// Probe: fp32 GEMM core with LDS-DMA operand staging (buffer_load ... lds), chunk-swizzled
// unpadded LDS rows, an NSTAGE-deep ring with counted vmcnt + one raw s_barrier per K-tile.
//   C[M,N] = A[M,K] * B[N,K]^T        (the 1x1-convolution shapes of the ResNet trunks)
// Stand-alone: hipcc --offload-arch=gfx950 -O3 scripts/dma_gemm_probe.hip -o build/dma_gemm_probe
//              build/dma_gemm_probe            (prints TF/s per shape, checks against a plain kernel)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void lds_void;

#define CHECK(x)                                                                  \
  do {                                                                            \
    hipError_t e = (x);                                                           \
    if (e != hipSuccess) {                                                        \
      printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); \
      exit(1);                                                                    \
    }                                                                             \
  } while (0)

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// BM x BN block tile, 4 waves (2x2), BK floats per stage, NSTAGE ring
template <int BM, int BN, int BK, int NSTAGE>
__global__ __launch_bounds__(256, 2) void dma_gemm(const float* __restrict__ A,
                                                   const float* __restrict__ B,
                                                   float* __restrict__ C, int M, int N, int K,
                                                   int tiles_n) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int WTM = BM / 2, WTN = BN / 2, MT = WTM / 32, NT = WTN / 32;
  constexpr int RB = BK * 4;              // bytes per LDS row
  constexpr int CPR = BK / 4;             // 16-byte chunks per row
  constexpr int RPI = 1024 / RB;          // rows filled by one wave-wide DMA instruction
  constexpr int RPB = 256 / RB;           // rows per 256-byte LDS bank row
  constexpr int A_BYTES = BM * RB, B_BYTES = BN * RB, STAGE_BYTES = A_BYTES + B_BYTES;
  constexpr int A_INSTR = BM / RPI / 4, B_INSTR = BN / RPI / 4;  // per wave
  constexpr int IPW = A_INSTR + B_INSTR;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, half = lane >> 5, l31 = lane & 31;

  // XCD-aware tile order (bijective)
  int tile;
  {
    const int bid = blockIdx.x, nwg = gridDim.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tile_m = tile / tiles_n, tile_n = tile - tile_m * tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(A), 0, (int)((long)M * K * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_b = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(B), 0, (int)((long)N * K * 4), 0x00020000);

  // DMA lane mapping: instruction j of this wave fills rows [ (j*4 + wave) * RPI, +RPI )
  const int drow = lane / CPR;                  // row inside the instruction's group
  const int dchunk = lane % CPR;                // PHYSICAL chunk this lane writes
  int a_voff[A_INSTR], b_voff[B_INSTR];
#pragma unroll
  for (int j = 0; j < A_INSTR; ++j) {
    const int row = (j * 4 + wave) * RPI + drow;
    const int c = dchunk ^ ((row / RPB) & (CPR - 1));  // logical chunk stored at this position
    const int m = m0 + row;
    a_voff[j] = m < M ? (int)(((long)m * K + c * 4) * 4) : (int)0x80000000;
  }
#pragma unroll
  for (int j = 0; j < B_INSTR; ++j) {
    const int row = (j * 4 + wave) * RPI + drow;
    const int c = dchunk ^ ((row / RPB) & (CPR - 1));
    const int n = n0 + row;
    b_voff[j] = n < N ? (int)(((long)n * K + c * 4) * 4) : (int)0x80000000;
  }

  auto issue = [&](int kt, int buf) {
    char* base = smem + buf * STAGE_BYTES;
    const int soff = kt * BK * 4;
#pragma unroll
    for (int j = 0; j < A_INSTR; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (lds_void*)(base + (j * 4 + wave) * 1024), 16,
                                           a_voff[j], soff, 0, 0);
#pragma unroll
    for (int j = 0; j < B_INSTR; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b,
                                           (lds_void*)(base + A_BYTES + (j * 4 + wave) * 1024), 16,
                                           b_voff[j], soff, 0, 0);
  };

  // fragment read offsets (bytes inside a stage), chunk swizzle folded in per k-group
  int a_rowoff[MT], a_sw[MT], b_rowoff[NT], b_sw[NT];
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int row = wm * WTM + i * 32 + l31;
    a_rowoff[i] = row * RB;
    a_sw[i] = (row / RPB) & (CPR - 1);
  }
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int row = wn * WTN + j * 32 + l31;
    b_rowoff[j] = A_BYTES + row * RB;
    b_sw[j] = (row / RPB) & (CPR - 1);
  }

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int KT = K / BK;
#pragma unroll
  for (int s = 0; s < NSTAGE - 1; ++s)
    if (s < KT) issue(s, s);

  for (int t = 0; t < KT; ++t) {
    // stage t landed (mine), then everyone's; the stages t+1 .. t+NSTAGE-2 may stay in flight
    if (t + NSTAGE - 2 < KT) wait_vmcnt<(NSTAGE - 2) * IPW>();
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    if (t + NSTAGE - 1 < KT) issue(t + NSTAGE - 1, (t + NSTAGE - 1) % NSTAGE);
    const char* cur = smem + (t % NSTAGE) * STAGE_BYTES;
#pragma unroll
    for (int g = 0; g < BK / 8; ++g) {
      f32x4 af[MT], bf[NT];
#pragma unroll
      for (int i = 0; i < MT; ++i)
        af[i] = *reinterpret_cast<const f32x4*>(cur + a_rowoff[i] + (((2 * g + half) ^ a_sw[i]) << 4));
#pragma unroll
      for (int j = 0; j < NT; ++j)
        bf[j] = *reinterpret_cast<const f32x4*>(cur + b_rowoff[j] + (((2 * g + half) ^ b_sw[j]) << 4));
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][e], bf[j][e], acc[i][j], 0, 0, 0);
    }
  }

  // epilogue: transpose through LDS in two halves of 64 rows, 16-byte row stores
  constexpr int LDC = BN + 4;
  float* Ct = reinterpret_cast<float*>(smem);
  constexpr int TPR = BN / 4, RPP = 256 / TPR;
  const int c4 = (tid % TPR) * 4;
#pragma unroll
  for (int ep = 0; ep < 2; ++ep) {
    __syncthreads();
    if (wm == ep) {
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            Ct[row * LDC + wn * WTN + j * 32 + l31] = acc[i][j][r];
          }
    }
    __syncthreads();
    if (n0 + c4 < N) {
#pragma unroll 4
      for (int rr = tid / TPR; rr < WTM; rr += RPP) {
        const int row = m0 + ep * WTM + rr;
        if (row >= M) break;
        *reinterpret_cast<f32x4*>(C + (long)row * N + n0 + c4) =
            *reinterpret_cast<const f32x4*>(Ct + rr * LDC + c4);
      }
    }
  }
#endif
}

__global__ void ref_gemm(const float* A, const float* B, float* C, int M, int N, int K) {
  const long i = blockIdx.x * 256L + threadIdx.x;
  if (i >= (long)M * N) return;
  const int m = i / N, n = i % N;
  float s = 0.f;
  for (int k = 0; k < K; ++k) s = fmaf(A[(long)m * K + k], B[(long)n * K + k], s);
  C[i] = s;
}

template <int BM, int BN, int BK, int NSTAGE>
float run(const float* A, const float* B, float* C, int M, int N, int K, int iters) {
  constexpr int smem_bytes = NSTAGE * (BM + BN) * BK * 4 > 64 * (BN + 4) * 4
                                 ? NSTAGE * (BM + BN) * BK * 4
                                 : 64 * (BN + 4) * 4;
  auto kern = dma_gemm<BM, BN, BK, NSTAGE>;
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                            hipFuncAttributeMaxDynamicSharedMemorySize, smem_bytes));
  const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i)
    hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n), dim3(256), smem_bytes, 0, A, B, C, M, N, K, tiles_n);
  CHECK(hipEventRecord(e0));
  for (int i = 0; i < iters; ++i)
    hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n), dim3(256), smem_bytes, 0, A, B, C, M, N, K, tiles_n);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  CHECK(hipGetLastError());
  float ms;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1e3f / iters;
}

int main() {
  struct Shape { const char* name; int M, K, N; };
  const Shape shapes[] = {{"l1_1x1_64_256", 262144, 64, 256},   {"l1_1x1_256_64", 262144, 256, 64},
                          {"l2_1x1_128_512", 65536, 128, 512},  {"l2_1x1_512_128", 65536, 512, 128},
                          {"l3_1x1_256_1024", 16384, 256, 1024}, {"l3_1x1_1024_256", 16384, 1024, 256},
                          {"l4_1x1_512_2048", 4096, 512, 2048}, {"l4_1x1_2048_512", 4096, 2048, 512},
                          {"gemm_4096^3", 4096, 4096, 4096}};
  const long maxA = 262144L * 256, maxB = 4096L * 4096, maxC = 262144L * 256;
  float *A, *B, *C, *R;
  CHECK(hipMalloc(&A, maxA * 4));
  CHECK(hipMalloc(&B, maxB * 4));
  CHECK(hipMalloc(&C, maxC * 4));
  CHECK(hipMalloc(&R, maxC * 4));
  std::vector<float> h(maxB > maxA ? maxB : maxA);
  srand(1);
  for (auto& v : h) v = (rand() / (float)RAND_MAX) * 2.f - 1.f;
  CHECK(hipMemcpy(A, h.data(), maxA * 4, hipMemcpyHostToDevice));
  for (auto& v : h) v = (rand() / (float)RAND_MAX) * 2.f - 1.f;
  CHECK(hipMemcpy(B, h.data(), maxB * 4, hipMemcpyHostToDevice));
  typedef float (*RunFn)(const float*, const float*, float*, int, int, int, int);
  struct Var { const char* name; RunFn fn; };
  const Var vars[] = {{"128x128 k32 s2", run<128, 128, 32, 2>}, {"128x128 k16 s4", run<128, 128, 16, 4>},
                      {"128x128 k16 s3", run<128, 128, 16, 3>}, {"128x128 k16 s2", run<128, 128, 16, 2>},
                      {"128x128 k32 s3", run<128, 128, 32, 3>}, {"128x64 k32 s3", run<128, 64, 32, 3>},
                      {"128x64 k32 s2", run<128, 64, 32, 2>},   {"128x64 k16 s4", run<128, 64, 16, 4>}};
  printf("%-16s", "shape (M,K,N)");
  for (const Var& v : vars) printf(" | %-16s", v.name);
  printf("   [us / TF/s]\n");
  for (const Shape& s : shapes) {
    const double fl = 2.0 * s.M * s.K * s.N;
    // correctness of the first variant on a sub-problem
    {
      const int M = s.M < 1024 ? s.M : 1024;
      hipLaunchKernelGGL(ref_gemm, dim3(((long)M * s.N + 255) / 256), dim3(256), 0, 0, A, B, R, M, s.N, s.K);
      run<128, 128, 32, 2>(A, B, C, M, s.N, s.K, 1);
      std::vector<float> c((long)M * s.N), r((long)M * s.N);
      CHECK(hipMemcpy(c.data(), C, c.size() * 4, hipMemcpyDeviceToHost));
      CHECK(hipMemcpy(r.data(), R, r.size() * 4, hipMemcpyDeviceToHost));
      double worst = 0;
      for (size_t i = 0; i < c.size(); ++i) worst = fmax(worst, fabs((double)c[i] - r[i]));
      run<128, 128, 16, 4>(A, B, C, M, s.N, s.K, 1);
      CHECK(hipMemcpy(c.data(), C, c.size() * 4, hipMemcpyDeviceToHost));
      double worst2 = 0;
      for (size_t i = 0; i < c.size(); ++i) worst2 = fmax(worst2, fabs((double)c[i] - r[i]));
      if (worst > 1e-3 * sqrt((double)s.K) || worst2 > 1e-3 * sqrt((double)s.K))
        printf("  !! mismatch vs reference: max|d| = %g / %g\n", worst, worst2);
    }
    printf("%-16s", s.name);
    for (const Var& v : vars) {
      const float t = v.fn(A, B, C, s.M, s.N, s.K, 20);
      printf(" | %7.1f %7.1f ", t, fl / t / 1e6);
    }
    printf("\n");
    fflush(stdout);
  }
  return 0;
}
