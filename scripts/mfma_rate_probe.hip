// Matrix-pipe issue-rate probe for gfx950: cycles per v_mfma_f32_32x32x16_bf16 in the access
// patterns conv_x3_kernel's matrix waves use.  hipcc --offload-arch=gfx950 -O3 -o probe this.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// MODE 0: 24 MFMAs / iteration, 4 accumulators, operands constant in registers
// MODE 1: + 12 ds_read_b128 per 24 MFMAs (fragments double-buffered as in the kernel)
// MODE 2: as 1 with one s_barrier per 48 MFMAs (all waves of the block)
// MODE 3: 24 MFMAs on ONE accumulator set of 2 (dependency distance 2)
template <int MODE>
__global__ __launch_bounds__(256) void probe(long long* out, int iters, float* sink) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 16384; i += blockDim.x) reinterpret_cast<unsigned*>(lds)[i] = 0x3f803f80u;
  __syncthreads();
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  bf16x8 fa[2][3], fb[2][3], ga[2][3], gb[2][3];
  const char* base = lds + (lane & 31) * 80 + (lane >> 5) * 16;
  auto rd = [&](bf16x8 (&a)[2][3], bf16x8 (&b)[2][3], int off) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        a[i][q] = *reinterpret_cast<const bf16x8*>(base + off + q * 10240 + i * 2560);
        b[i][q] = *reinterpret_cast<const bf16x8*>(base + off + 30720 + q * 10240 + i * 2560);
      }
  };
  auto mma = [&](const bf16x8 (&a)[2][3], const bf16x8 (&b)[2][3]) {
    constexpr int PA[6] = {2, 0, 1, 1, 0, 0};
    constexpr int PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
    for (int q = 0; q < 6; ++q)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int k = MODE == 3 ? (j & 1) : i * 2 + j;
          acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][PA[q]], b[j][PB[q]], acc[k], 0, 0, 0);
        }
  };
  rd(fa, fb, 0);
  rd(ga, gb, 32);
  __syncthreads();
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 1 || MODE == 2) rd(ga, gb, 32);
    __builtin_amdgcn_sched_barrier(0);
    mma(fa, fb);
    __builtin_amdgcn_sched_barrier(0);
    if (MODE == 2) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (MODE == 1 || MODE == 2) rd(fa, fb, (it & 1) * 61440 % 60000);
    __builtin_amdgcn_sched_barrier(0);
    mma(ga, gb);
    __builtin_amdgcn_sched_barrier(0);
  }
  const long long t1 = clock64();
  float s = 0.f;
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  if (s == 12345.f) sink[0] = s;
  if (tid == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}

template <int MODE>
void run(const char* name, int threads, int blocks) {
  long long* d;
  float* sink;
  hipMalloc(&d, 8);
  hipMalloc(&sink, 4);
  const int iters = 200;
  hipFuncSetAttribute(reinterpret_cast<const void*>(probe<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 122880);
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(threads), 122880, 0, d, iters, sink);
  hipDeviceSynchronize();
  long long h = 0;
  hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
  printf("%-58s threads %4d blocks %4d: %6.1f cycles / MFMA\n", name, threads, blocks, (double)h / (iters * 48.0));
  hipFree(d);
  hipFree(sink);
}

int main() {
  run<0>("registers only, 4 accumulators", 256, 256);
  run<0>("registers only, 4 accumulators, 1 block", 256, 1);
  run<0>("registers only, 4 accumulators, 1 wave", 64, 1);
  run<3>("registers only, 2 accumulators", 256, 256);
  run<1>("+ 12 ds_read_b128 per 24 MFMA", 256, 256);
  run<1>("+ 12 ds_read_b128 per 24 MFMA, 1 block", 256, 1);
  run<2>("+ ds_read + 1 s_barrier per 48 MFMA", 256, 256);
  return 0;
}
