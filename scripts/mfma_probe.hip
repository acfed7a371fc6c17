// Microbenchmark: fp32 MFMA (32x32x2) issue rate on gfx950 under the igemm kernel's
// conditions: W waves per workgroup-slot, optional ds_read_b128 per 4 MFMAs, optional barrier
// every 64 MFMAs.  Build: hipcc --offload-arch=gfx950 -O3 scripts/mfma_probe.hip -o scripts/mfma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>  // 0: MFMA only; 1: + ds_read frags; 2: + barrier per 64 MFMA
__global__ __launch_bounds__(256, 2) void probe(float* out, int iters, int lds_floats) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  for (int i = threadIdx.x; i < lds_floats; i += 256) smem[i] = (float)(i & 7) * 0.001f;
  __syncthreads();
  f32x16 acc[2][2];
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 2; ++j)
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int lane = threadIdx.x & 63;
  const float* base = smem + (lane & 31) * 36 + 4 * (lane >> 5);
  f32x4 a0 = {1.f, 2.f, 3.f, 4.f}, a1 = a0, b0 = a0, b1 = a0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      if (MODE >= 1) {
        a0 = *reinterpret_cast<const f32x4*>(base + 8 * g);
        a1 = *reinterpret_cast<const f32x4*>(base + 32 * 36 + 8 * g);
        b0 = *reinterpret_cast<const f32x4*>(base + 128 * 36 + 8 * g);
        b1 = *reinterpret_cast<const f32x4*>(base + 160 * 36 + 8 * g);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[e], b0[e], acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[e], b1[e], acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[e], b0[e], acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[e], b1[e], acc[1][1], 0, 0, 0);
      }
    }
    if (MODE >= 2) __syncthreads();
  }
  float s = 0.f;
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 2; ++j)
      for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name, int wgs, int lds_bytes, float* out) {
  const int iters = 2000;
  hipFuncSetAttribute(reinterpret_cast<const void*>(probe<MODE>),
                      hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  probe<MODE><<<wgs, 256, lds_bytes>>>(out, 10, 256 * 36);
  hipEventRecord(e0);
  probe<MODE><<<wgs, 256, lds_bytes>>>(out, iters, 256 * 36);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double flop = (double)wgs * 4 * iters * 64 * 4096.0;
  printf("%-28s wgs=%4d lds=%6d  %8.3f ms  %7.1f TF/s\n", name, wgs, lds_bytes, ms, flop / ms / 1e9);
}

int main() {
  float* out;
  hipMalloc(&out, 4096 * 256 * 4);
  const int small = 256 * 36 * 4;  // 36.9 KB (any number of WGs per CU up to regs)
  const int big = 73728;           // 2 WGs per CU
  const int huge = 120000;         // 1 WG per CU
  run<0>("mfma only", 512, big, out);
  run<0>("mfma only 1wg/cu", 256, huge, out);
  run<0>("mfma only 4096 wgs", 4096, big, out);
  run<1>("mfma + ds_read", 512, big, out);
  run<1>("mfma + ds_read 1wg/cu", 256, huge, out);
  run<2>("mfma + ds_read + barrier", 512, big, out);
  run<2>("same, 1wg/cu", 256, huge, out);
  run<2>("same, 3-4 wg/cu", 1024, small, out);
  return 0;
}
