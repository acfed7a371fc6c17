// How fast does HBM deliver a [rows x K] fp32 matrix when a workgroup owns 64 consecutive rows and
// walks them K-chunk by K-chunk (the access order of conv_u3 / conv_x3: per step every row gives
// CH bytes, the next step the next CH bytes of the same rows), as a function of CH?
//   hipcc --offload-arch=gfx950 -O3 scripts/rowchunk_probe.hip -o /tmp/rowchunk_probe && /tmp/rowchunk_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int CH16>   // 16-byte pieces per row and step: 8 = 128 B, 16 = 256 B, 32 = 512 B
__global__ __launch_bounds__(512) void walk(const f32x4* __restrict__ x, float* out, int rows, int K4, int depth) {
  // 512 threads: (512 / CH16) rows per pass, 64 rows per tile -> 64 * CH16 / 512 passes per step
  constexpr int RPP = 512 / CH16, PASSES = 64 / RPP;
  const int tid = threadIdx.x, r = tid / CH16, c = tid % CH16;
  f32x4 acc = {0, 0, 0, 0};
  for (int tile = blockIdx.x; tile < rows / 64; tile += gridDim.x) {
    const f32x4* base = x + (long)tile * 64 * K4;
    for (int k = 0; k < K4; k += CH16) {
#pragma unroll
      for (int p = 0; p < PASSES; ++p) {
        const f32x4 v = base[(long)(p * RPP + r) * K4 + k + c];
        acc += v;
      }
    }
  }
  if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.f) out[0] = acc[0];
}
int main() {
  const int K = 1024, rows = 16384 * 8;   // 512 MB: beyond the Infinity Cache
  f32x4* x; float* out;
  hipMalloc(&x, (size_t)rows * K * 4); hipMalloc(&out, 4);
  hipMemset(x, 0, (size_t)rows * K * 4);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  auto run = [&](auto kern, const char* name) {
    for (int it = 0; it < 2; ++it) {
      hipEventRecord(a);
      hipLaunchKernelGGL(kern, dim3(256), dim3(512), 0, 0, x, out, rows, K / 4, 0);
      hipEventRecord(b); hipEventSynchronize(b);
    }
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("%-28s %.3f ms  %.2f TB/s\n", name, ms, (double)rows * K * 4 / ms / 1e9);
  };
  run(walk<8>, "128 B per row and step");
  run(walk<16>, "256 B per row and step");
  run(walk<32>, "512 B per row and step");
  run(walk<64>, "1 KB per row and step");
  return 0;
}
