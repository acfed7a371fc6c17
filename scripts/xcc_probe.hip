// Which XCD / SE / CU does each workgroup of a launch land on?  (scripts/cumask_probe.py)
#include <hip/hip_runtime.h>
#include <stdint.h>

__global__ void xcc_probe_kernel(uint32_t* out, int spin) {
  if (threadIdx.x == 0) {
    const uint32_t xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);   // HW_REG_XCC_ID[3:0]
    const uint32_t hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);    // HW_REG_HW_ID
    out[blockIdx.x] = (xcc << 24) | (hw & 0xffffff);
  }
  // keep the workgroup resident for a while so that a launch spreads over the whole mask
  long long t0 = clock64();
  while (clock64() - t0 < spin) {
  }
}

extern "C" int xcc_probe(uint32_t* out, int nwg, int threads, int lds_bytes, int spin, void* stream) {
  hipLaunchKernelGGL(xcc_probe_kernel, dim3(nwg), dim3(threads), lds_bytes, (hipStream_t)stream, out,
                     spin);
  return (int)hipGetLastError();
}
