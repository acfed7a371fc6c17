// All the weight images a TRAINABLE trunk needs for one step, in ONE launch
// (vlnce_conv2d_prepare_weights): with MODEL.*_ENCODER.trainable the filters change every step, and
// per convolution the host used to issue eight small launches (OIHW -> OHWI, planes and fragments in
// the forward format, flip + transpose, planes and fragments of the data-gradient bank in format 1)
// -- ~900 launches of ~4 us each per step for the two ResNet-50 trunks, more GPU time than the
// bytes they move (profiles/r06_m_trainable_step_kernels.txt).  A job table on the device names
// every output; a work item is eight consecutive input channels of one (filter, tap).
#include "igemm_shared.h"

using namespace vlnce_detail;

namespace {

enum { WP_F32 = 0, WP_PLANES = 1, WP_FRAGMENTS = 2 };

__device__ __forceinline__ void store8(unsigned short* dst, const unsigned short (&v)[8]) {
  uint4 u;
  u.x = v[0] | ((unsigned)v[1] << 16);
  u.y = v[2] | ((unsigned)v[3] << 16);
  u.z = v[4] | ((unsigned)v[5] << 16);
  u.w = v[6] | ((unsigned)v[7] << 16);
  *reinterpret_cast<uint4*>(dst) = u;
}

__global__ __launch_bounds__(256) void weight_prep_kernel(const vlnce_weight_job* __restrict__ jobs,
                                                          const long* __restrict__ first_item,
                                                          int njobs, long total) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += gridDim.x * 256L) {
    int lo = 0, hi = njobs;  // the job whose item range holds i
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (first_item[mid] <= i) lo = mid; else hi = mid;
    }
    const vlnce_weight_job jb = jobs[lo];
    const long r = i - first_item[lo];
    const int T = jb.T, tr = jb.transposed;
    const int N = tr ? jb.Cin : jb.Cout;   // filters of the bank this job writes
    const int Cp = tr ? jb.Cout : jb.Cin;  // its channels
    int n, t, c0;
    long rest = 0;
    int lane = 0;
    if (jb.kind == WP_FRAGMENTS) {  // (nb, ks, lane) as pack_weights_kernel (conv_p3.hip)
      lane = (int)(r & 63);
      rest = r >> 6;
      const int KS = T * Cp / 16;
      const int ks = (int)(rest % KS), nb = (int)(rest / KS);
      const int s = ks & 1, ct = ks >> 1;
      t = ct % T;
      n = nb * 32 + (lane & 31);
      c0 = (ct / T) * 32 + s * 16 + (lane >> 5) * 8;
    } else {
      const int groups = Cp / 8;
      c0 = (int)(r % groups) * 8;
      const long q = r / groups;
      t = (int)(q % T);
      n = (int)(q / T);
    }
    float v[8];
    const float* w = jb.w_oihw;
#pragma unroll
    for (int e = 0; e < 8; ++e)
      v[e] = tr ? w[((long)(c0 + e) * jb.Cin + n) * T + (T - 1 - t)]   // [Cin,T,Cout], taps reversed
                : w[((long)n * jb.Cin + c0 + e) * T + t];
    if (jb.kind == WP_F32) {
      float* dst = reinterpret_cast<float*>(jb.dst) + ((long)n * T + t) * Cp + c0;
      *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
      *reinterpret_cast<float4*>(dst + 4) = make_float4(v[4], v[5], v[6], v[7]);
      continue;
    }
    unsigned short out[3][8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      unsigned short o3[3];
      if (jb.format == MATH_F16X3) split_weight<MATH_F16X3>(v[e], o3);
      else split_weight<MATH_BF16X6>(v[e], o3);
#pragma unroll
      for (int q = 0; q < 3; ++q) out[q][e] = o3[q];
    }
    unsigned short* base = reinterpret_cast<unsigned short*>(jb.dst);
    if (jb.kind == WP_PLANES) {
      const long count = (long)N * T * Cp;
      const long at = ((long)n * T + t) * Cp + c0;
#pragma unroll
      for (int q = 0; q < 3; ++q) store8(base + q * count + at, out[q]);
    } else {
      unsigned short* dst = base + ((rest * 3) * 64 + lane) * 8;
#pragma unroll
      for (int q = 0; q < 3; ++q) store8(dst + (long)q * 512, out[q]);
    }
  }
}

}  // namespace

extern "C" long vlnce_weight_job_items(const vlnce_weight_job* job) {
  if (!job || job->Cout <= 0 || job->Cin <= 0 || job->T <= 0) return -1;
  const int N = job->transposed ? job->Cin : job->Cout, Cp = job->transposed ? job->Cout : job->Cin;
  if (job->kind < WP_F32 || job->kind > WP_FRAGMENTS) return -1;
  if (Cp % 8 != 0) return -1;
  if (job->kind == WP_FRAGMENTS && (N % 32 != 0 || Cp % 32 != 0)) return -1;
  if (job->kind != WP_F32 && job->format != MATH_BF16X6 && job->format != MATH_F16X3) return -1;
  return (long)N * job->T * Cp / 8;
}

extern "C" int vlnce_conv2d_prepare_weights(const vlnce_weight_job* jobs_dev,
                                            const long* first_item_dev, int njobs,
                                            long total_items, vlnce_stream_t stream) {
  VLNCE_CHECK_ARG(jobs_dev && first_item_dev && njobs > 0 && total_items > 0,
                  "conv2d_prepare_weights: bad argument");
  long blocks = (total_items + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(weight_prep_kernel, dim3((unsigned)blocks), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), jobs_dev, first_item_dev, njobs,
                     total_items);
  VLNCE_CHECK_LAUNCH("conv2d_prepare_weights");
  return 0;
}
