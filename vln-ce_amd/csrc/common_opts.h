// Error string and dispatch options: the part of common.h that plain C++ (api.cpp) can include.
#pragma once
#include "../../include/vlnce_hip.h"

void vlnce_set_error(const char* fmt, ...);

// Dispatch options (vlnce_set_option / vlnce_get_option in include/vlnce_hip.h; table in api.cpp).
// The library never reads the environment: the host sets what it wants, tests set and restore.
enum {
  VLNCE_OPT_CONV_MATH,        // 0 = fp32 MFMA everywhere; non-zero = the plane kernels (the host packs the weights as
                              // 1 = three bf16 planes / six products, 2 = fp16 planes / three products: default)
  VLNCE_OPT_P3,               // conv_p3_kernel: 0 off, 1 every layer it covers, 2 KxK only (default), 3 1x1 only
  VLNCE_OPT_P3_TILE,          // 0 = by CU fill, 1..6 = forced tile
  VLNCE_OPT_S3,               // conv_s3_kernel: 0 off, 1 default rule, 2 every eligible shape
  VLNCE_OPT_U3,               // conv_u3_kernel: 0 off, 1 default rule, 2 / 3 = force 64- / 128-row tiles
  VLNCE_OPT_U3_WAVES,         // 8 (default) or 4
  VLNCE_OPT_X3_TILE,          // conv_x3_kernel: 0 = by CU fill, 1..4 = forced tile
  VLNCE_OPT_IGEMM_TILE,       // igemm_kernel: 0 = rule, 1 128x128, 2 128x64, 3 64x64
  VLNCE_OPT_IGEMM_NOBUF,      // 1 = no buffer-descriptor operand path
  VLNCE_OPT_IGEMM_NO_SPLITK,  // 1 = no split-K
  VLNCE_OPT_WGRAD_TILE,       // 64 (default) or 128
  VLNCE_OPT_ROLLOUT_ONE_XCD,  // 1 = the GRU rollout's workgroups on one XCD
  VLNCE_OPT_M3,               // conv_m3_kernel: 0 off, 1 the small launches (default), 2 every layer it covers
  VLNCE_OPT_COUNT
};
int vlnce_opt(int id);

// Options of ONE launch (vlnce_prologue.options: VLNCE_OPT_COUNT ints, negative = the process value):
// an entry point that takes them installs them for the calling thread while it dispatches.
struct VlnceOptScope {
  explicit VlnceOptScope(const int* per_call);
  ~VlnceOptScope();
  const int* prev;
};
