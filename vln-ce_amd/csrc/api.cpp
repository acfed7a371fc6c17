// Error string + version of libvlnce_hip.so.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <atomic>

#include "common_opts.h"

static thread_local char g_err[512] = "";

void vlnce_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* vlnce_last_error(void) { return g_err; }
// major*100 + minor; 1.x: round-1 ABI (centered normalisation vectors, dual-input prologue,
// backward / data-path / returns entry points).  Struct layouts only ever grow at the end.
// 134: vlnce_set_option / vlnce_get_option (the library no longer reads environment variables).
// 136: vlnce_epilogue.bn (train-mode BatchNorm column sums added by the convolution) +
// vlnce_bn_finalize_sums.
// 137: vlnce_stem7_fwd.
// 138: vlnce_action_head_fwd / _bwd.
// 139: vlnce_attn_fwd_shared / _bwd_shared, vlnce_segment_sum.
// 140: option "m3" (conv_m3_kernel).
// 141: vlnce_rnn_seq_fwd2 / _bwd2 / _wgrad, vlnce_linear_rows_fwd / _bwd, vlnce_ppo_loss, vlnce_prologue.options (per-launch dispatch options).
// 142: plane format 2 (fp16 planes, three plane products per multiply): vlnce_prologue.w_format, a
// `format` argument of vlnce_conv2d_split_weights / _pack_weights, `w_format` of vlnce_stem7_fwd;
// option "conv_math" defaults to 2.
// 143: vlnce_bn_bwd takes a workspace (vlnce_bn_bwd_workspace_floats): partial sums instead of atomics;
// vlnce_conv2d_prepare_weights / vlnce_weight_job (all weight images of a trainable trunk in one launch).
extern "C" int vlnce_version(void) { return 143; }

// ---- dispatch options: one int per name, process-wide, relaxed atomics (a tuning / test knob,
// not a synchronisation point: set them before the launches they are meant for)
namespace {
struct OptDef {
  const char* name;
  int def;
};
const OptDef kOpts[VLNCE_OPT_COUNT] = {
    {"conv_math", 2},   {"p3", 2},          {"p3_tile", 0},         {"s3", 1},
    {"u3", 1},          {"u3_waves", 8},    {"x3_tile", 0},         {"igemm_tile", 0},
    {"igemm_nobuf", 0}, {"igemm_no_splitk", 0}, {"wgrad_tile", 64}, {"rollout_one_xcd", 0},
    {"m3", 1},
};
std::atomic<int> g_opt[VLNCE_OPT_COUNT] = {
    {2}, {2}, {0}, {1}, {1}, {8}, {0}, {0}, {0}, {0}, {64}, {0}, {1},
};
int opt_index(const char* name) {
  if (name)
    for (int i = 0; i < VLNCE_OPT_COUNT; ++i)
      if (strcmp(name, kOpts[i].name) == 0) return i;
  return -1;
}
}  // namespace

namespace {
thread_local const int* t_call_opts = nullptr;   // vlnce_prologue.options of the launch being dispatched
}
VlnceOptScope::VlnceOptScope(const int* per_call) : prev(t_call_opts) { t_call_opts = per_call; }
VlnceOptScope::~VlnceOptScope() { t_call_opts = prev; }

int vlnce_opt(int id) {
  if (t_call_opts && t_call_opts[id] >= 0) return t_call_opts[id];
  return g_opt[id].load(std::memory_order_relaxed);
}

extern "C" int vlnce_option_count(void) { return VLNCE_OPT_COUNT; }

extern "C" int vlnce_option_index(const char* name) { return opt_index(name); }

extern "C" int vlnce_set_option(const char* name, int value) {
  const int i = opt_index(name);
  if (i < 0) {
    vlnce_set_error("vlnce_set_option: unknown option '%s'", name ? name : "(null)");
    return 1;
  }
  g_opt[i].store(value, std::memory_order_relaxed);
  return 0;
}

extern "C" int vlnce_get_option(const char* name, int* value) {
  const int i = opt_index(name);
  if (i < 0 || !value) {
    vlnce_set_error("vlnce_get_option: unknown option '%s'", name ? name : "(null)");
    return 1;
  }
  *value = g_opt[i].load(std::memory_order_relaxed);
  return 0;
}

extern "C" int vlnce_option_default(const char* name, int* value) {
  const int i = opt_index(name);
  if (i < 0 || !value) {
    vlnce_set_error("vlnce_option_default: unknown option '%s'", name ? name : "(null)");
    return 1;
  }
  *value = kOpts[i].def;
  return 0;
}
