// Error string + version of libvlnce_hip.so.
#include <stdarg.h>
#include <stdio.h>

#include "../../include/vlnce_hip.h"

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
extern "C" int vlnce_version(void) { return 133; }
