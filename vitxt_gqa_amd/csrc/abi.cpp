// ABI bookkeeping of libt2s_hip.so: version and the thread-local error slot.
#include <stdarg.h>
#include <stdio.h>

#include "../../include/t2s_hip.h"

static thread_local char g_err[512] = "";

void t2s_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int t2s_abi_version(void) { return T2S_ABI_VERSION; }
extern "C" const char* t2s_last_error(void) { return g_err; }
