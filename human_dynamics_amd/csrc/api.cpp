// libhmmr_hip.so: ABI version + per-thread error string (include/hmmr_hip.h).
#include <stdarg.h>
#include <stdio.h>

#include "hmmr_hip.h"

static thread_local char g_err[512] = "";

void hmmr_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int hmmr_abi_version(void) { return HMMR_ABI_VERSION; }
extern "C" const char* hmmr_last_error(void) { return g_err; }
