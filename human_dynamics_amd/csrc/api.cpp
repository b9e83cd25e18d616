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

// Development switches: one process-wide struct, all zero = product defaults.  Read by the launch code through
// hmmr_debug_state(); written only by hmmr_set_debug (tests and A/B tools), never from the environment.
static hmmr_debug_t g_debug = {};
const hmmr_debug_t* hmmr_debug_state() { return &g_debug; }
extern "C" void hmmr_set_debug(const hmmr_debug_t* d) { g_debug = d ? *d : hmmr_debug_t{}; }
extern "C" void hmmr_get_debug(hmmr_debug_t* d) { if (d) *d = g_debug; }

extern "C" int hmmr_abi_version(void) { return HMMR_ABI_VERSION; }
extern "C" const char* hmmr_last_error(void) { return g_err; }
