// libhmmr_hip.so: ABI version + per-thread error string (include/hmmr_hip.h).
#include <stdarg.h>
#include <stdio.h>

#include <atomic>

#include <hip/hip_runtime.h>

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

// ---- launch counters (include/hmmr_hip.h: hmmr_launch_counts); `which` indexes hmmr_launch_counts_t's fields
static std::atomic<unsigned long long> g_launches[5];
void hmmr_count_launch(int which) { if (which >= 0 && which < 5) g_launches[which].fetch_add(1ull, std::memory_order_relaxed); }
extern "C" void hmmr_launch_counts(hmmr_launch_counts_t* out, int clear) {
    unsigned long long v[5];
    for (int i = 0; i < 5; ++i) v[i] = clear ? g_launches[i].exchange(0ull, std::memory_order_relaxed) : g_launches[i].load(std::memory_order_relaxed);
    if (out) { out->unit_pair = v[0]; out->b1_unit = v[1]; out->tail_split = v[2]; out->conv3x3_stream = v[3]; out->conv1x1_stream = v[4]; }
}

// compute units of the stream's device, asked once per device (csrc/common.h); 256 (an MI355X) if the runtime will not say
int hmmr_cu_count(hipStream_t stream) {
    static std::atomic<int> cus[64];
    int dev = 0;
    if (!stream || hipStreamGetDevice(stream, &dev) != hipSuccess)
        if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    const int slot = dev & 63;
    int n = cus[slot].load(std::memory_order_relaxed);
    if (n <= 0) {
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 1) n = 256;
        cus[slot].store(n, std::memory_order_relaxed);
    }
    return n;
}

extern "C" int hmmr_abi_version(void) { return HMMR_ABI_VERSION; }
extern "C" const char* hmmr_last_error(void) { return g_err; }

// ---- run flags: every translation unit that splits values keeps a sticky device word (csrc/common.h) and registers its reader here
typedef int (*hmmr_flag_reader_t)(unsigned* flags, int clear);
static hmmr_flag_reader_t g_flag_readers[32];
static int g_n_flag_readers = 0;
void hmmr_register_flag_reader(hmmr_flag_reader_t fn) {
    if (g_n_flag_readers < 32) g_flag_readers[g_n_flag_readers++] = fn;
}
extern "C" int hmmr_run_flags(unsigned* flags, int clear) {
    if (!flags) { hmmr_set_error("hmmr_run_flags: null argument"); return -1; }
    // the readers copy with the null stream, which does not wait for non-blocking streams: drain the device first
    if (hipDeviceSynchronize() != hipSuccess) { hmmr_set_error("hmmr_run_flags: hipDeviceSynchronize failed"); return -2; }
    unsigned all = 0;
    for (int i = 0; i < g_n_flag_readers; ++i) {
        unsigned v = 0;
        if (g_flag_readers[i](&v, clear)) { hmmr_set_error("hmmr_run_flags: reading the device flag word failed"); return -2; }
        all |= v;
    }
    *flags = all;
    return 0;
}
