// Measurement aid (not on the product path): the rate at which this GPU sustains v_mfma_f32_32x32x16_f16 when EVERY SIMD issues
// nothing else -- one wave per SIMD, four independent accumulators, operands in registers.  bench.py times it with events and reports
// it next to the nominal 2.5 PFLOP/s: under its power cap an MI355X clocks 1.7-1.9 GHz on a dense fp16 MFMA stream, not 2.4 (DESIGN
// section 4.1.2), so this -- divided by three for the split format -- is the ceiling a perfect kernel would reach on the box at hand.
#include "common.h"
#include "hmmr_hip.h"

namespace {
__global__ __launch_bounds__(256, 1) void mfma_rate_kernel(float* out, int n8) {
    shalf8 a, b;
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = (shalf_t)(0.001f * (threadIdx.x & 63) + 0.125f * i); b[i] = (shalf_t)(1.0f + 0.0078125f * i); }
    f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
    for (int it = 0; it < n8; ++it) {
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %4, %5, %0\n\tv_mfma_f32_32x32x16_f16 %1, %4, %5, %1\n\t"
                     "v_mfma_f32_32x32x16_f16 %2, %4, %5, %2\n\tv_mfma_f32_32x32x16_f16 %3, %4, %5, %3\n\t"
                     "v_mfma_f32_32x32x16_f16 %0, %4, %5, %0\n\tv_mfma_f32_32x32x16_f16 %1, %4, %5, %1\n\t"
                     "v_mfma_f32_32x32x16_f16 %2, %4, %5, %2\n\tv_mfma_f32_32x32x16_f16 %3, %4, %5, %3"
                     : "+a"(c0), "+a"(c1), "+a"(c2), "+a"(c3) : "v"(a), "v"(b));
    }
    if (out) out[blockIdx.x * 256 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}

// The same stream on operands that CHANGE: four hashed fragment pairs taken in turn, so the operand buses and the multiplier arrays toggle the way
// they do on real tensors (the loop above multiplies the same two fragments for ever: the lowest-power case, hence the highest clock the cap allows).
__device__ __forceinline__ unsigned probe_hash(unsigned s) { s ^= s >> 16; s *= 0x7feb352du; s ^= s >> 15; s *= 0x846ca68bu; s ^= s >> 16; return s; }
__global__ __launch_bounds__(256, 1) void mfma_rate_toggle_kernel(float* out, int n8) {
    shalf8 a[4], b[4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            // fp16 bit patterns with random sign / mantissa and an exponent near 1.0 (no overflow of the fp32 sums over the loop: half of them negative)
            const unsigned h0 = probe_hash((threadIdx.x * 64 + k * 16 + i) * 2654435761u + blockIdx.x), h1 = probe_hash(h0);
            a[k][i] = __builtin_bit_cast(shalf_t, (unsigned short)((h0 & 0x83ffu) | 0x3800u));
            b[k][i] = __builtin_bit_cast(shalf_t, (unsigned short)((h1 & 0x83ffu) | 0x3800u));
        }
    f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
    for (int it = 0; it < n8; ++it) {
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %4, %8, %0\n\tv_mfma_f32_32x32x16_f16 %1, %5, %9, %1\n\t"
                     "v_mfma_f32_32x32x16_f16 %2, %6, %10, %2\n\tv_mfma_f32_32x32x16_f16 %3, %7, %11, %3\n\t"
                     "v_mfma_f32_32x32x16_f16 %0, %5, %10, %0\n\tv_mfma_f32_32x32x16_f16 %1, %6, %11, %1\n\t"
                     "v_mfma_f32_32x32x16_f16 %2, %7, %8, %2\n\tv_mfma_f32_32x32x16_f16 %3, %4, %9, %3"
                     : "+a"(c0), "+a"(c1), "+a"(c2), "+a"(c3)
                     : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]));
    }
    if (out) out[blockIdx.x * 256 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}

// The shader clock, sampled while OTHER work runs (round 6): one wave that reads s_memtime (shader-clock ticks) and s_memrealtime (the
// constant reference counter, hipDeviceAttributeWallClockRate) every `sleep` x 64 clocks and stores the pairs.  Launched on its own stream
// beside a measured region it shows what the part clocks at under THAT load: the unit pair of block 3 runs at 1.59 GHz with all 256 CUs
// busy and at 2.07 GHz without its trunk stores and shortcut requests -- which is what "the matrix pipes are 48 % busy"
// has to be read against.
__global__ __launch_bounds__(64, 1) void clock_probe_kernel(unsigned long long* samples, int n, int sleep, const volatile int* stop) {
    for (int i = 0; i < n; ++i) {
        const unsigned long long t = __builtin_amdgcn_s_memtime(), r = __builtin_amdgcn_s_memrealtime();
        if (threadIdx.x == 0) { samples[2 * i] = t; samples[2 * i + 1] = r; }
        if (stop && *stop) {                                 // the host's "enough": the remaining slots stay zero
            return;
        }
        for (int k = 0; k < sleep; ++k) __builtin_amdgcn_s_sleep(127);          // 127 x 64 clocks
    }
}
}  // namespace

// samples: 2 n device words (s_memtime, s_memrealtime pairs, zero-initialised by the caller); sleep: s_sleep(127) repetitions between samples
// (1 ~ 4-5 us); stop: NULL or a device int the caller sets non-zero (from another stream) to end the kernel early.
extern "C" int hmmr_clock_probe(unsigned long long* samples, int n, int sleep, const int* stop, void* stream) {
    HMMR_REQUIRE(samples && n > 0 && sleep >= 0, "hmmr_clock_probe: bad arguments");
    hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, samples, n, sleep, (const volatile int*)stop);
    HMMR_CHECK_HIP(hipGetLastError());
    return 0;
}

// One launch of `workgroups` x 4 waves, each issuing 8 * n8 MFMAs of 32 x 32 x 16 (32768 FLOP each).  out: NULL or workgroups * 256 floats.
// n8 < 0 (ABI 19): -n8 iterations of the form whose operands change from MFMA to MFMA.
extern "C" int hmmr_mfma_rate_probe(int workgroups, int n8, float* out, void* stream) {
    HMMR_REQUIRE(workgroups > 0 && n8 != 0, "hmmr_mfma_rate_probe: bad arguments");
    if (n8 < 0) {
        hipLaunchKernelGGL(mfma_rate_toggle_kernel, dim3((unsigned)workgroups), dim3(256), 0, (hipStream_t)stream, out, -n8);
        HMMR_CHECK_HIP(hipGetLastError());
        return 0;
    }
    hipLaunchKernelGGL(mfma_rate_kernel, dim3((unsigned)workgroups), dim3(256), 0, (hipStream_t)stream, out, n8);
    HMMR_CHECK_HIP(hipGetLastError());
    return 0;
}
