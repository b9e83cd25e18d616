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
}  // namespace

// One launch of `workgroups` x 4 waves, each issuing 8 * n8 MFMAs of 32 x 32 x 16 (32768 FLOP each).  out: NULL or workgroups * 256 floats.
extern "C" int hmmr_mfma_rate_probe(int workgroups, int n8, float* out, void* stream) {
    HMMR_REQUIRE(workgroups > 0 && n8 > 0, "hmmr_mfma_rate_probe: bad arguments");
    hipLaunchKernelGGL(mfma_rate_kernel, dim3((unsigned)workgroups), dim3(256), 0, (hipStream_t)stream, out, n8);
    HMMR_CHECK_HIP(hipGetLastError());
    return 0;
}
