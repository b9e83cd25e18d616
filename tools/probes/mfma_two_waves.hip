// Micro-benchmark (development aid, round 6): TWO waves per SIMD with different jobs -- what csrc/unit_pair.hip's wave-specialised
// form (unit_pair_ws_kernel) relies on.  Waves 0-3 of a 512-thread workgroup ("A") run [MFMA ; NFA vector fillers ; NRA/3 ds_read_b128],
// waves 4-7 ("B") run [MFMA ; NFB fillers ; NRB/3 ds_read_b128]; one workgroup per CU.  tools/probes/mfma_fillers.hip measured ONE wave
// per SIMD: 2 fillers per 32x32x16 MFMA are free, every further DEPENDENT one costs 8 cycles.  Question here: does the matrix pipe stay
// busy when the fillers sit in one wave and the other wave has none -- i.e. is MFMAs-per-SIMD-cycle ~1/32 for the pair?
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_two_waves.hip -o /tmp/mfma_two_waves && /tmp/mfma_two_waves
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) _Float16 h8;
typedef __attribute__((ext_vector_type(16))) float f16v;
typedef __attribute__((ext_vector_type(4))) float f4;

template <int NF, int NR> __device__ __forceinline__ void body(int iters, f16v& c0, f16v& c1, float& v0, float& v3, h8& a, h8& b) {
    float v1 = 1.5f, v2 = 0.25f;
    f4 r0 = {}, r1 = {};
    const unsigned addr = (threadIdx.x & 63) * 16;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 12; ++m) {
            if (m & 1) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c1) : "v"(a), "v"(b));
            else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c0) : "v"(a), "v"(b));
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                if (f & 1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v0) : "v"(v1), "v"(v2));
                else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v3) : "v"(v1), "v"(v2));
            }
            // NR ds_read_b128 per 3 MFMAs
            if ((m % 3) < NR) {
                if (m & 1) asm volatile("ds_read_b128 %0, %1" : "=v"(r0) : "v"(addr) : "memory");
                else asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(r1) : "v"(addr) : "memory");
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    v0 += r0[0] + r1[1];
}

template <int NFA, int NRA, int NFB, int NRB, int WAVES>
__global__ __launch_bounds__(WAVES * 64, 1) void k(float* out, unsigned long long* cyc, int iters) {
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(1.0f + i * 0.01f); }
    f16v c0 = {}, c1 = {};
    float v0 = threadIdx.x, v3 = 3.f;
    __shared__ float lds[4096];
    for (int i = threadIdx.x; i < 4096; i += WAVES * 64) lds[i] = i;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x < 256) body<NFA, NRA>(iters, c0, c1, v0, v3, a, b);
    else body<NFB, NRB>(iters, c0, c1, v0, v3, a, b);
    asm volatile("s_nop 7\n s_nop 7\n s_nop 7");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x + blockIdx.x * WAVES * 64] = c0[0] + c1[1] + v0 + v3;
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) cyc[threadIdx.x >> 6] = t1 - t0;
}

template <int NFA, int NRA, int NFB, int NRB, int WAVES> static void run(const char* name, float* out, unsigned long long* cyc) {
    const int iters = 1500;
    auto kern = k<NFA, NRA, NFB, NRB, WAVES>;
    hipLaunchKernelGGL(kern, dim3(256), dim3(WAVES * 64), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(256), dim3(WAVES * 64), 0, 0, out, cyc, iters);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c[8]; hipMemcpy(c, cyc, 64, hipMemcpyDeviceToHost);
    const double mf = (double)iters * 12.0 * (WAVES / 4);          // MFMAs per SIMD
    printf("%-46s A: %d fillers %d reads / B: %d fillers %d reads (%d waves): %6.1f ns per MFMA of the SIMD = %5.1f %% of 32 cycles at 2.4 GHz; ticks wave0 %llu wave%d %llu\n",
           name, NFA, NRA, NFB, NRB, WAVES, ms * 1e6 / mf, 100.0 * (32.0 / 2.4) / (ms * 1e6 / mf), c[0], WAVES - 1, c[WAVES - 1]);
}

int main() {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 64);
    run<0, 0, 0, 0, 4>("one wave per SIMD, MFMA only", out, cyc);
    run<4, 2, 4, 2, 4>("one wave per SIMD, 4 fillers + 2 reads", out, cyc);
    run<6, 2, 6, 2, 4>("one wave per SIMD, 6 fillers + 2 reads", out, cyc);
    run<0, 0, 0, 0, 8>("two waves per SIMD, MFMA only", out, cyc);
    run<4, 2, 4, 2, 8>("two waves, both 4 fillers + 2 reads", out, cyc);
    run<6, 2, 6, 2, 8>("two waves, both 6 fillers + 2 reads", out, cyc);
    run<8, 2, 0, 2, 8>("two waves, A 8 fillers, B none", out, cyc);
    run<8, 3, 4, 3, 8>("two waves, A 8 fillers, B 4, 3 reads each", out, cyc);
    run<12, 2, 0, 2, 8>("two waves, A 12 fillers, B none", out, cyc);
    run<8, 2, 8, 2, 8>("two waves, both 8 fillers + 2 reads", out, cyc);
    run<12, 2, 12, 2, 8>("two waves, both 12 fillers + 2 reads", out, cyc);
    return 0;
}
