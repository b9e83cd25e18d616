// Micro-benchmark (development aid): one wave per SIMD, a loop of [v_mfma_f32_32x32x16_f16 ; N fillers of one kind], cycles per MFMA.
// What does an instruction cost beside the matrix pipe when the wave is alone on its SIMD (csrc/unit_pair.hip's regime)?
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_fillers.hip -o /tmp/mfma_fillers && /tmp/mfma_fillers
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) _Float16 h8;
typedef __attribute__((ext_vector_type(16))) float f16v;

#define REP4(x) x x x x
#define REP8(x) REP4(x) REP4(x)

template <int KIND, int NF, bool TWOACC>
__global__ __launch_bounds__(256, 1) void k(float* out, unsigned long long* cyc, int iters, float one) {
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(1.0f + i * 0.01f); }
    f16v c0 = {}, c1 = {};
    float v0 = threadIdx.x, v1 = 1.5f, v2 = 0.25f, v3 = 3.f;
    unsigned u0 = threadIdx.x, u1 = 77;
    __shared__ float lds[1024];
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            if (TWOACC && (m & 1)) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c1) : "v"(a), "v"(b));
            else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c0) : "v"(a), "v"(b));
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v0) : "v"(v1), "v"(v2));
                if (KIND == 1) { if (f & 1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v0) : "v"(v1), "v"(v2)); else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v3) : "v"(v1), "v"(v2)); }
                if (KIND == 2) asm volatile("v_cvt_f16_f32 %0, %1" : "=v"(u0) : "v"(v0));
                if (KIND == 3) asm volatile("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(v3) : "v"(u0), "v"(v1), "v"(u1));
                if (KIND == 4) asm volatile("v_med3_f32 %0, %1, %2, %3" : "=v"(v3) : "v"(v0), "v"(v1), "v"(v2));
                if (KIND == 5) asm volatile("v_mov_b32 %0, %1" : "=v"(u0) : "v"(u1));
                if (KIND == 6) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(u0) : "v"(v0), "v"(v1));
                if (KIND == 7) asm volatile("ds_read_b128 %0, %1" : "=v"(*(float __attribute__((ext_vector_type(4)))*)&c1) : "v"((threadIdx.x & 63) * 16) : "memory");
                if (KIND == 8) asm volatile("s_nop 0");
                if (KIND == 9) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(u0) : "a"(u1));
                if (KIND == 10) asm volatile("s_add_u32 %0, %0, 1" : "+s"(u1));
            }
        }
    }
    asm volatile("s_nop 7\n s_nop 7\n s_nop 7");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x + blockIdx.x * 256] = c0[0] + c1[1] + v0 + v3 + u0;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int KIND, int NF, bool TWOACC> static void run(const char* name, float* out, unsigned long long* cyc) {
    const int iters = 2000;
    hipLaunchKernelGGL((k<KIND, NF, TWOACC>), dim3(256), dim3(256), 0, 0, out, cyc, iters, 1.0f);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<KIND, NF, TWOACC>), dim3(256), dim3(256), 0, 0, out, cyc, iters, 1.0f);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-14s fillers %d %s: %6.1f ticks / MFMA, %6.1f ns / MFMA\n", name, NF, TWOACC ? "2 acc" : "1 acc", (double)c / (iters * 8.0), ms * 1e6 / (iters * 8.0));
}
#define ROW(K, name) run<K, 0, false>(name, out, cyc); run<K, 2, false>(name, out, cyc); run<K, 4, false>(name, out, cyc); run<K, 6, false>(name, out, cyc); run<K, 8, false>(name, out, cyc); run<K, 4, true>(name, out, cyc); run<K, 6, true>(name, out, cyc);
int main() {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 8);
    ROW(0, "fma dep") ROW(1, "fma 2 chains") ROW(2, "cvt_f16_f32") ROW(3, "fma_mix") ROW(4, "med3") ROW(5, "mov") ROW(6, "cvt_pk") ROW(7, "ds_read_b128") ROW(8, "s_nop") ROW(9, "accvgpr_read") ROW(10, "salu")
    return 0;
}
