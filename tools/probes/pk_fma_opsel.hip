// Micro-benchmark (development aid, round 6): does `v_pk_fma_f32 D, A, B, C op_sel:[0,1,0] op_sel_hi:[1,0,1]` (the SLP vectoriser's
// "swapped B" form: D.lo = A.lo * B.hi + C.lo, D.hi = A.hi * B.lo + C.hi) lose its low product when OTHER work shares the CU?
// tools/tail_race_check.py + asm-level bisection (profiles/r06s_*) put the wrong bits of the SLP build of smpl_pose_kernel on exactly this
// instruction: o[10] of the chain step comes out as C.lo -- the product term missing -- in lanes 48-55, with EXEC = {lane i, lane 32 + i}, and only
// beside the ResNet's kernels.  Here: a victim kernel runs the instruction with that EXEC pattern (or full EXEC) on known values and counts wrong
// results per lane; a noise kernel (MFMA + LDS + global loads, its own stream) runs beside it or not.
//   hipcc --offload-arch=gfx950 -O3 -w tools/probes/pk_fma_opsel.hip -o /tmp/pk_fma_opsel && /tmp/pk_fma_opsel
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
typedef __attribute__((ext_vector_type(2))) float f2;
typedef __attribute__((ext_vector_type(8))) __bf16 bf8;
typedef __attribute__((ext_vector_type(16))) float f16v;
typedef __attribute__((ext_vector_type(4))) float f4;

__device__ __forceinline__ float mk(unsigned s) {           // a float in [-2, 2) from a hash
    s ^= s >> 16; s *= 0x7feb352du; s ^= s >> 15; s *= 0x846ca68bu; s ^= s >> 16;
    return (float)(int)(s & 0xffff) * (1.f / 16384.f) - 2.f;
}

// One victim kernel per instruction form: OPC (0 fma, 1 mul, 2 add), the modifier text, and what the text means: source s gives its register
// SsL to the low result and SsH to the high result (0 = low register of the pair, 1 = high).  SPARSE: EXEC = {lane i, lane 32 + i}, i = 16 .. 23
// (the chain step of smpl_pose_kernel); LDS_AROUND: operands through LDS (b128 / b64 reads), the result back (b128 write), like that step.
#define VICTIM(NAME, OPC, TEXT, S0L, S0H, S1L, S1H, S2L, S2H)                                                                              \
template <bool SPARSE, bool LDS_AROUND>                                                                                                    \
__global__ __launch_bounds__(256) void NAME(unsigned* bad, unsigned* first, int iters) {                                                   \
    __shared__ float sm[256 * 12];                                                                                                         \
    const int lane = threadIdx.x & 63, j = threadIdx.x & 31;                                                                               \
    unsigned nbad = 0;                                                                                                                     \
    for (int it = 0; it < iters; ++it) {                                                                                                   \
        const unsigned seed = (blockIdx.x * 256 + threadIdx.x) * 7919u + it * 104729u;                                                     \
        f2 a = {mk(seed), mk(seed + 1)}, b = {mk(seed + 2), mk(seed + 3)}, c = {mk(seed + 4), mk(seed + 5)};                               \
        if (LDS_AROUND) {                                                                                                                  \
            float* p = sm + threadIdx.x * 12;                                                                                              \
            p[0] = a[0]; p[1] = a[1]; p[2] = b[0]; p[3] = b[1]; p[4] = c[0]; p[5] = c[1];                                                  \
            __syncthreads();                                                                                                               \
        }                                                                                                                                  \
        const int i = 16 + (it & 7);                                                                                                       \
        unsigned long long d64 = 0ull;      /* (one 64-bit value: its halves are taken apart with integer arithmetic) */                   \
        if (!SPARSE || j == i) {                                                                                                           \
            if (LDS_AROUND) {                                                                                                              \
                const f4 v = *(const f4*)(sm + threadIdx.x * 12);                                                                          \
                const f2 w = *(const f2*)(sm + threadIdx.x * 12 + 4);                                                                      \
                a = f2{v[0], v[1]}; b = f2{v[2], v[3]}; c = w;                                                                             \
            }                                                                                                                              \
            if (OPC == 0) asm volatile("v_pk_fma_f32 %0, %1, %2, %3 " TEXT : "=v"(d64) : "v"(a), "v"(b), "v"(c));                          \
            else if (OPC == 1) asm volatile("v_pk_mul_f32 %0, %1, %2 " TEXT : "=v"(d64) : "v"(a), "v"(b));                                 \
            else if (OPC == 2) asm volatile("v_pk_add_f32 %0, %1, %2 " TEXT : "=v"(d64) : "v"(a), "v"(b));                                 \
            else asm volatile("v_pk_mov_b32 %0, %1, %2 " TEXT : "=v"(d64) : "v"(a), "v"(b));   /* D.lo = A[op_sel[0]], D.hi = B[op_sel[1]] */  \
            const unsigned d0 = (unsigned)d64, d1 = (unsigned)(d64 >> 32);                                                                 \
            if (LDS_AROUND) *(f4*)(sm + threadIdx.x * 12 + 8) = f4{__builtin_bit_cast(float, d0), __builtin_bit_cast(float, d1), 0.f, 0.f}; \
            float e0, e1;                                                                                                                  \
            if (OPC == 0) { e0 = __builtin_fmaf(a[S0L], b[S1L], c[S2L]); e1 = __builtin_fmaf(a[S0H], b[S1H], c[S2H]); }                    \
            else if (OPC == 1) { e0 = a[S0L] * b[S1L]; e1 = a[S0H] * b[S1H]; }                                                             \
            else if (OPC == 2) { e0 = a[S0L] + b[S1L]; e1 = a[S0H] + b[S1H]; }                                                             \
            else { e0 = a[S0L]; e1 = b[S1H]; }                                                                                             \
            asm volatile("" : "+v"(e0), "+v"(e1));                                                                                         \
            if (d0 != __builtin_bit_cast(unsigned, e0) || d1 != __builtin_bit_cast(unsigned, e1)) {                                        \
                ++nbad;                                                                                                                    \
                if (atomicAdd(&first[0], 1u) < 8) {                                                                                        \
                    const unsigned k = atomicAdd(&first[1], 1u);                                                                           \
                    if (k < 8) {                                                                                                           \
                        float* o = (float*)(first + 8 + k * 12);                                                                           \
                        o[0] = a[0]; o[1] = a[1]; o[2] = b[0]; o[3] = b[1]; o[4] = c[0]; o[5] = c[1];                                      \
                        o[6] = __builtin_bit_cast(float, d0); o[7] = __builtin_bit_cast(float, d1); o[8] = e0; o[9] = e1;                  \
                        first[8 + k * 12 + 10] = lane; first[8 + k * 12 + 11] = it;                                                        \
                    }                                                                                                                      \
                }                                                                                                                          \
            }                                                                                                                              \
        }                                                                                                                                  \
        if (LDS_AROUND) __syncthreads();                                                                                                   \
    }                                                                                                                                      \
    if (nbad) atomicAdd(&bad[lane], nbad);                                                                                                 \
}

VICTIM(fma_plain, 0, "", 0, 1, 0, 1, 0, 1)
VICTIM(fma_swap1, 0, "op_sel:[0,1,0] op_sel_hi:[1,0,1]", 0, 1, 1, 0, 0, 1)          // the SLP build's: source 1 swapped
VICTIM(fma_swap0, 0, "op_sel:[1,0,0] op_sel_hi:[0,1,1]", 1, 0, 0, 1, 0, 1)
VICTIM(fma_swap2, 0, "op_sel:[0,0,1] op_sel_hi:[1,1,0]", 0, 1, 0, 1, 1, 0)
VICTIM(fma_swap_all, 0, "op_sel:[1,1,1] op_sel_hi:[0,0,0]", 1, 0, 1, 0, 1, 0)
VICTIM(fma_hi0, 0, "op_sel:[1,0,0]", 1, 1, 0, 1, 0, 1)                             // high register of source 0 to both halves
VICTIM(fma_lo0, 0, "op_sel_hi:[0,1,1]", 0, 0, 0, 1, 0, 1)
VICTIM(fma_hi1, 0, "op_sel:[0,1,0]", 0, 1, 1, 1, 0, 1)
VICTIM(fma_lo1, 0, "op_sel_hi:[1,0,1]", 0, 1, 0, 0, 0, 1)                          // the shipped library's only form (the SMPL blend, hand-written)
VICTIM(fma_hi2, 0, "op_sel:[0,0,1]", 0, 1, 0, 1, 1, 1)
VICTIM(fma_lo2, 0, "op_sel_hi:[1,1,0]", 0, 1, 0, 1, 0, 0)
VICTIM(mul_plain, 1, "", 0, 1, 0, 1, 0, 0)
VICTIM(mul_swap0, 1, "op_sel:[1,0] op_sel_hi:[0,1]", 1, 0, 0, 1, 0, 0)
VICTIM(mul_swap1, 1, "op_sel:[0,1] op_sel_hi:[1,0]", 0, 1, 1, 0, 0, 0)
VICTIM(mul_hi0, 1, "op_sel:[1,0]", 1, 1, 0, 1, 0, 0)
VICTIM(mul_lo1, 1, "op_sel_hi:[1,0]", 0, 1, 0, 0, 0, 0)
VICTIM(add_plain, 2, "", 0, 1, 0, 1, 0, 0)
VICTIM(add_swap0, 2, "op_sel:[1,0] op_sel_hi:[0,1]", 1, 0, 0, 1, 0, 0)
VICTIM(add_swap1, 2, "op_sel:[0,1] op_sel_hi:[1,0]", 0, 1, 1, 0, 0, 0)
VICTIM(add_hi1, 2, "op_sel:[0,1]", 0, 1, 1, 1, 0, 0)
VICTIM(mov_01, 3, "op_sel:[0,1]", 0, 0, 0, 1, 0, 0)                                 // v_pk_mov_b32's default: D = (A.lo, B.hi)
VICTIM(mov_10, 3, "op_sel:[1,0]", 1, 0, 0, 0, 0, 0)                                 // (A.hi, B.lo): the SLP build's shuffles
VICTIM(mov_11, 3, "op_sel:[1,1]", 1, 0, 0, 1, 0, 0)
VICTIM(mov_00, 3, "op_sel:[0,0]", 0, 0, 0, 0, 0, 0)

// the neighbours: bf16 MFMAs, LDS reads / writes, global loads; 256 threads, 64 KB of LDS
__global__ __launch_bounds__(256) void noise(float* out, const float* in, int iters, int what) {
    extern __shared__ float lds[];
    for (int i = threadIdx.x; i < 16384; i += 256) lds[i] = i;
    __syncthreads();
    bf8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(1.0f + i * 0.01f); }
    f16v c0 = {}, c1 = {};
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        if (what & 1) {
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
            }
        }
        if (what & 2) {
            const f4 v = *(const f4*)(lds + ((threadIdx.x * 4 + it * 64) & 16380));
            acc += v[0] + v[3];
            lds[(threadIdx.x + it * 17) & 16383] = acc;
        }
        if (what & 4) acc += in[((size_t)blockIdx.x * 256 + threadIdx.x + (size_t)it * 65536) & ((1u << 24) - 1)];
        if (what & 8) {                                      // packed fp32 of its own
            f2 x = {acc, acc + 1.f}, y = {1.0001f, 0.9999f}, z = {0.5f, 0.25f};
#pragma unroll
            for (int m = 0; m < 8; ++m) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(y), "v"(z));
            acc = x[0] + x[1];
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = c0[0] + c1[5] + acc;
}

typedef void (*victim_t)(unsigned*, unsigned*, int);
static void run(const char* name, victim_t kern, int what, unsigned* bad, unsigned* first, float* out, float* in) {
    hipStream_t sv, sn;
    hipStreamCreateWithFlags(&sv, hipStreamNonBlocking); hipStreamCreateWithFlags(&sn, hipStreamNonBlocking);
    hipMemset(bad, 0, 64 * 4); hipMemset(first, 0, (8 + 8 * 12) * 4);
    hipDeviceSynchronize();
    const int reps = 40;
    for (int r = 0; r < reps; ++r) {
        if (what) hipLaunchKernelGGL(noise, dim3(512), dim3(256), 65536, sn, out, in, 3000, what);
        // the victim in small launches (48 workgroups like smpl_pose_kernel at 384 instances), several per noise launch
        for (int k = 0; k < 6; ++k) hipLaunchKernelGGL(kern, dim3(48), dim3(256), 0, sv, bad, first, 2000);
    }
    hipDeviceSynchronize();
    unsigned hb[64], hf[8 + 8 * 12];
    hipMemcpy(hb, bad, sizeof(hb), hipMemcpyDeviceToHost); hipMemcpy(hf, first, sizeof(hf), hipMemcpyDeviceToHost);
    unsigned long long tot = 0; for (int i = 0; i < 64; ++i) tot += hb[i];
    const double execs = (double)reps * 6 * 48 * 4 * 2000;
    printf("%-58s noise %2d: %llu wrong of %.3g wave-instructions", name, what, tot, execs);
    if (tot) {
        printf(" | lanes:");
        for (int i = 0; i < 64; ++i) if (hb[i]) printf(" %d:%u", i, hb[i]);
        printf("\n");
        for (unsigned k = 0; k < hf[1] && k < 4; ++k) {
            const float* o = (const float*)(hf + 8 + k * 12);
            printf("      lane %u it %u: A (%g, %g) B (%g, %g) C (%g, %g) -> got (%.9g, %.9g) expected (%.9g, %.9g)\n", hf[8 + k * 12 + 10], hf[8 + k * 12 + 11],
                   o[0], o[1], o[2], o[3], o[4], o[5], o[6], o[7], o[8], o[9]);
        }
    } else printf("\n");
    hipStreamDestroy(sv); hipStreamDestroy(sn);
}

int main(int argc, char** argv) {
    const bool quick = argc > 1 && !strcmp(argv[1], "--quick");       // tests/test_gpu_isa.py: the shipped forms and the SLP build's, beside MFMA + LDS + global loads
    unsigned *bad, *first; float *out, *in;
    hipMalloc(&bad, 64 * 4); hipMalloc(&first, (8 + 8 * 12) * 4); hipMalloc(&out, 512 * 256 * 4); hipMalloc(&in, (size_t)(1u << 24) * 4);
    hipMemset(in, 0, (size_t)(1u << 24) * 4);
    hipFuncSetAttribute((const void*)noise, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
#define FORM(K) run(#K ", full EXEC", K<false, false>, 7, bad, first, out, in)
    if (quick) { FORM(fma_plain); FORM(fma_lo1); FORM(fma_swap1); return 0; }
    // 1. the form the SLP build of smpl_pose_kernel holds, by neighbour (noise bits: 1 MFMA, 2 LDS, 4 global loads, 8 packed fp32)
    for (int what : {0, 1, 2, 4, 7, 15}) {
        run("fma, source 1 swapped, EXEC = {i, 32 + i}", fma_swap1<true, false>, what, bad, first, out, in);
        run("fma, source 1 swapped, EXEC = {i, 32 + i}, through LDS", fma_swap1<true, true>, what, bad, first, out, in);
        run("fma, source 1 swapped, full EXEC", fma_swap1<false, false>, what, bad, first, out, in);
        run("fma, plain, full EXEC", fma_plain<false, false>, what, bad, first, out, in);
    }
    // 2. every form, full EXEC, beside MFMA + LDS + global loads
    FORM(fma_plain); FORM(fma_swap1); FORM(fma_swap0); FORM(fma_swap2); FORM(fma_swap_all); FORM(fma_hi0); FORM(fma_lo0); FORM(fma_hi1); FORM(fma_lo1);
    FORM(fma_hi2); FORM(fma_lo2); FORM(mul_plain); FORM(mul_swap0); FORM(mul_swap1); FORM(mul_hi0); FORM(mul_lo1); FORM(add_plain); FORM(add_swap0); FORM(add_swap1); FORM(add_hi1);
    FORM(mov_01); FORM(mov_10); FORM(mov_11); FORM(mov_00);
    return 0;
}
