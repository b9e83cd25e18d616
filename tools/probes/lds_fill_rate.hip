// Micro-benchmark (development aid): how many bytes per second do ALL CUs together get from L2 / HBM into LDS -- the operand delivery
// ceiling a split-operand GEMM tile runs into (a 128 x 128 tile has 32 FLOP per operand byte, so 240 TFLOP/s is
// 7.5 TB/s into LDS).  Every workgroup (4 or 8 waves, 2 per CU) streams a region with global_load_lds_dwordx4 (1 KB per wave
// instruction) into a 64 KB LDS ring, 8 pieces in flight per wave, nothing consumes the data:
//   mode 0: every workgroup the SAME 1 MB (a filter bank: L2 hits, one L2 slice set per XCD)
//   mode 1: the workgroups of an XCD-sized group share a 2 MB region each (activation panels re-read by the N tiles of a GEMM)
//   mode 2: every workgroup its own 1 MB of a 1 GB buffer (HBM)
//   hipcc --offload-arch=gfx950 -O3 tools/probes/lds_fill_rate.hip -o /tmp/lds_fill_rate && /tmp/lds_fill_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int WAVES>
__global__ __launch_bounds__(WAVES * 64, 2) void fill(const char* src, long long region, long long stride_wg, int share, int passes) {
    extern __shared__ __attribute__((aligned(256))) char smem[];
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6), lane = threadIdx.x & 63;
    const char* base = src + (long long)(blockIdx.x / share) * stride_wg;
    const long long per_wave = region / WAVES;
    const char* p0 = base + wave * per_wave + lane * 16;
    char* dst = smem + wave * 8192;
    for (int pass = 0; pass < passes; ++pass) {
        for (long long off = 0; off < per_wave; off += 8192) {
            const char* p = p0 + off;
#pragma unroll
            for (int q = 0; q < 8; ++q)
                __builtin_amdgcn_global_load_lds((gptr_t)(p + q * 1024), (lptr_t)(dst + q * 1024), 16, 0, 0);
            __builtin_amdgcn_s_waitcnt((8 & 15) | (7 << 4) | (15 << 8) | ((8 >> 4) << 14));      // at most 8 pieces behind
        }
    }
    __builtin_amdgcn_s_waitcnt(0);
}

int main() {
    int cus = 0;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    const long long total = 1ll << 30;
    char* buf = nullptr;
    if (hipMalloc(&buf, total) != hipSuccess) return 1;
    hipMemset(buf, 1, total);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = 2 * cus;
    const struct { const char* name; long long region, stride; int share, passes; } modes[] = {
        {"same 1 MB for every workgroup (filters: L2 hits)", 1 << 20, 0, 1, 16},
        {"2 MB shared by 8 consecutive workgroups (panels re-read by N tiles)", 2 << 20, 2 << 20, 8, 8},
        {"own 1 MB per workgroup of a 1 GB buffer (HBM)", 1 << 20, 1 << 20, 1, 4},
    };
    for (int waves = 4; waves <= 8; waves += 4)
        for (const auto& m : modes) {
            auto launch = [&]() {
                if (waves == 4) hipLaunchKernelGGL(fill<4>, dim3(grid), dim3(256), 65536, 0, buf, m.region, m.stride, m.share, m.passes);
                else hipLaunchKernelGGL(fill<8>, dim3(grid), dim3(512), 65536, 0, buf, m.region, m.stride, m.share, m.passes);
            };
            hipFuncSetAttribute((const void*)fill<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
            hipFuncSetAttribute((const void*)fill<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
            launch();
            hipDeviceSynchronize();
            float best = 1e30f;
            for (int r = 0; r < 5; ++r) {
                hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                best = ms < best ? ms : best;
            }
            const double bytes = (double)grid * m.region * m.passes;
            printf("%d waves per workgroup, %d workgroups: %-70s %7.3f ms  %6.2f TB/s into LDS (%.1f B/cycle/CU at 2.1 GHz)\n", waves, grid, m.name, best,
                   bytes / best / 1e9, bytes / best / 1e-3 / cus / 2.1e9);
        }
    return 0;
}
