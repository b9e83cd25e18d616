// Micro-benchmark (development aid, round 6): is a VALU write of a ds_write's DATA registers, issued right behind the store, safe while
// other waves of the CU keep the LDS busy?  The SLP-vectorised smpl_pose_kernel (plain -O3) contains exactly
//     ds_write_b128 v1, v[2:5]      ;  v_pk_add_f32 v[4:5], ...      (the store's data registers overwritten by the next instruction)
// and returned wrong values in the last quarter of a wave only when it ran beside other kernels.  Here wave 0 of every workgroup stores
// a known pattern with ds_write_b128 and overwrites two of the four data registers in the very next instruction (variants: v_pk_add_f32,
// v_mov_b32 x 2, v_add_f32 x 2, or -- the control -- nothing), reads the tile back and counts wrong dwords; waves 1 .. W-1 stream
// ds_write_b128 / ds_read_b128 through other parts of the LDS meanwhile (W = 1: the wave is alone).
//   hipcc --offload-arch=gfx950 -O3 tools/probes/lds_store_war.hip -o /tmp/lds_store_war && /tmp/lds_store_war
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) float f4;
typedef __attribute__((ext_vector_type(2))) float f2;

template <int KIND, int WAVES>
__global__ __launch_bounds__(WAVES * 64, 1) void k(unsigned* bad, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[WAVES * 64 * 4 * 4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned wrong = 0;
    if (wave == 0) {
        const unsigned addr = (unsigned)(unsigned long long)(__attribute__((address_space(3))) void*)lds + lane * 16;
        for (int it = 0; it < iters; ++it) {
            f4 v = {(float)(it * 4 + 0) + lane * 0.5f, (float)(it * 4 + 1) + lane * 0.5f, (float)(it * 4 + 2) + lane * 0.5f, (float)(it * 4 + 3) + lane * 0.5f};
            f2 junk = {-7777.f, -8888.f};
            f4 back;
            // (the data registers are named: the overwriting instruction addresses a part of the store's register quad)
#define WAR_SETUP "v_mov_b32 v40, %1\n v_mov_b32 v41, %2\n v_mov_b32 v42, %3\n v_mov_b32 v43, %4\n s_nop 4\n ds_write_b128 %0, v[40:43]\n"
#define WAR_ARGS : : "v"(addr), "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(junk), "v"(junk[0]) : "v40", "v41", "v42", "v43", "memory"
            if (KIND == 0) asm volatile(WAR_SETUP "s_waitcnt lgkmcnt(0)" WAR_ARGS);                                                  // control: nothing behind the store
            if (KIND == 1) asm volatile(WAR_SETUP "v_pk_add_f32 v[42:43], %5, %5\n s_waitcnt lgkmcnt(0)" WAR_ARGS);                 // the last register pair of the quad
            if (KIND == 2) asm volatile(WAR_SETUP "v_mov_b32 v43, %6\n s_waitcnt lgkmcnt(0)" WAR_ARGS);
            if (KIND == 3) asm volatile(WAR_SETUP "v_add_f32 v43, %6, %6\n s_waitcnt lgkmcnt(0)" WAR_ARGS);
            if (KIND == 4) asm volatile(WAR_SETUP "v_pk_add_f32 v[40:41], %5, %5\n s_waitcnt lgkmcnt(0)" WAR_ARGS);                 // the first pair
            if (KIND == 5) asm volatile(WAR_SETUP "s_nop 1\n v_pk_add_f32 v[42:43], %5, %5\n s_waitcnt lgkmcnt(0)" WAR_ARGS);       // two wait states in between
            asm volatile("ds_read_b128 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(back) : "v"(addr) : "memory");
#pragma unroll
            for (int e = 0; e < 4; ++e) wrong += back[e] != (float)(it * 4 + e) + lane * 0.5f;
        }
    } else {
        // the other waves: keep the LDS store path and the array busy
        const unsigned addr = (unsigned)(unsigned long long)(__attribute__((address_space(3))) void*)lds + (wave * 64 + lane) * 16 * 4;
        f4 a = {1.f, 2.f, 3.f, 4.f}, b;
        for (int it = 0; it < iters * 4; ++it) {
            asm volatile("ds_write_b128 %1, %2\n ds_write_b128 %1, %2 offset:16\n ds_read_b128 %0, %1 offset:32\n ds_write_b128 %1, %2 offset:48\n s_waitcnt lgkmcnt(0)"
                         : "=v"(b) : "v"(addr), "v"(a) : "memory");
            a[0] += b[1] * 0.f;
        }
        if (a[0] == -1.f) wrong = 1;
    }
    if (wrong) atomicAdd(bad + (wave == 0 ? (lane >> 4) : 4), wrong);        // per quarter of wave 0
}

template <int KIND, int WAVES> static void run(const char* name, unsigned* bad) {
    hipMemset(bad, 0, 32);
    hipLaunchKernelGGL((k<KIND, WAVES>), dim3(512), dim3(WAVES * 64), 0, 0, bad, 20000);
    hipDeviceSynchronize();
    unsigned h[8]; hipMemcpy(h, bad, 32, hipMemcpyDeviceToHost);
    printf("%-44s %d waves per workgroup: wrong dwords read back by wave 0, per quarter of the wave (lanes 0-15 | 16-31 | 32-47 | 48-63): %u %u %u %u\n", name, WAVES, h[0], h[1], h[2], h[3]);
}
int main() {
    unsigned* bad; hipMalloc(&bad, 32);
    run<0, 1>("control (nothing behind the store)", bad); run<0, 8>("control (nothing behind the store)", bad);
    run<1, 1>("v_pk_add_f32 into the data registers", bad); run<1, 4>("v_pk_add_f32 into the data registers", bad); run<1, 8>("v_pk_add_f32 into the data registers", bad);
    run<2, 1>("v_mov_b32 into a data register", bad); run<2, 8>("v_mov_b32 into a data register", bad);
    run<3, 1>("v_add_f32 into a data register", bad); run<3, 8>("v_add_f32 into a data register", bad);
    run<4, 1>("v_pk_add_f32 into the FIRST pair", bad); run<4, 8>("v_pk_add_f32 into the FIRST pair", bad);
    run<5, 8>("s_nop 1, then v_pk_add_f32 (last pair)", bad);
    return 0;
}
