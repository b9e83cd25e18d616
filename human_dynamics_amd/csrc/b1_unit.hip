// A whole stride-1 bottleneck unit of block 1 (64 -> 256 -> 64 channels, 56 x 56 pixels) for gfx950, f16x3 ("split") operands:
//
//     h2     = relu(bn2(conv2_3x3(h1)))                                (bottleneck_v2 `conv2`, SAME, stride 1)
//     trunk' = conv3({h2 [, xp]}) * scale3 + shift3 [+ shortcut]       (`conv3` + add; with xp the unit's conv shortcut is folded in)
//     h1'    = relu(bn1'(conv1'(relu(bn_pre'(trunk')))))               (the NEXT unit's `preact` + `conv1`)
//
// in ONE kernel (slim resnet_v2.bottleneck as invoked at src/models.py:65-75).  Round 5's replacement of the
// CONV2 form of bottleneck_split.hip (8 x 8 pixel tiles, 24 barriers per 64 pixels, filters as per-wave fragments from L2:
// 0.59-0.61 ms per unit at 2.4-3.6 TB/s, neither HBM- nor MFMA-bound).  What bounds block 1 is the trunk: 2.5 KB of HBM
// traffic per pixel (shortcut in, trunk out, h1 in, h1' out) against 408-504 MFMAs per 32 pixels, i.e. 0.41 / 0.29 ms of
// HBM time against 0.17 / 0.21 ms of matrix time per unit -- so the kernel is built to keep HBM requests in flight while
// some wave of the CU computes:
//   * TWO workgroups per CU (4 waves each, 256 registers per lane, 79 KB of LDS): while one runs its conv2 (matrix work,
//     no HBM traffic to speak of) the other streams its trunk chunks; inside a workgroup the code is plain (compiler-
//     scheduled) -- the second wave of every SIMD is what fills its stalls;
//   * a workgroup owns 128 consecutive pixels of the flattened [img][y][x] order, a WAVE owns 32 of them for the whole
//     unit.  conv2: K is chunk-major in 16-channel chunks (K step kt = chunk kt / 9, tap kt % 9 -- the order of
//     conv3x3_stream.hip, so the results equal its launches bit for bit); the chunk of every pixel the tile can touch
//     (128 + 2 W + 2 rows of 64 bytes, slots XOR-swizzled by (row >> 2) & 3) is DMA'd into one of two patch buffers, a
//     tap is a row shift, out-of-image taps read a zero row of the same bank;
//   * the wave's conv2 result (32 px x 64 channels) gets its BN + ReLU, is split and turned from the MFMA's D layout
//     into B-operand fragments IN REGISTERS (one v_permlane32_swap per register pair, as unit_pair.hip): h2 never exists
//     in LDS or HBM.  With a folded shortcut the unit's pre-activated input xp is a second register panel;
//   * ALL filters of the unit -- conv2 (144 KB), conv3 (64 / 128 KB), conv1' (64 KB) -- are ONE stream of 2 KB MFMA
//     A-operand fragments in the order the kernel consumes them (packing.pack_b1_unit_stream), DMA'd through a 5-slab
//     LDS ring (8 KB slabs, four in flight) that every wave reads: 272 / 336 KB of L2 -> LDS traffic per 128 pixels
//     (the old form: 4 KB per PIXEL of per-wave fragment reads);
//   * conv3's output channels are walked 32 at a time (= one K step of conv1'), software-pipelined: the MFMAs of conv3 chunk c + 1 are
//     issued in front of the epilogue of chunk c (the stream holds the slabs in that order: A(0) | A(1) B(0) | A(2) B(1) | ...).  The shortcut chunk is DMA'd two chunks
//     ahead into a wave-private staging tile (which aliases the patch buffers: chunk 0 is requested while conv2 still
//     runs on the other buffer), the trunk chunk is written over it in place, leaves as coalesced 16-byte row stores
//     and is pre-activated in registers for conv1';
//   * every wave issues the same vector-memory instructions (rows beyond M read row 0 and store to a dump page), so the
//     waits on the ring / patch / shortcut are COUNTED s_waitcnt vmcnt(N), N from a compile-time table of the schedule.
// Rounding points, product order (w.hi x.lo, w.lo x.hi, w.hi x.hi) and K order are those of the launches it replaces
// (conv3x3_stream tiles 19 / 20, then gemm_conv.hip's conv3 and conv1): bit-identical, tested per kernel and through the ResNet.
#include <type_traits>
#include <utility>

#include "common.h"
#include "hmmr_hip.h"

namespace {

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

__device__ u32x4 g_b1_dump[64];             // 1 KB: where the stores of rows beyond M go

struct B1Args {
    const char* h1;                         // [M][64] split rows (256 B): conv2's input
    const char* stream;                     // packing.pack_b1_unit_stream
    const float* scale2; const float* shift2;        // [64]: conv2's folded BN
    const bsplit_t* xp;                     // KC3B: [M][64], the folded shortcut's operand
    const float* scale3; const float* shift3; const float* pre_scale; const float* pre_shift;     // [256]
    const bsplit_t* res; int ldr;           // RES: the shortcut, rows of ldr elements
    bsplit_t* out;                          // [M][256]
    const float* scale1; const float* shift1; int relu1;     // [64]
    bsplit_t* out_h1;                       // [M][64]
    int M, H, W, n_tiles;
    unsigned long long* ts;                 // probe build (bit 64): [tiles][4 waves][16] s_memtime stamps, or NULL
};

// Development build only (tools/b1_probe_build.sh, -DB1_PROBE_BITS=n): drop the MFMAs (1), the HBM traffic (2: every tile reads tile 0's
// pixels and stores to the dump page), the waits and barriers of the slab steps (4) or the epilogue arithmetic (8) at COMPILE time, to
// see what bounds the kernel; 16: no filter stream (the ring keeps whatever it holds); 32: no fragment reads out of the ring.  Results are garbage in those modes; the product build compiles the switches away.
#ifndef B1_NT_RES
#define B1_NT_RES 0         /* A/B: the shortcut chunks (read exactly once) as non-temporal requests */
#endif
#ifndef B1_NT_ST
#define B1_NT_ST 0          /* A/B: the trunk stores as non-temporal stores */
#endif
#ifndef B1_PROBE_BITS
#define B1_PROBE_BITS 0
#endif
#define B1_PROBE(bit) (((B1_PROBE_BITS) & (bit)) != 0)
#define B1_STAMP(k) do { if (B1_PROBE(64) && a.ts) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); if ((threadIdx.x & 63) == 0) a.ts[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 16 + (k)] = t_; } } while (0)

constexpr int B1_BM = 128, B1_DEPTH = 256, B1_NCH = B1_DEPTH / 32;
constexpr int B1_NS = 5, B1_SLAB = 8192, B1_RING = B1_NS * B1_SLAB;
constexpr int B1_NPP = 4;                                   // 64-row pieces of a patch: 128 + 2 * 56 + 2 = 242 rows
constexpr int B1_PROWS = B1_NPP * 64 + 16;                   // + the zero rows
constexpr int B1_PBUF = B1_PROWS * 64;
constexpr int B1_OFF_P = B1_RING;                            // two patch buffers; later the staging tiles: wave w, tile b at b * PBUF + w * 4096
constexpr int B1_OFF_C = B1_OFF_P + 2 * B1_PBUF;             // scale3, shift3, pre_scale, pre_shift [256]; scale2, shift2, scale1, shift1 [64]
constexpr int B1_LDS = B1_OFF_C + 4 * B1_DEPTH * 4 + 4 * 64 * 4;
static_assert(2 * B1_LDS <= 160 * 1024, "two workgroups per CU");
constexpr int B1_CONV2_SLABS = 18;                           // 36 K steps of 4 KB

// ---- the schedule of vector-memory instructions per wave, by slab step t (the step that READS slab t): first the ring's DMA of
// slab t + 4 (2 instructions), then the step's other requests.  KC3: conv3's 16-wide K chunks (4, or 8 with a folded shortcut); NA = KC3 / 4
// slabs hold the conv3 fragments A(c) of one 32-channel chunk c of conv3's output, one slab its conv1' fragments B(c).  The tail's slabs
// come in the order the software pipeline consumes them -- conv3 of chunk c + 1 is issued BEFORE the epilogue of chunk c:
//     A(0) | A(1) B(0) | A(2) B(1) | ... | A(7) B(6) | B(7)
constexpr int b1_total(int kc3) { return B1_CONV2_SLABS + B1_NCH * (kc3 / 4 + 1); }
constexpr int b1_step_a(int c, int kc3) { return c == 0 ? B1_CONV2_SLABS : B1_CONV2_SLABS + kc3 / 4 + (c - 1) * (kc3 / 4 + 1); }   // first slab of A(c)
constexpr int b1_step_b(int c, int kc3) {                    // the slab of B(c)
    return c + 1 < B1_NCH ? B1_CONV2_SLABS + 2 * (kc3 / 4) + c * (kc3 / 4 + 1) : b1_total(kc3) - 1;
}
constexpr int b1_chunk_of_b(int t, int kc3) {                // c if step t is B(c), else -1
    for (int c = 0; c < B1_NCH; ++c) if (b1_step_b(c, kc3) == t) return c;
    return -1;
}
constexpr int b1_ring_ops(int t, int kc3) { return (t + (B1_NS - 1) < b1_total(kc3) && !B1_PROBE(16)) ? 2 : 0; }
// requests of step t behind its ring DMA -- first group: shortcut chunk 1 (step 18); second group: the patch of chunk 2 / 3 (steps 5 / 9),
// shortcut chunk 0 (step 14), and in a chunk's B step its 4 trunk stores + the shortcut chunk two ahead
constexpr int b1_extra_first(int t, int kc3, bool res) { return (t == B1_CONV2_SLABS && res) ? 4 : 0; }
constexpr int b1_extra_last(int t, int kc3, bool res) {
    if (t == 5 || t == 9) return 4;
    if (t == 14) return res ? 4 : 0;
    const int c = b1_chunk_of_b(t, kc3);
    return c < 0 ? 0 : 4 + ((res && c + 2 < B1_NCH) ? 4 : 0);
}
// instructions issued from the start of step 0 through step t's ring DMA (phase 0), first group (1), last group (2)
constexpr int b1_cum(int t, int phase, int kc3, bool res) {
    int n = 0;
    for (int u = 0; u <= t; ++u) {
        n += b1_ring_ops(u, kc3);
        if (u < t || phase >= 1) n += b1_extra_first(u, kc3, res);
        if (u < t || phase >= 2) n += b1_extra_last(u, kc3, res);
    }
    return n;
}
// what may stay in flight at the wait of step s: everything issued after the youngest request step s depends on -- all of step s - 4
// (its ring DMA is slab s; a patch or shortcut chunk 0 requested there is first read in step s or later) and, in a B step, the
// shortcut chunk its epilogue adds (chunk 1: the first group of step 18; chunk c >= 2: the last group of B(c - 2))
constexpr int b1_wait_n(int s, int kc3, bool res) {
    int cover = b1_cum(s - 4, 2, kc3, res);
    const int c = b1_chunk_of_b(s, kc3);
    if (res && c == 1) { const int v = b1_cum(B1_CONV2_SLABS, 1, kc3, res); cover = v > cover ? v : cover; }
    if (res && c >= 2) { const int v = b1_cum(b1_step_b(c - 2, kc3), 2, kc3, res); cover = v > cover ? v : cover; }
    return b1_cum(s - 1, 2, kc3, res) - cover;
}

// s_waitcnt vmcnt(N) lgkmcnt(0) (gfx9 encoding: vmcnt[3:0] | expcnt[6:4] | lgkmcnt[11:8] | vmcnt_hi[15:14])
template <int N> __device__ __forceinline__ void b1_wait() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
    __builtin_amdgcn_s_waitcnt((N & 15) | (7 << 4) | (0 << 8) | ((N >> 4) << 14));
}

template <int V> using b1_ic = std::integral_constant<int, V>;
// f(b1_ic<0>{}), f(b1_ic<1>{}), ... in order
template <class F, int... I> __device__ __forceinline__ void b1_for(F&& f, std::integer_sequence<int, I...>) { (f(b1_ic<I>{}), ...); }

struct xfrag { shalf8 hi, lo; };            // MFMA B-operand fragment: 32 pixels x 16 channels

// D layout of a 32-channel block (nh / nl[g][i]: the packed hi / lo halves of channels 8 g + 4 lh + 2 i, + 1 of pixel lr) -> the two
// B-operand fragments of its 16-wide K chunks (lane (lr, lh): channels 16 k + 8 lh .. + 7): one v_permlane32_swap per register pair
__device__ __forceinline__ void b1_d_to_b(const shalf2 (&nh)[4][2], const shalf2 (&nl)[4][2], xfrag (&th)[2]) {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        unsigned fh[4], fl[4];
#pragma unroll
        for (int d = 0; d < 2; ++d) {
            const u32x2 sh = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, nh[2 * k][d]),
                                                              __builtin_bit_cast(unsigned, nh[2 * k + 1][d]), false, false);
            const u32x2 sl = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, nl[2 * k][d]),
                                                              __builtin_bit_cast(unsigned, nl[2 * k + 1][d]), false, false);
            fh[d] = sh[0]; fh[2 + d] = sh[1];
            fl[d] = sl[0]; fl[2 + d] = sl[1];
        }
        th[k].hi = __builtin_bit_cast(shalf8, u32x4{fh[0], fh[1], fh[2], fh[3]});
        th[k].lo = __builtin_bit_cast(shalf8, u32x4{fl[0], fl[1], fl[2], fl[3]});
    }
}
// ---- the epilogues' arithmetic.  Two waves per SIMD share one vector issue port, and the unit does ~18 values per lane and 32 output
// channels: the instruction COUNT is what the kernel's skeleton costs (the first version, written with split4 and float conversions,
// spent 22 vector instructions per value: 0.32 ms of a 0.6 ms launch with the MFMAs and the HBM traffic taken out, profiles/r05b).
// split2_mix / split4_mix (common.h): the split in that form.
__device__ __forceinline__ void b1_split2(float c0, float c1, unsigned& h, unsigned& l) { split2_mix(c0, c1, h, l); }
// the value a packed (hi, lo) pair holds: hi + lo, exact before its one rounding to fp32 (= (float)hi + (float)lo); low / high half of the dwords
__device__ __forceinline__ float b1_sum_lo(unsigned h, unsigned l) { return split_sum_lo(h, l); }
__device__ __forceinline__ float b1_sum_hi(unsigned h, unsigned l) { return split_sum_hi(h, l); }
__device__ __forceinline__ void b1_split4(const float (&v)[4], float lo_clamp, unsigned (&h)[2], unsigned (&l)[2], float& satm) { split4_mix(v, lo_clamp, h, l, satm); }

// mma3 (common.h), or under probe bit 1 something that only keeps its operands alive
__device__ __forceinline__ f32x16 b1_mma3(const wfrag& w, const shalf8& xh, const shalf8& xl, f32x16 c) {
    if constexpr (B1_PROBE(1)) { c[0] += (float)w.hi[0] + (float)w.lo[0] + (float)xh[0] + (float)xl[0]; return c; }
    else return mma3(w, xh, xl, c);
}

// KC3B: 16-wide K chunks of conv3 that come from xp (0, or 4: the folded shortcut); RES: a shortcut tensor is added
template <int KC3B, bool RES>
__global__ __launch_bounds__(256, 2) void b1_unit_kernel(const B1Args a) {
    constexpr int KC3 = 4 + KC3B, TOTAL = b1_total(KC3);
    constexpr int NS = B1_NS, SLAB = B1_SLAB, NCH = B1_NCH, DEPTH = B1_DEPTH;
    static_assert(!(RES && KC3B), "either a shortcut tensor or a folded one");

    extern __shared__ __attribute__((aligned(256))) char smem[];
    B1_STAMP(0);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 31, lh = lane >> 5;
    const int m0 = B1_PROBE(2) ? 0 : xcd_remap(blockIdx.x, a.n_tiles) * B1_BM;
    const int W = a.W, HW = a.H * a.W;
    const int mbase = m0 + wave * 32;                           // this wave's first pixel
    const int lane16 = lane * 16;

    // ---- the filter stream: slab -> ring slot slab % 5; each wave moves a quarter (2 x 1 KB)
    const char* gstream = a.stream + wave * 2048 + lane16;
    auto ring_dma = [&](int slab) {                             // slab is a constant after unrolling
        if (B1_PROBE(16)) return;                               // (probe: nothing enters the ring after the prologue... nor in it)
        const char* src = gstream + (long long)slab * SLAB;
        char* dst = smem + (slab % NS) * SLAB + wave * 2048;
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)dst, 16, 0, 0);       // (the instruction offset moves both addresses)
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)dst, 16, 1024, 0);
    };
    // ---- patch DMA: piece q of this wave = rows 64 q + 16 wave .. + 15 of the patch (row r = input pixel m0 - W - 1 + r), four
    // lanes per row (one coalesced 64-byte chunk), source slot swizzled.  Rows outside the tensor read its first / last pixel: only
    // out-of-image taps (which read a zero row instead) and pixels >= M see them
    const char* pptr[B1_NPP];
#pragma unroll
    for (int q = 0; q < B1_NPP; ++q) {
        const int r = 64 * q + 16 * wave + (lane >> 2);
        int px = m0 - W - 1 + r;
        px = px < 0 ? 0 : (px >= a.M ? a.M - 1 : px);
        pptr[q] = a.h1 + (long long)px * 256 + (((lane & 3) ^ ((r >> 2) & 3)) << 4);
    }
    auto patch_dma = [&](int c, int buf) {
        char* dst = smem + B1_OFF_P + buf * B1_PBUF + wave * 1024;
#pragma unroll
        for (int q = 0; q < B1_NPP; ++q)
            __builtin_amdgcn_global_load_lds((gptr_t)(pptr[q] + c * 64), (lptr_t)(dst + q * 4096), 16, 0, 0);
    };
    // ---- staging tiles of this wave (rows = pixels, 128 B = 8 slots of 16 B: hi / lo of 4 groups interleaved, slot XOR-swizzled by
    // (row >> 1) & 7).  A row piece q covers rows 8 q .. 8 q + 7: lane L = (row 8 q + (L >> 3), physical slot L & 7)
    const int rsub = lane >> 3, pslot = lane & 7;
    auto stg_of = [&](int b) -> char* { return smem + B1_OFF_P + b * B1_PBUF + wave * 4096; };
    const bsplit_t* rrow[4];
    bsplit_t* orow[4];
    bsplit_t* hrow[4];
    int ostep[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int r = 8 * q + rsub, m = mbase + r;
        const int ls = pslot ^ ((r >> 1) & 7);
        const bool ok = m < a.M && !B1_PROBE(2);
        orow[q] = ok ? a.out + (long long)m * DEPTH + ls * 4 : (bsplit_t*)g_b1_dump + lane * 4;
        hrow[q] = ok ? a.out_h1 + (long long)m * 64 + ls * 4 : (bsplit_t*)g_b1_dump + lane * 4;
        ostep[q] = ok ? 32 : 0;
        rrow[q] = RES ? a.res + (long long)(ok ? m : 0) * a.ldr + ls * 4 : nullptr;
    }
    auto res_dma = [&](int chunk, int b) {
        if constexpr (RES) {
            char* dst = stg_of(b);
#pragma unroll
            for (int q = 0; q < 4; ++q)
                __builtin_amdgcn_global_load_lds((gptr_t)(rrow[q] + chunk * 32), (lptr_t)(dst + q * 1024), 16, 0, B1_NT_RES ? 2 : 0);
        }
    };

    // ---- prologue: the constants (one value of each array per thread, requested up front: a load under a branch would cost its own
    // round trip), four slabs, the first two patch chunks, the zero rows
    const float c_s3 = (a.scale3 ? a.scale3 : a.pre_scale)[tid], c_b3 = (a.shift3 ? a.shift3 : a.pre_shift)[tid];
    const float c_ps = a.pre_scale[tid], c_pb = a.pre_shift[tid];
    const float c_64 = (tid < 128 ? (tid < 64 ? a.scale2 : a.shift2) : (tid < 192 ? a.scale1 : a.shift1))[tid & 63];
#pragma unroll
    for (int s_ = 0; s_ < NS - 1; ++s_) ring_dma(s_);
    patch_dma(0, 0);
    patch_dma(1, 1);
    float* sS3 = (float*)(smem + B1_OFF_C);
    float* sB3 = sS3 + DEPTH;
    float* sPS = sB3 + DEPTH;
    float* sPB = sPS + DEPTH;
    float* sS2 = sPB + DEPTH;                                   // scale2, shift2, scale1, shift1: 64 floats each, contiguous
    float* sB2 = sS2 + 64;
    float* sS1 = sB2 + 64;
    float* sB1 = sS1 + 64;
    {
        sS3[tid] = a.scale3 ? c_s3 : 1.0f;
        sB3[tid] = a.shift3 ? c_b3 : 0.0f;
        sPS[tid] = c_ps;
        sPB[tid] = c_pb;
        sS2[tid] = c_64;
        if (tid < 128) {                                        // 16 zero rows behind the data rows of either patch buffer
            const int pl = tid >> 6, o = tid & 63;
            *(u32x4*)(smem + B1_OFF_P + pl * B1_PBUF + B1_NPP * 64 * 64 + o * 16) = u32x4{0u, 0u, 0u, 0u};
        }
    }
    // the folded shortcut's operand: this wave's 32 pixels x 64 channels as B-operand fragments (lane (lr, lh): channels 16 kc + 8 lh ..)
    xfrag xh[KC3];
    if constexpr (KC3B > 0) {
        const int m = mbase + lr;
        const long long row = m < a.M ? m : 0;
#pragma unroll
        for (int kc = 0; kc < KC3B; ++kc) {
            const bsplit_t* p = a.xp + row * 64 + (2 * kc + lh) * 8;
            xh[4 + kc].hi = *(const shalf8*)p;
            xh[4 + kc].lo = *((const shalf8*)p + 1);
        }
    }
    // border flags of this lane's pixel (SAME padding: a tap that leaves the image reads a zero row)
    bool top, bot, lef, rig;
    {
        const int m = mbase + lr;
        const int rem = m % HW, y = rem / W, x = rem - y * W;
        top = y == 0; bot = y == a.H - 1; lef = x == 0; rig = x == W - 1;
    }
    const int rb0 = wave * 32 + lr;

    f32x16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_waitcnt(0);                              // everything landed (vmcnt(0) lgkmcnt(0) expcnt(0))
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);

    B1_STAMP(1);
    // ---- the start of slab step S: slab S has landed for every wave, slab S - 1 is free
    auto slab_step = [&](auto s_c) {
        constexpr int S = decltype(s_c)::value;
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (B1_PROBE(4)) b1_wait<63>();
        else if constexpr (S >= NS - 1) b1_wait<b1_wait_n(S, KC3, RES)>();
        else b1_wait<63>();                                     // (lgkmcnt(0) alone: slabs 0 .. 3 landed in the prologue)
        if (!B1_PROBE(4)) __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (S + NS - 1 < TOTAL) ring_dma(S + NS - 1);
        if constexpr (S == 5) patch_dma(2, 0);                  // (buffer 0: chunk 0, last read in K step 8 = slab 4)
        if constexpr (S == 9) patch_dma(3, 1);                  // (buffer 1: chunk 1, last read in K step 17 = slab 8)
        if constexpr (S == 14) res_dma(0, 0);                   // (buffer 0: chunk 2, last read in K step 26 = slab 13)
        if constexpr (S == B1_CONV2_SLABS) res_dma(1, 1);       // (buffer 1: chunk 3, last read in K step 35 = slab 17)
        __builtin_amdgcn_sched_barrier(0);
    };
    // fragment f (0 .. 3) of slab S: hi plane, lo plane
    auto ring_frag = [&](int slab, int f) -> wfrag {
        const char* p = smem + (slab % NS) * SLAB + f * 2048 + lane16;
        wfrag w;
        if constexpr (B1_PROBE(32)) { w.hi = shalf8{}; w.lo = shalf8{}; w.hi[0] = (shalf_t)(float)slab; return w; }     // (probe: no fragment reads)
        w.hi = *(const shalf8*)p;
        w.lo = *(const shalf8*)(p + 1024);
        return w;
    };

    // ---- conv2: 36 K steps (chunk KT / 9 of 16 channels, tap KT % 9), two per slab.  Software pipeline over slabs: the fragments of
    // slab s + 1 (filters out of the ring, pixels out of the patch) are requested BEFORE the MFMAs of slab s issue, so LDS latency, the
    // wait on the ring and the barrier sit behind matrix work instead of in front of it
    struct Slab2 { wfrag w[4]; shalf8 xh[2], xl[2]; };         // two K steps: filters (K step, row block), pixels (K step)
    auto read_slab2 = [&](auto s_c, Slab2& f) {
        constexpr int S = decltype(s_c)::value;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int kt = 2 * S + h, tap = kt % 9, ky = tap / 3, kx = tap % 3, buf = (kt / 9) & 1;      // (constants after unrolling)
            const unsigned rowp = (unsigned)(rb0 + ky * W + kx);
            const unsigned ph = ((2u * lh) ^ ((rowp >> 2) & 3u)) << 4;
            unsigned ad = B1_OFF_P + buf * B1_PBUF + rowp * 64 + ph;
            if (tap != 4) {
                const bool outside = (ky == 0 && top) || (ky == 2 && bot) || (kx == 0 && lef) || (kx == 2 && rig);
                const unsigned zb = B1_OFF_P + buf * B1_PBUF + B1_NPP * 64 * 64 + (rowp & 3u) * 64 + ph;     // the zero row of the same bank
                ad = outside ? zb : ad;
            }
            f.xh[h] = *(const shalf8*)(smem + ad);
            f.xl[h] = *(const shalf8*)(smem + (ad ^ 16u));
#pragma unroll
            for (int j = 0; j < 2; ++j) f.w[2 * h + j] = ring_frag(S, 2 * h + j);
        }
        __builtin_amdgcn_sched_barrier(0);                      // (the requests stay in front of the MFMAs that follow)
    };
    auto mma_slab2 = [&](const Slab2& f) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[j] = b1_mma3(f.w[2 * h + j], f.xh[h], f.xl[h], acc[j]);
    };
    struct Slab1 { wfrag w[4]; };                               // four fragments of a tail slab
    auto read_slab1 = [&](int slab, Slab1& f) {
#pragma unroll
        for (int k = 0; k < 4; ++k) f.w[k] = ring_frag(slab, k);
        __builtin_amdgcn_sched_barrier(0);
    };
    Slab2 fa, fb;
    Slab1 fA;                                                   // conv3 fragments of the chunk about to start
    slab_step(b1_ic<0>{});
    read_slab2(b1_ic<0>{}, fa);
    auto slab_pair = [&](auto s_c) {                            // slabs S (in fa) and S + 1 (in fb), S even
        constexpr int S = decltype(s_c)::value;
        slab_step(b1_ic<S + 1>{});
        read_slab2(b1_ic<S + 1>{}, fb);
        mma_slab2(fa);
        slab_step(b1_ic<S + 2>{});                              // (S + 2 = 18: the first slab of the tail, conv3 chunk 0)
        if constexpr (S + 2 < B1_CONV2_SLABS) read_slab2(b1_ic<S + 2>{}, fa);
        else read_slab1(S + 2, fA);
        mma_slab2(fb);
    };
    b1_for([&](auto i_c) { slab_pair(b1_ic<2 * decltype(i_c)::value>{}); }, std::make_integer_sequence<int, B1_CONV2_SLABS / 2>{});

    B1_STAMP(2);
    // ---- conv2's epilogue: folded BN + ReLU, split, D layout -> the h2 panel as B-operand fragments
    float satm = 0.f;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        shalf2 nh[4][2], nl[4][2];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int n = 32 * j + 8 * g + 4 * lh;
            const f32x4 s4 = *(const f32x4*)(sS2 + n), b4 = *(const f32x4*)(sB2 + n);
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaf(acc[j][4 * g + e], s4[e], b4[e]);
            unsigned h[2], l[2];
            b1_split4(v, 0.f, h, l, satm);                      // (ReLU and the clamp to the fp16 range are one v_med3_f32)
#pragma unroll
            for (int i = 0; i < 2; ++i) { nh[g][i] = __builtin_bit_cast(shalf2, h[i]); nl[g][i] = __builtin_bit_cast(shalf2, l[i]); }
        }
        xfrag t2[2];
        b1_d_to_b(nh, nl, t2);
        xh[2 * j] = t2[0]; xh[2 * j + 1] = t2[1];
    }

    B1_STAMP(3);
    // ---- the tail: conv3's output channels 32 at a time
    f32x16 acc2[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[j][r] = 0.f;
    const int sw = (lr >> 1) & 7;
    // conv3 of chunk C into `dst` (zeroed here): K chunks in order over {h2, xp}.  Its first slab's fragments are in fA already
    auto conv3 = [&](auto c_c, f32x16& dst) {
        constexpr int C = decltype(c_c)::value, SA = b1_step_a(C, KC3);
#pragma unroll
        for (int r = 0; r < 16; ++r) dst[r] = 0.f;
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) dst = b1_mma3(fA.w[kc], xh[kc].hi, xh[kc].lo, dst);
        if constexpr (KC3B > 0) {
            slab_step(b1_ic<SA + 1>{});
            Slab1 fA2;
            read_slab1(SA + 1, fA2);
#pragma unroll
            for (int kc = 0; kc < 4; ++kc) dst = b1_mma3(fA2.w[kc], xh[4 + kc].hi, xh[4 + kc].lo, dst);
        }
    };
    f32x16 accP, accQ;                                         // conv3 accumulators of even / odd chunks
    conv3(b1_ic<0>{}, accP);
    auto chunk = [&](auto c_c) {
        constexpr int C = decltype(c_c)::value, SB = b1_step_b(C, KC3), B = C & 1;
        f32x16& acc1 = (C & 1) ? accQ : accP;
        // software pipeline: the MFMAs of conv3 chunk C + 1 are issued in front of the epilogue of chunk C, which then runs beside them
        if constexpr (C + 1 < NCH) {
            slab_step(b1_ic<b1_step_a(C + 1, KC3)>{});
            read_slab1(b1_step_a(C + 1, KC3), fA);
            conv3(b1_ic<C + 1>{}, (C & 1) ? accP : accQ);
        }
        // the conv1' fragments of this chunk are requested in front of the epilogue
        slab_step(b1_ic<SB>{});
        Slab1 fB;
        read_slab1(SB, fB);
        // * scale3 + shift3 (+ shortcut), split, IN PLACE into the staging tile; the pre-activation of the STORED value
        char* stg = stg_of(B);
        shalf2 nh[4][2], nl[4][2];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int ch = C * 32 + 8 * g + 4 * lh;
            const f32x4 s4 = *(const f32x4*)(sS3 + ch), b4 = *(const f32x4*)(sB3 + ch);
            char* ph = stg + lr * 128 + (((2 * g) ^ sw) << 4) + 8 * lh;
            char* pl = stg + lr * 128 + (((2 * g + 1) ^ sw) << 4) + 8 * lh;
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaf(acc1[4 * g + e], s4[e], b4[e]);
            if constexpr (RES) {
                const unsigned long long h = *(const unsigned long long*)ph, l = *(const unsigned long long*)pl;
                v[0] += b1_sum_lo((unsigned)h, (unsigned)l);
                v[1] += b1_sum_hi((unsigned)h, (unsigned)l);
                v[2] += b1_sum_lo((unsigned)(h >> 32), (unsigned)(l >> 32));
                v[3] += b1_sum_hi((unsigned)(h >> 32), (unsigned)(l >> 32));
            }
            unsigned oh[2], ol[2];
            if constexpr (B1_PROBE(8)) {
                oh[0] = __float_as_uint(v[0]); oh[1] = __float_as_uint(v[1]); ol[0] = __float_as_uint(v[2]); ol[1] = __float_as_uint(v[3]);
            } else b1_split4(v, -HMMR_SPLIT_MAX, oh, ol, satm);
            *(unsigned long long*)ph = (unsigned long long)oh[0] | ((unsigned long long)oh[1] << 32);
            *(unsigned long long*)pl = (unsigned long long)ol[0] | ((unsigned long long)ol[1] << 32);
            if constexpr (B1_PROBE(8)) {
#pragma unroll
                for (int i = 0; i < 2; ++i) { nh[g][i] = __builtin_bit_cast(shalf2, oh[i]); nl[g][i] = __builtin_bit_cast(shalf2, ol[i]); }
                continue;
            }
            // the next unit's pre-activation of the value AS STORED (hi + lo), ReLU'd, clamped and split again
            const f32x4 ps = *(const f32x4*)(sPS + ch), pb = *(const f32x4*)(sPB + ch);
            float y[4];
            y[0] = fmaf(b1_sum_lo(oh[0], ol[0]), ps[0], pb[0]);
            y[1] = fmaf(b1_sum_hi(oh[0], ol[0]), ps[1], pb[1]);
            y[2] = fmaf(b1_sum_lo(oh[1], ol[1]), ps[2], pb[2]);
            y[3] = fmaf(b1_sum_hi(oh[1], ol[1]), ps[3], pb[3]);
            unsigned qh[2], ql[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                b1_split2(split_relu(y[2 * i]), split_relu(y[2 * i + 1]), qh[i], ql[i]);
                nh[g][i] = __builtin_bit_cast(shalf2, qh[i]); nl[g][i] = __builtin_bit_cast(shalf2, ql[i]);
            }
        }
        xfrag th[2];
        b1_d_to_b(nh, nl, th);
        // the trunk chunk leaves as 16-byte row pieces; the shortcut chunk two ahead is requested into the tile it leaves
        {
            u32x4 xr[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) xr[q] = *(const u32x4*)(stg + q * 1024 + lane16);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (B1_NT_ST) __builtin_nontemporal_store(xr[q], (u32x4*)orow[q]); else *(u32x4*)orow[q] = xr[q];
                orow[q] += ostep[q];
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (C + 2 < NCH) res_dma(C + 2, B);
            __builtin_amdgcn_sched_barrier(0);
        }
        // conv1' K step C: K chunks 2 C, 2 C + 1 against both row blocks
#pragma unroll
        for (int kcl = 0; kcl < 2; ++kcl)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc2[j] = b1_mma3(fB.w[kcl * 2 + j], th[kcl].hi, th[kcl].lo, acc2[j]);
        B1_STAMP(4 + C);
    };
    b1_for(chunk, std::make_integer_sequence<int, NCH>{});

    // ---- conv1' epilogue: BN (+ ReLU), split, through this wave's staging tiles, coalesced stores
    __builtin_amdgcn_sched_barrier(0);
    const float lo1 = a.relu1 ? 0.f : -HMMR_SPLIT_MAX;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        char* stg = stg_of(j);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int n2 = j * 32 + 8 * g + 4 * lh;
            const f32x4 s4 = *(const f32x4*)(sS1 + n2), b4 = *(const f32x4*)(sB1 + n2);
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaf(acc2[j][4 * g + e], s4[e], b4[e]);
            unsigned oh[2], ol[2];
            b1_split4(v, lo1, oh, ol, satm);
            *(unsigned long long*)(stg + lr * 128 + (((2 * g) ^ sw) << 4) + 8 * lh) = (unsigned long long)oh[0] | ((unsigned long long)oh[1] << 32);
            *(unsigned long long*)(stg + lr * 128 + (((2 * g + 1) ^ sw) << 4) + 8 * lh) = (unsigned long long)ol[0] | ((unsigned long long)ol[1] << 32);
        }
        u32x4 xr[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) xr[q] = *(const u32x4*)(stg + q * 1024 + lane16);
#pragma unroll
        for (int q = 0; q < 4; ++q) *(u32x4*)(hrow[q] + (ostep[q] ? j * 32 : 0)) = xr[q];
    }
    split_flag_max(satm);
    B1_STAMP(12);
}

template <int KC3B, bool RES>
int launch_b1(const B1Args& base, hipStream_t stream) {
    B1Args a = base;
    a.n_tiles = (a.M + B1_BM - 1) / B1_BM;
    auto kern = b1_unit_kernel<KC3B, RES>;
    static DeviceOnce once;
    if (const unsigned long long bit = once.due()) {
        HMMR_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, B1_LDS));
        once.mark(bit);
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)a.n_tiles), dim3(256), B1_LDS, stream, a);
    HMMR_CHECK_HIP(hipGetLastError());
    return 0;
}

}  // namespace

// bytes of the filter stream of a block-1 unit (packing.pack_b1_unit_stream): conv2's 36 K steps of 4 KB, then per 32 channels of
// conv3's output kc3 = (64 + c_xp) / 16 fragments of conv3 and 4 of conv1', 2 KB each
extern "C" size_t hmmr_b1_unit_stream_bytes(int c_xp) {
    return (size_t)B1_CONV2_SLABS * B1_SLAB + (size_t)B1_NCH * (size_t)((64 + c_xp) / 16 + 4) * 2048;
}

// hmmr_bottleneck_tail with unit_stream set (called from bottleneck_split.hip)
int hmmr_b1_unit_split(const hmmr_tail_desc_t* d, hipStream_t stream) {
    HMMR_REQUIRE(d->h1 && !d->h2 && d->unit_stream && d->out && d->out_h1 && d->pre_scale && d->pre_shift && d->scale1 && d->shift1 &&
                 d->scale2 && d->shift2 && !d->out_pre && !d->res_strided && d->m > 0 && d->conv2_stride <= 1,
                 "hmmr_bottleneck_tail (f16x3, unit_stream): needs h1, the unit's filter stream, conv2's and the next conv1's constants, "
                 "the next unit's preact, out, out_h1 and a dense shortcut");
    HMMR_REQUIRE(d->c_mid == 64 && d->depth == 256 && d->n2 == 64 && d->hin > 0 && d->win > 0 && d->win <= 56 && d->m % (d->hin * d->win) == 0,
                 "hmmr_bottleneck_tail (f16x3, unit_stream): the 64 -> 256 -> 64 shape on whole images at most 56 pixels wide (got %d, %d, %d, %d x %d)",
                 d->c_mid, d->depth, d->n2, d->hin, d->win);
    const bool folded = d->xp != nullptr;
    HMMR_REQUIRE(folded != (d->res != nullptr), "hmmr_bottleneck_tail (f16x3, unit_stream): either a shortcut tensor (res) or a folded one (xp)");
    HMMR_REQUIRE(!folded || d->c_xp == 64, "hmmr_bottleneck_tail (f16x3, unit_stream): a folded shortcut has 64 input channels (got %d)", d->c_xp);
    HMMR_REQUIRE(folded || d->ldr >= d->depth, "hmmr_bottleneck_tail: residual row stride < depth");
    B1Args a = {};
    a.h1 = (const char*)d->h1; a.stream = (const char*)d->unit_stream; a.scale2 = d->scale2; a.shift2 = d->shift2;
    a.xp = (const bsplit_t*)d->xp;
    a.scale3 = d->scale3; a.shift3 = d->shift3; a.pre_scale = d->pre_scale; a.pre_shift = d->pre_shift;
    a.res = (const bsplit_t*)d->res; a.ldr = d->ldr; a.out = (bsplit_t*)d->out;
    a.scale1 = d->scale1; a.shift1 = d->shift1; a.relu1 = d->relu1; a.out_h1 = (bsplit_t*)d->out_h1;
    a.M = d->m; a.H = d->hin; a.W = d->win;
    if (B1_PROBE(64))       // (development build: the stamp buffer's address travels in the reserved debug words, like the other probe builds')
        a.ts = (unsigned long long*)(((unsigned long long)(unsigned)hmmr_debug_state()->reserved[1] << 32) | (unsigned)hmmr_debug_state()->reserved[0]);
    return folded ? launch_b1<4, false>(a, stream) : launch_b1<0, true>(a, stream);
}
