// Fused unit pair of a wide ResNet-v2 bottleneck block (blocks 2-3) for gfx950, f16x3 ("split") operands:
//
//     trunk' = conv3({h2 [, xp]}) * scale3 + shift3 [+ shortcut]     (1x1; bottleneck_v2 `conv3` + add; with xp the unit's conv
//                                                                      shortcut is folded into the same GEMM)
//     h1'    = relu(bn1'(conv1'(relu(bn_pre'(trunk')))))             (the NEXT unit's `preact` + `conv1`)
//
// in ONE kernel (slim resnet_v2.bottleneck as invoked at src/models.py:65-75), bit-identical to the two
// hmmr_conv_gemm launches it replaces.  bottleneck_split.hip does this for block 1 with the h2 panel in LDS and the filters
// as per-wave fragments from L2; with c_mid = 256 that panel does not fit, and the filter stream per pixel was 4x too high
// (64-pixel tiles: 0.72 ms against 0.28 ms for the two launches, DESIGN section 5.1 (0')).  This kernel turns the roles
// around:
//   * a WAVE owns 32 pixels for the whole unit pair and keeps their state in REGISTERS: the h2 panel as MFMA B-operand
//     fragments (32 px x 256 K x 4 B = 128 registers per lane) and the conv1' accumulators (32 px x 256 channels = 128
//     registers).  One wave per SIMD, 512 registers per lane (the accumulators and the panel sit in the AGPR half);
//   * a workgroup = 4 such waves = 128 pixels; the only thing they share is the FILTER STREAM: the host packs W3 and W1'
//     as one flat sequence of 2 KB MFMA A-operand fragments (hi plane | lo plane, lane-linear) in exactly the order the
//     kernel consumes them, the waves DMA it into a 7-slab LDS ring (16 KB slabs, global_load_lds, five slabs in flight:
//     L2-hit loads queue behind the kernel's own HBM traffic in a CU's memory pipeline, so the ring covers ~2 us) and every
//     wave reads every fragment (2 ds_read_b128, conflict-free).  2 MB per 128 pixels: half the L2 -> LDS bytes per FLOP of
//     the two launches it replaces;
//   * conv3's output channels are walked 32 at a time (= one MFMA row block = one K step of conv1').  Chunk c's
//     accumulator gets `* scale3 + shift3 + shortcut`, is split and written IN PLACE into a wave-private staging tile that
//     the shortcut chunk was DMA'd into two iterations earlier, leaves as coalesced 16-byte stores (= the trunk tensor),
//     and is pre-activated IN REGISTERS: the D layout of the MFMA (lane = pixel, 4 consecutive channels per register
//     group) becomes the B-operand layout of conv1' (lane = pixel, 8 consecutive channels) with one v_permlane32_swap per
//     register pair -- the pre-activated trunk never touches LDS or HBM;
//   * software pipeline over chunks: iteration `it` issues the MFMAs of conv3 chunk it and of conv1' K step it-2
//     ALTERNATELY (the dependent accumulator chains of either never run back to back) while the vector unit does the
//     epilogue of chunk it-1.  Rounding points, product order (x.lo*w.hi, x.hi*w.lo, x.hi*w.hi) and K order are those of
//     gemm_conv.hip: the results are bit-identical to the two launches (tested through the whole ResNet).
// Round 5: PERSISTENT WORKGROUPS for the block-2 shapes once a launch is at least two rounds of workgroups (one workgroup per CU, tiles
// b, b + grid, b + 2 grid ...: a.tpw of them): the ring is left streaming across the tile boundary (the stream is periodic and the loop already requests NS - 1 slabs past
// the end), the constants stay in LDS, so the second tile's prologue is its panel + first shortcut chunk instead of 96 KB of ring + both
// (the prologue was 20 k of a block-2 tile's ~75 k cycles).  The conv1' constants then live behind the other constants, not in the ring.
// Every wave executes the same sequence of vector-memory instructions (invalid rows read row 0 and store to a dump page),
// so the waits on the ring are COUNTED s_waitcnt vmcnt(N) with N a compile-time function of the position in the
// iteration.
#include <type_traits>

#include "common.h"
#include "hmmr_hip.h"

namespace {

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
typedef __attribute__((ext_vector_type(2))) _Float16 shalf2_t;

__device__ u32x4 g_pair_dump[64];          // 1 KB: where the stores of rows beyond M (and of the pipeline's fill / drain iterations) go

struct PairArgs {
    const bsplit_t* src[2]; int src_ld[2];  // conv3's K: KC3A 16-wide chunks from rows of src[0], KC3B from rows of src[1] (row strides in elements)
    const char* stream;                     // the filter stream (packing.pack_pair_stream)
    const float* scale3; const float* shift3; const float* pre_scale; const float* pre_shift;   // [DEPTH]
    const bsplit_t* res; int ldr;           // RES: the shortcut, rows of ldr elements
    bsplit_t* out;                          // [M][DEPTH]
    const float* scale1; const float* shift1; int relu1;
    bsplit_t* out_h1;                       // [M][N2]
    int M;
    int tpw;                                // tiles (of 128 pixels) per workgroup (tiles b, b + grid, ...): 1, or more for the shapes with TWO (see the kernel)
    float one;                              // 1.0f (a run-time value: see the epilogue)
    int probe;                              // always 0 in the product build (HMMR_GEMM_PROBE below)
    unsigned long long* ts;                 // probe build: [blocks][4 waves][8] s_memtime stamps, or NULL
};

// Development build only (-DHMMR_GEMM_PROBE, tools/probe_build.sh): drop the MFMAs (1), the ring's DMA and waits (2), its barriers (4),
// the epilogue arithmetic (8), the end-of-iteration stores / shortcut requests (16) or the fragment reads (32) to see what bounds
// the kernel.  Results are garbage in those modes; the product build compiles the switches away.
#ifdef HMMR_GEMM_PROBE
#ifdef HMMR_PAIR_PROBE_BITS     // compile-time switches: run-time ones put ~10 taken branches into every unit of the loop
#define PAIR_PROBE(a, bit) (((HMMR_PAIR_PROBE_BITS) & (bit)) != 0)
#else
#define PAIR_PROBE(a, bit) (((a).probe & (bit)) != 0)
#endif
#define PAIR_STAMP(k) do { if (a.ts) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); if ((threadIdx.x & 63) == 0) a.ts[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 8 + (k)] = t_; } } while (0)
#else
#define PAIR_PROBE(a, bit) false
#define PAIR_STAMP(k) do { } while (0)
#endif

constexpr int PAIR_NS = 7;                  // ring slabs
constexpr int PAIR_SLAB = 16384;            // 8 fragments of 2 KB

// fragment i of an iteration's FT = NA + NB fragments belongs to conv3 (A) when this holds (packing.pair_is_a is the same rule)
__host__ __device__ constexpr bool pair_is_a(int i, int na, int ft) { return ((i + 1) * na) / ft > (i * na) / ft; }
__host__ __device__ constexpr int pair_a_before(int i, int na, int ft) { return (i * na) / ft; }
// vector-memory instructions a wave issues after the DMA of slab s+1 and before the wait of step s at position p of the CL-step
// iteration: the slab DMAs of NS - 3 steps (4 each) and every end-of-iteration block (eops) in between
__host__ __device__ constexpr int pair_wait_n(int p, int cl, int ns, int eops) {
    int n = 4 * (ns - 3);
    for (int d = 1; d <= ns - 2; ++d)
        if ((((p - d) % cl) + cl) % cl == cl - 1) n += eops;
    return n;
}

// s_waitcnt vmcnt(N) alone (gfx9 encoding: vmcnt[3:0] | expcnt[6:4] | lgkmcnt[11:8] | vmcnt_hi[15:14])
template <int N> __device__ __forceinline__ void wait_vm() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
    __builtin_amdgcn_s_waitcnt((N & 15) | (7 << 4) | (15 << 8) | ((N >> 4) << 14));
}
template <int CL, int NS, int EOPS> __device__ __forceinline__ void wait_pos(int p) {      // p is a constant after unrolling
    if (p == 0) wait_vm<pair_wait_n(0, CL, NS, EOPS)>();
    if constexpr (CL > 1) { if (p == 1) wait_vm<pair_wait_n(1, CL, NS, EOPS)>(); }
    if constexpr (CL > 2) { if (p == 2) wait_vm<pair_wait_n(2, CL, NS, EOPS)>(); }
    if constexpr (CL > 3) { if (p == 3) wait_vm<pair_wait_n(3, CL, NS, EOPS)>(); }
    if constexpr (CL > 4) { if (p == 4) wait_vm<pair_wait_n(4, CL, NS, EOPS)>(); }
    if constexpr (CL > 5) { if (p == 5) wait_vm<pair_wait_n(5, CL, NS, EOPS)>(); }
    static_assert(CL <= 6, "at most 6 slabs per iteration");
}

struct xfrag { shalf8 hi, lo; };            // MFMA B-operand fragment of 32 pixels x 16 channels

// LDS accesses of the loop as inline assembly.  hipcc waits lgkmcnt(0) wherever LDS data is first used -- never a counted wait -- which
// drains this unit's fragment requests whenever an older value is touched (~100 cycles with the matrix pipe idle, every unit: profiles/
// r04a).  These the compiler does not track: the loop places its own COUNTED waits (LDS operations of a wave complete in order).
template <int OFF> __device__ __forceinline__ shalf8 lds_rd128h(unsigned addr) {
    shalf8 v; asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF)); return v;
}
template <int OFF> __device__ __forceinline__ f32x4 lds_rd128f(unsigned addr) {
    f32x4 v; asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF)); return v;
}
template <int OFF> __device__ __forceinline__ unsigned long long lds_rd64(unsigned addr) {
    unsigned long long v; asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF)); return v;
}
template <int OFF> __device__ __forceinline__ void lds_wr64(unsigned addr, unsigned long long v) {
    asm volatile("ds_write_b64 %0, %1 offset:%2" : : "v"(addr), "v"(v), "n"(OFF) : "memory");
}
// fragment f (0..7) of the slab at LDS address `base` (already + 16 * lane): hi plane, lo plane
__device__ __forceinline__ wfrag lds_frag(unsigned base, int f) {      // f is a constant after unrolling
    wfrag w;
#define PAIR_F(k) if (f == k) { w.hi = lds_rd128h<k * 2048>(base); w.lo = lds_rd128h<k * 2048 + 1024>(base); }
    PAIR_F(0) PAIR_F(1) PAIR_F(2) PAIR_F(3) PAIR_F(4) PAIR_F(5) PAIR_F(6) PAIR_F(7)
#undef PAIR_F
    return w;
}
// one plane (h: 0 = hi, 1 = lo) of fragment f
__device__ __forceinline__ shalf8 lds_frag_half(unsigned base, int f, int h) {
    shalf8 v = {};
#define PAIR_F(k) if (f == k) { if (h == 0) v = lds_rd128h<k * 2048>(base); else v = lds_rd128h<k * 2048 + 1024>(base); }
    PAIR_F(0) PAIR_F(1) PAIR_F(2) PAIR_F(3) PAIR_F(4) PAIR_F(5) PAIR_F(6) PAIR_F(7)
#undef PAIR_F
    return v;
}
// s_waitcnt lgkmcnt(n) alone, n a constant after unrolling
__device__ __forceinline__ void wait_lgkm(int n) {
#define PAIR_W(k) if (n == k) __builtin_amdgcn_s_waitcnt((63 & 15) | (7 << 4) | (k << 8) | ((63 >> 4) << 14));
    PAIR_W(0) PAIR_W(1) PAIR_W(2) PAIR_W(3) PAIR_W(4) PAIR_W(5) PAIR_W(6) PAIR_W(7) PAIR_W(8) PAIR_W(9) PAIR_W(10) PAIR_W(11) PAIR_W(12)
    PAIR_W(13) PAIR_W(14) PAIR_W(15)
#undef PAIR_W
}

// mma3 (common.h) on two independent accumulators, their MFMAs issued alternately: neither dependent chain runs back to back
__device__ __forceinline__ void mma3x2(const wfrag& w0, const xfrag& x0, f32x16& c0, const wfrag& w1, const xfrag& x1, f32x16& c1) {
    c0 = mfma_split(w0.hi, x0.lo, c0);
    c1 = mfma_split(w1.hi, x1.lo, c1);
    c0 = mfma_split(w0.lo, x0.hi, c0);
    c1 = mfma_split(w1.lo, x1.hi, c1);
    c0 = mfma_split(w0.hi, x0.hi, c0);
    c1 = mfma_split(w1.hi, x1.hi, c1);
}

// KC3A / KC3B: 16-wide K chunks of conv3 from src[0] (h2) / src[1] (xp, the folded shortcut's operand); DEPTH = conv3's output
// channels; N2 = conv1' output channels; RES: a shortcut tensor is added
template <int KC3A, int KC3B, int DEPTH, int N2, bool RES>
__global__ __launch_bounds__(256, 1) void unit_pair_kernel(const PairArgs a) {
    constexpr int KC3 = KC3A + KC3B, NCH = DEPTH / 32, NF2 = N2 / 32;
    constexpr int NA = KC3, NB = 2 * NF2, FT = NA + NB;        // fragments per iteration: conv3 chunk / conv1' K step
    static_assert(FT % 8 == 0, "an iteration is a whole number of slabs");
    constexpr int CL = FT / 8;                                 // slabs (= steps) per iteration
    constexpr int NS = PAIR_NS, SLAB = PAIR_SLAB;
    constexpr int TOTAL = (NCH + 2) * CL;                      // slabs of the stream
    constexpr int OFF_STG = NS * SLAB;                         // [4 waves][2][4 KB]: 32 px x 32 channels x 4 B
    constexpr int OFF_C = OFF_STG + 4 * 2 * 4096;              // scale3, shift3, pre_scale, pre_shift: [DEPTH] floats each
    constexpr bool TWO = DEPTH <= 512;                         // room for conv1's constants behind them: the ring may stay busy across tiles
    constexpr int OFF_C1 = OFF_C + 4 * DEPTH * 4;              // TWO: scale1, shift1 [N2]
    constexpr int EOPS = RES ? 8 : 4;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    PAIR_STAMP(0);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 31, lh = lane >> 5;
    const int tpw = TWO ? a.tpw : 1;
    int mbase = blockIdx.x * 128 + wave * 32;                  // this wave's first pixel (of the workgroup's first tile: tiles b, b + grid, ...)

    float* sS3 = (float*)(smem + OFF_C);
    float* sB3 = sS3 + DEPTH;
    float* sPS = sB3 + DEPTH;
    float* sPB = sPS + DEPTH;
    // the constants: four floats of each array per thread, REQUESTED here and stored behind the ring's and the panel's requests -- as a
    // loop of scalar loads (one global round trip per 256 channels before anything else was in flight) this was half of the prologue
    static_assert(DEPTH <= 1024 && DEPTH % 4 == 0, "one f32x4 per thread");
    const int ci4 = tid * 4, cl4 = ci4 < DEPTH ? ci4 : DEPTH - 4;     // (unconditional requests: a load under a branch makes hipcc wait vmcnt(0))
    f32x4 c_s3 = *(const f32x4*)((a.scale3 ? a.scale3 : a.pre_scale) + cl4);
    f32x4 c_b3 = *(const f32x4*)((a.shift3 ? a.shift3 : a.pre_shift) + cl4);
    const f32x4 c_ps = *(const f32x4*)(a.pre_scale + cl4);
    const f32x4 c_pb = *(const f32x4*)(a.pre_shift + cl4);
    static_assert(N2 <= 256, "one conv1' constant pair per thread");
    float c_s1 = a.scale1[tid < N2 ? tid : 0], c_b1 = a.shift1[tid < N2 ? tid : 0];     // (for the h1' epilogue: no round trip at the end)

    // ---- the filter stream: slab `slab` -> ring slot; each wave moves a quarter (4 x 1 KB)
    // (uniform base + one 32-bit lane offset: the scalar-base addressing mode, no vector address arithmetic per request)
    // (one 64-bit address and one M0 per slab; the four pieces are instruction offsets)
    const char* gstream = a.stream + wave * 4096 + lane * 16;
    auto dma_slab = [&](int slab, int slot_) {
        const char* src = gstream + (long long)slab * SLAB;
        char* dst = smem + slot_ * SLAB + wave * 4096;
        // (the instruction offset moves the global AND the LDS address)
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)dst, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)dst, 16, 1024, 0);
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)dst, 16, 2048, 0);
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)dst, 16, 3072, 0);
    };
    // ---- staging tile of this wave: rows = pixels (128 B = 8 slots of 16 B: [hi 4 groups interleaved with lo]), slot XOR-swizzled by
    // (row >> 1) & 7.  A DMA / row piece q covers rows 8q .. 8q+7, lane L = (row 8q + (L >> 3), physical slot L & 7).
    const int rsub = lane >> 3, pslot = lane & 7;
    auto stg_of = [&](int b) -> char* { return smem + OFF_STG + (wave * 2 + b) * 4096; };
    auto dma_res = [&](int chunk, int b) {
        if constexpr (RES) {
            char* dst = stg_of(b);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int r = 8 * q + rsub, m = mbase + r;
                const int ls = pslot ^ ((r >> 1) & 7);
                const bsplit_t* src = a.res + (long long)(m < a.M ? m : 0) * a.ldr + chunk * 32 + ls * 4;
                __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(dst + q * 1024), 16, 0, 0);
            }
        }
    };

#pragma unroll
    for (int s_ = 0; s_ < NS - 1; ++s_) dma_slab(s_, s_);
    if constexpr (TWO) {
        // the constants go to LDS ONCE, in front of the tile loop (inside it their registers would stay live across its back edge:
        // the folded form then spilled); their requests have been out since the top of the kernel
        if (!a.scale3) c_s3 = f32x4{1.f, 1.f, 1.f, 1.f};
        if (!a.shift3) c_b3 = f32x4{0.f, 0.f, 0.f, 0.f};
        if (ci4 < DEPTH) {
            *(f32x4*)(sS3 + ci4) = c_s3; *(f32x4*)(sB3 + ci4) = c_b3; *(f32x4*)(sPS + ci4) = c_ps; *(f32x4*)(sPB + ci4) = c_pb;
        }
        if (tid < N2) { ((float*)(smem + OFF_C1))[tid] = c_s1; ((float*)(smem + OFF_C1))[N2 + tid] = c_b1; }
    }
    float satm = 0.f;                                          // largest |trunk value| of this thread: the flag is raised once per tile
    int slot = 0, dslab = NS - 1;                              // ring state: persists across the tiles of a workgroup
    // ---- a wave's h2 panel as B-operand fragments: lane (lr, lh) = pixel lr, channels 16 kc + 8 lh .. + 7.  Requested for the first
    // tile here, for every further tile of a persistent workgroup in front of the previous tile's h1' epilogue (13 k cycles of cover)
    xfrag xh[KC3];
    auto load_panel = [&](int mb) {
        const int m = mb + lr;
        const long long row = m < a.M ? m : 0;
#pragma unroll
        for (int kc = 0; kc < KC3; ++kc) {
            const bsplit_t* p = kc < KC3A ? a.src[0] + row * a.src_ld[0] + (2 * kc + lh) * 8
                                          : a.src[1] + row * a.src_ld[1] + (2 * (kc - KC3A) + lh) * 8;
            xh[kc].hi = *(const shalf8*)p;
            xh[kc].lo = *((const shalf8*)p + 1);
        }
    };
    load_panel(mbase);
#pragma unroll 1
    for (int t = 0; t < tpw; ++t, mbase += gridDim.x * 128) {
    if (t > 0 && (blockIdx.x + t * gridDim.x) * 128 >= a.M) break;          // (the last round of tiles is a partial one)
    dma_res(0, 0);
    if constexpr (RES) {                                       // the pipeline's fill iteration reads tile 1 before any shortcut chunk landed in it: finite values
#pragma unroll
        for (int q = 0; q < 4; ++q) *(u32x4*)(stg_of(1) + q * 1024 + lane * 16) = u32x4{0u, 0u, 0u, 0u};
    }

    {
        if constexpr (!TWO) {
            if (!a.scale3) c_s3 = f32x4{1.f, 1.f, 1.f, 1.f};
            if (!a.shift3) c_b3 = f32x4{0.f, 0.f, 0.f, 0.f};
            if (ci4 < DEPTH) {
                *(f32x4*)(sS3 + ci4) = c_s3; *(f32x4*)(sB3 + ci4) = c_b3; *(f32x4*)(sPS + ci4) = c_ps; *(f32x4*)(sPB + ci4) = c_pb;
            }
        }
        // the panel is read by MFMAs only (B operand): keep it in the AGPR half of the register file, where the conv1' accumulators
        // already are -- left to itself the allocator parks part of it there anyway and copies it back before every use
#pragma unroll
        for (int kc = 0; kc < KC3; ++kc) asm volatile("" : "+a"(xh[kc].hi), "+a"(xh[kc].lo));
    }

    f32x16 acc2[NF2];
#pragma unroll
    for (int j = 0; j < NF2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[j][r] = 0.f;
    f32x16 acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc1[r] = 0.f;
    xfrag th[2];                                               // conv1' operand of K step it-2 (kcl = 0, 1)
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int e = 0; e < 8; ++e) { th[k].hi[e] = (shalf_t)0.f; th[k].lo[e] = (shalf_t)0.f; }

    __builtin_amdgcn_sched_barrier(0);
    PAIR_STAMP(1);
    __builtin_amdgcn_s_waitcnt(0);                             // everything landed (vmcnt(0) lgkmcnt(0) expcnt(0))
    __syncthreads();
    PAIR_STAMP(2);

    const int lane16 = lane * 16;
    const unsigned lds0 = (unsigned)(unsigned long long)(lptr_t)smem;      // LDS byte address of smem (0 unless something is linked in front)
    const int sw = (lr >> 1) & 7;
    const float one = a.one;                                   // 1.0f the compiler cannot fold: fma(h, one, l) is ONE v_fma_mix_f32
    // row pieces of the staging tile <-> global rows: piece q = rows 8q + rsub, this lane's 16 bytes = logical slot ls of the row
    // running pointers, advanced by one chunk (32 channels) per iteration: where the next shortcut chunk is read (chunk 1 first: chunk 0
    // is requested in the prologue) and where the next trunk chunk goes (rows beyond M: the dump page, not advanced)
    const bsplit_t* rrow[RES ? 4 : 1];
    bsplit_t* orow[4];
    int ostep[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int r = 8 * q + rsub, m = mbase + r;
        const int ls = pslot ^ ((r >> 1) & 7);
        const bool ok = m < a.M;
        orow[q] = ok ? a.out + (long long)m * DEPTH + ls * 4 : (bsplit_t*)g_pair_dump + lane * 4;
        ostep[q] = ok ? 32 : 0;
        if constexpr (RES) rrow[q] = a.res + (long long)(ok ? m : 0) * a.ldr + ls * 4 + 32;
    }
    constexpr int NU = 4 * CL;                                 // units (two fragments, six MFMAs) per iteration
    constexpr int LDU = NU >= 8 ? 2 : 1;                       // a group's LDS reads are requested this many units before its first piece
#ifdef HMMR_GEMM_PROBE
    unsigned long long ut[NU + 3] = {};                        // probe bit 64: cycles per unit (+ iteration head, tail, step boundary), summed over the iterations
    unsigned long long tprev = 0;
#define PAIR_UT(k) do { if (PAIR_PROBE(a, 64)) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); ut[k] += t_ - tprev; tprev = t_; } } while (0)
#else
#define PAIR_UT(k) do { } while (0)
#endif
    // ---- LDS addresses of this lane
    const unsigned fbase0 = lds0 + lane16;                                  // + slot * SLAB: fragment reads
    const unsigned cbase0 = lds0 + OFF_C + lh * 16;                         // + chunk * 128: this lane's 4 channels of group 0 in scale3 (the other
                                                                            //   groups / vectors: instruction offsets)
    unsigned soff[4][2];                                                    // staging tile 0 of this wave: (pixel lr, group g, hi / lo plane, channels 4 lh ..)
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) soff[g][pl] = lds0 + OFF_STG + wave * 8192 + lr * 128 + (((2 * g + pl) ^ sw) << 4) + 8 * lh;

    // what the epilogue of a chunk reads out of LDS (its constants, the shortcut chunk), per group of 8 channels
    f32x4 cs3[4], cb3[4], cps[4], cpb[4];
    unsigned long long rh[RES ? 4 : 1], rl[RES ? 4 : 1];
    // element j of group g's reads: 0..3 = scale3, shift3, pre_scale, pre_shift of its 4 channels, 4 / 5 = the shortcut's hi / lo halves
    auto group_load = [&](int e_, auto b_c, int g, int j) {    // b_c: the staging tile (e_ & 1), a compile-time constant
        constexpr int B = decltype(b_c)::value;
        const int ec_ = e_ < 0 ? 0 : (e_ >= NCH ? NCH - 1 : e_);
        const unsigned cb = cbase0 + ec_ * 128;
#define PAIR_G(k) if (g == k) { \
            if (j == 0) cs3[k] = lds_rd128f<k * 32>(cb); \
            if (j == 1) cb3[k] = lds_rd128f<k * 32 + DEPTH * 4>(cb); \
            if (j == 2) cps[k] = lds_rd128f<k * 32 + DEPTH * 8>(cb); \
            if (j == 3) cpb[k] = lds_rd128f<k * 32 + DEPTH * 12>(cb); \
            if constexpr (RES) { if (j == 4) rh[k] = lds_rd64<B * 4096>(soff[k][0]); if (j == 5) rl[k] = lds_rd64<B * 4096>(soff[k][1]); } }
        PAIR_G(0) PAIR_G(1) PAIR_G(2) PAIR_G(3)
#undef PAIR_G
    };
    auto group_loads = [&](int e_, auto b_c, int g) {
#pragma unroll
        for (int j = 0; j < (RES ? 6 : 4); ++j) group_load(e_, b_c, g, j);
    };
    // LDS operations a unit issues behind its 4 fragment reads: a group's reads (B), a group's two trunk writes (C)
    auto n_b = [](int v) -> int {
        v = ((v % NU) + NU) % NU;
        bool gl = v == NU - LDU;
        for (int g = 1; g < 4; ++g) gl = gl || v == (g * NU) / 4 - LDU;
        return gl ? (RES ? 6 : 4) : 0;
    };
    auto n_c = [](int v) -> int {
        v = ((v % NU) + NU) % NU;
        int n = 0;
        for (int pc = 2; pc < 16; pc += 4) n += ((pc * NU) / 16 == v) ? 2 : 0;
        return n;
    };

    wfrag wq[4][2];                                            // fragments of the units u, u+1, u+2 (slot u & 3): requested TWO units ahead
#pragma unroll
    for (int u = 0; u < 2; ++u) { wq[u][0] = lds_frag(fbase0 + slot * SLAB, 2 * u); wq[u][1] = lds_frag(fbase0 + slot * SLAB, 2 * u + 1); }
    group_loads(-1, std::integral_constant<int, 1>{}, 0);
    wait_lgkm(0);
    __builtin_amdgcn_sched_barrier(0);

    // One iteration: the MFMAs of conv3 chunk `it` (-> accN) and of conv1' K step it - 2 (-> acc2), the epilogue of chunk it - 1
    // (accumulator accO) beside them.  Called for even and odd `it` alternately with the two accumulators swapped, so neither is copied.
    auto iteration = [&](int it, auto par_c, f32x16& accN, f32x16& accO) {
        constexpr int PAR = decltype(par_c)::value;            // it & 1
        constexpr int BE = 1 - PAR;                            // staging tile of chunk e = it - 1
#ifdef HMMR_GEMM_PROBE
        if (PAIR_PROBE(a, 64)) tprev = __builtin_amdgcn_s_memtime();
#endif
        const int e = it - 1;                                  // chunk whose epilogue runs in this iteration
        const bool ev = e >= 0 && e < NCH;
        char* stg = stg_of(BE);
#pragma unroll
        for (int r = 0; r < 16; ++r) accN[r] = 0.f;
        shalf2 nh[4][2], nl[4][2];                             // pre-activated chunk e as packed halves, D layout (group g = channels 8g + 4lh ..)
        shalf2 oh[4][2], ol[4][2];                             // the trunk chunk as packed halves
        // Piece pc = 2 (2g + i) + stage of the epilogue: channels 2i, 2i+1 of group g.  Stage 0: conv3's epilogue (scale, shift,
        // + shortcut) and the split; stage 1: the pre-activation of the STORED value (= preact_slot_split of the consumer side, bit
        // for bit) and its split.  hi + lo and v - hi are single v_fma_mix instructions (exact product, one rounding: the value of
        // the convert / add sequences of gemm_conv.hip).  ~10 vector instructions each: a single wave issues one per ~8 cycles,
        // so a unit of six MFMAs covers about two dozen (tools/probes/mfma_fillers.hip).
        auto epilogue_piece = [&](int pc) {
            const int g = pc >> 2, i = (pc >> 1) & 1;
            if ((pc & 1) == 0) {
                shalf2 h2v, l2v;
                if constexpr (RES) {
                    h2v = __builtin_bit_cast(shalf2, (unsigned)(rh[g] >> (32 * i)));
                    l2v = __builtin_bit_cast(shalf2, (unsigned)(rl[g] >> (32 * i)));
                }
                float c[2], vraw[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    float v = fmaf(accO[4 * g + 2 * i + j], cs3[g][2 * i + j], cb3[g][2 * i + j]);
                    if constexpr (RES) v += __builtin_fmaf((float)h2v[j], one, (float)l2v[j]);
                    c[j] = split_clamp(v);
                    vraw[j] = v;
                }
                satm = sat_acc(satm, vraw[0], vraw[1]);          // (one v_maximum3_f32, NaN-propagating: hmmr_run_flags)
                shalf2 ph = {(shalf_t)c[0], (shalf_t)c[1]};
                asm volatile("" : "+v"(ph));                   // (one v_cvt_pk_f16_f32; its halves are read in place below)
                const shalf2 pl = {(shalf_t)__builtin_fmaf((float)ph[0], -one, c[0]), (shalf_t)__builtin_fmaf((float)ph[1], -one, c[1])};
                oh[g][i] = ph; ol[g][i] = pl;
                if (i == 1 && !PAIR_PROBE(a, 1024)) {          // the group is complete: in place into the staging tile
                    const unsigned long long wh = (unsigned long long)__builtin_bit_cast(unsigned, oh[g][0]) |
                                                  ((unsigned long long)__builtin_bit_cast(unsigned, oh[g][1]) << 32);
                    const unsigned long long wl = (unsigned long long)__builtin_bit_cast(unsigned, ol[g][0]) |
                                                  ((unsigned long long)__builtin_bit_cast(unsigned, ol[g][1]) << 32);
#define PAIR_S(k) if (g == k) { lds_wr64<BE * 4096>(soff[k][0], wh); lds_wr64<BE * 4096>(soff[k][1], wl); }
                    PAIR_S(0) PAIR_S(1) PAIR_S(2) PAIR_S(3)
#undef PAIR_S
                }
            } else {
                const shalf2 ph = oh[g][i], pl = ol[g][i];
                float y[2];
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    y[j] = split_relu(fmaf(__builtin_fmaf((float)ph[j], one, (float)pl[j]), cps[g][2 * i + j], cpb[g][2 * i + j]));
                shalf2 qh = {(shalf_t)y[0], (shalf_t)y[1]};
                asm volatile("" : "+v"(qh));
                nh[g][i] = qh;
                nl[g][i] = shalf2{(shalf_t)__builtin_fmaf((float)qh[0], -one, y[0]), (shalf_t)__builtin_fmaf((float)qh[1], -one, y[1])};
            }
        };

#pragma unroll
        for (int p = 0; p < CL; ++p) {
            // ---- step: slab s (ring slot `slot`) is read; slab s+1 must be visible before anybody prefetches out of it
            if (p == 0) PAIR_UT(NU);                           // (iteration head)
            __builtin_amdgcn_sched_barrier(0);
            if (!PAIR_PROBE(a, 2)) wait_pos<CL, NS, EOPS>(p);
            if (!PAIR_PROBE(a, 4)) __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            const int prev = slot == 0 ? NS - 1 : slot - 1;
            if (!PAIR_PROBE(a, 2)) dma_slab(dslab, prev);      // slab s + NS - 1 into the slot everybody left before this barrier
            dslab = dslab + 1 == TOTAL ? 0 : dslab + 1;
            const int nslot = slot + 1 == NS ? 0 : slot + 1;
            const unsigned fcur = fbase0 + slot * SLAB, fnxt = fbase0 + nslot * SLAB;
            PAIR_UT(NU + 2);                                   // (wait + barrier + DMA issue)
#pragma unroll
            for (int u = 0; u < 4; ++u) {                      // two fragments at a time
                const int i0 = p * 8 + 2 * u, i1 = i0 + 1;
                const int uu = p * 4 + u;
                // ---- everything this unit uses has landed: its fragments (requested two units ago) and, if it runs a group's
                // first piece, that group's reads (LDU units ago).  Issued since, and allowed to stay in flight: the writes of
                // unit uu - 2 (LDU 2) and all of unit uu - 1.
                __builtin_amdgcn_sched_barrier(0);
                wait_lgkm((LDU == 2 ? n_c(uu - 2) + 4 + n_b(uu - 1) : 0) + n_c(uu - 1));
                __builtin_amdgcn_sched_barrier(0);
                const wfrag w0 = wq[uu & 3][0], w1 = wq[uu & 3][1];
                const bool a0 = pair_is_a(i0, NA, FT), a1 = pair_is_a(i1, NA, FT);
                const int ka0 = pair_a_before(i0, NA, FT), ka1 = pair_a_before(i1, NA, FT);
                const int kb0 = i0 - ka0, kb1 = i1 - ka1;          // conv1' fragment kb: K chunk kb / NF2 of the step, row block kb % NF2
                // MFMA k of the unit: the three products (x.lo*w.hi, x.hi*w.lo, x.hi*w.hi) of the two fragments, alternately when
                // they feed different accumulators (neither dependent chain runs back to back), else in K order
                auto mfma_k = [&](int k) {
                    const int fi = (a0 && a1) ? k / 3 : (k & 1), j = (a0 && a1) ? k % 3 : (k >> 1);
                    const wfrag& w = fi ? w1 : w0;
                    const bool isa = fi ? a1 : a0;
                    const int ka = fi ? ka1 : ka0, kb = fi ? kb1 : kb0;
                    const shalf8& wop = (j == 1) ? w.lo : w.hi;
                    // (the empty asm ties the MFMA into the ordered chain of this unit's LDS requests: without it the instruction
                    //  selector is free to emit the six MFMAs and the reads as two clumps, and the barriers below preserve that)
                    if (isa) {
                        const shalf8& xop = (j == 0) ? xh[ka].lo : xh[ka].hi;
                        accN = mfma_split(wop, xop, accN);
                        asm volatile("" : "+a"(accN));
                    } else {
                        const shalf8& xop = (j == 0) ? th[kb / NF2].lo : th[kb / NF2].hi;
                        acc2[kb % NF2] = mfma_split(wop, xop, acc2[kb % NF2]);
                        asm volatile("" : "+a"(acc2[kb % NF2]));
                    }
                };
                // LDS read j of the unit: 0..3 = the two planes of the two fragments of unit uu + 2 (past the slab: out of slab s+1,
                // published by this step's barrier); 4.. = a group's reads, if this unit carries them
                const bool gl0 = uu == NU - LDU;
                int glg = 0;
#pragma unroll
                for (int g = 1; g < 4; ++g) if (uu == (g * NU) / 4 - LDU) glg = g;
                const int nrd = (PAIR_PROBE(a, 32) ? 0 : 4) + ((gl0 || glg) && !PAIR_PROBE(a, 512) ? (RES ? 6 : 4) : 0);
                auto lds_read_j = [&](int j) {
                    if (!PAIR_PROBE(a, 32)) {
                        if (j < 4) {
                            const int f = (u < 2 ? 2 * u + 4 : 2 * u - 4) + (j >> 1);
                            shalf8 v = lds_frag_half(u < 2 ? fcur : fnxt, f, j & 1);
                            if (j & 1) wq[(uu + 2) & 3][j >> 1].lo = v; else wq[(uu + 2) & 3][j >> 1].hi = v;
                            return;
                        }
                        j -= 4;
                    }
                    if (gl0) {
                        // group 0 of the NEXT iteration's epilogue: the shortcut chunk `it` was requested at the end of iteration
                        // it - 1, and this wave has issued the 4 CL slab requests of this iteration since (in-order completion)
                        if (RES && j == 4) wait_vm<4 * CL>();
                        group_load(it, std::integral_constant<int, PAR>{}, 0, j);
                    } else {
                        group_load(e, std::integral_constant<int, BE>{}, glg, j);
                    }
                };
                // one MFMA, then this gap's share of the reads (a ds_read_b128 holds the wave's LDS port for 16 cycles: more than two
                // behind one MFMA leave the matrix pipe idle); vector and scalar instructions may move across the gaps
                int jr = 0;
#pragma unroll
                for (int k = 0; k < 6; ++k) {
                    if (!PAIR_PROBE(a, 1)) mfma_k(k);
                    const int r = (nrd * (k + 1)) / 6 - (nrd * k) / 6;
#pragma unroll
                    for (int q = 0; q < 2; ++q)
                        if (q < r) lds_read_j(jr + q);
                    jr += r;
                    __builtin_amdgcn_sched_barrier(0x6);
                }
                // the epilogue of chunk e beside them: 16 pieces spread evenly over the NU units
#pragma unroll
                for (int pc = 0; pc < 16; ++pc)
                    if (uu == (pc * NU) / 16 && !PAIR_PROBE(a, 8) && !PAIR_PROBE(a, (pc & 1) ? 128 : 256)) epilogue_piece(pc);
                __builtin_amdgcn_sched_barrier(0);             // a unit's work stays in its unit
                PAIR_UT(uu);
            }
            slot = nslot;
        }
        // ---- D layout -> B-operand layout of conv1': one permlane32_swap per register pair (see the header)
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
            asm volatile("" : "+v"(th[k].hi), "+v"(th[k].lo)); // (assembled ONCE into the register quads the MFMAs read, not per use)
        }

        // ---- end-of-iteration block: the trunk chunk e leaves as 16-byte row pieces, the shortcut chunk e+2 is requested into the tile.
        // (every wave issues these 4 + 4 instructions in every iteration: the counted waits above rely on it.  The row reads are
        //  ordinary loads: the compiler waits lgkmcnt(0) in front of the stores, so the next iteration starts with nothing in flight)
        if (PAIR_PROBE(a, 16)) return;
        PAIR_UT(NU + 1);                                       // (swaps)
        u32x4 xr[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) xr[q] = *(const u32x4*)(stg + q * 1024 + lane16);
        if (ev) {
#pragma unroll
            for (int q = 0; q < 4; ++q) { *(u32x4*)orow[q] = xr[q]; orow[q] += ostep[q]; }
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) *(u32x4*)((bsplit_t*)g_pair_dump + lane * 4) = xr[q];
        }
        if constexpr (RES) {
            // chunk e + 2 (the last two iterations re-request the last chunk: harmless, and the instruction count stays uniform)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                __builtin_amdgcn_global_load_lds((gptr_t)rrow[q], (lptr_t)(stg + q * 1024), 16, 0, 0);
            if (e + 3 < NCH) {
#pragma unroll
                for (int q = 0; q < 4; ++q) rrow[q] += 32;
            }
        }
    };
    static_assert(NCH % 2 == 0, "the loop runs the NCH + 2 iterations in pairs");
    f32x16 accB;
#pragma unroll
    for (int r = 0; r < 16; ++r) accB[r] = 0.f;
    for (int it = 0; it <= NCH + 1; it += 2) {
        iteration(it, std::integral_constant<int, 0>{}, acc1, accB);
        iteration(it + 1, std::integral_constant<int, 1>{}, accB, acc1);
    }

    // ---- conv1' epilogue: BN (+ ReLU), split, through this wave's staging tile, coalesced stores
    // (the accumulators leave the loop as AGPRs: without this the copies into the vector file the epilogue needs are made at the
    //  end of EVERY iteration, 128 wasted instructions each)
#pragma unroll
    for (int j = 0; j < NF2; ++j) asm volatile("" : "+a"(acc2[j]));
    PAIR_STAMP(3);
    split_flag_max(satm);
#ifdef HMMR_GEMM_PROBE
    if (PAIR_PROBE(a, 64) && a.ts && lane == 0 && blockIdx.x == 0 && wave == 0)
        for (int k = 0; k < NU + 3; ++k) a.ts[4096 * 4 * 8 + k] = ut[k];
#endif
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_waitcnt(0);                             // (the last shortcut request must not land in the tile any more)
    PAIR_STAMP(4);
    if constexpr (TWO) {                                       // the next tile's panel: its registers are free since conv3's last chunk
        if (t + 1 < tpw) load_panel(mbase + (int)gridDim.x * 128);          // (past the last tile: rows clamp to row 0, never used)
    }
    // conv1's folded BN constants: through the (now idle) ring, one global round trip for the workgroup instead of one per row block
    // (TWO: behind the other constants since the prologue -- the ring is busy with the next tile's slabs)
    float* sS1 = (float*)(smem + (TWO ? OFF_C1 : 0));
    float* sB1 = sS1 + N2;
    if constexpr (!TWO) {
        __syncthreads();                                       // every wave has left the loop: the ring is free
        asm volatile("" : "+v"(c_s1), "+v"(c_b1));             // (requested in the prologue)
        if (tid < N2) { sS1[tid] = c_s1; sB1[tid] = c_b1; }
        __syncthreads();
    }
    char* stg = stg_of(0);
    float satmax = 0.f;
    const float lo1 = a.relu1 ? 0.f : -HMMR_SPLIT_MAX;
    bsplit_t* hrow[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int r = 8 * q + rsub, m = mbase + r;
        const int ls = pslot ^ ((r >> 1) & 7);
        hrow[q] = m < a.M ? a.out_h1 + (long long)m * N2 + ls * 4 : (bsplit_t*)g_pair_dump + lane * 4;
    }
#pragma unroll
    for (int of = 0; of < NF2; ++of) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int n2 = of * 32 + 8 * g + 4 * lh;
            const f32x4 s4 = *(const f32x4*)(sS1 + n2), b4 = *(const f32x4*)(sB1 + n2);
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = fmaf(acc2[of][4 * g + j], s4[j], b4[j]);
            // (ReLU and the clamp to the fp16 range as one v_med3_f32, the halves in mixed-precision FMA form: split4's bits at half its
            //  instruction count -- this epilogue is 128 values per lane in block 3 with nothing to hide behind)
            unsigned h2[2], l2[2];
            split4_mix(v, lo1, h2, l2, satmax);
            *(unsigned long long*)(stg + (of & 1) * 4096 + lr * 128 + (((2 * g) ^ sw) << 4) + 8 * lh) = (unsigned long long)h2[0] | ((unsigned long long)h2[1] << 32);
            *(unsigned long long*)(stg + (of & 1) * 4096 + lr * 128 + (((2 * g + 1) ^ sw) << 4) + 8 * lh) = (unsigned long long)l2[0] | ((unsigned long long)l2[1] << 32);
        }
        // (the two staging tiles alternate: the row reads of block `of` do not hold up the writes of block of + 1)
        u32x4 xr[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) xr[q] = *(const u32x4*)(stg + (of & 1) * 4096 + q * 1024 + lane16);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int m = mbase + 8 * q + rsub;
            *(u32x4*)(hrow[q] + (m < a.M ? of * 32 : 0)) = xr[q];
        }
    }
    split_flag_max(satmax);
    PAIR_STAMP(5);
    satm = 0.f;
    }   // tiles of this workgroup
}

// =====================================================================================================================================
// Round 6: the WAVE-SPECIALISED form of the same unit pair -- TWO waves per SIMD with different jobs (hmmr_debug_t.pair_form).
//
// What bounds the kernel above (profiles/r04_unit_pair_probe_study.log, r05w): one wave per SIMD issues the MFMAs, the ~200 vector
// instructions of a chunk's epilogue, the fragment reads, the ring's waits / barriers / DMA requests and the end-of-iteration block
// (permlane swaps, row reads, stores, shortcut requests) from ONE in-order instruction stream: the matrix pipe is busy 45 % of the loop,
// a unit of six MFMAs takes 285 cycles instead of 192, a step boundary 200, the end-of-iteration block ~1200.  Two such workgroups per
// CU do not fit: the state of 32 pixels (h2 panel + conv1' accumulators) is 256 registers per lane.  So the state is SPLIT between the
// two waves a SIMD can hold at 256 registers each, and with it the work:
//   * wave A (waves 0-3 of the 512-thread workgroup; 32 pixels each) holds the h2 PANEL (B-operand fragments in the AGPR half) and runs
//     conv3: per 32-channel chunk its MFMAs and the first half of the epilogue -- * scale3 + shift3 + shortcut, clamp, split -- written IN
//     PLACE into the pair's staging tile the shortcut chunk was DMA'd into (hi / lo slots of 16 bytes, rows of 128 bytes, XOR-swizzled);
//   * wave B (waves 4-7; wave w + 4 shares pixels and SIMD with wave w) holds the conv1' ACCUMULATORS and does everything else: the ring's
//     DMA requests and counted waits, and per chunk, from the staging tile A filled one iteration earlier: the trunk rows out as 16-byte
//     stores, the tile's 16-byte slots read back AS conv1' B-OPERAND FRAGMENTS (a slot is 8 consecutive channels of a pixel: the D-layout
//     -> operand-layout change that cost a permlane swap per register pair is an addressing mode here), the NEXT unit's pre-activation
//     of the stored value in registers (the same fma_mix / med3 / cvt sequence: same bits), the shortcut request three chunks ahead into
//     the tile just emptied, and conv1's MFMAs; at the end the h1' epilogue.
// A has no vector-memory instruction at all; B's are counted (ws_wait_n).  The two waves of a SIMD meet only at the ring's barrier, once
// per 8 fragments; three staging tiles per pair (chunk c in tile c % 3) give the shortcut DMA two iterations of lead.  The filter stream
// is the one packing.pack_pair_stream writes for the kernel above (A reads the conv3 fragments of a slab, B the conv1' fragments); the
// products, their order and every rounding point are unchanged: bit-identical to the one-wave form and to the two launches (tested).
// tools/probes/mfma_two_waves.hip (profiles/r06b) is the measurement behind the split: what two waves per SIMD buy over one.
constexpr int WS_EOPS = 8;                   // B's per-iteration block: 4 trunk stores + 4 shortcut requests
constexpr int WS_PE = 1;                     // ... issued in step 1 of an iteration (behind that step's slab requests)
constexpr int WS_PF = 3;                     // fragments requested this many units ahead (a ring of 4 fragment registers sets per wave)
template <int DEPTH> struct ws_ring { static constexpr int NS = DEPTH > 512 ? 5 : 6; };      // ring slabs that fit beside 48 KB of tiles
__host__ __device__ constexpr int ws_mod(int a, int m) { return ((a % m) + m) % m; }
__host__ __device__ constexpr int ws_min(int a, int b) { return a < b ? a : b; }
// B's vector-memory stream: step q issues D_q (4 slab requests) after its barrier, and E_q (WS_EOPS) behind them when q % cl == WS_PE.
// The wait in front of the barrier of step s (position p = s % cl) must have retired D of step s - (ns - 2) (slab s + 1: read from this
// barrier on) and, at p == 0, E of step s - 2 cl + WS_PE (its shortcut chunk is read by wave A from this barrier on): the number of
// requests issued since the younger of the two.
__host__ __device__ constexpr bool ws_is_e(int q, int cl) { return ws_mod(q, cl) == WS_PE % cl; }
__host__ __device__ constexpr int ws_wait_n(int p, int cl, int ns) {
    int n1 = ws_is_e(p - (ns - 2), cl) ? WS_EOPS : 0;
    for (int d = 1; d <= ns - 3; ++d) n1 += 4 + (ws_is_e(p - d, cl) ? WS_EOPS : 0);
    if (p != 0) return n1;
    int n2 = 0;
    for (int d = 1; d <= 2 * cl - WS_PE % cl - 1; ++d) n2 += 4 + (ws_is_e(p - d, cl) ? WS_EOPS : 0);
    return ws_min(n1, n2);
}
template <int CL, int NS> __device__ __forceinline__ void ws_wait_pos(int p) {       // p is a constant after unrolling
    if (p == 0) wait_vm<ws_wait_n(0, CL, NS)>();
    if constexpr (CL > 1) { if (p == 1) wait_vm<ws_wait_n(1, CL, NS)>(); }
    if constexpr (CL > 2) { if (p == 2) wait_vm<ws_wait_n(2, CL, NS)>(); }
    if constexpr (CL > 3) { if (p == 3) wait_vm<ws_wait_n(3, CL, NS)>(); }
    static_assert(CL >= 2 && CL <= 4, "2 .. 4 slabs per iteration");
}
// position (0 .. ft - 1, + ft per iteration ahead) of the k-th conv3 (want_a) / conv1' fragment counted from the start of an iteration
__host__ __device__ constexpr int ws_pos(int k, int na, int ft, bool want_a) {
    const int per = want_a ? na : ft - na;
    const int ahead = k / per;
    k -= ahead * per;
    int seen = 0;
    for (int i = 0; i < ft; ++i)
        if (pair_is_a(i, na, ft) == want_a) { if (seen == k) return i + ahead * ft; ++seen; }
    return -1;
}
// ---- the LDS request streams of the two roles, for the counted lgkmcnt waits.  A unit = one fragment = three MFMAs:
//   [the unit's extra reads (PRE)] [wait] [MFMAs, arithmetic] [the unit's writes (POST)] [the two reads of the fragment WS_PF units ahead]
// (the fragment goes into the register set the PREVIOUS unit's MFMAs read: a unit of distance between the last reader and the request).
// LDS operations of a wave complete in order, so "everything issued before X has landed" = "at most (operations issued after X) are
// outstanding".  A unit waits for its own fragment (requested at the end of unit u - WS_PF) and for whatever extra data it consumes.
// wave A (na units per iteration, four epilogue groups of na / 4 units): the group's constants + shortcut (4 reads) go out in the
// group's first unit, its arithmetic and its two trunk writes run WSA_DC units later
__host__ __device__ constexpr int wsa_dc(int na) { return na / 4 >= 4 ? 3 : 1; }
__host__ __device__ constexpr int wsa_pre(int v, int na) { return ws_mod(v, na) % (na / 4) == 0 ? 4 : 0; }
__host__ __device__ constexpr int wsa_post(int v, int na) { return ws_mod(v, na) % (na / 4) == wsa_dc(na) ? 2 : 0; }
__host__ __device__ constexpr int wsa_wait(int u, int na) {
    int n = wsa_pre(u, na);
    for (int v = u - WS_PF + 1; v <= u - 1; ++v) n += wsa_pre(v, na) + wsa_post(v, na) + 2;
    if (ws_mod(u, na) % (na / 4) == wsa_dc(na)) {               // the group's reads went out at the top of unit x
        const int x = u - wsa_dc(na);
        int m = wsa_post(x, na) + 2 + wsa_pre(u, na);
        for (int v = x + 1; v <= u - 1; ++v) m += wsa_pre(v, na) + wsa_post(v, na) + 2;
        n = ws_min(n, m);
    }
    return ws_min(n, 15);
}
// wave B (nb units per iteration): unit 0 reads the K-chunk-1 operand slots of the finished tile + their constants (6), unit 1 its rows
// (4); units 2, 3 pre-activate (need unit 0's), unit 4 stores the rows; unit nb / 2 reads the K-chunk-0 slots of the tile wave A is half
// way through (6), units nb / 2 + 2, + 3 pre-activate them for the next iteration
__host__ __device__ constexpr int wsb_pre(int v, int nb) { v = ws_mod(v, nb); return v == 0 ? 6 : v == 1 ? 4 : v == nb / 2 ? 6 : 0; }
__host__ __device__ constexpr int wsb_after_pre(int x, int u, int nb) {
    int m = 2 + wsb_pre(u, nb);
    for (int v = x + 1; v <= u - 1; ++v) m += wsb_pre(v, nb) + 2;
    return m;
}
__host__ __device__ constexpr int wsb_wait(int u, int nb) {
    int n = wsb_pre(u, nb);
    for (int v = u - WS_PF + 1; v <= u - 1; ++v) n += wsb_pre(v, nb) + 2;
    const int uu = ws_mod(u, nb);
    if (uu == 2 || uu == 3) n = ws_min(n, wsb_after_pre(0, uu, nb));
    if (uu == 4) n = ws_min(n, wsb_after_pre(1, 4, nb));
    if (uu == nb / 2 + 2 || uu == nb / 2 + 3) n = ws_min(n, wsb_after_pre(nb / 2, uu, nb));
    return ws_min(n, 15);
}
// the wait of unit u (a constant after unrolling) with its count as a template argument: left as a function of a run-time-typed u the
// compiler evaluates the counting loops at RUN time and switches over the sixteen encodings
template <int N> __device__ __forceinline__ void wait_lgkm_c() {
    static_assert(N >= 0 && N <= 15, "lgkmcnt is a 4-bit counter");
    __builtin_amdgcn_s_waitcnt((63 & 15) | (7 << 4) | (N << 8) | ((63 >> 4) << 14));
}
template <int NU, bool IS_A> __device__ __forceinline__ void ws_wait_unit(int u) {
#define WS_WU(k) if constexpr (NU > k) { if (u == k) wait_lgkm_c<IS_A ? wsa_wait(k, NU) : wsb_wait(k, NU)>(); }
    WS_WU(0) WS_WU(1) WS_WU(2) WS_WU(3) WS_WU(4) WS_WU(5) WS_WU(6) WS_WU(7) WS_WU(8) WS_WU(9) WS_WU(10) WS_WU(11) WS_WU(12) WS_WU(13) WS_WU(14) WS_WU(15)
#undef WS_WU
    static_assert(NU <= 16, "at most 16 units per iteration");
}
template <int OFF> __device__ __forceinline__ u32x4 lds_rd128u(unsigned addr) {
    u32x4 v; asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF)); return v;
}

// probe build (tools/pair_probe_build.sh): s_memtime stamps [blocks][8 waves][8] -- 0 start, 1 in front of the prologue's wait, 2 loop entry, 3 loop
// exit, 4 end (nothing inside the loop: s_memtime is a scalar-memory instruction, and reading its result drains the wave's LDS
// requests); bits of HMMR_PAIR_PROBE_BITS drop the MFMAs (1), the ring's requests and waits (2), the barriers (4), wave A's epilogue
// (8), wave B's per-iteration pieces (16), the fragment reads (32), wave B's pre-activation arithmetic (64), the units' counted LDS
// waits of wave A (128) / wave B (256), wave A's MFMAs only (512), wave B's MFMAs only (1024); -DWS_PRIO_A / -DWS_PRIO_B: s_setprio of the roles
#ifdef HMMR_GEMM_PROBE
#define WS_STAMP(k) do { if (a.ts) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); if ((threadIdx.x & 63) == 0) a.ts[(blockIdx.x * 8 + (threadIdx.x >> 6)) * 8 + (k)] = t_; } } while (0)
#define WS_RSTAMP(k) do { if (a.ts) { const unsigned long long t_ = __builtin_amdgcn_s_memrealtime(); if ((threadIdx.x & 63) == 0) a.ts[(blockIdx.x * 8 + (threadIdx.x >> 6)) * 8 + (k)] = t_; } } while (0)   /* the constant 100 MHz counter: slots 6 / 7 = start / end, so (t4 - t0) / (r7 - r6) x 100 MHz = the clock this wave ran at */
#define WS_ACC_BEGIN() const unsigned long long tb_ = a.ts ? __builtin_amdgcn_s_memtime() : 0ull
#define WS_ACC_END(var) do { if (a.ts) var += __builtin_amdgcn_s_memtime() - tb_; } while (0)
#define WS_ACC_STORE(k, var) do { if (a.ts && (threadIdx.x & 63) == 0) a.ts[(blockIdx.x * 8 + (threadIdx.x >> 6)) * 8 + (k)] = var; } while (0)
#else
#define WS_STAMP(k) do { } while (0)
#define WS_RSTAMP(k) do { } while (0)
#define WS_ACC_BEGIN() do { } while (0)
#define WS_ACC_END(var) do { } while (0)
#define WS_ACC_STORE(k, var) do { } while (0)
#endif

template <int KC3, int DEPTH, int N2>
__global__ __launch_bounds__(512, 1) void unit_pair_ws_kernel(const PairArgs a) {
    constexpr int NCH = DEPTH / 32, NF2 = N2 / 32;
    constexpr int NA = KC3, NB = 2 * NF2, FT = NA + NB;
    static_assert(FT % 8 == 0 && NA % 8 == 0 && NB % 8 == 0, "an iteration is a whole number of slabs; the fragment ring's indices are static");
    constexpr int CL = FT / 8, NS = ws_ring<DEPTH>::NS, SLAB = PAIR_SLAB;
    constexpr int TOTAL = (NCH + 2) * CL;
    constexpr int OFF_STG = NS * SLAB;                         // [4 pairs][3 tiles][4 KB]
    constexpr int OFF_C = OFF_STG + 4 * 3 * 4096;              // scale3, shift3, pre_scale, pre_shift [DEPTH] floats each
    constexpr int OFF_C1 = OFF_C + 4 * DEPTH * 4;              // scale1, shift1 [N2]
    static_assert(NCH % 2 == 0 && NCH >= 4, "the loop runs the NCH + 2 iterations in pairs; three shortcut chunks requested ahead");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    WS_STAMP(0);
    WS_RSTAMP(6);
    unsigned long long t_wait = 0ull;
    (void)t_wait;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pr = wave & 3;                                   // the pair = the 32-pixel block
    const bool roleB = wave >= 4;
    const int lr = lane & 31, lh = lane >> 5;
    const int mbase = blockIdx.x * 128 + pr * 32;
    const unsigned lds0 = (unsigned)(unsigned long long)(lptr_t)smem;
    const int lane16 = lane * 16;
    const int sw = (lr >> 1) & 7;

    // ---- constants -> LDS (requested first, stored by wave A's threads behind the other requests)
    float* sS3 = (float*)(smem + OFF_C);
    static_assert(DEPTH / 4 <= 256 && N2 <= 256, "wave A's 256 threads store the constants");
    const int ci4 = tid * 4, cl4 = ci4 < DEPTH ? ci4 : DEPTH - 4;

    const unsigned tile0 = lds0 + OFF_STG + pr * (3 * 4096);   // this pair's three staging tiles (LDS byte address)
    char* const tilep = smem + OFF_STG + pr * (3 * 4096);
    const unsigned fbase0 = lds0 + lane16;                     // + slot * SLAB: fragment reads

    if (!roleB) {
        // =============================================================================================================== wave A
        f32x4 c_s3 = *(const f32x4*)((a.scale3 ? a.scale3 : a.pre_scale) + cl4);
        f32x4 c_b3 = *(const f32x4*)((a.shift3 ? a.shift3 : a.pre_shift) + cl4);
        const f32x4 c_ps = *(const f32x4*)(a.pre_scale + cl4);
        const f32x4 c_pb = *(const f32x4*)(a.pre_shift + cl4);
        const float c_s1 = a.scale1[tid < N2 ? tid : 0], c_b1 = a.shift1[tid < N2 ? tid : 0];
        xfrag xh[KC3];
        {
            const int m = mbase + lr;
            const long long row = m < a.M ? m : 0;
#pragma unroll
            for (int kc = 0; kc < KC3; ++kc) {
                const bsplit_t* p = a.src[0] + row * a.src_ld[0] + (2 * kc + lh) * 8;
                xh[kc].hi = *(const shalf8*)p;
                xh[kc].lo = *((const shalf8*)p + 1);
            }
        }
        if (!a.scale3) c_s3 = f32x4{1.f, 1.f, 1.f, 1.f};
        if (!a.shift3) c_b3 = f32x4{0.f, 0.f, 0.f, 0.f};
        if (ci4 < DEPTH) {
            *(f32x4*)(sS3 + ci4) = c_s3; *(f32x4*)(sS3 + DEPTH + ci4) = c_b3;
            *(f32x4*)(sS3 + 2 * DEPTH + ci4) = c_ps; *(f32x4*)(sS3 + 3 * DEPTH + ci4) = c_pb;
        }
        if (tid < N2) { ((float*)(smem + OFF_C1))[tid] = c_s1; ((float*)(smem + OFF_C1))[N2 + tid] = c_b1; }
        // the register file of a kernel is ONE split for both roles: wave B needs 128 accumulator registers, so wave A's 128-register
        // panel + 32 accumulator registers keep to the same 128 by leaving the panel's last chunks in the vector half (MFMA operands may
        // come from either)
        constexpr int KC_A = KC3 > 12 ? 12 : KC3;
#pragma unroll
        for (int kc = 0; kc < KC3; ++kc) {
            if (kc < KC_A) asm volatile("" : "+a"(xh[kc].hi), "+a"(xh[kc].lo));
            else asm volatile("" : "+v"(xh[kc].hi), "+v"(xh[kc].lo));
        }
        f32x16 acc1, accB;
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc1[r] = 0.f; accB[r] = 0.f; }
        __builtin_amdgcn_sched_barrier(0);
        WS_STAMP(1);
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
        WS_STAMP(2);

#ifdef WS_PRIO_A
        __builtin_amdgcn_s_setprio(WS_PRIO_A);
#endif
        float satm = 0.f;
        int slot = 0;
        const float one = a.one;
        // this lane's 8 bytes of (pixel lr, group g, hi / lo plane) in tile 0 of the pair: channels 8 g + 4 lh .. + 3
        unsigned soff[4][2];
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) soff[g][pl] = tile0 + lr * 128 + (((2 * g + pl) ^ sw) << 4) + 8 * lh;
        const unsigned cbase0 = lds0 + OFF_C + lh * 16;        // + chunk * 128 (+ g * 32): scale3; shift3 at + DEPTH * 4

        wfrag wq[4];                                           // fragment of unit u in wq[u & 3], requested WS_PF units ahead
#pragma unroll
        for (int k = 0; k < WS_PF; ++k) {
            static_assert(ws_pos(WS_PF - 1, NA, FT, true) < 8, "the first fragments sit in the first slab");
            wq[k] = lds_frag(fbase0 + slot * SLAB, ws_pos(0, NA, FT, true) + k * (FT / NA));
        }

        auto iterA = [&](int it, f32x16& accN, f32x16& accO) {
            const int e = it - 1;
            const bool ev = e >= 0 && e < NCH;
            const int ec = e < 0 ? 0 : (e >= NCH ? NCH - 1 : e);
            const unsigned toff = (unsigned)(ws_mod(e, 3) * 4096);      // tile of chunk e
            const unsigned cb = cbase0 + ec * 128;
#pragma unroll
            for (int r = 0; r < 16; ++r) accN[r] = 0.f;
            f32x4 cs3v, cb3v;
            unsigned long long rhv = 0ull, rlv = 0ull;
            constexpr int U4 = NA / 4, DC = wsa_dc(NA);         // units per epilogue group; its arithmetic runs DC units behind its reads
#pragma unroll
            for (int p = 0; p < CL; ++p) {
                __builtin_amdgcn_sched_barrier(0);
                if (p == 0) wait_lgkm(0);                      // this wave's trunk writes of the last iteration: B reads them behind this barrier
                if (!PAIR_PROBE(a, 4)) __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                const int nslot = slot + 1 == NS ? 0 : slot + 1;
                const unsigned fcur = fbase0 + slot * SLAB, fnxt = fbase0 + nslot * SLAB;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int i = p * 8 + j;
                    if (!pair_is_a(i, NA, FT)) continue;
                    const int ua = pair_a_before(i, NA, FT);   // unit = conv3 K chunk
                    // ---- requests: a group's constants + shortcut when its turn starts (the fragment WS_PF units ahead: at the unit's end)
                    const int gl = (ua % U4 == 0) ? ua / U4 : -1;
                    if (gl >= 0 && !PAIR_PROBE(a, 8)) {
#define WS_GL(k) if (gl == k) { \
                            cs3v = lds_rd128f<k * 32>(cb); cb3v = lds_rd128f<k * 32 + DEPTH * 4>(cb); \
                            rhv = lds_rd64<0>(soff[k][0] + toff); rlv = lds_rd64<0>(soff[k][1] + toff); }
                        WS_GL(0) WS_GL(1) WS_GL(2) WS_GL(3)
#undef WS_GL
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if (PAIR_PROBE(a, 32) || PAIR_PROBE(a, 8)) wait_lgkm(0); else if (!PAIR_PROBE(a, 128)) ws_wait_unit<NA, true>(ua);
                    __builtin_amdgcn_sched_barrier(0);
                    // (past the last chunk the stream holds zero fragments: the MFMAs run on them rather than behind a branch per unit)
                    if (!PAIR_PROBE(a, 1) && !PAIR_PROBE(a, 512)) {
                        const wfrag& w = wq[ua & 3];
                        accN = mfma_split(w.hi, xh[ua].lo, accN);
                        accN = mfma_split(w.lo, xh[ua].hi, accN);
                        accN = mfma_split(w.hi, xh[ua].hi, accN);
                    }
                    asm volatile("" : "+a"(accN));
                    // ---- the group whose reads went out DC units ago: conv3's epilogue + the shortcut, clamp, split, in place
                    const int gp = (ua % U4 == DC) ? ua / U4 : -1;
                    if (gp >= 0 && ev && !PAIR_PROBE(a, 8)) {
                        unsigned oh[2], ol[2];
#pragma unroll
                        for (int i2 = 0; i2 < 2; ++i2) {
                            const shalf2 h2v = __builtin_bit_cast(shalf2, (unsigned)(rhv >> (32 * i2)));
                            const shalf2 l2v = __builtin_bit_cast(shalf2, (unsigned)(rlv >> (32 * i2)));
                            float c[2], vraw[2];
#pragma unroll
                            for (int jj = 0; jj < 2; ++jj) {
                                float v = fmaf(accO[4 * gp + 2 * i2 + jj], cs3v[2 * i2 + jj], cb3v[2 * i2 + jj]);
                                v += __builtin_fmaf((float)h2v[jj], one, (float)l2v[jj]);
                                c[jj] = split_clamp(v);
                                vraw[jj] = v;
                            }
                            satm = sat_acc(satm, vraw[0], vraw[1]);
                            shalf2 ph = {(shalf_t)c[0], (shalf_t)c[1]};
                            asm volatile("" : "+v"(ph));
                            const shalf2 pl = {(shalf_t)__builtin_fmaf((float)ph[0], -one, c[0]), (shalf_t)__builtin_fmaf((float)ph[1], -one, c[1])};
                            oh[i2] = __builtin_bit_cast(unsigned, ph);
                            ol[i2] = __builtin_bit_cast(unsigned, pl);
                        }
                        const unsigned long long wh = (unsigned long long)oh[0] | ((unsigned long long)oh[1] << 32);
                        const unsigned long long wl = (unsigned long long)ol[0] | ((unsigned long long)ol[1] << 32);
#define WS_S(kk) if (gp == kk) { lds_wr64<0>(soff[kk][0] + toff, wh); lds_wr64<0>(soff[kk][1] + toff, wl); }
                        WS_S(0) WS_S(1) WS_S(2) WS_S(3)
#undef WS_S
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    {
                        // (the fragment WS_PF conv3 fragments after position i: the conv3 fragments sit at every FT / NA-th position)
                        constexpr int STEP = FT / NA;
                        static_assert(NA * STEP == FT && ws_pos(1, NA, FT, true) - ws_pos(0, NA, FT, true) == STEP, "evenly interleaved streams");
                        const int ix = i + WS_PF * STEP;
                        const int pn = ix / 8, jn = ix % 8;     // slab (relative to this iteration's first) and fragment in it
                        if (!PAIR_PROBE(a, 32)) wq[(ua + WS_PF) & 3] = lds_frag(pn == p ? fcur : fnxt, jn);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                slot = nslot;
            }
            // the next iteration's first fragments are in flight into registers the compiler knows nothing asynchronous about: let them
            // land before anything (a loop-carried copy) may touch them
            __builtin_amdgcn_sched_barrier(0);
            wait_lgkm(0);
            __builtin_amdgcn_sched_barrier(0);
        };
        for (int it = 0; it <= NCH + 1; it += 2) {
            iterA(it, acc1, accB);
            iterA(it + 1, accB, acc1);
        }
        WS_STAMP(3);
        split_flag_max(satm);
        WS_STAMP(4);
        WS_RSTAMP(7);
        WS_ACC_STORE(5, t_wait);
        return;
    }

    // =================================================================================================================== wave B
    const char* gstream = a.stream + pr * 4096 + lane * 16;
    auto dma_slab = [&](int slab, int slot_) {
        const char* src = gstream + (long long)slab * SLAB;
        char* dst = smem + slot_ * SLAB + pr * 4096;
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)dst, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)dst, 16, 1024, 0);
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)dst, 16, 2048, 0);
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)dst, 16, 3072, 0);
    };
    const int rsub = lane >> 3, pslot = lane & 7;
    // row pieces of a staging tile <-> global rows: piece q = rows 8 q + rsub, this lane's 16 bytes = logical slot ls of the row
    const bsplit_t* rrow[4];                                   // where the next shortcut chunk is read (running)
    bsplit_t* orow[4];                                         // where the next trunk chunk goes
    int ostep[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int r = 8 * q + rsub, m = mbase + r;
        const int ls = pslot ^ ((r >> 1) & 7);
        const bool ok = m < a.M;
        orow[q] = ok ? a.out + (long long)m * DEPTH + ls * 4 : (bsplit_t*)g_pair_dump + lane * 4;
        ostep[q] = ok ? 32 : 0;
        rrow[q] = a.res + (long long)(ok ? m : 0) * a.ldr + ls * 4;
    }
    auto dma_res = [&](char* tile, int ahead) {                // the shortcut chunk `ahead` chunks past rrow -> tile
#pragma unroll
        for (int q = 0; q < 4; ++q)
            __builtin_amdgcn_global_load_lds((gptr_t)(rrow[q] + ahead * 32), (lptr_t)(tile + q * 1024), 16, 0, 0);
    };
#pragma unroll
    for (int s_ = 0; s_ < NS - 1; ++s_) dma_slab(s_, s_);
    // shortcut chunks 0, 1, 2 -> tiles 0, 1, 2.  Iteration `it` then requests chunk it + 1 into tile (it + 1) % 3 -- iterations 0 and 1
    // repeat chunks 1 and 2 (the same bytes into the same tile: every iteration issues the same requests, the waits are counted)
#pragma unroll
    for (int c = 0; c < 3; ++c) dma_res(tilep + c * 4096, c);
#pragma unroll
    for (int q = 0; q < 4; ++q) rrow[q] += 32;                 // -> chunk 1: what iteration 0 requests
    f32x16 acc2[NF2];
#pragma unroll
    for (int j = 0; j < NF2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[j][r] = 0.f;
    xfrag th[2];                                               // conv1's B operand: K chunks 0, 1 of the step
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int e8 = 0; e8 < 8; ++e8) { th[k].hi[e8] = (shalf_t)0.f; th[k].lo[e8] = (shalf_t)0.f; }
    __builtin_amdgcn_sched_barrier(0);
    WS_STAMP(1);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    WS_STAMP(2);

#ifdef WS_PRIO_B
    __builtin_amdgcn_s_setprio(WS_PRIO_B);
#endif
    int slot = 0, dslab = NS - 1;
    // conv1' operand slots of this lane in tile 0: K chunk kcl = group 2 kcl + lh of the row, its hi and its lo slot
    unsigned toffB[2][2];
#pragma unroll
    for (int kcl = 0; kcl < 2; ++kcl)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) toffB[kcl][pl] = tile0 + lr * 128 + (((2 * (2 * kcl + lh) + pl) ^ sw) << 4);
    const unsigned pbase0 = lds0 + OFF_C + 2 * DEPTH * 4 + lh * 32;      // + chunk * 128 + kcl * 64: pre_scale of this lane's 8 channels; pre_shift at + DEPTH * 4
    wfrag wq[4];
#pragma unroll
    for (int k = 0; k < WS_PF; ++k) {
        static_assert(ws_pos(WS_PF - 1, NA, FT, false) < 8, "the first fragments sit in the first slab");
        wq[k] = lds_frag(fbase0 + slot * SLAB, ws_pos(0, NA, FT, false) + k * (FT / NB));
    }
    // the pre-activation of 8 stored trunk values (hi / lo slots rh / rl, constants ps / pb) -> 4 dwords of hi halves, 4 of lo halves
    // (pairs k0 .. k0 + 1 of the four): the consumer-side arithmetic of common.h preact_slot_split, value for value
    auto preact_pairs = [&](const u32x4& rh_, const u32x4& rl_, const f32x4 (&ps_)[2], const f32x4 (&pb_)[2], int k0, unsigned (&fh)[4], unsigned (&fl)[4]) {
#pragma unroll
        for (int k = k0; k < k0 + 2; ++k) {
            const float s0 = split_sum_lo(rh_[k], rl_[k]), s1 = split_sum_hi(rh_[k], rl_[k]);
            const float y0 = split_relu(fmaf(s0, ps_[k >> 1][2 * (k & 1)], pb_[k >> 1][2 * (k & 1)]));
            const float y1 = split_relu(fmaf(s1, ps_[k >> 1][2 * (k & 1) + 1], pb_[k >> 1][2 * (k & 1) + 1]));
            split2_mix(y0, y1, fh[k], fl[k]);
        }
    };
    constexpr int H = NB / 2;                                  // units per K chunk of conv1'
    static_assert(NB / CL == 4 && H >= 4, "four conv1' fragments per slab; unit 4 opens step 1");

#pragma unroll 1
    for (int it = 0; it <= NCH + 1; ++it) {
        const int e2 = it - 2;                                 // the trunk chunk consumed here = conv1's K step
        const bool kv = e2 >= 0 && e2 < NCH;
        const int e2c = e2 < 0 ? 0 : (e2 >= NCH ? NCH - 1 : e2);
        const int e1c = it - 1 < 0 ? 0 : (it - 1 >= NCH ? NCH - 1 : it - 1);
        const unsigned toff = (unsigned)(ws_mod(it + 1, 3) * 4096);     // tile of chunk it - 2 == tile of chunk it + 1
        const unsigned toff1 = (unsigned)(ws_mod(it + 2, 3) * 4096);    // tile of chunk it - 1: wave A is writing it during this iteration
        u32x4 xr[4], rawh, rawl;
        f32x4 psv[2], pbv[2];
        unsigned fh[4], fl[4];
#pragma unroll
        for (int p = 0; p < CL; ++p) {
            __builtin_amdgcn_sched_barrier(0);
            if (!PAIR_PROBE(a, 2)) ws_wait_pos<CL, NS>(p);
            if (!PAIR_PROBE(a, 4)) __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            const int prev = slot == 0 ? NS - 1 : slot - 1;
            if (!PAIR_PROBE(a, 2)) dma_slab(dslab, prev);
            dslab = dslab + 1 == TOTAL ? 0 : dslab + 1;
            const int nslot = slot + 1 == NS ? 0 : slot + 1;
            const unsigned fcur = fbase0 + slot * SLAB, fnxt = fbase0 + nslot * SLAB;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int i = p * 8 + j;
                if (pair_is_a(i, NA, FT)) continue;
                const int ub = i - pair_a_before(i, NA, FT);   // unit = conv1' fragment kb: K chunk kb / NF2 of the step, row block kb % NF2
                // ---- this unit's extra reads
                if (!PAIR_PROBE(a, 16)) {
                    if (ub == 0) {                             // K chunk 1 of the finished tile (chunk it - 2): operand slots + constants
                        const unsigned pb_ = pbase0 + e2c * 128;
                        rawh = lds_rd128u<0>(toffB[1][0] + toff); rawl = lds_rd128u<0>(toffB[1][1] + toff);
                        psv[0] = lds_rd128f<64>(pb_); psv[1] = lds_rd128f<64 + 16>(pb_);
                        pbv[0] = lds_rd128f<DEPTH * 4 + 64>(pb_); pbv[1] = lds_rd128f<DEPTH * 4 + 64 + 16>(pb_);
                    }
                    if (ub == 1) {                             // its rows
                        xr[0] = lds_rd128u<0>(tile0 + toff + lane16); xr[1] = lds_rd128u<1024>(tile0 + toff + lane16);
                        xr[2] = lds_rd128u<2048>(tile0 + toff + lane16); xr[3] = lds_rd128u<3072>(tile0 + toff + lane16);
                    }
                    if (ub == H) {                             // K chunk 0 of the tile wave A is half way through (chunk it - 1: its groups 0, 1 are written)
                        const unsigned pb_ = pbase0 + e1c * 128;
                        rawh = lds_rd128u<0>(toffB[0][0] + toff1); rawl = lds_rd128u<0>(toffB[0][1] + toff1);
                        psv[0] = lds_rd128f<0>(pb_); psv[1] = lds_rd128f<16>(pb_);
                        pbv[0] = lds_rd128f<DEPTH * 4>(pb_); pbv[1] = lds_rd128f<DEPTH * 4 + 16>(pb_);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                if (PAIR_PROBE(a, 32) || PAIR_PROBE(a, 16)) wait_lgkm(0); else if (!PAIR_PROBE(a, 256)) ws_wait_unit<NB, false>(ub);
                __builtin_amdgcn_sched_barrier(0);
                if (!PAIR_PROBE(a, 1) && !PAIR_PROBE(a, 1024)) {       // (iterations 0, 1: zero fragments in the stream)
                    const wfrag& w = wq[ub & 3];
                    const xfrag& x = th[ub / NF2];
                    acc2[ub % NF2] = mfma_split(w.hi, x.lo, acc2[ub % NF2]);
                    acc2[ub % NF2] = mfma_split(w.lo, x.hi, acc2[ub % NF2]);
                    acc2[ub % NF2] = mfma_split(w.hi, x.hi, acc2[ub % NF2]);
                }
                asm volatile("" : "+a"(acc2[ub % NF2]));
                // ---- this unit's share of the per-iteration work
                if (!PAIR_PROBE(a, 16)) {
                    if ((ub == 2 || ub == H + 2) && !PAIR_PROBE(a, 64)) preact_pairs(rawh, rawl, psv, pbv, 0, fh, fl);
                    if ((ub == 3 || ub == H + 3) && !PAIR_PROBE(a, 64)) {
                        preact_pairs(rawh, rawl, psv, pbv, 2, fh, fl);
                        // K chunk 1: this iteration's second half reads it; K chunk 0: the NEXT iteration's first half (this iteration's is done)
                        xfrag& t = th[ub == 3 ? 1 : 0];
                        t.hi = __builtin_bit_cast(shalf8, u32x4{fh[0], fh[1], fh[2], fh[3]});
                        t.lo = __builtin_bit_cast(shalf8, u32x4{fl[0], fl[1], fl[2], fl[3]});
                        asm volatile("" : "+v"(t.hi), "+v"(t.lo));
                    }
                    if (ub == 4) {                             // the rows out, the tile re-armed with the shortcut chunk three ahead
                        if (kv) {
#pragma unroll
                            for (int q = 0; q < 4; ++q) { *(u32x4*)orow[q] = xr[q]; orow[q] += ostep[q]; }
                        } else {
#pragma unroll
                            for (int q = 0; q < 4; ++q) *(u32x4*)((bsplit_t*)g_pair_dump + lane * 4) = xr[q];
                        }
                        dma_res(tilep + toff, 0);               // shortcut chunk it + 1 (past the end: the last chunk again)
                        if (it + 2 < NCH) {
#pragma unroll
                            for (int q = 0; q < 4; ++q) rrow[q] += 32;
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                {
                    constexpr int STEP = FT / NB;
                    static_assert(NB * STEP == FT && ws_pos(1, NA, FT, false) - ws_pos(0, NA, FT, false) == STEP, "evenly interleaved streams");
                    const int ix = i + WS_PF * STEP;
                    const int pn = ix / 8, jn = ix % 8;
                    if (!PAIR_PROBE(a, 32)) wq[(ub + WS_PF) & 3] = lds_frag(pn == p ? fcur : fnxt, jn);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            slot = nslot;
        }
        __builtin_amdgcn_sched_barrier(0);
        wait_lgkm(0);                                           // (the prefetched fragments: see wave A)
        __builtin_amdgcn_sched_barrier(0);
    }

    // ---- conv1' epilogue: BN (+ ReLU), split, through this pair's staging tiles 0 / 1, coalesced stores
#pragma unroll
    for (int j = 0; j < NF2; ++j) asm volatile("" : "+a"(acc2[j]));
    __builtin_amdgcn_sched_barrier(0);
    WS_STAMP(3);
    __builtin_amdgcn_s_waitcnt(0);                             // (the last shortcut requests must not land in the tiles any more)
    float* sS1 = (float*)(smem + OFF_C1);
    float* sB1 = sS1 + N2;
    float satmax = 0.f;
    const float lo1 = a.relu1 ? 0.f : -HMMR_SPLIT_MAX;
    bsplit_t* hrow[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int r = 8 * q + rsub, m = mbase + r;
        const int ls = pslot ^ ((r >> 1) & 7);
        hrow[q] = m < a.M ? a.out_h1 + (long long)m * N2 + ls * 4 : (bsplit_t*)g_pair_dump + lane * 4;
    }
#pragma unroll
    for (int of = 0; of < NF2; ++of) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int n2 = of * 32 + 8 * g + 4 * lh;
            const f32x4 s4 = *(const f32x4*)(sS1 + n2), b4 = *(const f32x4*)(sB1 + n2);
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = fmaf(acc2[of][4 * g + j], s4[j], b4[j]);
            unsigned h2[2], l2[2];
            split4_mix(v, lo1, h2, l2, satmax);
            *(unsigned long long*)(tilep + (of & 1) * 4096 + lr * 128 + (((2 * g) ^ sw) << 4) + 8 * lh) = (unsigned long long)h2[0] | ((unsigned long long)h2[1] << 32);
            *(unsigned long long*)(tilep + (of & 1) * 4096 + lr * 128 + (((2 * g + 1) ^ sw) << 4) + 8 * lh) = (unsigned long long)l2[0] | ((unsigned long long)l2[1] << 32);
        }
        u32x4 xr[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) xr[q] = *(const u32x4*)(tilep + (of & 1) * 4096 + q * 1024 + lane16);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int m = mbase + 8 * q + rsub;
            *(u32x4*)(hrow[q] + (m < a.M ? of * 32 : 0)) = xr[q];
        }
    }
    split_flag_max(satmax);
    WS_STAMP(4);
    WS_RSTAMP(7);
    WS_ACC_STORE(5, t_wait);
}

template <int KC3, int DEPTH, int N2>
int launch_pair_ws(const PairArgs& a, hipStream_t stream) {
    constexpr int lds = ws_ring<DEPTH>::NS * PAIR_SLAB + 4 * 3 * 4096 + 4 * DEPTH * (int)sizeof(float) + 2 * N2 * (int)sizeof(float);
    static_assert(lds <= 160 * 1024, "LDS");
    auto kern = unit_pair_ws_kernel<KC3, DEPTH, N2>;
    static DeviceOnce once;
    if (const unsigned long long bit = once.due()) {
        HMMR_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        once.mark(bit);
    }
    PairArgs b = a;
    b.tpw = 1;
    hipLaunchKernelGGL(kern, dim3((unsigned)((a.M + 127) / 128)), dim3(512), lds, stream, b);
    HMMR_CHECK_HIP(hipGetLastError());
    return 0;
}

template <int KC3A, int KC3B, int DEPTH, int N2, bool RES>
int launch_pair(const PairArgs& a, hipStream_t stream) {
    constexpr int lds = PAIR_NS * PAIR_SLAB + 4 * 2 * 4096 + 4 * DEPTH * (int)sizeof(float) + (DEPTH <= 512 ? 2 * N2 * (int)sizeof(float) : 0);
    static_assert(lds <= 160 * 1024, "LDS");
    auto kern = unit_pair_kernel<KC3A, KC3B, DEPTH, N2, RES>;
    static DeviceOnce once;
    if (const unsigned long long bit = once.due()) {
        HMMR_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        once.mark(bit);
    }
    // persistent workgroups (block-2 shapes) once the launch is at least two rounds of one workgroup per CU: below that a second tile only
    // lengthens the launch (hmmr_debug_t.pair_two_tile_min moves the switch: tests run both forms on one batch)
    PairArgs b = a;
    const int n_tiles = (a.M + 127) / 128;
    const int two_min = hmmr_debug_state()->pair_two_tile_min > 0 ? hmmr_debug_state()->pair_two_tile_min : 512;
    int grid = n_tiles;
    b.tpw = 1;
    if (DEPTH <= 512 && n_tiles >= two_min) {
        const int cus = hmmr_cu_count(stream);                   // (cached per device; the STREAM's device, not the current one)
        grid = n_tiles < cus ? (n_tiles + 1) / 2 : cus;          // (forced on for a short launch: two tiles per workgroup)
        b.tpw = (n_tiles + grid - 1) / grid;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), lds, stream, b);
    HMMR_CHECK_HIP(hipGetLastError());
    return 0;
}

}  // namespace

// bytes of the filter stream of a unit pair (packing.pack_pair_stream): (depth / 32 + 2) iterations of (kc3 + n2 / 16) fragments of 2 KB
extern "C" size_t hmmr_pair_stream_bytes(int kc3, int depth, int n2) {
    return (size_t)(depth / 32 + 2) * (size_t)(kc3 + n2 / 16) * 2048;
}

// hmmr_bottleneck_tail with pair_stream set (called from bottleneck_split.hip)
int hmmr_unit_pair_split(const hmmr_tail_desc_t* d, hipStream_t stream) {
    HMMR_REQUIRE(d->h2 && !d->h1 && d->pair_stream && d->out && d->out_h1 && d->pre_scale && d->pre_shift && d->scale1 && d->shift1 &&
                 !d->out_pre && !d->res_strided && d->m > 0,
                 "hmmr_bottleneck_tail (f16x3, pair_stream): needs h2, the filter stream, out, out_h1, the next unit's preact and conv1 "
                 "constants and a dense shortcut");
    const bool folded = d->xp != nullptr;
    HMMR_REQUIRE(folded != (d->res != nullptr), "hmmr_bottleneck_tail (f16x3, pair_stream): either a shortcut tensor (res) or a folded one (xp)");
    HMMR_REQUIRE(folded || d->ldr >= d->depth, "hmmr_bottleneck_tail: residual row stride < depth");
    PairArgs a = {};
    a.src[0] = (const bsplit_t*)d->h2; a.src_ld[0] = d->c_mid;
    a.src[1] = (const bsplit_t*)d->xp; a.src_ld[1] = d->c_xp;
    a.stream = (const char*)d->pair_stream;
    a.scale3 = d->scale3; a.shift3 = d->shift3; a.pre_scale = d->pre_scale; a.pre_shift = d->pre_shift;
    a.res = (const bsplit_t*)d->res; a.ldr = d->ldr; a.out = (bsplit_t*)d->out;
    a.scale1 = d->scale1; a.shift1 = d->shift1; a.relu1 = d->relu1; a.out_h1 = (bsplit_t*)d->out_h1; a.M = d->m; a.one = 1.0f;
    a.probe = 0;
#ifdef HMMR_GEMM_PROBE
    a.probe = hmmr_debug_state()->gemm_probe;
    a.ts = (unsigned long long*)(((unsigned long long)(unsigned)hmmr_debug_state()->reserved[1] << 32) | (unsigned)hmmr_debug_state()->reserved[0]);
#endif
    // round 6: the wave-specialised form (two waves per SIMD) for the shapes with a shortcut tensor is built, bit-identical and measured
    // EQUAL to the one-wave-per-SIMD form (profiles/r06_pair_ws_*: 0.252 against 0.260 ms in block 3, 0.334 against 0.333 in block 2,
    // standalone at 257 frames), so the round-4 form stays the default and hmmr_debug_t.pair_form = 2 selects this one (tests run both)
    const bool ws = hmmr_debug_state()->pair_form == 2;
    if (d->c_mid == 256 && d->depth == 1024 && d->n2 == 256 && !folded)
        return ws ? launch_pair_ws<16, 1024, 256>(a, stream) : launch_pair<16, 0, 1024, 256, true>(a, stream);
    if (d->c_mid == 128 && d->depth == 512 && d->n2 == 128 && !folded)
        return ws ? launch_pair_ws<8, 512, 128>(a, stream) : launch_pair<8, 0, 512, 128, true>(a, stream);
    if (d->c_mid == 128 && d->depth == 512 && d->n2 == 128 && folded && d->c_xp == 256) return launch_pair<8, 16, 512, 128, false>(a, stream);
    hmmr_set_error("hmmr_bottleneck_tail (f16x3, pair_stream): supported shapes are 256 -> 1024 -> 256 and 128 -> 512 -> 128 "
                   "(the latter also with a folded 256-channel shortcut); got %d, %d, %d%s", d->c_mid, d->depth, d->n2, folded ? " folded" : "");
    return -1;
}
