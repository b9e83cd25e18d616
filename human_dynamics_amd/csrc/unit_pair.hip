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
    if (d->c_mid == 256 && d->depth == 1024 && d->n2 == 256 && !folded) return launch_pair<16, 0, 1024, 256, true>(a, stream);
    if (d->c_mid == 128 && d->depth == 512 && d->n2 == 128 && !folded) return launch_pair<8, 0, 512, 128, true>(a, stream);
    if (d->c_mid == 128 && d->depth == 512 && d->n2 == 128 && folded && d->c_xp == 256) return launch_pair<8, 16, 512, 128, false>(a, stream);
    hmmr_set_error("hmmr_bottleneck_tail (f16x3, pair_stream): supported shapes are 256 -> 1024 -> 256 and 128 -> 512 -> 128 "
                   "(the latter also with a folded 256-channel shortcut); got %d, %d, %d%s", d->c_mid, d->depth, d->n2, folded ? " folded" : "");
    return -1;
}
