// 1x1 convolutions of the ResNet's late blocks for gfx950, f16x3 ("split") operands, as ONE MFMA STREAM PER SIMD with BOTH operands
// in rings (hmmr_conv_desc_t.k_order = 2 with kh = kw = 1; tiles 22 .. 26).  slim resnet_v2.bottleneck `conv1` / `shortcut` as invoked
// at src/models.py:65-75.
//
// What the 8-wave tiles of gemm_conv.hip measure on these layers (block 4: 12.6 k pixels, K = 1024 / 2048; block3/unit_1's shortcut +
// conv1): 215-240 TFLOP/s with the matrix pipes a quarter busy.  tools/probes/lds_fill_rate.hip (profiles/r05q) shows why: L2 DELIVERS
// 33 TB/s into LDS, but a two-stage ring with two workgroups per CU has one K step of each workgroup in flight, and a step takes an L2 /
// fabric round trip (1.86 us per 32-channel step in block4/unit_2's conv1, 0.4 us of it matrix time).  This kernel keeps D - 1 K steps
// of BOTH operands in flight, the way csrc/conv3x3_stream.hip does for its filters:
//   * ONE wave per SIMD (4-wave workgroups, one per CU; tile 26: two), FM x FN accumulators of 32 x 32 in the AGPR half, two fragment sets in the VGPR
//     half; a K step is 16 channels = 3 FM FN MFMAs and one barrier, the next step's fragment reads between the MFMAs;
//   * filters: the stream of MFMA A-operand fragments of packing.pack_conv1x1_stream ([128-channel tile][K step][4 row blocks][hi plane |
//     lo plane] of 1 KB) through a ring of D slabs of 8 KB, D steps ahead;
//   * pixels: K step kt of every pixel of the tile is 64 bytes of its row ([hi k0-7 | lo k0-7 | hi k8-15 | lo k8-15]); four lanes per
//     row DMA them (one coalesced 64-byte chunk per pixel) into a ring of D slabs of BM rows, the 16-byte slots XOR-swizzled with
//     (row >> 2) & 3 on the global side so that the 16 lanes of a ds_read_b128 group hit 16 distinct slots; fragment addresses are
//     per-lane constants (no taps, no borders), ring slot and row block are an add and an instruction offset;
//   * waits are COUNTED (s_waitcnt vmcnt(N)): a step issues P = 2 + BM / 64 requests per wave for stage kt + D and ends when stage
//     kt + 2 has landed, (D - 2) P requests younger than it still in flight.
// Epilogue 0 is that of the 3x3 stream kernel (folded BN, ReLU, split, 16-byte row stores through wave-private staging tiles), with
// hmmr_conv_desc_t's column split: N tiles from n_split on go to out_b (the conv shortcut and conv1 of a block's first unit as one launch).
// Epilogue 1 is the conv3 form (res / out2 / in2 of the descriptor): K may continue in a second tensor, the shortcut's 32 x 32 blocks are
// DMA'd ahead into wave-private tiles and added before the split, the next unit's pre-activation of the stored value is a second output.
// Tile 26 runs TWO workgroups per CU (rings 3 deep, 256 registers per wave): one workgroup's prologue, epilogue and round trips under the
// other's loop -- for short K loops and the conv3 form's HBM-heavy epilogue.  DESIGN.md section 4.1.3 has the measurements.
// Products and their order per output element: (w.hi x.lo, w.lo x.hi, w.hi x.hi) per 16-channel K step, steps in channel order -- the
// same for every tile of this kernel (they differ from gemm_conv.hip's 32-channel steps by fp32 rounding of the accumulation only).
#include <type_traits>
#include <utility>

#include "common.h"
#include "hmmr_hip.h"

namespace {

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ u32x4 g_s1_dump[64 + 32];       // where the stores of rows beyond M go (a lane's 16 bytes, up to 4 row blocks of 128 B further)

struct S1Args {
    const char* in;                         // [M][C] split rows
    const char* wstream;                    // packing.pack_conv1x1_stream
    const float* scale; const float* shift; // [cout]
    void* out; int ldo, relu;               // N tiles [0, split_tiles)
    void* out_b; int ldo_b, relu_b;         // N tiles [split_tiles, tiles_n): channel n - 128 split_tiles
    int split_tiles;
    // the conv3 form (EPI = 1): a second operand source along K (K steps nk1 .. nk - 1 are in2's rows of C2 channels), a shortcut added
    // before the split (rows of ldo elements, like out), and the next unit's pre-activation of the stored value as a second output
    const char* in2; int C2, nk1;
    const char* res;
    void* out2; const float* scale2; const float* shift2;
    int M, C, nk;                           // nk = (C + C2) / 16 K steps
    int tiles_n, n_tiles;
    long long nt_stride;                    // bytes of one 128-channel tile of the stream: nk x 8 KB
};

// Development build only (tools/s1_probe_build.sh): drop the MFMAs (1), the fragment reads (2), every DMA request and the waits on them (4), the
// pixel requests alone (8; the waits then cover the filter ring only by accident), the loop's barrier (16) at COMPILE time; 32: the pixel
// requests read coalesced (wrong) bytes.  Results are garbage in those modes; the product build compiles the switches away.
#ifndef S1_PROBE_BITS
#define S1_PROBE_BITS 0
#endif
#define S1_PROBE(bit) (((S1_PROBE_BITS) & (bit)) != 0)

// s_waitcnt vmcnt(N) lgkmcnt(0) (gfx9 encoding: vmcnt[3:0] | expcnt[6:4] | lgkmcnt[11:8] | vmcnt_hi[15:14])
template <int N> __device__ __forceinline__ void s1_wait() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
    __builtin_amdgcn_s_waitcnt((N & 15) | (7 << 4) | (0 << 8) | ((N >> 4) << 14));
}

// the loop's LDS reads as inline assembly: the compiler does not track them; a step ends with one lgkmcnt(0) of its own
template <int OFF> __device__ __forceinline__ shalf8 s1_rd(unsigned addr) {
    shalf8 v; asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF)); return v;
}

struct xfrag { shalf8 hi, lo; };            // MFMA B-operand fragment: 32 pixels x 16 channels

template <int V> using s1_ic = std::integral_constant<int, V>;

// a group of D steps; steps past the last one (cin / 16 need not be a multiple of D) are skipped by a scalar branch
template <typename F, int... S> __device__ __forceinline__ void s1_group(F&& f, int kt0, int nk, std::integer_sequence<int, S...>) {
    ((kt0 + S < nk ? f(s1_ic<S>{}, kt0 + S) : (void)0), ...);
}

template <typename F, int... Bs> __device__ __forceinline__ void s1_blocks(F&& f, std::integer_sequence<int, Bs...>) { (f(s1_ic<Bs>{}), ...); }

// FM x FN accumulators per wave, WGM x WGN waves: the tile is 32 WGM FM pixels x 128 channels; D: ring depth of both operands
// EPI 0: scale / shift / relu, column split.  EPI 1: conv3 of a bottleneck unit -- IN2: the folded shortcut's operand, RES: the shortcut tensor,
// OUT2: the next unit's pre-activation
// OCC 2: two workgroups per CU (256 registers per wave, rings 3 deep): one workgroup's prologue, epilogue and request latency under the other's loop
template <int FM, int FN, int WGM, int WGN, int D, int EPI = 0, bool IN2 = false, bool RES = false, bool OUT2 = false, int OCC = 1>
__global__ __launch_bounds__(256, OCC) void conv1x1_stream_kernel(const S1Args a) {
    constexpr int NB = WGN * FN;                                // row blocks of 32 output channels per tile
    static_assert(WGM * WGN == 4 && NB == 4 && FM * FN <= 16 && D >= 3 && D <= 6, "4 waves, 128 channels, at most 16 accumulators");
    constexpr int U = D % 2 ? 2 * D : D;                        // steps per unrolled group: ring slot S % D and fragment set S & 1 are compile-time
    constexpr int R = WGM * FM, BM = 32 * R;
    constexpr int NPX = (BM + 63) / 64;                         // 64-row pieces of a pixel slab (16 rows per wave and piece)
    constexpr int XSLAB = NPX * 64 * 64;                        // rows of 64 bytes
    constexpr int SLAB = NB * 2048, RING = D * SLAB;
    constexpr int RW = NB / 2;                                  // 1 KB pieces of a filter slab each wave moves
    constexpr int P = RW + NPX;                                 // requests per wave and stage
    constexpr int CST = RING + D * XSLAB;                       // this tile's folded BN constants (EPI 1: 2 KB, scale2 / shift2 behind them)
    static_assert(EPI == 0 || (FN == 2 && !(IN2 && RES)), "the conv3 form is built for the 2 x 2 wave arrangements");
    constexpr int NR = 2 * (FM + FN), NG = 3 * FM * FN;         // fragment reads and MFMAs (= gaps) of a step
    static_assert((RING + D * XSLAB + 2048) * OCC <= 160 * 1024, "LDS");
    static_assert(D * P < 64, "vmcnt");
    static_assert(2048 * (FM - 1) + 16 < 65536 && (D - 1) * SLAB + 3 * 2048 + 1024 < 65536, "instruction offsets");
    static_assert(NR <= NG - 4 && NPX + 2 <= NG, "one fragment read and one request per gap");

    extern __shared__ __attribute__((aligned(256))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 31, lh = lane >> 5;
    const int L = xcd_remap(blockIdx.x, a.n_tiles);
    const int mt = L / a.tiles_n, nt = L - mt * a.tiles_n;
    const int m0 = mt * BM;
    const int nk = a.nk;
    const int wm = wave / WGN, wn = wave - wm * WGN;
    const unsigned lds0 = (unsigned)(unsigned long long)(lptr_t)smem;

    // ---- this tile's folded BN constants (128 scales, 128 shifts): loaded before any DMA request is in flight, stored behind the rings
    const float cv = tid < 128 ? a.scale[nt * 128 + tid] : a.shift[nt * 128 + tid - 128];
    float cv2 = 0.f;
    if constexpr (OUT2) cv2 = tid < 128 ? a.scale2[nt * 128 + tid] : a.shift2[nt * 128 + tid - 128];

    // ---- pixel DMA: piece q of this wave = rows 64 q + 16 wave .. + 15 of the slab, four lanes per row; rows beyond M read the last pixel
    unsigned poff[NPX], poff2[IN2 ? NPX : 1];
#pragma unroll
    for (int q = 0; q < NPX; ++q) {
        const int r = 64 * q + 16 * wave + (lane >> 2);
        int px = m0 + r;
        px = (px >= a.M ? a.M - 1 : px) - m0;                   // relative to the tile's first pixel: a 32-bit offset whatever the tensor's size
        poff[q] = (unsigned)px * (unsigned)(a.C * 4) + ((((unsigned)lane & 3u) ^ (((unsigned)r >> 2) & 3u)) << 4);
        if (S1_PROBE(32)) poff[q] = (unsigned)(px & ~15) * (unsigned)(a.C * 4) + lane * 16;       // (probe: a coalesced KB of the wrong bytes)
        if constexpr (IN2) poff2[q] = (unsigned)px * (unsigned)(a.C2 * 4) + ((((unsigned)lane & 3u) ^ (((unsigned)r >> 2) & 3u)) << 4);
    }
    const unsigned long long in_t = (unsigned long long)a.in + (unsigned long long)m0 * (unsigned long long)(a.C * 4);
    const unsigned long long in2_t = IN2 ? (unsigned long long)a.in2 + (unsigned long long)m0 * (unsigned long long)(a.C2 * 4) : 0ull;
    auto x_base = [&](int s) {                                  // K step s of the tile's first pixel: a uniform address the compiler keeps in scalar registers
        unsigned long long ub = in_t + (unsigned long long)(s * 64);      // (so that a piece is base + 32-bit lane offset)
        if constexpr (IN2) { if (s >= a.nk1) ub = in2_t + (unsigned long long)((s - a.nk1) * 64); }
        asm volatile("" : "+s"(ub));
        return (const char*)ub;
    };
    auto x_piece = [&](int q, const char* xb, int slot, bool first) {      // q, slot are constants after unrolling; first: the step reads `in`
        char* dst = smem + RING + slot * XSLAB + q * 4096 + wave * 1024;
        unsigned off = poff[q];
        if constexpr (IN2) off = first ? poff[q] : poff2[q];
        __builtin_amdgcn_global_load_lds((gptr_t)(xb + off), (lptr_t)dst, 16, 0, 0);
    };
    // ---- filter stream: K step s -> ring slot s % D; each wave moves a quarter (one row block, hi and lo plane)
    const char* gw = a.wstream + (long long)nt * a.nt_stride + wave * (RW * 1024) + lane * 16;
    auto ring_dma = [&](int s, int slot) {
        const char* src = gw + (long long)s * SLAB;
        char* dst = smem + slot * SLAB + wave * (RW * 1024);
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)dst, 16, 0, 0);          // (the instruction offset moves both addresses)
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)dst, 16, 1024, 0);
    };
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s_ = 0; s_ < D; ++s_) {
        ring_dma(s_, s_);
        const char* xb = x_base(s_);
#pragma unroll
        for (int q = 0; q < NPX; ++q) x_piece(q, xb, s_, !IN2 || s_ < a.nk1);
    }
    __builtin_amdgcn_sched_barrier(0);
    ((float*)(smem + CST))[tid] = cv;
    if constexpr (OUT2) ((float*)(smem + CST))[256 + tid] = cv2;

    f32x16 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
            asm volatile("" : "+a"(acc[i][j]));
        }

    const unsigned vW = lds0 + wn * FN * 2048 + lane * 16;      // filter fragments: + slot * SLAB + j * 2048 (+ 1024: lo plane)
    // pixel fragments: row rb0 of a slab, slots 2 lh (hi) and 2 lh + 1 (lo) before the swizzle; row block i 2 KB further
    const unsigned rb0 = (unsigned)(wm * FM * 32 + lr);
    const unsigned ax0 = lds0 + RING + rb0 * 64 + (((2u * lh) ^ ((rb0 >> 2) & 3u)) << 4);
    xfrag fx[2][FM];
    wfrag fw[2][FN];
    unsigned adcur = 0, adcurl = 0;

    // read number r of the NR fragment halves of a step (filters first) into set `set`; slot_ = that step's ring slot
    auto read_one = [&](int set, int r, int slot_) {            // all constants after unrolling
        if (r < 2 * FN) {
            const int j = r >> 1, pl = r & 1;
#define S1_F(SL, J, PL) if (slot_ == SL && j == J && pl == PL) { shalf8 v = s1_rd<SL * SLAB + J * 2048 + PL * 1024>(vW); if (PL) fw[set][J].lo = v; else fw[set][J].hi = v; }
#define S1_FJ(SL, J) S1_F(SL, J, 0) S1_F(SL, J, 1)
#define S1_FS(SL) if constexpr (SL < D) { S1_FJ(SL, 0) if constexpr (FN > 1) { S1_FJ(SL, 1) } if constexpr (FN > 2) { S1_FJ(SL, 2) S1_FJ(SL, 3) } }
            S1_FS(0) S1_FS(1) S1_FS(2) S1_FS(3) S1_FS(4) S1_FS(5)
#undef S1_FS
#undef S1_FJ
#undef S1_F
        } else {
            const int i = (r - 2 * FN) >> 1, pl = (r - 2 * FN) & 1;
#define S1_X(I) if constexpr (I < FM) { if (i == I) { if (pl) fx[set][I].lo = s1_rd<I * 2048>(adcurl); else fx[set][I].hi = s1_rd<I * 2048>(adcur); } }
            S1_X(0) S1_X(1) S1_X(2) S1_X(3) S1_X(4) S1_X(5) S1_X(6) S1_X(7) S1_X(8) S1_X(9) S1_X(10) S1_X(11) S1_X(12) S1_X(13) S1_X(14) S1_X(15)
#undef S1_X
        }
    };
    auto slot_setup = [&](int slot_) {
        adcur = ax0 + slot_ * XSLAB;
        adcurl = adcur ^ 16u;
    };

    // ---- prologue: stage 0, the fragments of step 0, then stage 1
    __builtin_amdgcn_sched_barrier(0);
    s1_wait<(D - 1) * P>();
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    slot_setup(0);
#pragma unroll
    for (int r = 0; r < NR; ++r) read_one(0, r, 0);
    __builtin_amdgcn_sched_barrier(0);
    s1_wait<(D - 2) * P>();
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);

    // One K step, S = its position in a group of D (ring slot S of both rings, fragment set S & 1).  Every MFMA is followed by a gap of at
    // most a few other instructions: gap 0 = the filter ring's DMA (stage kt + D into the slot of stage kt, whose fragments were read
    // during step kt - 1), gap 1 = the next step's pixel fragment address, gaps 2 ... = the pixel ring's pieces of stage kt + D, one per
    // gap, and the next step's fragment reads, spread evenly.  Past the last stage the requests go on (for the last stage again, into a
    // slot nobody reads any more): the counted waits stay the same for every step and the loop has no branches but its own
    auto step = [&](auto s_c, int kt) {
        constexpr int S = decltype(s_c)::value, CUR = S & 1, NXT = CUR ^ 1, SNEXT = (S + 1) % D;
        const int sn = kt + D < nk ? kt + D : nk - 1;
        const char* xsrc = x_base(sn);
        const bool xfirst = !IN2 || sn < a.nk1;
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int j = 0; j < FN; ++j) {
                    const shalf8& wa = p == 1 ? fw[CUR][j].lo : fw[CUR][j].hi;
                    const shalf8& xb = p == 0 ? fx[CUR][i].lo : fx[CUR][i].hi;
                    if (!S1_PROBE(1)) acc[i][j] = mfma_split(wa, xb, acc[i][j]);
                    const int g = (3 * i + p) * FN + j;
                    if (g == 0 && !S1_PROBE(4)) ring_dma(sn, S % D);
                    if (g == 1) slot_setup(SNEXT);
                    if (g >= 2 && g < 2 + NPX && !S1_PROBE(4) && !S1_PROBE(8)) x_piece(g - 2 < NPX ? g - 2 : 0, xsrc, S % D, xfirst);
#pragma unroll
                    for (int r = 0; r < NR; ++r)
                        if (2 + (r * (NG - 4)) / NR == g && !S1_PROBE(2)) read_one(NXT, r, SNEXT);
                    __builtin_amdgcn_sched_barrier(0);
                }
        // stage kt + 2 (read during step kt + 1) has landed; stages kt + 3 .. kt + D stay in flight
        if (S1_PROBE(4)) __builtin_amdgcn_s_waitcnt(63 | (7 << 4) | (0 << 8) | (3 << 14));      // lgkmcnt(0) alone
        else if (S1_PROBE(8)) s1_wait<(D - 2) * RW>();
        else s1_wait<(D - 2) * P>();
        __builtin_amdgcn_sched_barrier(0);
        if (!S1_PROBE(16)) __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    for (int kt0 = 0; kt0 < nk; kt0 += U) s1_group(step, kt0, nk, std::make_integer_sequence<int, U>{});
    // (the requests of the last steps -- repeats of the last stage -- land in the rings the epilogue is about to reuse)
    __builtin_amdgcn_sched_barrier(0);
    s1_wait<0>();
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);


    if constexpr (EPI == 1) {
        // ---- the conv3 epilogue: bias (folded row scale), + the shortcut (its 32 x 32 blocks DMA'd two blocks ahead into wave-private LDS tiles
        // with the staging tiles' swizzle, read back in the D layout), split -> out; the next unit's pre-activation of the STORED value
        // (hi + lo as the consumer would read it: gemm_conv.hip's out2), ReLU, split -> out2.  Per wave: out staging 2 x 4 KB | out2
        // staging 2 x 4 KB | shortcut tiles 5 x 4 KB, requested PD = 4 blocks ahead (one wave per SIMD: a block's arithmetic is 0.5 us, a
        // request's round trip 2 us, and nothing else runs meanwhile).  Vector-memory order per block b: [shortcut b + PD] wait(b) ... [stores of b]
        // (OCC 2: the other workgroup of the CU covers the round trips: one staging tile per output, shortcut blocks one ahead)
        constexpr int B = FM * FN, NST = OUT2 ? 8 : 4, PD = OCC == 2 ? 1 : 4, RT = PD + 1, NSTG = OCC == 2 ? 1 : 2;
        constexpr int WVB = (2 * NSTG + RT) * 4096;
        char* wv = smem + wave * WVB;
        static_assert(4 * WVB <= RING + D * XSLAB, "the epilogue's tiles live in the idle rings");
        const int rsub = lane >> 3, pslot = lane & 7, sw = (lr >> 1) & 7;
        const int nb = nt * 128 + wn * FN * 32;
        const float lo_clamp = a.relu ? 0.f : -HMMR_SPLIT_MAX;
        // byte offset of this lane's 16-byte slot of row 8 q + rsub of row block i (out, out2 and the shortcut share the row stride); rows
        // beyond M: the last row for loads, the dump for stores
        // (relative to the tile's first row: 32 bits whatever the tensors' size)
        const long long row0 = (long long)m0 * a.ldo * 4;
        const char* const res_t = RES ? a.res + row0 : nullptr;
        char* const out_t = (char*)a.out + row0;
        char* const out2_t = OUT2 ? (char*)a.out2 + row0 : nullptr;
        auto row_off = [&](int i, int q, bool& valid) {
            const int r = 8 * q + rsub;
            int m = m0 + (wm * FM + i) * 32 + r;
            valid = m < a.M;
            m = (valid ? m : a.M - 1) - m0;
            return ((unsigned)m * (unsigned)a.ldo + (unsigned)(nb + (pslot ^ ((r >> 1) & 7)) * 4)) * 4u;
        };
        auto res_dma = [&](int b) {                             // b is a constant after unrolling
            const int i = b / FN, j = b % FN;
            char* dst = wv + 2 * NSTG * 4096 + (b % RT) * 4096;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                bool valid;
                const unsigned off = row_off(i, q, valid);
                __builtin_amdgcn_global_load_lds((gptr_t)(res_t + off + j * 128), (lptr_t)(dst + q * 1024), 16, 0, 0);
            }
        };
        // the constants of this wave's channels: in registers (OCC 1), or read per use where registers are short (OCC 2)
        constexpr bool CREG = OCC == 1;
        f32x4 s4[CREG ? FN : 1][4], b4[CREG ? FN : 1][4], s2[CREG && OUT2 ? FN : 1][4], b2[CREG && OUT2 ? FN : 1][4];
        const float* const cst = (const float*)(smem + CST) + wn * FN * 32 + 4 * lh;
        if constexpr (CREG) {
#pragma unroll
            for (int j = 0; j < FN; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    s4[j][g] = *(const f32x4*)(cst + j * 32 + 8 * g);
                    b4[j][g] = *(const f32x4*)(cst + 128 + j * 32 + 8 * g);
                    if constexpr (OUT2) {
                        s2[j][g] = *(const f32x4*)(cst + 256 + j * 32 + 8 * g);
                        b2[j][g] = *(const f32x4*)(cst + 384 + j * 32 + 8 * g);
                    }
                }
        }
        asm volatile("" ::: "memory");
        if constexpr (RES) s1_blocks([&](auto b_c) { res_dma(decltype(b_c)::value); }, std::make_integer_sequence<int, (PD < B ? PD : B)>{});
        float satmax = 0.f;
        auto block = [&](auto b_c) {
            constexpr int b = decltype(b_c)::value, i = b / FN, j = b % FN;
            if constexpr (RES) {
                asm volatile("" ::: "memory");
                if constexpr (b + PD < B) res_dma(b + PD);
                // younger than shortcut block b: the shortcut blocks b + 1 .. b + PD and the stores of blocks b - PD .. b - 1
                constexpr int ahead = (B - 1 - b) < PD ? (B - 1 - b) : PD, behind = b < PD ? b : PD;
                s1_wait<4 * ahead + NST * behind>();
                asm volatile("" ::: "memory");
            }
            char* to = wv + (b % NSTG) * 4096;
            char* t2 = wv + NSTG * 4096 + (b % NSTG) * 4096;
            const char* rt = wv + 2 * NSTG * 4096 + (b % RT) * 4096;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const unsigned oh_ = lr * 128 + (((2 * g) ^ sw) << 4) + 8 * lh, ol_ = lr * 128 + (((2 * g + 1) ^ sw) << 4) + 8 * lh;
                float c[4];
                f32x4 sv, bv, s2v = {}, b2v = {};
                if constexpr (CREG) {
                    sv = s4[j][g]; bv = b4[j][g];
                    if constexpr (OUT2) { s2v = s2[j][g]; b2v = b2[j][g]; }
                } else {
                    sv = *(const f32x4*)(cst + j * 32 + 8 * g); bv = *(const f32x4*)(cst + 128 + j * 32 + 8 * g);
                    if constexpr (OUT2) { s2v = *(const f32x4*)(cst + 256 + j * 32 + 8 * g); b2v = *(const f32x4*)(cst + 384 + j * 32 + 8 * g); }
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) c[e] = fmaf(acc[i][j][4 * g + e], sv[e], bv[e]);
                if constexpr (RES) {
                    const unsigned long long rh = *(const unsigned long long*)(rt + oh_), rl = *(const unsigned long long*)(rt + ol_);
                    c[0] += split_sum_lo((unsigned)rh, (unsigned)rl);
                    c[1] += split_sum_hi((unsigned)rh, (unsigned)rl);
                    c[2] += split_sum_lo((unsigned)(rh >> 32), (unsigned)(rl >> 32));
                    c[3] += split_sum_hi((unsigned)(rh >> 32), (unsigned)(rl >> 32));
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    satmax = sat_acc(satmax, c[e]);
                    c[e] = __builtin_amdgcn_fmed3f(c[e], lo_clamp, HMMR_SPLIT_MAX);
                }
                unsigned h01, l01, h23, l23;
                split2_mix(c[0], c[1], h01, l01);
                split2_mix(c[2], c[3], h23, l23);
                *(unsigned long long*)(to + oh_) = (unsigned long long)h01 | ((unsigned long long)h23 << 32);
                *(unsigned long long*)(to + ol_) = (unsigned long long)l01 | ((unsigned long long)l23 << 32);
                if constexpr (OUT2) {
                    float u[4];
                    u[0] = fmaf(split_sum_lo(h01, l01), s2v[0], b2v[0]);
                    u[1] = fmaf(split_sum_hi(h01, l01), s2v[1], b2v[1]);
                    u[2] = fmaf(split_sum_lo(h23, l23), s2v[2], b2v[2]);
                    u[3] = fmaf(split_sum_hi(h23, l23), s2v[3], b2v[3]);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        satmax = sat_acc_signed(satmax, u[e]);
                        u[e] = __builtin_amdgcn_fmed3f(u[e], 0.f, HMMR_SPLIT_MAX);
                    }
                    unsigned p01, q01, p23, q23;
                    split2_mix(u[0], u[1], p01, q01);
                    split2_mix(u[2], u[3], p23, q23);
                    *(unsigned long long*)(t2 + oh_) = (unsigned long long)p01 | ((unsigned long long)p23 << 32);
                    *(unsigned long long*)(t2 + ol_) = (unsigned long long)q01 | ((unsigned long long)q23 << 32);
                }
            }
            u32x4 xr[4], x2[OUT2 ? 4 : 1];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                xr[q] = *(const u32x4*)(to + q * 1024 + lane * 16);
                if constexpr (OUT2) x2[q] = *(const u32x4*)(t2 + q * 1024 + lane * 16);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                bool valid;
                const unsigned off = row_off(i, q, valid);
                *(u32x4*)(valid ? out_t + off + j * 128 : (char*)g_s1_dump + lane * 16) = xr[q];
                if constexpr (OUT2) *(u32x4*)(valid ? out2_t + off + j * 128 : (char*)g_s1_dump + lane * 16) = x2[q];
            }
        };
        s1_blocks(block, std::make_integer_sequence<int, B>{});
        split_flag_max(satmax);
        return;
    }

    // ---- epilogue: D layout (lane = pixel, 4 consecutive channels per register group) -> folded BN, ReLU, split -> this wave's
    // staging tiles (rows of 128 B, slot XOR-swizzled by (row >> 1) & 7; the rings are idle) -> 16-byte row stores
    const bool second = nt >= a.split_tiles;
    bsplit_t* const obase = (bsplit_t*)(second ? a.out_b : a.out);
    const int ldo = second ? a.ldo_b : a.ldo;
    const int relu = second ? a.relu_b : a.relu;
    char* stg = smem + wave * 8192;
    const int rsub = lane >> 3, pslot = lane & 7, sw = (lr >> 1) & 7;
    const int nb = (second ? nt - a.split_tiles : nt) * 128 + wn * FN * 32;
    const float lo_clamp = relu ? 0.f : -HMMR_SPLIT_MAX;
    f32x4 s4[FN][4], b4[FN][4];
    {
        const float* cst = (const float*)(smem + CST) + wn * FN * 32 + 4 * lh;
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                s4[j][g] = *(const f32x4*)(cst + j * 32 + 8 * g);
                b4[j][g] = *(const f32x4*)(cst + 128 + j * 32 + 8 * g);
            }
    }
    float satmax = 0.f;
    int blk = 0;
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        const int mb = m0 + (wm * FM + i) * 32;
        bsplit_t* orow[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = 8 * q + rsub, m = mb + r;
            const int ls = pslot ^ ((r >> 1) & 7);
            orow[q] = m < a.M ? obase + (long long)m * ldo + nb + ls * 4 : (bsplit_t*)g_s1_dump + lane * 4;
        }
#pragma unroll
        for (int j = 0; j < FN; ++j, ++blk) {
            char* tile = stg + (blk & 1) * 4096;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float c[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v = fmaf(acc[i][j][4 * g + e], s4[j][g][e], b4[j][g][e]);
                    satmax = sat_acc(satmax, v);
                    c[e] = __builtin_amdgcn_fmed3f(v, lo_clamp, HMMR_SPLIT_MAX);
                }
                unsigned h01, l01, h23, l23;
                split2_mix(c[0], c[1], h01, l01);
                split2_mix(c[2], c[3], h23, l23);
                const unsigned long long oh = (unsigned long long)h01 | ((unsigned long long)h23 << 32);
                const unsigned long long ol = (unsigned long long)l01 | ((unsigned long long)l23 << 32);
                *(unsigned long long*)(tile + lr * 128 + (((2 * g) ^ sw) << 4) + 8 * lh) = oh;
                *(unsigned long long*)(tile + lr * 128 + (((2 * g + 1) ^ sw) << 4) + 8 * lh) = ol;
            }
            u32x4 xr[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) xr[q] = *(const u32x4*)(tile + q * 1024 + lane * 16);
#pragma unroll
            for (int q = 0; q < 4; ++q) *(u32x4*)(orow[q] + j * 32) = xr[q];
        }
    }
    split_flag_max(satmax);
}

template <int FM, int FN, int WGM, int WGN, int D, int EPI = 0, bool IN2 = false, bool RES = false, bool OUT2 = false, int OCC = 1>
int launch_s1(const S1Args& base, hipStream_t stream) {
    S1Args a = base;
    constexpr int BM = 32 * WGM * FM, NPX = (BM + 63) / 64;
    constexpr int lds = D * 8192 + D * NPX * 4096 + 2048;
    HMMR_REQUIRE(a.nk >= D, "hmmr_conv_gemm: k_order 2 (1x1): cin must be at least %d channels for this tile", 16 * D);
    a.n_tiles = ((a.M + BM - 1) / BM) * a.tiles_n;
    auto kern = conv1x1_stream_kernel<FM, FN, WGM, WGN, D, EPI, IN2, RES, OUT2, OCC>;
    static DeviceOnce once;
    if (const unsigned long long bit = once.due()) {
        HMMR_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        once.mark(bit);
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)a.n_tiles), dim3(256), lds, stream, a);
    HMMR_CHECK_HIP(hipGetLastError());
    return 0;
}

// the conv3 form of a tile shape: which of in2 / res / out2 the launch has picks the instantiation
template <int FM, int FN, int WGM, int WGN, int D, int OCC = 1>
int launch_s1_c3(const S1Args& a, hipStream_t stream) {
    if (a.in2) return a.out2 ? launch_s1<FM, FN, WGM, WGN, D, 1, true, false, true, OCC>(a, stream) : launch_s1<FM, FN, WGM, WGN, D, 1, true, false, false, OCC>(a, stream);
    if (a.res) return a.out2 ? launch_s1<FM, FN, WGM, WGN, D, 1, false, true, true, OCC>(a, stream) : launch_s1<FM, FN, WGM, WGN, D, 1, false, true, false, OCC>(a, stream);
    return a.out2 ? launch_s1<FM, FN, WGM, WGN, D, 1, false, false, true, OCC>(a, stream) : launch_s1<FM, FN, WGM, WGN, D, 0, false, false, false, OCC>(a, stream);
}

}  // namespace

// bytes of the filter stream of a 1x1 layer (packing.pack_conv1x1_stream): per tile of 128 output channels, cin / 16 K steps of 8 KB
extern "C" size_t hmmr_conv1x1_stream_bytes(int cin, int cout) {
    return (size_t)(cout / 128) * (size_t)(cin / 16) * 8192;
}

// hmmr_conv_gemm with k_order = 2 and a 1x1 filter (called from gemm_conv.hip)
int hmmr_conv1x1_stream(const hmmr_conv_desc_t* d, hipStream_t stream) {
    const long long M = (long long)d->n_img * d->ho * d->wo;
    HMMR_REQUIRE(d->in_dtype == HMMR_F16X3 && d->out_dtype == HMMR_F16X3, "hmmr_conv_gemm: k_order 2 with a 1x1 filter is built for split (f16x3) tensors");
    HMMR_REQUIRE(d->kh == 1 && d->kw == 1 && d->sy == 1 && d->sx == 1 && d->py == 0 && d->px == 0 && d->ho == d->hin && d->wo == d->win &&
                 d->in_px_stride == d->cin && d->in_row_stride == d->win * d->cin && d->in_img_stride == (int64_t)d->hin * d->win * d->cin &&
                 d->cin % 16 == 0 && d->cout % 128 == 0 && d->scale && d->shift && d->out && !d->pro_scale && d->split_k <= 1 && d->batch <= 1,
                 "hmmr_conv_gemm: k_order 2 (1x1) is for stride-1 1x1 convolutions over a dense [M][cin] tensor, cin %% 16 == 0, cout %% 128 == 0, "
                 "scale, shift and out given: no pro_scale, split_k, batch");
    const bool c3 = d->res || d->out2 || d->in2;               // the conv3 form
    HMMR_REQUIRE(!c3 || (!d->out_b && !(d->res && d->in2) && (!d->res || (!d->res_strided && d->ldr == d->ldo)) && (!d->in2 || (d->cin2 > 0 && d->cin2 % 16 == 0)) &&
                         (!d->out2 || (d->scale2 && d->shift2)) && d->ldo % 32 == 0),
                 "hmmr_conv_gemm: k_order 2 (1x1) with res / out2 / in2 (the conv3 form): no out_b, not res and in2 together, res dense with ldr == ldo, "
                 "cin2 %% 16 == 0, scale2 + shift2 with out2");
    HMMR_REQUIRE(!d->out_b || (d->n_split % 128 == 0 && d->n_split > 0 && d->n_split < d->cout), "hmmr_conv_gemm: k_order 2 (1x1): n_split must be a multiple of 128 inside (0, cout)");
    HMMR_REQUIRE(M < (1ll << 31), "hmmr_conv_gemm: k_order 2 (1x1): more than 2^31 pixels");
    if (M <= 0) return 0;
    hmmr_count_launch(HMMR_COUNT_CONV1X1_STREAM);
    S1Args a = {};
    a.in = (const char*)d->in; a.wstream = (const char*)d->w; a.scale = d->scale; a.shift = d->shift;
    a.out = d->out; a.ldo = d->ldo; a.relu = d->relu;
    a.tiles_n = d->cout / 128;
    a.split_tiles = d->out_b ? d->n_split / 128 : a.tiles_n;
    a.out_b = d->out_b; a.ldo_b = d->ldo_b; a.relu_b = d->relu_b;
    a.M = (int)M; a.C = d->cin; a.nk1 = d->cin / 16;
    a.in2 = (const char*)d->in2; a.C2 = d->in2 ? d->cin2 : 0;
    a.nk = a.nk1 + a.C2 / 16;
    a.res = (const char*)d->res; a.out2 = d->out2; a.scale2 = d->scale2; a.shift2 = d->shift2;
    a.nt_stride = (long long)a.nk * 8192;
    int tile = d->tile;
    if (!tile) {
        // the library's choice: fewest rounds of 256 workgroups x (row blocks per tile + ~1.5 for a tile's prologue and epilogue); the 7 x 1 / 8 x 1
        // wave tiles read 16-18 fragments per 21-24 MFMAs and pay ~15 % for it (profiles/r05r: LDS bandwidth)
        // (29, round 6: a 128-pixel tile for SHORT launches -- block 4's conv1 at 64 frames is 98 row blocks x 4 channel tiles: 52 workgroups of tile 25)
        static const int cand[6][3] = {{25, 8, 6}, {24, 14, 4}, {26, 8, 3}, {22, 7, 6}, {23, 8, 6}, {29, 4, 6}};      // tile, row blocks, ring depth
        const long long rbs = (M + 31) / 32;
        double best = 0;
        for (const auto& cd : cand) {
            if (a.nk < cd[2] || (c3 && cd[0] != 24 && cd[0] != 25 && cd[0] != 26)) continue;      // (the conv3 form's epilogue tiles live in the idle rings: a 128-pixel tile's are too small)
            const long long tiles = ((rbs + cd[1] - 1) / cd[1]) * a.tiles_n;
            double cost = (double)((tiles + 255) / 256) * (cd[1] + 1.5) * (cd[0] == 22 || cd[0] == 23 ? 1.15 : 1.0);
            // tile 26 (two workgroups per CU): a round is 512 workgroups and takes two tiles' time, less what one workgroup's prologue,
            // epilogue and round trips hide under the other's loop -- measured (profiles/r05v): 0.8 of it for the conv3 form (its epilogue is
            // HBM time), 0.88 for a short K loop, nothing gained on a long one
            if (cd[0] == 26) cost = (double)((tiles + 511) / 512) * 2.0 * (cd[1] + 1.5) * (c3 ? 0.8 : a.nk <= 32 ? 0.88 : 1.05);
            if (!tile || cost < best) { tile = cd[0]; best = cost; }
        }
    }
    HMMR_REQUIRE(tile, "hmmr_conv_gemm: k_order 2 (1x1): cin must be at least 64 channels (the rings are 4 K steps deep)");
    if (c3) {
        HMMR_REQUIRE(tile >= 24 && tile <= 26, "hmmr_conv_gemm: k_order 2 (1x1): the conv3 form (res / out2 / in2) runs tiles 24 .. 26, not %d", tile);
        return tile == 24 ? launch_s1_c3<7, 2, 2, 2, 4>(a, stream) : tile == 25 ? launch_s1_c3<4, 2, 2, 2, 6>(a, stream) : launch_s1_c3<4, 2, 2, 2, 3, 2>(a, stream);
    }
    switch (tile) {
    case 22: return launch_s1<7, 1, 1, 4, 6>(a, stream);       // 224 pixels, every wave all of them and 32 of the 128 channels
    case 23: return launch_s1<8, 1, 1, 4, 6>(a, stream);       // 256 pixels, likewise
    case 24: return launch_s1<7, 2, 2, 2, 4>(a, stream);       // 448 pixels, waves 2 x 2
    case 25: return launch_s1<4, 2, 2, 2, 6>(a, stream);       // 256 pixels, waves 2 x 2
    case 26: return launch_s1<4, 2, 2, 2, 3, 0, false, false, false, 2>(a, stream);       // 256 pixels, TWO workgroups per CU (rings 3 deep)
    case 29: return launch_s1<2, 2, 2, 2, 6>(a, stream);       // 128 pixels, waves 2 x 2: short launches
    default: break;
    }
    hmmr_set_error("hmmr_conv_gemm: k_order 2 (1x1) runs tiles 22 .. 26 and 29, not %d", tile);
    return -1;
}
