// 3x3 / stride 1 / SAME convolutions of the ResNet's conv2 layers for gfx950, f16x3 ("split") operands, as ONE MFMA STREAM PER SIMD
// (hmmr_conv_desc_t.k_order = 2; tiles 12 .. 21).  slim resnet_v2.bottleneck `conv2` as invoked at src/models.py:65-75.
//
// What the 8-wave patch tiles of gemm_conv.hip (k_order 1) measure: the matrix pipes are busy 45 % of a launch, 394 workgroups go to
// 256 CUs (block 3), and a launch without its MFMAs is still 80 % as long -- two waves per SIMD that meet at a barrier every 24
// MFMAs spend their time on their own skeleton.  This kernel is built the way csrc/unit_pair.hip is:
//   * ONE wave per SIMD (4-wave workgroups, one per CU) with all 512 registers: FM x FN accumulators of 32 x 32 in the AGPR half
//     (up to 16), two sets of operand fragments in the VGPR half.  A K step (one tap x 16 channels) is 3 FM FN MFMAs -- 42 for
//     the 7 x 2 wave tile -- and ONE barrier; the (FM + FN) x 2 fragment reads of the next step sit between them, at most one
//     ds_read_b128 per MFMA pair.  LDS reads per MFMA are half those of a 64 x 64 wave tile;
//   * the tile is R x 32 pixels by 128 output channels with R chosen per layer so that the launch is a whole number of rounds of
//     256 workgroups (R = 14: 226 tiles in block 3, 450 in block 2; the 256 x 128 tile: 394 and 788);
//   * filters: the host packs them as the stream of MFMA A-operand fragments the kernel consumes (packing.pack_conv3x3_stream:
//     [128-channel tile][K step][4 row blocks][hi plane | lo plane] of 1 KB, lane-linear), the waves DMA a K step (8 KB) into
//     a 6-slab ring six steps ahead (global_load_lds; waits are COUNTED s_waitcnt vmcnt(N)); reads are conflict-free with no
//     address arithmetic at all (ring slot and row block are instruction offsets);
//   * pixels: K is chunk-major in 16-channel chunks (K step kt = chunk kt / 9, tap kt % 9).  A chunk of every input pixel the tile
//     can touch is DMA'd once into a PATCH (two buffers: chunk c + 1 lands while chunk c is consumed): rows of 64 bytes
//     [hi k0-7 | lo k0-7 | hi k8-15 | lo k8-15], four lanes per row = one coalesced chunk, the 16-byte slots XOR-swizzled with
//     (row >> 2) & 3 (the 16 lanes of a ds_read_b128 group hit 16 distinct slots for any tap shift).  Patch row r holds input pixel
//     m0 - (W + 1) + r of the flattened [img][y][x] order, so tap (ky, kx) of tile pixel i is row i + ky W + kx for EVERY pixel: one
//     address pair per step, the FM row blocks are instruction offsets.  The taps are unrolled (18 steps = two chunks: ring slot,
//     fragment set, patch buffer and tap are compile-time).  SAME padding is an address select: per row block four 32-bit wave masks
//     in SCALAR registers (pixels in the top / bottom row, left / right column); a lane whose tap is outside its image reads a zero
//     row of the same bank instead (s_mov vcc + two v_cndmask per fragment, nothing for the centre tap);
//   * bf16 tensors (SPLIT = false): the same rows and fragments hold 32 channels = two 16-wide MFMA chunks; two MFMAs per
//     accumulator and step, K steps of 32 channels, bf16 rows out.
// DESIGN.md section 4.1.2 has the probe studies (profiles/r04a / r04b / r04d) behind each of these choices.
// Products and their order per output element: (w.hi x.lo, w.lo x.hi, w.hi x.hi) per K step, K steps in stream order -- the
// same for every tile shape, so every tile of this kernel produces the same bits (they differ from k_order 0 / 1 by the fp32
// rounding of a different summation order only).
#include <type_traits>

#include "common.h"
#include "hmmr_hip.h"

namespace {

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ u32x4 g_s3_dump[64 + 32];       // where the stores of rows beyond M go (a lane's 16 bytes, up to 4 row blocks of 128 B further)

struct S3Args {
    const char* in;                         // [M][C] split rows
    const char* wstream;                    // packing.pack_conv3x3_stream
    const float* scale; const float* shift; // [cout]
    void* out; int ldo;                     // bsplit_t (split) or bf16_t rows
    int M, H, W, C, nc;                     // nc = C / 16
    int relu, tiles_n, n_tiles;
    long long nt_stride;                    // bytes of one 128-channel tile of the stream: 9 nc x 8 KB
    unsigned long long* ts;                 // probe build: s_memtime stamps, or NULL
};

// Development build only (tools/s3_probe_build.sh): drop the MFMAs (1), the fragment reads (2), the DMA requests and their waits (4),
// the loop's barrier (8), the per-step address arithmetic (16), the waits on vector memory alone (32) or the patch requests alone (64) at COMPILE time; 128: the patch requests read coalesced (wrong) bytes, and stamp s_memtime around the phases.  Results are
// garbage in those modes; the product build compiles the switches away.
#ifndef S3_PROBE_BITS
#define S3_PROBE_BITS 0
#endif
#define S3_PROBE(bit) (((S3_PROBE_BITS) & (bit)) != 0)
#ifdef HMMR_GEMM_PROBE
#define S3_STAMP(k) do { if (a.ts) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); if ((threadIdx.x & 63) == 0) a.ts[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 8 + (k)] = t_; } } while (0)
#else
#define S3_STAMP(k) do { } while (0)
#endif

// s_waitcnt vmcnt(N) lgkmcnt(0) (gfx9 encoding: vmcnt[3:0] | expcnt[6:4] | lgkmcnt[11:8] | vmcnt_hi[15:14])
template <int N> __device__ __forceinline__ void s3_wait() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
    __builtin_amdgcn_s_waitcnt((N & 15) | (7 << 4) | (0 << 8) | ((N >> 4) << 14));
}

// the loop's LDS reads as inline assembly: the compiler does not track them (it would wait lgkmcnt(0) at the first use of any of
// them); the step ends with one lgkmcnt(0) of its own
template <int OFF> __device__ __forceinline__ shalf8 s3_rd(unsigned addr) {
    shalf8 v; asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF)); return v;
}

struct xfrag { shalf8 hi, lo; };            // MFMA B-operand fragment: 32 pixels x 16 channels

constexpr int S3_NS = 6;                    // ring slabs; a slab = one K step of a tile's NB row blocks (32 channels each) x (hi 1 KB | lo 1 KB)

template <int V> using s3_ic = std::integral_constant<int, V>;

// FM x FN accumulators per wave, WGM x WGN waves: the tile is 32 WGM FM pixels x 32 WGN FN channels (128 or 64); WMAX: the widest image
// SPLIT: f16x3 operands (a K step = 16 channels as hi and lo halves: three MFMAs per product); false: bf16 operands (the same 64-byte
// patch rows and 2 KB fragments hold 32 channels = two 16-wide MFMA chunks: two MFMAs per accumulator and step, the "hi" / "lo" reads
// of the split form are chunk 0 / chunk 1 here)
template <int FM, int FN, int WGM, int WGN, int WMAX, bool SPLIT = true>
__global__ __launch_bounds__(256, 1) void conv3x3_stream_kernel(const S3Args a) {
    constexpr int NP = SPLIT ? 3 : 2;                           // MFMAs per accumulator and step
    constexpr int ESZ = SPLIT ? 4 : 2;                          // bytes per channel of a tensor row
    constexpr int NB = WGN * FN;                                // row blocks of 32 output channels per tile
    static_assert(WGM * WGN == 4 && (NB == 4 || NB == 2) && FM * FN <= 16, "4 waves, 128 or 64 channels, at most 16 accumulators");
    constexpr int R = WGM * FM, BM = 32 * R;
    constexpr int NPP = (BM + 2 * WMAX + 2 + 63) / 64;          // 64-row pieces of a patch (BM + 2 W + 2 rows)
    constexpr int NS = S3_NS, SLAB = NB * 2048, RING = NS * SLAB;
    constexpr int RW = NB / 2;                                  // 1 KB pieces of a slab each wave moves
    constexpr int ZROWS = 32 * (FM - 1) + 16;                   // zero rows behind the data rows of a patch buffer
    constexpr int PBUF = (NPP * 64 + ZROWS) * 64;               // patch rows of 64 bytes: [hi k0][lo k0][hi k1][lo k1], slots XOR-swizzled by (row >> 2) & 3
    constexpr int NR = 2 * (FM + FN), NG = NP * FM * FN;         // fragment reads and MFMAs (= gaps) of a step
    static_assert(2048 * (FM - 1) + 16 < 65536, "instruction offsets");
    static_assert(NR <= NG - 4, "one fragment read per gap");

    extern __shared__ __attribute__((aligned(256))) char smem[];
    S3_STAMP(0);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 31, lh = lane >> 5;
    const int L = xcd_remap(blockIdx.x, a.n_tiles);
    const int mt = L / a.tiles_n, nt = L - mt * a.tiles_n;
    const int m0 = mt * BM;
    const int W = a.W, HW = a.H * a.W;
    const int base = m0 - W - 1;                                // input pixel of patch row 0
    const int nc = a.nc, nk = 9 * nc;
    const int wm = wave / WGN, wn = wave - wm * WGN;
    const unsigned lds0 = (unsigned)(unsigned long long)(lptr_t)smem;

    // ---- patch DMA: piece q of this wave = rows 64 q + 16 wave .. + 15, four lanes per row (one coalesced 64-byte chunk; as quarter
    // planes with one lane per row the same bytes cost four times the L1 transactions and 5 % of the kernel, profiles/r04d).  Rows
    // outside the tensor read its first / last pixel instead: only out-of-image taps (which read a zero row) and pixels >= M see them
    const char* pptr[NPP];
#pragma unroll
    for (int q = 0; q < NPP; ++q) {
        const int r = 64 * q + 16 * wave + (lane >> 2);
        int px = base + r;
        px = px < 0 ? 0 : (px >= a.M ? a.M - 1 : px);
        pptr[q] = a.in + ((long long)px * a.C) * ESZ + (((lane & 3) ^ ((r >> 2) & 3)) << 4);
        if (S3_PROBE(128)) pptr[q] = a.in + ((long long)(px & ~15) * a.C) * ESZ + lane * 16;     // (probe: a coalesced KB of the wrong bytes)
    }
    auto patch_piece = [&](int q, int c, int buf) {             // q is a constant after unrolling
        char* dst = smem + RING + buf * PBUF + wave * 1024;
        __builtin_amdgcn_global_load_lds((gptr_t)(pptr[q] + c * 64), (lptr_t)(dst + q * 4096), 16, 0, 0);
    };
    // ---- filter stream: K step s -> ring slot s % 6; each wave moves a quarter (128 channels: one row block, hi and lo plane)
    const char* gw = a.wstream + (long long)nt * a.nt_stride + wave * (RW * 1024) + lane * 16;
    auto ring_dma = [&](int s, int slot) {
        const char* src = gw + (long long)s * SLAB;
        char* dst = smem + slot * SLAB + wave * (RW * 1024);
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)dst, 16, 0, 0);          // (the instruction offset moves both addresses)
        if constexpr (RW == 2) __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)dst, 16, 1024, 0);
    };
#pragma unroll
    for (int q = 0; q < NPP; ++q) patch_piece(q, 0, 0);
#pragma unroll
    for (int s_ = 0; s_ < NS; ++s_) ring_dma(s_, s_);

    // ---- zero rows of the two patch buffers (never written again)
    for (int i = tid; i < 2 * ZROWS * 4; i += 256) {
        const int pl = i / (ZROWS * 4), o = i - pl * (ZROWS * 4);
        *(u32x4*)(smem + RING + pl * PBUF + NPP * 64 * 64 + o * 16) = u32x4{0u, 0u, 0u, 0u};
    }
    // ---- this tile's folded BN constants (128 scales, 128 shifts) behind the patch buffers: the epilogue reads them from LDS
    {
        float* cst = (float*)(smem + RING + 2 * PBUF);
        if (tid < 64 * NB) cst[tid] = tid < 32 * NB ? a.scale[nt * (32 * NB) + tid] : a.shift[nt * (32 * NB) + tid - 32 * NB];
    }
    // ---- wave masks (32 bits: both k halves of a wave hold the same 32 pixels) of the pixels on an image border, per row block
    unsigned mtop[FM], mbot[FM], mlef[FM], mrig[FM];
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        const int m = m0 + (wm * FM + i) * 32 + lr;
        const int rem = m % HW, y = rem / W, x = rem - y * W;
        mtop[i] = (unsigned)__builtin_amdgcn_ballot_w64(y == 0);
        mbot[i] = (unsigned)__builtin_amdgcn_ballot_w64(y == a.H - 1);
        mlef[i] = (unsigned)__builtin_amdgcn_ballot_w64(x == 0);
        mrig[i] = (unsigned)__builtin_amdgcn_ballot_w64(x == W - 1);
    }

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
    // pixel fragments: row rb0 + tap shift of a patch buffer, slots 2 lh (hi) and 2 lh + 1 (lo) before the swizzle; row block i 2 KB further
    const int rb0 = wm * FM * 32 + lr;
    const unsigned lh2 = SPLIT ? 2 * lh : lh;                   // the first read's slot before the swizzle (split: hi of k half lh; bf16: k half lh of chunk 0)
    constexpr unsigned XL = SPLIT ? 16u : 32u;                  // ... and the second read's: ^ XL (split: the lo slot; bf16: chunk 1)
    xfrag fx[2][FM];
    wfrag fw[2][FN];
    unsigned A0 = 0, A0l = 0, Zb = 0, Zbl = 0, adcur = 0, adcurl = 0;

    // read number r of the NR fragment halves of the step with tap TAP_ (filters first) into set `set`; slot_ = that step's ring slot
    auto read_one = [&](auto tap_c, int set, int r, int slot_) {           // set, r, slot_ are constants after unrolling
        constexpr int TAP_ = decltype(tap_c)::value, KY = TAP_ / 3, KX = TAP_ % 3;
        if (r < 2 * FN) {
            const int j = r >> 1, pl = r & 1;
#define S3_F(SL, J, PL) if (slot_ == SL && j == J && pl == PL) { shalf8 v = s3_rd<SL * SLAB + J * 2048 + PL * 1024>(vW); if (PL) fw[set][J].lo = v; else fw[set][J].hi = v; }
#define S3_FJ(SL, J) S3_F(SL, J, 0) S3_F(SL, J, 1)
#define S3_FS(SL) S3_FJ(SL, 0) if constexpr (FN > 1) { S3_FJ(SL, 1) } if constexpr (FN > 2) { S3_FJ(SL, 2) S3_FJ(SL, 3) }
            S3_FS(0) S3_FS(1) S3_FS(2) S3_FS(3) S3_FS(4) S3_FS(5)
#undef S3_FS
#undef S3_FJ
#undef S3_F
        } else {
            const int i = (r - 2 * FN) >> 1, pl = (r - 2 * FN) & 1;
            if (pl == 0) {
                // SAME padding: lanes whose tap leaves the image read the zero row of their bank
                if (KY == 1 && KX == 1) { adcur = A0; adcurl = A0l; }
                else {
                    const int ii = i < FM ? i : 0;
                    const unsigned my = KY == 0 ? mtop[ii] : (KY == 2 ? mbot[ii] : 0u);
                    const unsigned mx = KX == 0 ? mlef[ii] : (KX == 2 ? mrig[ii] : 0u);
                    asm volatile("s_mov_b32 vcc_lo, %2\n\ts_mov_b32 vcc_hi, %2\n\tv_cndmask_b32 %0, %3, %4, vcc\n\tv_cndmask_b32 %1, %5, %6, vcc"
                                 : "=&v"(adcur), "=v"(adcurl) : "s"(my | mx), "v"(A0), "v"(Zb), "v"(A0l), "v"(Zbl) : "vcc");
                }
            }
#define S3_X(I) if constexpr (I < FM) { if (i == I) { if (pl) fx[set][I].lo = s3_rd<I * 2048>(adcurl); else fx[set][I].hi = s3_rd<I * 2048>(adcur); } }
            S3_X(0) S3_X(1) S3_X(2) S3_X(3) S3_X(4) S3_X(5) S3_X(6) S3_X(7) S3_X(8) S3_X(9) S3_X(10) S3_X(11) S3_X(12) S3_X(13) S3_X(14) S3_X(15)
#undef S3_X
        }
    };
    auto tap_setup = [&](auto tap_c, int buf_) {                // buf_ is a constant after unrolling
        constexpr int TAP_ = decltype(tap_c)::value, KY = TAP_ / 3, KX = TAP_ % 3;
        const unsigned rowp = (unsigned)(rb0 + KY * W + KX);
        const unsigned ph = (lh2 ^ ((rowp >> 2) & 3u)) << 4;    // this lane's hi slot in that row (the lo slot: ^ 16)
        const unsigned pb = lds0 + RING + buf_ * PBUF;
        A0 = pb + rowp * 64 + ph;
        A0l = A0 ^ XL;
        Zb = pb + NPP * 64 * 64 + (rowp & 3u) * 64 + ph;       // the zero row of the same bank (row & 3 and the slot are what the bank is made of)
        Zbl = Zb ^ XL;
    };

    // ---- prologue: patch 0 and stage 0, then the fragments of step 0
    __builtin_amdgcn_sched_barrier(0);
    s3_wait<RW * (NS - 1)>();
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    tap_setup(s3_ic<0>{}, 0);
#pragma unroll
    for (int r = 0; r < NR; ++r) read_one(s3_ic<0>{}, 0, r, 0);
    __builtin_amdgcn_sched_barrier(0);
    s3_wait<RW * (NS - 2)>();                                   // ... stage 1 too, and the fragments of step 0
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    S3_STAMP(1);

    // One K step, S = its position in a pair of chunks (18 steps: tap S % 9, ring slot S % 6, fragment set S & 1, patch buffer S / 9).
    // Every MFMA is followed by a gap that holds at most a few other instructions, so the matrix pipe never waits for the wave's
    // own bookkeeping: gap 0 = the ring's DMA (stage kt + 6 into the slot of stage kt, whose fragments were read during step
    // kt - 1), gap 1 = the addresses of the next step's pixel fragments, gaps 2 ... = that step's fragment reads, spread evenly;
    // with tap 0, the patch of the next chunk goes out one piece per gap
    auto step = [&](auto s_c, int kt, int c) {
        constexpr int S = decltype(s_c)::value, TAP = S % 9, SUB = S % NS, CUR = S & 1, NXT = CUR ^ 1;
        constexpr int S1 = (S + 1) % 18, TAP1 = S1 % 9, SNEXT = S1 % NS, BUF1 = S1 / 9, BUFP = (S / 9) ^ 1;
        const bool more = kt + NS < nk;
        const bool pat = TAP == 0 && c + 1 < nc;
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int p = 0; p < NP; ++p)
#pragma unroll
                for (int j = 0; j < FN; ++j) {
                    if constexpr (SPLIT) {
                        const shalf8& wa = p == 1 ? fw[CUR][j].lo : fw[CUR][j].hi;
                        const shalf8& xb = p == 0 ? fx[CUR][i].lo : fx[CUR][i].hi;
                        if (!S3_PROBE(1)) acc[i][j] = mfma_split(wa, xb, acc[i][j]);
                    } else {
                        const shalf8& wa = p ? fw[CUR][j].lo : fw[CUR][j].hi;
                        const shalf8& xb = p ? fx[CUR][i].lo : fx[CUR][i].hi;
                        if (!S3_PROBE(1)) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wa), __builtin_bit_cast(bf16x8, xb), acc[i][j], 0, 0, 0);
                    }
                    const int g = (NP * i + p) * FN + j;
                    if (g == 0 && more && !S3_PROBE(4)) ring_dma(kt + NS, SUB);
                    if (g == 1 && !S3_PROBE(16)) tap_setup(s3_ic<TAP1>{}, BUF1);           // (past the last step: a harmless read of stale LDS)
                    if (TAP == 0 && g >= 2 && g < 2 + NPP && pat && !S3_PROBE(4) && !S3_PROBE(64)) patch_piece(g - 2 < NPP ? g - 2 : 0, c + 1, BUFP);
#pragma unroll
                    for (int r = 0; r < NR; ++r)
                        if (2 + (r * (NG - 4)) / NR == g && !S3_PROBE(2)) read_one(s3_ic<TAP1>{}, NXT, r, SNEXT);
                    __builtin_amdgcn_sched_barrier(0);
                }
        // stage kt + 2 (read during step kt + 1) has landed; younger requests stay in flight: stages kt + 3 .. kt + 6 and, through
        // taps 0 .. 4, the patch of the next chunk (issued behind the ring stage of tap 0; vmcnt retires in order)
        if (S3_PROBE(4) || S3_PROBE(32)) __builtin_amdgcn_s_waitcnt(63 | (7 << 4) | (0 << 8) | (3 << 14));      // lgkmcnt(0) alone
        else if (more) { if (TAP <= 4 && c + 1 < nc && !S3_PROBE(64)) s3_wait<RW * (NS - 2) + NPP>(); else s3_wait<RW * (NS - 2)>(); }
        else s3_wait<0>();
        __builtin_amdgcn_sched_barrier(0);
        if (!S3_PROBE(8)) __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    auto chunk_pair = [&](int kt0, int c0, auto... S) { (step(S, kt0 + decltype(S)::value, c0 + decltype(S)::value / 9), ...); };
    for (int c0 = 0; c0 < nc; c0 += 2)
        chunk_pair(18 * (c0 >> 1), c0, s3_ic<0>{}, s3_ic<1>{}, s3_ic<2>{}, s3_ic<3>{}, s3_ic<4>{}, s3_ic<5>{}, s3_ic<6>{}, s3_ic<7>{}, s3_ic<8>{},
                   s3_ic<9>{}, s3_ic<10>{}, s3_ic<11>{}, s3_ic<12>{}, s3_ic<13>{}, s3_ic<14>{}, s3_ic<15>{}, s3_ic<16>{}, s3_ic<17>{});
    S3_STAMP(2);

    // ---- epilogue: D layout (lane = pixel, 4 consecutive channels per register group) -> folded BN, ReLU, split -> this wave's
    // staging tiles (rows of 128 B, slot XOR-swizzled by (row >> 1) & 7; the ring is idle) -> 16-byte row stores.  One wave per
    // SIMD: every instruction here is exposed, so the arithmetic is 4.5 instructions per value -- fma, one med3 (ReLU and the
    // fp16 clamp together), half a packed convert for the hi halves, one v_fma_mix per lo half (c - hi, rounded to fp16, written
    // straight into its half of the pair), half a max3 for the saturation flag
    char* stg = smem + wave * 8192;
    const int rsub = lane >> 3, pslot = lane & 7, sw = (lr >> 1) & 7;
    const int nb = nt * (32 * NB) + wn * FN * 32;
    const float lo_clamp = a.relu ? 0.f : -HMMR_SPLIT_MAX;
    f32x4 s4[FN][4], b4[FN][4];
    {
        const float* cst = (const float*)(smem + RING + 2 * PBUF) + wn * FN * 32 + 4 * lh;
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                s4[j][g] = *(const f32x4*)(cst + j * 32 + 8 * g);
                b4[j][g] = *(const f32x4*)(cst + 32 * NB + j * 32 + 8 * g);
            }
    }
    if constexpr (SPLIT) {
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
                orow[q] = m < a.M ? (bsplit_t*)a.out + (long long)m * a.ldo + nb + ls * 4 : (bsplit_t*)g_s3_dump + lane * 4;
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
                    const unsigned h01 = __builtin_bit_cast(unsigned, shalf2{(shalf_t)c[0], (shalf_t)c[1]});
                    const unsigned h23 = __builtin_bit_cast(unsigned, shalf2{(shalf_t)c[2], (shalf_t)c[3]});
                    // lo = fp16(c - hi): c - hi is exact in fp32, so the mixed-precision fma rounds once, like the cast of split4 (common.h)
                    unsigned l01, l23;
                    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l01) : "v"(h01), "v"(c[0]));
                    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l01) : "v"(h01), "v"(c[1]));
                    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l23) : "v"(h23), "v"(c[2]));
                    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l23) : "v"(h23), "v"(c[3]));
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
    } else {
        // bf16 rows: 32 channels = 64 bytes per pixel and accumulator; lane (pixel lr, k half lh) holds channels 8 g + 4 lh .. + 3 of group g
        // as 8 bytes; a 2 KB staging tile per accumulator, rows of 64 B, 16-byte slot g XOR (row >> 2) & 3
        int blk = 0;
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            const int mb = m0 + (wm * FM + i) * 32;
            bf16_t* orow[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int r = 16 * q + (lane >> 2), m = mb + r;
                const int ls = (lane & 3) ^ ((r >> 2) & 3);
                orow[q] = m < a.M ? (bf16_t*)a.out + (long long)m * a.ldo + nb + ls * 8 : (bf16_t*)g_s3_dump + lane * 8;
            }
#pragma unroll
            for (int j = 0; j < FN; ++j, ++blk) {
                char* tile = stg + (blk & 1) * 4096;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
                    bf16x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float v = fmaf(acc[i][j][4 * g + e], s4[j][g][e], b4[j][g][e]);
                        if (a.relu) v = fmaxf(v, 0.f);
                        o[e] = (bf16_t)v;
                    }
                    *(bf16x4*)(tile + lr * 64 + ((g ^ ((lr >> 2) & 3)) << 4) + 8 * lh) = o;
                }
                u32x4 xr[2];
#pragma unroll
                for (int q = 0; q < 2; ++q) xr[q] = *(const u32x4*)(tile + q * 1024 + lane * 16);
#pragma unroll
                for (int q = 0; q < 2; ++q) *(u32x4*)(orow[q] + j * 32) = xr[q];
            }
        }
    }
    S3_STAMP(3);
}

template <int FM, int FN, int WGM, int WGN, int WMAX, bool SPLIT>
int launch_s3_t(const S3Args& base, int cout, hipStream_t stream) {
    S3Args a = base;
    constexpr int NB = WGN * FN, BM = 32 * WGM * FM, NPP = (BM + 2 * WMAX + 2 + 63) / 64;
    constexpr int lds = S3_NS * NB * 2048 + 8 * (NPP * 64 + 32 * (FM - 1) + 16) * 16 + 1024;
    static_assert(lds <= 160 * 1024, "LDS");
    a.tiles_n = cout / (32 * NB);
    a.n_tiles = ((a.M + BM - 1) / BM) * a.tiles_n;
    a.nt_stride = (long long)9 * a.nc * (NB * 2048);
    auto kern = conv3x3_stream_kernel<FM, FN, WGM, WGN, WMAX, SPLIT>;
    static DeviceOnce once;
    if (const unsigned long long bit = once.due()) {
        HMMR_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        once.mark(bit);
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)a.n_tiles), dim3(256), lds, stream, a);
    HMMR_CHECK_HIP(hipGetLastError());
    return 0;
}

template <int FM, int FN, int WGM, int WGN, int WMAX>
int launch_s3(const S3Args& a, int cout, bool split, hipStream_t stream) {
    return split ? launch_s3_t<FM, FN, WGM, WGN, WMAX, true>(a, cout, stream) : launch_s3_t<FM, FN, WGM, WGN, WMAX, false>(a, cout, stream);
}

}  // namespace

// bytes of the filter stream of a 3x3 layer (packing.pack_conv3x3_stream): per tile of 128 output channels (64 when cout is 64), 9 cin / 16
// K steps of 2 KB per 32 channels
extern "C" size_t hmmr_conv3x3_stream_bytes(int cin, int cout) {
    return (size_t)(cout / 32) * (size_t)(9 * (cin / 16)) * 2048;
}

// hmmr_conv_gemm with k_order = 2 (called from gemm_conv.hip, which has checked the descriptor's geometry)
int hmmr_conv3x3_stream(const hmmr_conv_desc_t* d, hipStream_t stream) {
    const bool split = d->in_dtype == HMMR_F16X3 && d->out_dtype == HMMR_F16X3;
    HMMR_REQUIRE(split || (d->in_dtype == HMMR_BF16 && d->out_dtype == HMMR_BF16), "hmmr_conv_gemm: k_order 2 is built for split (f16x3) and bf16 tensors");
    HMMR_REQUIRE(split || d->cin % 64 == 0, "hmmr_conv_gemm: k_order 2 with bf16 tensors needs cin %% 64 == 0 (K steps of 32 channels, taken in pairs)");
    HMMR_REQUIRE(d->cin % 32 == 0 && (d->cout % 128 == 0 || d->cout == 64) && d->win <= 56 && d->scale && d->shift,
                 "hmmr_conv_gemm: k_order 2 needs cin %% 32 == 0, cout %% 128 == 0 (or cout = 64), an image at most 56 pixels wide and scale + shift");
    hmmr_count_launch(HMMR_COUNT_CONV3X3_STREAM);
    S3Args a = {};
    a.in = (const char*)d->in; a.wstream = (const char*)d->w; a.scale = d->scale; a.shift = d->shift;
    a.out = d->out; a.ldo = d->ldo;
    a.M = d->n_img * d->ho * d->wo; a.H = d->hin; a.W = d->win; a.C = d->cin; a.nc = split ? d->cin / 16 : d->cin / 32;      // K chunks: 64 bytes of a pixel's row
    a.relu = d->relu;
#ifdef HMMR_GEMM_PROBE
    a.ts = (unsigned long long*)(((unsigned long long)(unsigned)hmmr_debug_state()->reserved[1] << 32) | (unsigned)hmmr_debug_state()->reserved[0]);
#endif
    int tile = d->tile;
    const bool narrow = d->cout == 64, wide = d->win > 28;
    if (!tile) {
        // the library's choice: fewest rounds of 256 workgroups x (row blocks per tile + ~1.5 for a tile's prologue and epilogue)
        // (27 / 28, round 6: 128- and 192-pixel tiles for SHORT launches -- at the reference's own batch sizes, 64 frames of FeatureExtractor
        //  and the 160 of Tester.predict, blocks 3-4 are 98 - 245 row blocks: 56 - 140 workgroups of the 224-pixel tile on 256 CUs)
        static const int cand[9][3] = {{12, 14, 0}, {13, 8, 0}, {15, 12, 0}, {16, 10, 0}, {21, 7, 0}, {27, 4, 0}, {28, 6, 0}, {19, 20, 1}, {20, 16, 1}};
        const long long rbs = (a.M + 31) / 32, nts = narrow ? 1 : d->cout / 128;
        double best = 0;
        for (const auto& cd : cand) {
            if ((cd[2] != 0) != narrow || ((cd[0] == 21 || cd[0] == 27 || cd[0] == 28) && !split)) continue;
            const long long tiles = ((rbs + cd[1] - 1) / cd[1]) * nts;
            const double cost = (double)((tiles + 255) / 256) * (cd[1] + 1.5);
            if (!tile || cost < best) { tile = cd[0]; best = cost; }
        }
    }
    HMMR_REQUIRE((tile == 19 || tile == 20) == narrow && (!wide || narrow), "hmmr_conv_gemm: k_order 2: tiles 12 .. 18, 21, 27, 28 take cout %% 128 == 0 and images up to 28 pixels wide, "
                 "tiles 19 / 20 cout = 64 and up to 56 (tile %d, cout %d, win %d)", tile, d->cout, d->win);
    switch (tile) {
    case 12: return launch_s3<7, 2, 2, 2, 28>(a, d->cout, split, stream);       // 448 pixels
    case 13: return launch_s3<4, 2, 2, 2, 28>(a, d->cout, split, stream);       // 256
    case 14: return launch_s3<8, 2, 2, 2, 28>(a, d->cout, split, stream);       // 512
    case 15: return launch_s3<6, 2, 2, 2, 28>(a, d->cout, split, stream);       // 384
    case 16: return launch_s3<5, 2, 2, 2, 28>(a, d->cout, split, stream);       // 320
    case 17: return launch_s3<4, 4, 4, 1, 28>(a, d->cout, split, stream);       // 512, waves split the pixels
    case 18: return launch_s3<3, 4, 4, 1, 28>(a, d->cout, split, stream);       // 384
    case 19: return launch_s3<5, 2, 4, 1, 56>(a, d->cout, split, stream);       // 640 pixels x 64 channels
    case 20: return launch_s3<4, 2, 4, 1, 56>(a, d->cout, split, stream);       // 512 x 64
    case 21:                                                                    // 224 pixels, every wave all of them and 32 of the 128 channels
        HMMR_REQUIRE(split, "hmmr_conv_gemm: k_order 2, tile 21 (7 x 1 accumulators per wave) is built for split tensors: with two MFMAs per step "
                     "the bf16 form has no room for its 16 fragment reads");
        return launch_s3_t<7, 1, 1, 4, 28, true>(a, d->cout, stream);
    case 27:                                                                    // 128 pixels x 128 channels (2 x 2 accumulators per wave): short launches
    case 28:                                                                    // 192 pixels (3 x 2)
        HMMR_REQUIRE(split, "hmmr_conv_gemm: k_order 2, tiles 27 / 28 (2 x 2 and 3 x 2 accumulators per wave) are built for split tensors: with two MFMAs "
                     "per step the bf16 form has no room for their fragment reads");
        return tile == 27 ? launch_s3_t<2, 2, 2, 2, 28, true>(a, d->cout, stream) : launch_s3_t<3, 2, 2, 2, 28, true>(a, d->cout, stream);
    default: break;
    }
    hmmr_set_error("hmmr_conv_gemm: k_order 2 runs tiles 12 .. 21, 27 and 28, not %d", tile);
    return -1;
}
