// Fused tail of a ResNet-v2 bottleneck unit for gfx950 (bf16 operands):
//
//     trunk' = conv3(h2) + bias + shortcut                      (1x1, c_mid -> depth; bottleneck_v2 `conv3` + add)
//     h1'    = relu(bn1'(conv1'(relu(bn_pre'(trunk')))))        (the NEXT unit's `preact` + `conv1`)
//
// in ONE kernel (slim resnet_v2.bottleneck as invoked at src/models.py:65-75).  In
// the layer-per-launch schedule the next unit's conv1 re-reads the whole trunk tensor from HBM
// (0.41 GB per block-1 unit at 257 frames, a quarter of the unit's traffic); here the trunk tile
// never leaves the CU between the two GEMMs.
//
// One workgroup (8 waves) owns 128 pixels and walks the `depth` output channels of conv3 in
// chunks of 64, which are exactly the K steps of conv1':
//
//   per chunk  (1) conv3 chunk: 4 MFMAs per wave out of LDS (H2 tile x W3 chunk, K = c_mid = 64)
//              (2) + bias + shortcut, rounded to bf16 IN PLACE in an LDS tile that was pre-filled
//                  with the shortcut chunk
//              (3) the tile is streamed out (coalesced 16-B stores = the trunk tensor) and
//                  pre-activated in place (relu(x*s+b), bf16): it is now K step `chunk` of conv1'
//              (4) conv1' K step: 4 MFMAs per wave (P tile x W1' chunk) into accumulators that live
//                  across the chunks
//   at the end the conv1' accumulators get BN + ReLU and leave through LDS as coalesced stores.
//
// MFMA operands are swapped (weights = A, activations = B), so a lane owns 4 CONSECUTIVE CHANNELS
// of one pixel (D[i][j]: i = channel, j = pixel) -- 8 bytes of a pixel row, which makes steps (2)
// and the final epilogue plain ds_read_b64 / ds_write_b64 without an fp32 transposition buffer.
// a*b commutes exactly and the K order is unchanged, so every value equals the one the separate
// conv3 and (fused-preact) conv1 launches of gemm_conv.hip produce, bit for bit (tested).
//
// All global->LDS staging goes through registers (ordinary loads, ds_write_b128): hipcc then counts
// its own vmcnt waits and `__syncthreads()` stays a bare s_barrier; the shortcut chunk is requested
// TWO chunks ahead (HBM latency), the weight chunks one ahead (L2).
// LDS: H2 16 KB + W3 chunk 8 KB + W1' chunk 8 KB + 2 x 16 KB trunk tiles + 4 KB constants = 68 KB
// -> two workgroups per CU (block 2: 64-pixel tiles, 72 KB).
//
// Optional phases, all selected at compile time and all bit-identical to the launches they replace:
//   CONV2  the unit's 3x3 conv2 (stride 1 or 2, + BN + ReLU) runs first, over h1, as 9 x c_mid/64 K steps
//          through two LDS stages in the region the tail uses afterwards: h2 never exists in HBM;
//   SC     (block1/unit_1, 64-channel input) the conv shortcut is computed per chunk from the unit's input
//          tile instead of being loaded: the shortcut tensor is neither written nor read;
//   !PH2   no next conv1 (a block's stride-2 last unit): the chunks are streamed out raw and/or
//          pre-activated and the launch ends.
// So a whole bottleneck unit of blocks 1-2 is one launch:
//   [conv shortcut] + conv2 + conv3 + add + [next preact + next conv1].
#include "common.h"
#include "hmmr_hip.h"

namespace {

struct TailArgs {
    const bf16_t* h2; const bf16_t* w3; const float* scale3; const float* shift3;
    const bf16_t* res; bf16_t* out;
    const float* pre_scale; const float* pre_shift;
    const bf16_t* w1; const float* scale1; const float* shift1; bf16_t* out_h1;
    int M, depth, ldr, res_strided, Wo, HoWo;
    long long res_img_stride; int res_row_stride, res_px_stride;
    int relu1;
    // conv2 in front (CONV2 kernels): h1 [n][H][W][c_mid], 3x3 SAME stride 1, folded BN + ReLU
    const bf16_t* h1; const bf16_t* w2; const float* scale2; const float* shift2; int H, W, S;   // S = conv2 stride
    bf16_t* out_pre;        // PH2 == false: the pre-activated trunk goes to HBM here (may be NULL), `out` may be NULL
    // conv shortcut computed in the kernel (SC kernels): shortcut = xp [M][64] x wsc [depth][64] + shift_sc, rounded
    // to bf16 like the tensor the separate launch would have written; `res` is not read
    const bf16_t* xp; const bf16_t* wsc; const float* shift_sc;
};

constexpr int NT = 512;

__device__ __forceinline__ f32x16 mma16(const bf16x8& a, const bf16x8& b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ float bf_lo(unsigned v) { return __uint_as_float(v << 16); }
__device__ __forceinline__ float bf_hi(unsigned v) { return __uint_as_float(v & 0xffff0000u); }
__device__ __forceinline__ unsigned pack_bf(float a, float b) {
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
    bf16x2 pk = {(bf16_t)a, (bf16_t)b};
    return __builtin_bit_cast(unsigned, pk);
}

// BM pixels per workgroup, CM = conv3's K, NCH = depth / 64 chunks, N2 = conv1' output channels.
//   block1: <128, 64, 4, 64>    block2: <64, 128, 8, 128>
// Every LDS tile is a stack of [rows][64 bf16] sub-tiles (128-B rows, 16-B slots XOR-swizzled with
// (row>>1)&7, exactly the operand image of gemm_conv.hip).
template <int BM, int CM, int NCH, int N2, bool SC = false>
struct TailCfg {
    static constexpr int KT1 = CM / 64;                        // K steps of conv3
    static constexpr int RB = BM / 64;                         // 64-row blocks of a pixel tile
    static constexpr int H2_BYTES = BM * CM * 2, W3_BYTES = 64 * CM * 2, W1_BYTES = N2 * 128, P_BYTES = BM * 128;
    static constexpr int O_BYTES = BM * N2 * 2;
    static constexpr int OFF_H2 = 0;                           // later: the conv1' output tile
    static constexpr int OFF_W3 = OFF_H2 + (H2_BYTES > O_BYTES ? H2_BYTES : O_BYTES);
    static constexpr int OFF_W1 = OFF_W3 + W3_BYTES;
    static constexpr int OFF_P0 = OFF_W1 + W1_BYTES, OFF_P1 = OFF_P0 + P_BYTES;
    // SC: one trunk tile only (nothing is pre-filled); the P1 slot holds the shortcut's input tile [BM][64] and a
    // [64][64] chunk of its filters follows
    static constexpr int OFF_XP = OFF_P1, OFF_WS = OFF_P1 + P_BYTES;
    static constexpr int OFF_C = OFF_WS + (SC ? 64 * 128 : 0);
    static constexpr int LDS = OFF_C + (SC ? 5 : 4) * NCH * 64 * (int)sizeof(float);
    static constexpr int TM = BM / 32;                         // 32-pixel blocks per tile
    static_assert((N2 / 32) * TM == 8, "conv1' tile must map one 32x32 block to each of the 8 waves");
    static_assert(2 * TM <= 8, "conv3 chunk needs at most 8 waves");
};

// PH2 = false: no next-unit conv1 -- the launch ends after the trunk chunk has been streamed out (raw and/or
// pre-activated): the stride-2 last unit of a block, whose successor also owns a conv shortcut.
template <int BM, int CM, int NCH, int N2, bool CONV2, bool SC, bool PH2>
__global__ __launch_bounds__(NT, 4) void bottleneck_tail_kernel(const TailArgs a) {
    static_assert(PH2 || (CONV2 && !SC), "the single-phase tail is used with conv2 in front and a loaded shortcut");
    static_assert(!SC || (BM == 128 && CM == 64), "the in-kernel shortcut is written for c_in = 64 (block1/unit_1)");
    static_assert(!CONV2 || (CM / 32) * (BM / 32) == 8, "conv2 tile must map one 32x32 block to each of the 8 waves");
    typedef TailCfg<BM, CM, NCH, N2, SC> Cfg;
    constexpr int KT1 = Cfg::KT1, RB = Cfg::RB, TM = Cfg::TM;
    constexpr int depth = NCH * 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m0 = blockIdx.x * BM;

    // ---- staging geometry (as gemm_conv.hip): 8 lanes per 128-B row; one pass of the workgroup fills 64 rows;
    // thread (r0, pslot) owns PHYSICAL slot pslot of row r0 of every pass and fills it with LOGICAL slot lslot
    const int pslot = tid & 7, r0 = tid >> 3;
    const int lslot = pslot ^ ((r0 >> 1) & 7);
    const int st_off = r0 * 128 + pslot * 16;

    float* sScale3 = (float*)(smem + Cfg::OFF_C);
    float* sBias3 = sScale3 + depth;
    float* sPreS = sBias3 + depth;
    float* sPreB = sPreS + depth;
    for (int i = tid; i < depth; i += NT) {
        sScale3[i] = a.scale3 ? a.scale3[i] : 1.0f;
        sBias3[i] = a.shift3 ? a.shift3[i] : 0.0f;
        sPreS[i] = a.pre_scale[i];
        sPreB[i] = a.pre_shift[i];
        if constexpr (SC) (sPreB + depth)[i] = a.shift_sc ? a.shift_sc[i] : 0.0f;
    }
    [[maybe_unused]] const float* sBiasSc = sPreB + depth;

    // pixel rows of this thread and their shortcut addresses (flat rows, or x[:, ::s, ::s] of the unit input)
    bool rok[RB]; long long roff[RB];
#pragma unroll
    for (int p = 0; p < RB; ++p) {
        const int m = m0 + r0 + 64 * p;
        rok[p] = m < a.M;
        const int mm = rok[p] ? m : 0;
        if (a.res_strided) {
            const int img = mm / a.HoWo, rem = mm - img * a.HoWo;
            const int oy = rem / a.Wo, ox = rem - oy * a.Wo;
            roff[p] = (long long)img * a.res_img_stride + (long long)oy * a.res_row_stride + (long long)ox * a.res_px_stride;
        } else {
            roff[p] = (long long)mm * a.ldr;
        }
    }
    auto load_res = [&](int nc, u32x4 (&r)[RB]) {
#pragma unroll
        for (int p = 0; p < RB; ++p) r[p] = *(const u32x4*)(a.res + roff[p] + nc * 64 + lslot * 8);
    };
    auto store_rows = [&](int off, const u32x4 (&r)[RB]) {       // a [BM][64] tile
#pragma unroll
        for (int p = 0; p < RB; ++p) *(u32x4*)(smem + off + st_off + p * (64 * 128)) = r[p];
    };
    auto load_w3 = [&](int nc, u32x4 (&r)[KT1]) {                // rows = 64 channels of chunk nc, KT1 K steps
#pragma unroll
        for (int kt = 0; kt < KT1; ++kt) r[kt] = *(const u32x4*)(a.w3 + (long long)(nc * 64 + r0) * CM + kt * 64 + lslot * 8);
    };
    auto store_w3 = [&](const u32x4 (&r)[KT1]) {
#pragma unroll
        for (int kt = 0; kt < KT1; ++kt) *(u32x4*)(smem + Cfg::OFF_W3 + kt * (64 * 128) + st_off) = r[kt];
    };
    auto load_w1 = [&](int nc, u32x4 (&r)[N2 / 64]) {            // rows = N2 output channels, K step nc
#pragma unroll
        for (int p = 0; p < N2 / 64; ++p) r[p] = *(const u32x4*)(a.w1 + (long long)(r0 + 64 * p) * depth + nc * 64 + lslot * 8);
    };
    auto store_w1 = [&](const u32x4 (&r)[N2 / 64]) {
#pragma unroll
        for (int p = 0; p < N2 / 64; ++p) *(u32x4*)(smem + Cfg::OFF_W1 + st_off + p * (64 * 128)) = r[p];
    };

    // ---- fragment / epilogue geometry: wave (wn, wm) owns channels [32 wn, +32) x pixels [32 wm, +32);
    // conv3 chunks have 64 channels, so only the waves with wn < 2 work in that phase
    const int wn = wave / TM, wm = wave % TM;
    const int lr = lane & 31, lh = lane >> 5;
    const int fsw = (lr >> 1) & 7;
    auto frag = [&](int tile_off, int row, int kc) {
        return *(const bf16x8*)(smem + tile_off + row * 128 + (((2 * kc + lh) ^ fsw) << 4));
    };
    const int prow = (wm * 32 + lr) * 128;                    // this lane's pixel row inside a [BM][64] tile

    // ---- prologue: weight chunks 0 and shortcut chunks 0, 1 are requested first (they fly while conv2 runs)
    u32x4 rres[2][RB], rw3[KT1], rw1[N2 / 64];
    [[maybe_unused]] u32x4 rws;
    auto load_ws = [&](int nc) { return *(const u32x4*)(a.wsc + (long long)(nc * 64 + r0) * 64 + lslot * 8); };
    load_w3(0, rw3);
    if constexpr (PH2) load_w1(0, rw1);
    if constexpr (SC) {
        // the shortcut's operand tile and filter chunk 0 go to LDS right away (their slots are not used by conv2)
        u32x4 rx[RB];
#pragma unroll
        for (int p = 0; p < RB; ++p) rx[p] = *(const u32x4*)(a.xp + (long long)(rok[p] ? m0 + r0 + 64 * p : 0) * 64 + lslot * 8);
        rws = load_ws(0);
        store_rows(Cfg::OFF_XP, rx);
        *(u32x4*)(smem + Cfg::OFF_WS + st_off) = rws;
        if (NCH > 1) rws = load_ws(1);
    } else {
        load_res(0, rres[0]);
        if (NCH > 1) load_res(1, rres[1]);
    }
    if constexpr (CONV2) {
        // ---- conv2: 3x3 SAME stride 1 over h1 [.., CM], K steps = 9 taps x CM/64 channel blocks, D[channel][pixel];
        // two LDS stages of {pixels [BM][64], W2 block [CM][64]} in the region the tail uses afterwards, operands
        // staged through registers two and three K steps ahead (zero for the out-of-image taps)
        constexpr int STAGE = BM * 128 + CM * 128, NK = 9 * KT1, WP = CM / 64;
        const bf16_t* pbase[RB]; unsigned pmask[RB];
#pragma unroll
        for (int p = 0; p < RB; ++p) {
            const int m = rok[p] ? m0 + r0 + 64 * p : 0;
            // output pixel (oy, ox) of the Ho x Wo grid reads input pixels (oy*S - 1 + ky, ox*S - 1 + kx): SAME for
            // stride 1, slim's conv2d_same (pad 1 + VALID) for stride 2
            const int img = m / a.HoWo, rem = m - img * a.HoWo;
            const int oy = rem / a.Wo, ox = rem - oy * a.Wo;
            pbase[p] = a.h1 + (((long long)img * a.H + oy * a.S) * a.W + ox * a.S) * CM + lslot * 8;
            unsigned mk = 0u;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int iy = oy * a.S - 1 + t / 3, ix = ox * a.S - 1 + t % 3;
                if ((unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W) mk |= 1u << t;
            }
            pmask[p] = rok[p] ? mk : 0u;
        }
        u32x4 ra[2][RB], rb[2][WP];                            // two register sets: K steps k+2 and k+3 in flight
        auto load_step = [&](int k, int set) {
            const int t = k / KT1, kt = k % KT1;
            const int off = ((t / 3 - 1) * a.W + (t % 3 - 1)) * CM + kt * 64;
#pragma unroll
            for (int p = 0; p < RB; ++p) {
                u32x4 v = {0u, 0u, 0u, 0u};
                if ((pmask[p] >> t) & 1u) v = *(const u32x4*)(pbase[p] + off);
                ra[set][p] = v;
            }
#pragma unroll
            for (int p = 0; p < WP; ++p) rb[set][p] = *(const u32x4*)(a.w2 + (long long)(r0 + 64 * p) * (9 * CM) + k * 64 + lslot * 8);
        };
        auto store_step = [&](int stage, int set) {
            store_rows(stage * STAGE, ra[set]);
#pragma unroll
            for (int p = 0; p < WP; ++p) *(u32x4*)(smem + stage * STAGE + BM * 128 + st_off + p * (64 * 128)) = rb[set][p];
        };
        f32x16 acc0;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc0[r] = 0.f;
        load_step(0, 0);
        store_step(0, 0);
        load_step(1, 1);
        load_step(2, 0);
        __syncthreads();
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            const int cur = k & 1;
            if (k + 1 < NK) store_step(cur ^ 1, (k + 1) & 1);
            if (k + 3 < NK) load_step(k + 3, (k + 1) & 1);
#pragma unroll
            for (int kc = 0; kc < 4; ++kc)
                acc0 = mma16(frag(cur * STAGE + BM * 128, wn * 32 + lr, kc), frag(cur * STAGE, wm * 32 + lr, kc), acc0);
            __syncthreads();
        }
        // BN + ReLU, bf16 -> the H2 tile (all stage reads are behind the barrier above)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int n = wn * 32 + 8 * g + 4 * lh, cl = n & 63;
            const f32x4 s4 = *(const f32x4*)(a.scale2 + n), b4 = *(const f32x4*)(a.shift2 + n);
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = fmaxf(fmaf(acc0[4 * g + j], s4[j], b4[j]), 0.f);
            char* q = smem + Cfg::OFF_H2 + (n >> 6) * (BM * 128) + prow + (((cl >> 3) ^ fsw) << 4) + 8 * lh;
            *(unsigned long long*)q = (unsigned long long)pack_bf(v[0], v[1]) | ((unsigned long long)pack_bf(v[2], v[3]) << 32);
        }
    } else {
        u32x4 rh[KT1][RB];
#pragma unroll
        for (int kt = 0; kt < KT1; ++kt)
#pragma unroll
            for (int p = 0; p < RB; ++p) {
                const int m = rok[p] ? m0 + r0 + 64 * p : 0;
                rh[kt][p] = *(const u32x4*)(a.h2 + (long long)m * CM + kt * 64 + lslot * 8);
            }
#pragma unroll
        for (int kt = 0; kt < KT1; ++kt) store_rows(Cfg::OFF_H2 + kt * (BM * 128), rh[kt]);
    }
    store_w3(rw3);
    if constexpr (PH2) store_w1(rw1);
    if constexpr (!SC) store_rows(Cfg::OFF_P0, rres[0]);
    // weight chunks are re-requested the moment their registers are free (right after the ds_write of
    // the previous chunk), so each request has a whole chunk iteration to come back from L2
    if (NCH > 1) { load_w3(1, rw3); if constexpr (PH2) load_w1(1, rw1); }
    __syncthreads();

    f32x16 acc2;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc2[r] = 0.f;

#pragma unroll
    for (int nc = 0; nc < NCH; ++nc) {
        const int pcur = (!SC && (nc & 1)) ? Cfg::OFF_P1 : Cfg::OFF_P0;
        [[maybe_unused]] const int pnxt = (nc & 1) ? Cfg::OFF_P0 : Cfg::OFF_P1;
        // (a) requests for later chunks
        if constexpr (!SC) if (nc + 2 < NCH) load_res(nc + 2, rres[nc & 1]);
        if (wn < 2) {
            // (b) conv3 chunk: D[channel][pixel]
            f32x16 acc1;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc1[r] = 0.f;
#pragma unroll
            for (int kt = 0; kt < KT1; ++kt)
#pragma unroll
                for (int kc = 0; kc < 4; ++kc)
                    acc1 = mma16(frag(Cfg::OFF_W3 + kt * (64 * 128), wn * 32 + lr, kc),
                                 frag(Cfg::OFF_H2 + kt * (BM * 128), wm * 32 + lr, kc), acc1);
            // (b') SC: the shortcut chunk itself, D[channel][pixel] in the same lane layout
            [[maybe_unused]] f32x16 accs;
            if constexpr (SC) {
#pragma unroll
                for (int r = 0; r < 16; ++r) accs[r] = 0.f;
#pragma unroll
                for (int kc = 0; kc < 4; ++kc)
                    accs = mma16(frag(Cfg::OFF_WS, wn * 32 + lr, kc), frag(Cfg::OFF_XP, wm * 32 + lr, kc), accs);
            }
            // (c) + bias + shortcut, rounded to bf16, in place (lane: 4 consecutive channels x 4 groups of its pixel)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int cl = wn * 32 + 8 * g + 4 * lh;             // channel inside the chunk
                char* p = smem + pcur + prow + ((((cl >> 3)) ^ fsw) << 4) + 8 * lh;
                const f32x4 s4 = *(const f32x4*)(sScale3 + nc * 64 + cl), b4 = *(const f32x4*)(sBias3 + nc * 64 + cl);
                unsigned r01, r23;
                if constexpr (SC) {      // what the separate shortcut launch stores: bf16(acc + bias)
                    const f32x4 c4 = *(const f32x4*)(sBiasSc + nc * 64 + cl);
                    r01 = pack_bf(accs[4 * g + 0] + c4[0], accs[4 * g + 1] + c4[1]);
                    r23 = pack_bf(accs[4 * g + 2] + c4[2], accs[4 * g + 3] + c4[3]);
                } else {
                    const unsigned long long rr = *(const unsigned long long*)p;
                    r01 = (unsigned)rr; r23 = (unsigned)(rr >> 32);
                }
                // the epilogue arithmetic of gemm_conv.hip, rounding for rounding: fma(acc, scale, shift) + shortcut
                // (scale3 absent -> 1.0f: fma(acc, 1, b) == acc + b exactly)
                float v0 = fmaf(acc1[4 * g + 0], s4[0], b4[0]), v1 = fmaf(acc1[4 * g + 1], s4[1], b4[1]);
                float v2 = fmaf(acc1[4 * g + 2], s4[2], b4[2]), v3 = fmaf(acc1[4 * g + 3], s4[3], b4[3]);
                v0 += bf_lo(r01); v1 += bf_hi(r01); v2 += bf_lo(r23); v3 += bf_hi(r23);
                const unsigned long long o = (unsigned long long)pack_bf(v0, v1) | ((unsigned long long)pack_bf(v2, v3) << 32);
                *(unsigned long long*)p = o;
            }
        }
        __syncthreads();
        // (e) stream the trunk chunk out and pre-activate it in place; stage the next shortcut / W3 chunks
        {
            const f32x4 s0 = *(const f32x4*)(sPreS + nc * 64 + lslot * 8), s1 = *(const f32x4*)(sPreS + nc * 64 + lslot * 8 + 4);
            const f32x4 b0 = *(const f32x4*)(sPreB + nc * 64 + lslot * 8), b1 = *(const f32x4*)(sPreB + nc * 64 + lslot * 8 + 4);
#pragma unroll
            for (int p = 0; p < RB; ++p) {
                char* q = smem + pcur + st_off + p * (64 * 128);
                const u32x4 x = *(const u32x4*)q;
                if (rok[p] && (PH2 || a.out)) *(u32x4*)(a.out + (long long)(m0 + r0 + 64 * p) * depth + nc * 64 + lslot * 8) = x;
                if constexpr (!PH2) { if (!a.out_pre) continue; }
                u32x4 y;
                y[0] = pack_bf(fmaxf(fmaf(bf_lo(x[0]), s0[0], b0[0]), 0.f), fmaxf(fmaf(bf_hi(x[0]), s0[1], b0[1]), 0.f));
                y[1] = pack_bf(fmaxf(fmaf(bf_lo(x[1]), s0[2], b0[2]), 0.f), fmaxf(fmaf(bf_hi(x[1]), s0[3], b0[3]), 0.f));
                y[2] = pack_bf(fmaxf(fmaf(bf_lo(x[2]), s1[0], b1[0]), 0.f), fmaxf(fmaf(bf_hi(x[2]), s1[1], b1[1]), 0.f));
                y[3] = pack_bf(fmaxf(fmaf(bf_lo(x[3]), s1[2], b1[2]), 0.f), fmaxf(fmaf(bf_hi(x[3]), s1[3], b1[3]), 0.f));
                if constexpr (PH2) *(u32x4*)q = y;
                else if (rok[p]) *(u32x4*)(a.out_pre + (long long)(m0 + r0 + 64 * p) * depth + nc * 64 + lslot * 8) = y;
            }
            if (nc + 1 < NCH) {
                if constexpr (SC) {
                    *(u32x4*)(smem + Cfg::OFF_WS + st_off) = rws;
                    if (nc + 2 < NCH) rws = load_ws(nc + 2);
                } else {
                    store_rows(pnxt, rres[(nc + 1) & 1]);
                }
                store_w3(rw3);
                if (nc + 2 < NCH) load_w3(nc + 2, rw3);
            }
        }
        __syncthreads();
        if constexpr (PH2) {
            // (g) conv1' K step `nc`
#pragma unroll
            for (int kc = 0; kc < 4; ++kc) acc2 = mma16(frag(Cfg::OFF_W1, wn * 32 + lr, kc), frag(pcur, wm * 32 + lr, kc), acc2);
            __syncthreads();
            if (nc + 1 < NCH) {
                store_w1(rw1);
                if (nc + 2 < NCH) load_w1(nc + 2, rw1);
            }
        }
    }
    if constexpr (!PH2) return;

    // ---- conv1' epilogue: BN (+ReLU), bf16, through the H2 region ([BM][64] sub-tiles), coalesced stores
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int n2 = wn * 32 + 8 * g + 4 * lh;
        const f32x4 s4 = *(const f32x4*)(a.scale1 + n2), b4 = *(const f32x4*)(a.shift1 + n2);
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            v[j] = fmaf(acc2[4 * g + j], s4[j], b4[j]);
            if (a.relu1) v[j] = fmaxf(v[j], 0.f);
        }
        const int cl = n2 & 63;
        char* p = smem + Cfg::OFF_H2 + (n2 >> 6) * (BM * 128) + prow + (((cl >> 3) ^ fsw) << 4) + 8 * lh;
        *(unsigned long long*)p = (unsigned long long)pack_bf(v[0], v[1]) | ((unsigned long long)pack_bf(v[2], v[3]) << 32);
    }
    __syncthreads();
#pragma unroll
    for (int st = 0; st < N2 / 64; ++st)
#pragma unroll
        for (int p = 0; p < RB; ++p)
            if (rok[p])
                *(u32x4*)(a.out_h1 + (long long)(m0 + r0 + 64 * p) * N2 + st * 64 + lslot * 8) =
                    *(const u32x4*)(smem + Cfg::OFF_H2 + st * (BM * 128) + st_off + p * (64 * 128));
}

template <int BM, int CM, int NCH, int N2, bool CONV2, bool SC = false, bool PH2 = true>
int launch_tail(const TailArgs& a, hipStream_t stream) {
    typedef TailCfg<BM, CM, NCH, N2, SC> Cfg;
    static_assert(!CONV2 || 2 * (BM * 128 + CM * 128) <= Cfg::OFF_C, "conv2 stages must fit below the constants");
    auto kern = bottleneck_tail_kernel<BM, CM, NCH, N2, CONV2, SC, PH2>;
    static DeviceOnce once;              // per kernel instantiation, per device
    if (const unsigned long long bit = once.due()) {
        HMMR_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS));
        once.mark(bit);
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)((a.M + BM - 1) / BM)), dim3(NT), Cfg::LDS, stream, a);
    HMMR_CHECK_HIP(hipGetLastError());
    return 0;
}

}  // namespace

int hmmr_bottleneck_tail_split(const hmmr_tail_desc_t* d, hipStream_t stream);      // bottleneck_split.hip

extern "C" int hmmr_bottleneck_tail(const hmmr_tail_desc_t* d, void* stream) {
    HMMR_REQUIRE(d && (d->h2 || d->h1) && (d->w3 || d->pair_stream || d->unit_stream) && (d->res || d->xp), "hmmr_bottleneck_tail: null argument");
    if (d->dtype == HMMR_F16X3) return hmmr_bottleneck_tail_split(d, (hipStream_t)stream);
    const bool ph2 = d->w1 != nullptr;
    HMMR_REQUIRE(!ph2 || (d->out && d->pre_scale && d->pre_shift && d->scale1 && d->shift1 && d->out_h1 && !d->out_pre),
                 "hmmr_bottleneck_tail: the next conv1 needs out, pre_scale/pre_shift, scale1/shift1, out_h1 (and no out_pre)");
    HMMR_REQUIRE(ph2 || (d->h1 && !d->xp && (d->out || d->out_pre) && (!d->out_pre || (d->pre_scale && d->pre_shift))),
                 "hmmr_bottleneck_tail: without a next conv1 (w1 == NULL) the launch needs conv2 in front (h1), a loaded "
                 "shortcut, and out and/or out_pre (+ pre_scale/pre_shift)");
    const bool conv2 = d->h1 != nullptr;
    HMMR_REQUIRE(!conv2 || (!d->h2 && d->w2 && d->scale2 && d->shift2 && d->hin > 0 && d->win > 0 &&
                            d->ho > 0 && d->wo > 0 && d->m % (d->ho * d->wo) == 0 &&
                            (d->conv2_stride <= 1 ? (d->ho == d->hin && d->wo == d->win)
                                                  : (d->conv2_stride == 2 && d->ho == (d->hin + 1) / 2 && d->wo == (d->win + 1) / 2))),
                 "hmmr_bottleneck_tail: conv2 in front needs h1, w2, scale2, shift2, hin, win, ho, wo (stride 1 or 2; h2 == NULL)");
    HMMR_REQUIRE(d->dtype == HMMR_BF16, "hmmr_bottleneck_tail: bf16 operands only");
    const bool b1 = d->c_mid == 64 && d->depth == 256 && (!ph2 || d->n2 == 64);
    const bool b2 = d->c_mid == 128 && d->depth == 512 && (!ph2 || d->n2 == 128);
    HMMR_REQUIRE(b1 || b2, "hmmr_bottleneck_tail: supported shapes are 64 -> 256 -> 64 and 128 -> 512 -> 128 (got %d, %d, %d)",
                 d->c_mid, d->depth, d->n2);
    HMMR_REQUIRE(d->m > 0, "hmmr_bottleneck_tail: empty launch");
    const bool sc = d->xp != nullptr;
    HMMR_REQUIRE(!sc || (conv2 && !d->res && d->wsc && d->c_mid == 64 && d->depth == 256 && d->n2 == 64),
                 "hmmr_bottleneck_tail: the in-kernel conv shortcut (xp, wsc) needs conv2 in front, res == NULL and the "
                 "64 -> 256 -> 64 shape with a 64-channel shortcut input");
    HMMR_REQUIRE(sc || d->res_strided || d->ldr >= d->depth, "hmmr_bottleneck_tail: residual row stride < depth");
    HMMR_REQUIRE(!d->res_strided || (d->ho > 0 && d->wo > 0), "hmmr_bottleneck_tail: strided residual needs ho, wo");
    TailArgs a;
    a.h2 = (const bf16_t*)d->h2; a.w3 = (const bf16_t*)d->w3; a.scale3 = d->scale3; a.shift3 = d->shift3;
    a.res = (const bf16_t*)d->res; a.out = (bf16_t*)d->out; a.pre_scale = d->pre_scale; a.pre_shift = d->pre_shift;
    a.w1 = (const bf16_t*)d->w1; a.scale1 = d->scale1; a.shift1 = d->shift1; a.out_h1 = (bf16_t*)d->out_h1;
    a.M = d->m; a.depth = d->depth; a.ldr = d->ldr; a.res_strided = d->res_strided;
    a.Wo = d->wo > 0 ? d->wo : 1; a.HoWo = d->ho > 0 ? d->ho * d->wo : 1;
    a.res_img_stride = d->res_img_stride; a.res_row_stride = d->res_row_stride; a.res_px_stride = d->res_px_stride;
    a.relu1 = d->relu1;
    a.h1 = (const bf16_t*)d->h1; a.w2 = (const bf16_t*)d->w2; a.scale2 = d->scale2; a.shift2 = d->shift2;
    a.H = d->hin; a.W = d->win; a.S = d->conv2_stride > 1 ? d->conv2_stride : 1; a.out_pre = (bf16_t*)d->out_pre;
    if (conv2 && d->ho > 0) { a.Wo = d->wo; a.HoWo = d->ho * d->wo; }
    if (!ph2)
        return b1 ? launch_tail<128, 64, 4, 64, true, false, false>(a, (hipStream_t)stream)
                  : launch_tail<64, 128, 8, 128, true, false, false>(a, (hipStream_t)stream);
    a.xp = (const bf16_t*)d->xp; a.wsc = (const bf16_t*)d->wsc; a.shift_sc = d->shift_sc;
    if (sc) return launch_tail<128, 64, 4, 64, true, true>(a, (hipStream_t)stream);
    if (conv2)
        return b1 ? launch_tail<128, 64, 4, 64, true>(a, (hipStream_t)stream)
                  : launch_tail<64, 128, 8, 128, true>(a, (hipStream_t)stream);
    return b1 ? launch_tail<128, 64, 4, 64, false>(a, (hipStream_t)stream)
              : launch_tail<64, 128, 8, 128, false>(a, (hipStream_t)stream);
}
