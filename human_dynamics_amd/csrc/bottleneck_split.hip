// Fused tail of a ResNet-v2 bottleneck unit for gfx950, f16x3 ("split") operands:
//
//     trunk' = conv3({h2 [, xp]}) + bias [+ shortcut]               (1x1; bottleneck_v2 `conv3` + add; with xp the unit's
//                                                                     conv shortcut is folded into the same GEMM)
//     h1'    = relu(bn1'(conv1'(relu(bn_pre'(trunk')))))            (the NEXT unit's `preact` + `conv1`)
//
// in ONE kernel (slim resnet_v2.bottleneck as invoked at src/models.py:65-75): the next conv1 does not
// re-read the trunk (the widest tensor of the unit, 4 B/element in this mode) from HBM.  Same algorithm as
// bottleneck.hip (bf16): a workgroup owns BM pixels and walks conv3's output channels in chunks of 64, which are the K
// steps of conv1'; MFMA operands are swapped (weights = A, activations = B) so a lane owns 4 consecutive channels of
// one pixel and every read-modify-write is lane-local.  What differs for split operands:
//   * every LDS tile is a PAIR of fp16 planes (hi, lo), each in the [rows][64 halves] / XOR-swizzled geometry of
//     gemm_conv.hip; a product is three MFMAs (w.hi*x.lo, w.lo*x.hi, w.hi*x.hi: gemm_conv.hip's order with the
//     operands swapped, so the sums are bit-identical to the separate launches);
//   * the filters never enter LDS: the packer stores them FRAGMENT-MAJOR ([32-row block][16-wide K chunk][lane]
//     [hi 16 B | lo 16 B]), a wave's A operand is one coalesced 2 KB read from L2 per (block, chunk) -- that keeps the
//     workgroup at 36-52 KB of LDS (3-4 per CU) where bottleneck.hip's layout would need 132 KB;
//   * workgroups are 4 waves on 64 pixels (168-VGPR budget at three waves per SIMD).
// Global tensors keep the interleaved split layout ([hi8][lo8] per 8 channels): a 64-channel chunk of a row is 256
// contiguous bytes = 16 slots, staged by 16 lanes (even slots -> hi plane, odd -> lo plane).
#include "common.h"
#include "hmmr_hip.h"

namespace {

struct SplitTailArgs {
    const bsplit_t* src[2]; int src_ld[2];     // conv3's K, 64 channels at a time: rows of src[i] with stride src_ld[i] elements
    const char* w3f;                            // fragment-major [depth / 32][KS * 4][64 lanes][32 B]
    const float* scale3; const float* shift3;
    const bsplit_t* res; int ldr;               // RES: the shortcut, rows of ldr elements
    bsplit_t* out;                              // [M][depth]
    const float* pre_scale; const float* pre_shift;
    const char* w1f;                            // fragment-major [N2 / 32][depth / 16][64 lanes][32 B]
    const float* scale1; const float* shift1; int relu1;
    bsplit_t* out_h1;                           // [M][N2]
    int M;
    // CONV2: the unit's 3x3 conv2 (stride 1, SAME) runs in front, over h1 [n][H][W][64]; src[0] is not read
    const bsplit_t* h1; int H, W;               // H, W multiples of 8: a workgroup owns an 8 x 8 pixel tile
    const char* w2f;                            // conv2 filters packed [64+][9 * 64] like every hmmr_layer_t.w, K = (ky, kx, ci)
    const float* scale2; const float* shift2;
};

__device__ __forceinline__ float bf_lo(unsigned v) { return shalf_lo(v); }
__device__ __forceinline__ float bf_hi(unsigned v) { return shalf_hi(v); }

// KS: 64-channel K sub-tiles of conv3 (1: h2 of block 1; 2: h2 of block 2, or block 1's {h2, xp} with the shortcut folded)
// NCH = depth / 64, N2 = conv1' output channels, RES: a shortcut tensor is added (false: it is folded into conv3's K)
// Block 1 (NCH 4) fits the 168-VGPR budget of three workgroups per CU (0.45 -> 0.38 ms per launch); block 2's longer K and two
// conv1' blocks per wave do not (spills, 0.32 -> 0.41 ms), so it runs two per CU.
// CONV2 (block 1): the workgroup's 64 pixels are an 8 x 8 tile of one image; its 10 x 10 h1 patch (zeros outside the
// image) sits in the P region while conv2 runs (9 taps read as shifted fragments of the patch, the tap's filters staged
// in the H2 tile region), and conv2's BN + ReLU output becomes the H2 tile: h2 never exists in HBM.
template <int KS, int NCH, int N2, bool RES, bool CONV2>
__global__ __launch_bounds__(256, NCH > 4 ? 2 : 3) void tail_split_kernel(const SplitTailArgs a) {
    constexpr int BM = 64, NT = 256, depth = NCH * 64;
    constexpr int PLANE = BM * 128;                  // one fp16 plane of a [64 rows][64 channels] tile
    constexpr int PPLANE = 104 * 128;                // one plane of the 10 x 10 pixel patch (CONV2)
    constexpr int OFF_H2 = 0;                        // KS tiles x (hi, lo); later the conv1' output tiles
    // CONV2: only K tile 0 (h2, produced in the launch) sits in the H2 region -- which holds the conv2 filter tap while
    // conv2 runs -- and a second K tile (the folded shortcut's operand) follows the P planes inside the patch region
    constexpr int NH2 = CONV2 ? 1 : KS;
    constexpr int OFF_P = OFF_H2 + NH2 * 2 * PLANE;  // the trunk chunk (hi, lo); CONV2: first the h1 patch
    constexpr int XREG = CONV2 ? (2 * PPLANE > KS * 2 * PLANE ? 2 * PPLANE : KS * 2 * PLANE) : 2 * PLANE;
    constexpr int OFF_C = OFF_P + XREG;              // 4 x depth floats
    auto ktile = [&](int ks) { return (CONV2 && ks > 0) ? OFF_P + ks * 2 * PLANE : OFF_H2 + ks * 2 * PLANE; };
    static_assert(!CONV2 || N2 == 64, "conv2 in front is written for block 1");
    constexpr int J2 = N2 / 64;                      // conv1' 32-row blocks per wave
    static_assert(J2 <= (CONV2 ? 1 : KS), "the conv1' output tiles reuse the H2 region");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m0 = blockIdx.x * BM;

    // ---- staging geometry: 16 lanes per 256-byte row chunk (slot s: even = hi half, odd = lo half of group s >> 1),
    // 16 rows per pass, 4 passes
    const int s = tid & 15, rr = tid >> 4;
    const int plane = s & 1, L = s >> 1;
    auto slot_of = [&](int tile_off, int row) -> char* {
        return smem + tile_off + plane * PLANE + row * 128 + ((L ^ ((row >> 1) & 7)) << 4);
    };
    // tile pixel p -> row of the [M][..] tensors: 64 consecutive rows, or (CONV2) pixel (p >> 3, p & 7) of an 8 x 8 image tile
    int img = 0, ty = 0, tx = 0;
    if constexpr (CONV2) {
        const int tw = a.W >> 3, tpi = (a.H >> 3) * tw;
        img = blockIdx.x / tpi;
        const int t = blockIdx.x - img * tpi;
        ty = t / tw; tx = t - ty * tw;
    }
    bool rok[4]; long long grow[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int pix = rr + 16 * p;
        const int m = CONV2 ? ((img * a.H + 8 * ty + (pix >> 3)) * a.W + 8 * tx + (pix & 7)) : m0 + pix;
        rok[p] = m < a.M;
        grow[p] = rok[p] ? m : 0;                    // tail rows read row 0 and are never stored
    }

    unsigned long long satmask = 0ull;               // lanes of this wave that split a value beyond the fp16 range (common.h: hmmr_run_flags)
    float* sScale3 = (float*)(smem + OFF_C);
    float* sBias3 = sScale3 + depth;
    float* sPreS = sBias3 + depth;
    float* sPreB = sPreS + depth;
    for (int i = tid; i < depth; i += NT) {
        sScale3[i] = a.scale3 ? a.scale3[i] : 1.0f;
        sBias3[i] = a.shift3 ? a.shift3[i] : 0.0f;
        sPreS[i] = a.pre_scale[i];
        sPreB[i] = a.pre_shift[i];
    }

    // ---- fragment geometry: wave (wn, wm) owns channels [32 wn, +32) of a chunk x pixels [32 wm, +32)
    const int wn = wave >> 1, wm = wave & 1;
    const int lr = lane & 31, lh = lane >> 5;
    const int fsw = (lr >> 1) & 7;
    const int prow = (wm * 32 + lr) * 128;
    auto xfrag = [&](int plane_off, int kc) {
        return *(const shalf8*)(smem + plane_off + prow + (((2 * kc + lh) ^ fsw) << 4));
    };
    // Filter fragments straight from L2, requested a whole chunk before their use.  ALLW (block 2: 256-VGPR budget of two
    // workgroups per CU): every fragment of a chunk (KS x 4 of conv3, J2 x 4 of conv1') is held at once; otherwise four of
    // each, and the second K tile / row block is requested mid-chunk (an exposed L2 round trip that three workgroups per
    // CU cover in block 1 -- and that took 17 k cycles per chunk in block 2 before ALLW).
    constexpr bool ALLW = NCH > 4;
    constexpr int G3 = ALLW ? KS : 1, G1 = ALLW ? J2 : 1;
    wfrag w3[G3 * 4], w1[G1 * 4];
    auto load_w3 = [&](int nc, int ks) {             // row block 2 nc + wn, K chunks 4 ks .. 4 ks + 3
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) {
            const shalf8* p = (const shalf8*)(a.w3f + ((long long)((2 * nc + wn) * (KS * 4) + ks * 4 + kc) * 64 + lane) * 32);
            w3[(ALLW ? ks * 4 : 0) + kc].hi = p[0]; w3[(ALLW ? ks * 4 : 0) + kc].lo = p[1];
        }
    };
    auto load_w1 = [&](int nc, int j) {              // row block wn * J2 + j, K chunks 4 nc .. 4 nc + 3
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) {
            const shalf8* p = (const shalf8*)(a.w1f + ((long long)((wn * J2 + j) * (depth / 16) + 4 * nc + kc) * 64 + lane) * 32);
            w1[(ALLW ? j * 4 : 0) + kc].hi = p[0]; w1[(ALLW ? j * 4 : 0) + kc].lo = p[1];
        }
    };
    auto load_w3_chunk = [&](int nc) {
#pragma unroll
        for (int ks = 0; ks < G3; ++ks) load_w3(nc, ks);
    };
    auto load_w1_chunk = [&](int nc) {
#pragma unroll
        for (int j = 0; j < G1; ++j) load_w1(nc, j);
    };
    constexpr int AHEAD = 2;                         // shortcut chunks in flight
    u32x4 rres[AHEAD][RES ? 4 : 1];
    auto load_res = [&](int nc, u32x4 (&r)[RES ? 4 : 1]) {
        if constexpr (RES) {
#pragma unroll
            for (int p = 0; p < 4; ++p) r[p] = *(const u32x4*)(a.res + grow[p] * a.ldr + nc * 64 + s * 4);
        }
    };

    // ---- prologue
    load_res(0, rres[0]);
    if (AHEAD > 1 && NCH > 1) load_res(1, rres[AHEAD - 1]);
    if constexpr (CONV2) {
        // the 10 x 10 h1 patch -> LDS planes (16 lanes per pixel, 7 passes; zeros outside the image)
#pragma unroll
        for (int pass = 0; pass < 7; ++pass) {
            const int pix = rr + 16 * pass;
            if (pix < 100) {
                const int py = pix / 10, px = pix - 10 * py;
                const int iy = 8 * ty - 1 + py, ix = 8 * tx - 1 + px;
                u32x4 v = {0u, 0u, 0u, 0u};
                if ((unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W)
                    v = *(const u32x4*)(a.h1 + ((long long)(img * a.H + iy) * a.W + ix) * 64 + s * 4);
                *(u32x4*)(smem + OFF_P + plane * PPLANE + pix * 128 + ((L ^ ((pix >> 1) & 7)) << 4)) = v;
            }
        }
        // conv2's filters, one tap ([64 out][64 in] = a [64 rows][256 B] tile like every other) at a time: the tile of
        // tap t+1 is requested into registers before the MFMAs of tap t and written to LDS behind them -- 16 KB per tap and
        // workgroup through the address path instead of 32 KB of per-wave fragments (which made it the limiter)
        const bsplit_t* w2 = (const bsplit_t*)a.w2f;                  // packed [64+][9 * 64], K = (ky, kx, ci)
        u32x4 rw2[4];
        auto load_w2 = [&](int tap) {
#pragma unroll
            for (int p = 0; p < 4; ++p) rw2[p] = *(const u32x4*)(w2 + (long long)(rr + 16 * p) * (9 * 64) + tap * 64 + s * 4);
        };
        auto store_w2 = [&]() {
#pragma unroll
            for (int p = 0; p < 4; ++p) *(u32x4*)slot_of(OFF_H2, rr + 16 * p) = rw2[p];
        };
        load_w2(0);
        store_w2();
        __syncthreads();
        const int pq = wm * 32 + lr;                                  // this lane's pixel of the tile
        const int brow = (pq >> 3) * 10 + (pq & 7);                   // its patch row for tap (0, 0)
        const int wrow = (wn * 32 + lr) * 128;                        // this lane's filter row of the tap tile
        f32x16 acc0;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc0[r] = 0.f;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            if (tap + 1 < 9) load_w2(tap + 1);
            const int row = brow + (tap / 3) * 10 + tap % 3;
            const char* ph = smem + OFF_P + row * 128;
            const int sw = (row >> 1) & 7;
#pragma unroll
            for (int kc = 0; kc < 4; ++kc) {
                wfrag w;
                w.hi = *(const shalf8*)(smem + OFF_H2 + wrow + (((2 * kc + lh) ^ fsw) << 4));
                w.lo = *(const shalf8*)(smem + OFF_H2 + PLANE + wrow + (((2 * kc + lh) ^ fsw) << 4));
                const shalf8 xh = *(const shalf8*)(ph + (((2 * kc + lh) ^ sw) << 4));
                const shalf8 xl = *(const shalf8*)(ph + PPLANE + (((2 * kc + lh) ^ sw) << 4));
                acc0 = mma3(w, xh, xl, acc0);
            }
            __syncthreads();                                          // every wave is done with this tap's filters
            if (tap + 1 < 9) {
                store_w2();
                __syncthreads();
            }
        }
        // BN + ReLU, split -> the H2 tile (lane: 4 consecutive channels x 4 groups of its pixel)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int n = wn * 32 + 8 * g + 4 * lh;
            const f32x4 s4 = *(const f32x4*)(a.scale2 + n), b4 = *(const f32x4*)(a.shift2 + n);
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = fmaxf(fmaf(acc0[4 * g + j], s4[j], b4[j]), 0.f);
            char* q = smem + OFF_H2 + prow + (((n >> 3) ^ fsw) << 4) + 8 * lh;
            unsigned long long oh, ol;
            split4(v, oh, ol, satmask);
            *(unsigned long long*)q = oh;
            *(unsigned long long*)(q + PLANE) = ol;
        }
        __syncthreads();                                              // every wave is done with the patch: P is free
    }
    load_w3_chunk(0);
    load_w1_chunk(0);
#pragma unroll
    for (int ks = CONV2 ? 1 : 0; ks < KS; ++ks) {
        u32x4 v[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) v[p] = *(const u32x4*)(a.src[ks] + grow[p] * a.src_ld[ks] + s * 4);
#pragma unroll
        for (int p = 0; p < 4; ++p) *(u32x4*)slot_of(ktile(ks), rr + 16 * p) = v[p];
    }

    f32x16 acc2[J2];
#pragma unroll
    for (int j = 0; j < J2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[j][r] = 0.f;

#pragma unroll
    for (int nc = 0; nc < NCH; ++nc) {
        // (1) the shortcut chunk -> the P planes (pre-fill for the read-modify-write below)
        if constexpr (RES) {
#pragma unroll
            for (int p = 0; p < 4; ++p) *(u32x4*)slot_of(OFF_P, rr + 16 * p) = rres[nc % AHEAD][p];
        }
        __syncthreads();
        // (2) conv3 chunk: D[channel][pixel]
        f32x16 acc1;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc1[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            if (!ALLW && ks > 0) { __builtin_amdgcn_sched_barrier(0); load_w3(nc, ks); }
#pragma unroll
            for (int kc = 0; kc < 4; ++kc)
                acc1 = mma3(w3[(ALLW ? ks * 4 : 0) + kc], xfrag(ktile(ks), kc), xfrag(ktile(ks) + PLANE, kc), acc1);
        }
        __builtin_amdgcn_sched_barrier(0);               // (the reload must not be hoisted above the MFMAs: it would double the live fragment registers)
        if (nc + 1 < NCH) load_w3_chunk(nc + 1);
        // + bias (+ shortcut), split, in place: a lane owns 4 consecutive channels x 4 groups of its pixel
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int cl = wn * 32 + 8 * g + 4 * lh;                 // channel inside the chunk
            char* ph = smem + OFF_P + prow + (((cl >> 3) ^ fsw) << 4) + 8 * lh;
            const f32x4 s4 = *(const f32x4*)(sScale3 + nc * 64 + cl), b4 = *(const f32x4*)(sBias3 + nc * 64 + cl);
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = fmaf(acc1[4 * g + j], s4[j], b4[j]);    // (scale3 absent -> 1.0f: == acc + b)
            if constexpr (RES) {
                const unsigned long long h = *(const unsigned long long*)ph, l = *(const unsigned long long*)(ph + PLANE);
                v[0] += bf_lo((unsigned)h) + bf_lo((unsigned)l);
                v[1] += bf_hi((unsigned)h) + bf_hi((unsigned)l);
                v[2] += bf_lo((unsigned)(h >> 32)) + bf_lo((unsigned)(l >> 32));
                v[3] += bf_hi((unsigned)(h >> 32)) + bf_hi((unsigned)(l >> 32));
            }
            unsigned long long oh, ol;
            split4(v, oh, ol, satmask);
            *(unsigned long long*)ph = oh;
            *(unsigned long long*)(ph + PLANE) = ol;
        }
        __syncthreads();
        // (3) stream the trunk chunk out and pre-activate it in place (hi / lo lanes of a group are neighbours)
        {
            const int ch = nc * 64 + L * 8 + 4 * plane;              // this lane's four channels (preact_slot_split's convention)
            const f32x4 ps = *(const f32x4*)(sPreS + ch), pb = *(const f32x4*)(sPreB + ch);
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                char* q = slot_of(OFF_P, rr + 16 * p);
                const u32x4 x = *(const u32x4*)q;
                if (rok[p]) *(u32x4*)(a.out + grow[p] * depth + nc * 64 + s * 4) = x;
                *(u32x4*)q = preact_slot_split(x, ps, pb, plane);
            }
        }
        __syncthreads();
        // (4) conv1' K step `nc`
#pragma unroll
        for (int j = 0; j < J2; ++j) {
            if (!ALLW && j > 0) { __builtin_amdgcn_sched_barrier(0); load_w1(nc, j); }
#pragma unroll
            for (int kc = 0; kc < 4; ++kc) acc2[j] = mma3(w1[(ALLW ? j * 4 : 0) + kc], xfrag(OFF_P, kc), xfrag(OFF_P + PLANE, kc), acc2[j]);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (nc + 1 < NCH) load_w1_chunk(nc + 1);
        // the shortcut chunk two ahead is requested BEHIND the next chunk's filter fragments: a wave's loads return in order, and an
        // L2-resident fragment issued behind this HBM request would only arrive with it (every chunk then waits an HBM round trip)
        if constexpr (RES) {
            __builtin_amdgcn_sched_barrier(0);
            if (nc + AHEAD < NCH) load_res(nc + AHEAD, rres[nc % AHEAD]);
        }
        __syncthreads();                                             // P is free for the next chunk's pre-fill
    }

    // ---- conv1' epilogue: BN (+ ReLU), split, through the H2 region, coalesced stores
#pragma unroll
    for (int j = 0; j < J2; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int n2 = (wn * J2 + j) * 32 + 8 * g + 4 * lh;
            const f32x4 s4 = *(const f32x4*)(a.scale1 + n2), b4 = *(const f32x4*)(a.shift1 + n2);
            float v[4];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                v[jj] = fmaf(acc2[j][4 * g + jj], s4[jj], b4[jj]);
                if (a.relu1) v[jj] = fmaxf(v[jj], 0.f);
            }
            char* ph = smem + OFF_H2 + (n2 >> 6) * 2 * PLANE + prow + ((((n2 & 63) >> 3) ^ fsw) << 4) + 8 * lh;
            unsigned long long oh, ol;
            split4(v, oh, ol, satmask);
            *(unsigned long long*)ph = oh;
            *(unsigned long long*)(ph + PLANE) = ol;
        }
    split_flag(satmask != 0ull);
    __syncthreads();
#pragma unroll
    for (int t = 0; t < J2; ++t)
#pragma unroll
        for (int p = 0; p < 4; ++p)
            if (rok[p])
                *(u32x4*)(a.out_h1 + grow[p] * N2 + t * 64 + s * 4) = *(const u32x4*)slot_of(OFF_H2 + t * 2 * PLANE, rr + 16 * p);
}

template <int KS, int NCH, int N2, bool RES, bool CONV2 = false>
int launch_split_tail(const SplitTailArgs& a, hipStream_t stream) {
    constexpr int xreg = CONV2 ? (2 * 104 * 128 > KS * 2 * 64 * 128 ? 2 * 104 * 128 : KS * 2 * 64 * 128) : 2 * 64 * 128;
    constexpr int lds = (CONV2 ? 1 : KS) * 2 * 64 * 128 + xreg + 4 * NCH * 64 * (int)sizeof(float);
    auto kern = tail_split_kernel<KS, NCH, N2, RES, CONV2>;
    static DeviceOnce once;
    if (const unsigned long long bit = once.due()) {
        HMMR_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        once.mark(bit);
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)((a.M + 63) / 64)), dim3(256), lds, stream, a);     // (CONV2: M = images x H x W, 64 | H W)
    HMMR_CHECK_HIP(hipGetLastError());
    return 0;
}

}  // namespace

int hmmr_unit_pair_split(const hmmr_tail_desc_t* d, hipStream_t stream);       // unit_pair.hip
int hmmr_b1_unit_split(const hmmr_tail_desc_t* d, hipStream_t stream);         // b1_unit.hip

// hmmr_bottleneck_tail for HMMR_F16X3 (called from bottleneck.hip).  w3 / w1 are FRAGMENT-MAJOR here (hmmr_hip.h).
int hmmr_bottleneck_tail_split(const hmmr_tail_desc_t* d, hipStream_t stream) {
    if (d->pair_stream) { hmmr_count_launch(HMMR_COUNT_UNIT_PAIR); return hmmr_unit_pair_split(d, stream); }
    if (d->unit_stream) { hmmr_count_launch(HMMR_COUNT_B1_UNIT); return hmmr_b1_unit_split(d, stream); }
    hmmr_count_launch(HMMR_COUNT_TAIL_SPLIT);
    HMMR_REQUIRE((d->h2 != nullptr) != (d->h1 != nullptr) && d->w1 && d->out && d->out_h1 && d->pre_scale && d->pre_shift && d->scale1 &&
                 d->shift1 && !d->out_pre && !d->res_strided,
                 "hmmr_bottleneck_tail (f16x3): needs h2 or h1 (conv2 in front), the next conv1, out, out_h1 and a dense shortcut");
    const bool conv2 = d->h1 != nullptr;
    HMMR_REQUIRE(!conv2 || (d->w2 && d->scale2 && d->shift2 && d->conv2_stride <= 1 && d->hin > 0 && d->win > 0 && d->hin % 8 == 0 &&
                            d->win % 8 == 0 && d->m % (d->hin * d->win) == 0 && d->c_mid == 64 && d->depth == 256 && d->n2 == 64),
                 "hmmr_bottleneck_tail (f16x3): conv2 in front needs the 64 -> 256 -> 64 shape, stride 1, w2 (packed [cout][9 * c_mid] like every filter bank), "
                 "scale2, shift2 and an image grid that is a multiple of 8 x 8");
    const bool folded = d->xp != nullptr;             // {h2, xp} x [W3 | Wsc]: the conv shortcut inside conv3's K
    HMMR_REQUIRE(folded != (d->res != nullptr), "hmmr_bottleneck_tail (f16x3): either a shortcut tensor (res) or a folded one (xp)");
    SplitTailArgs a = {};
    a.w3f = (const char*)d->w3; a.scale3 = d->scale3; a.shift3 = d->shift3;
    a.res = (const bsplit_t*)d->res; a.ldr = d->ldr; a.out = (bsplit_t*)d->out;
    a.pre_scale = d->pre_scale; a.pre_shift = d->pre_shift;
    a.w1f = (const char*)d->w1; a.scale1 = d->scale1; a.shift1 = d->shift1; a.relu1 = d->relu1;
    a.out_h1 = (bsplit_t*)d->out_h1; a.M = d->m;
    HMMR_REQUIRE(d->m > 0, "hmmr_bottleneck_tail: empty launch");
    HMMR_REQUIRE(folded || d->ldr >= d->depth, "hmmr_bottleneck_tail: residual row stride < depth");
    if (d->c_mid == 64 && d->depth == 256 && d->n2 == 64) {
        a.src[0] = (const bsplit_t*)d->h2; a.src_ld[0] = 64;
        a.h1 = (const bsplit_t*)d->h1; a.H = d->hin; a.W = d->win; a.w2f = (const char*)d->w2; a.scale2 = d->scale2; a.shift2 = d->shift2;
        if (folded) {
            a.src[1] = (const bsplit_t*)d->xp; a.src_ld[1] = 64;
            return conv2 ? launch_split_tail<2, 4, 64, false, true>(a, stream) : launch_split_tail<2, 4, 64, false>(a, stream);
        }
        return conv2 ? launch_split_tail<1, 4, 64, true, true>(a, stream) : launch_split_tail<1, 4, 64, true>(a, stream);
    }
    if (d->c_mid == 128 && d->depth == 512 && d->n2 == 128 && !folded) {
        a.src[0] = (const bsplit_t*)d->h2; a.src_ld[0] = 128;
        a.src[1] = (const bsplit_t*)d->h2 + 64; a.src_ld[1] = 128;
        return launch_split_tail<2, 8, 128, true>(a, stream);
    }
    hmmr_set_error("hmmr_bottleneck_tail (f16x3): supported shapes are 64 -> 256 -> 64 (optionally with a folded 64-channel "
                   "shortcut) and 128 -> 512 -> 128 (got %d, %d, %d)", d->c_mid, d->depth, d->n2);
    return -1;
}
