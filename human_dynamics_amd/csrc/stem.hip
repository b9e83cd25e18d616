// Fused ResNet-v2-50 stem for gfx950: slim `conv2d_same(x, 64, 7, stride=2)` (explicit zero pad 3,
// VALID, + bias, no BN/ReLU) -> `max_pool2d([3,3], stride=2, padding=SAME)` (pad bottom/right only)
// -> block1/unit_1 `preact` BN + ReLU, in ONE kernel.  fp32 RGB images in, the
// pooled pre-activated [n,56,56,64] tensor out: the 112x112x64 conv map (1.6 MB/frame in bf16) and
// the re-packed RGBX image never touch HBM.
//
// One workgroup (4 waves) per 8x8 tile of POOLED pixels of one image:
//   1. the 39x39 input patch it needs is read from the fp32 image (zeros outside the image = the
//      explicit padding), converted to the operand type and stored in LDS as RGBX (8/16 B per pixel);
//   2. the 64 filters, packed [64][tap ky][8 px x RGBX] (kx = 7 and X are zero), go to LDS with
//      16-B-padded rows (conflict-free ds_read_b128 across 16 filters);
//   3. the 17x17 conv pixels of the tile are the rows of an implicit GEMM whose A fragments are read
//      STRAIGHT out of the patch: 8 consecutive k of tap ky = 2 neighbouring RGBX pixels = one
//      aligned 16-byte LDS read -- no im2col buffer.  M = 289 (10 MFMA tiles of 32), N = 64, K = 7x32;
//   4. conv + bias is staged in LDS in the activation type (the rounding point of the unfused path),
//      then each lane max-pools 3x3/2 windows of 8 channels, applies scale/shift + ReLU, and writes
//      16 B (bf16) / 32 B (fp32) of a 128/256-B pixel row.
#include <type_traits>
#include <utility>

#include "common.h"
#include "hmmr_hip.h"

namespace {
constexpr int IMG = 224, CONV = 112, POOL = 56, CO = 64;
constexpr int PT = 8;                 // pooled tile edge
constexpr int CT = 2 * PT + 1;        // conv tile edge (17)
constexpr int IP = 2 * CT + 5;        // input patch edge (39)
constexpr int IPW = 40;               // patch row stride in pixels
constexpr int NPIX = CT * CT;         // 289 conv pixels per tile
constexpr int MT = (NPIX + 31) / 32;  // 10 MFMA row tiles
constexpr int TAPK = 32;              // K elements per tap: 8 pixels x RGBX
constexpr int WK = 256;               // packed filter row in HBM: 8 taps x 32 (tap 7 is zero, unused here)

template <typename T> struct StemLds {
    static constexpr int PXB = 4 * (int)sizeof(T);                 // bytes per RGBX pixel
    static constexpr int PATCH = IP * IPW * PXB;
    static constexpr int WROW = 7 * TAPK * (int)sizeof(T) + 16;    // padded filter row
    static constexpr int WTS = CO * WROW;
    static constexpr int CST = MT * 32 * CO * (int)sizeof(T);      // conv staging (overlaps the filters)
    static constexpr int TOTAL = PATCH + (WTS > CST ? WTS : CST);
};

template <typename T> struct StemFrag;
template <> struct StemFrag<float> { typedef f32x4 type; };
template <> struct StemFrag<bf16_t> { typedef bf16x8 type; };

__device__ __forceinline__ f32x16 stem_mma(const f32x4& a, const f32x4& b, f32x16 c) {
#pragma unroll
    for (int j = 0; j < 4; ++j) c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[j], c, 0, 0, 0);
    return c;
}
__device__ __forceinline__ f32x16 stem_mma(const bf16x8& a, const bf16x8& b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ void store_rgbx(float* o, float r, float g, float b) {
    *(f32x4*)o = f32x4{r, g, b, 0.f};
}
__device__ __forceinline__ void store_rgbx(bf16_t* o, float r, float g, float b) {
    typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
    *(bf16x4*)o = bf16x4{(bf16_t)r, (bf16_t)g, (bf16_t)b, (bf16_t)0.f};
}

template <typename T>
__global__ __launch_bounds__(256) void stem_fused_kernel(const float* __restrict__ img, const T* __restrict__ wts,
                                                         const float* __restrict__ bias,
                                                         const float* __restrict__ pscale,
                                                         const float* __restrict__ pshift, T* __restrict__ out,
                                                         int n_real, const T* __restrict__ w1,
                                                         const float* __restrict__ s1, const float* __restrict__ b1,
                                                         T* __restrict__ out_h1) {
    typedef StemLds<T> LD;
    typedef typename StemFrag<T>::type frag_t;
    constexpr int CPT = (int)sizeof(T);           // 32-byte MFMA chunks per tap: 2 (bf16) / 4 (fp32)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* s_patch = smem;
    char* s_w = smem + LD::PATCH;
    T* s_c = (T*)(smem + LD::PATCH);              // reuses the filter region after the MFMAs

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ty = blockIdx.x / (POOL / PT), tx = blockIdx.x % (POOL / PT);
    const int n = blockIdx.y;
    const int cy0 = 2 * PT * ty, cx0 = 2 * PT * tx;          // first conv row/col of the tile
    const int iy0 = 2 * cy0 - 3, ix0 = 2 * cx0 - 3;          // first input row/col of the patch

    // ---- 1. input patch -> LDS (RGBX, operand type); zeros outside the image and for zero-tail images.
    // Work item = 4 pixels starting at a global column that is a multiple of 4: 12 floats = three
    // aligned 16-byte loads (image rows are 2688 B, a multiple of 16).  The patch starts at column
    // ix0 = 32*tx - 3, so groups 32*tx - 4 + 4*q (q = 0..10) cover it; a group is entirely inside or
    // outside the image (224 is a multiple of 4).
    const float* im = img + (long long)n * IMG * IMG * 3;
    for (int i = tid; i < IP * 11; i += 256) {
        const int py = i / 11, q = i - py * 11;
        const int gy = iy0 + py, gx = ix0 - 1 + 4 * q;
        float f[12];
#pragma unroll
        for (int e = 0; e < 12; ++e) f[e] = 0.f;
        if (n < n_real && (unsigned)gy < (unsigned)IMG && (unsigned)gx < (unsigned)IMG) {
            const f32x4* p = (const f32x4*)(im + (gy * IMG + gx) * 3);
            const f32x4 v0 = p[0], v1 = p[1], v2 = p[2];
#pragma unroll
            for (int e = 0; e < 4; ++e) { f[e] = v0[e]; f[4 + e] = v1[e]; f[8 + e] = v2[e]; }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int px = 4 * q - 1 + e;                     // patch column of this pixel
            if ((unsigned)px < (unsigned)IPW) {
                T* o = (T*)(s_patch + (py * IPW + px) * LD::PXB);
                store_rgbx(o, f[3 * e], f[3 * e + 1], f[3 * e + 2]);
            }
        }
    }
    // ---- 2. filters -> LDS, rows padded by 16 B
    constexpr int SLOTS_PER_ROW = 7 * TAPK * (int)sizeof(T) / 16;
    for (int i = tid; i < CO * SLOTS_PER_ROW; i += 256) {
        const int row = i / SLOTS_PER_ROW, sl = i - row * SLOTS_PER_ROW;
        *(u32x4*)(s_w + row * LD::WROW + sl * 16) = *(const u32x4*)((const char*)(wts + (long long)row * WK) + sl * 16);
    }
    __syncthreads();

    // ---- 3. implicit GEMM: wave w owns row tiles w, w+4, w+8 (all 64 output channels)
    const int lr = lane & 31, lh = lane >> 5;
    constexpr int MPW = (MT + 3) / 4;             // row tiles per wave (3)
    f32x16 acc[MPW][2];
#pragma unroll
    for (int i = 0; i < MPW; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    int abase[MPW];
#pragma unroll
    for (int i = 0; i < MPW; ++i) {
        int p = (wave + 4 * i) * 32 + lr;
        if (p >= NPIX) p = 0;                     // padding rows of the last tile: computed, never used
        const int cy = p / CT, cx = p - cy * CT;
        abase[i] = ((2 * cy) * IPW + 2 * cx) * LD::PXB + lh * 16;
    }
    const int bbase = lr * LD::WROW + lh * 16;
#pragma unroll 1
    for (int ky = 0; ky < 7; ++ky) {
#pragma unroll
        for (int c = 0; c < CPT; ++c) {
            const frag_t b0 = *(const frag_t*)(s_w + bbase + ky * TAPK * (int)sizeof(T) + c * 32);
            const frag_t b1 = *(const frag_t*)(s_w + bbase + 32 * LD::WROW + ky * TAPK * (int)sizeof(T) + c * 32);
#pragma unroll
            for (int i = 0; i < MPW; ++i) {
                if (wave + 4 * i < MT) {
                    const frag_t a = *(const frag_t*)(s_patch + abase[i] + ky * IPW * LD::PXB + c * 32);
                    acc[i][0] = stem_mma(a, b0, acc[i][0]);
                    acc[i][1] = stem_mma(a, b1, acc[i][1]);
                }
            }
        }
    }
    __syncthreads();                              // every wave is done with the filters: reuse as staging

    // ---- 4a. conv + bias -> LDS [pixel][64] in the activation type
#pragma unroll
    for (int i = 0; i < MPW; ++i) {
        if (wave + 4 * i < MT) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int ch = j * 32 + lr;
                const float bch = bias[ch];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int p = (wave + 4 * i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    s_c[p * CO + ch] = elem_traits<T>::from_f32(acc[i][j][r] + bch);
                }
            }
        }
    }
    __syncthreads();

    // ---- 4b. 3x3/2 max pool (TF SAME: rows/cols past 111 do not exist) + preact BN + ReLU
    for (int it = tid; it < PT * PT * (CO / 8); it += 256) {
        const int v8 = it & 7, pp = it >> 3;
        const int py = pp / PT, px = pp - py * PT;
        float m[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) m[j] = -3.0e38f;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            if (cy0 + 2 * py + dy >= CONV) continue;
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                if (cx0 + 2 * px + dx >= CONV) continue;
                float v[8];
                load8(s_c + ((2 * py + dy) * CT + 2 * px + dx) * CO + v8 * 8, v);
#pragma unroll
                for (int j = 0; j < 8; ++j) m[j] = fmaxf(m[j], v[j]);
            }
        }
        float sc[8], sh[8];
        load8(pscale + v8 * 8, sc); load8(pshift + v8 * 8, sh);
#pragma unroll
        for (int j = 0; j < 8; ++j) m[j] = fmaxf(fmaf(m[j], sc[j], sh[j]), 0.f);   // one explicit fma: = maxpool_bn_relu_kernel
        store8(out + (((long long)n * POOL + PT * ty + py) * POOL + PT * tx + px) * CO + v8 * 8, m);
        if constexpr (sizeof(T) == 2) {
            // block1/unit_1's conv1 follows in this launch: keep the tile as its operand image ([64 pixels][64 ch],
            // 128-B rows, 16-B slots XOR-swizzled like gemm_conv.hip) in the patch region, which is free by now
            if (out_h1) store8((T*)(s_patch + pp * 128 + ((v8 ^ ((pp >> 1) & 7)) << 4)), m);
        }
    }
    if constexpr (sizeof(T) == 2) {
        if (!out_h1) return;
        // ---- 5. conv1 of block1/unit_1 (1x1, 64 -> 64, BN + ReLU) on the tile just written: D[channel][pixel]
        // = W1 (A operand, fragments straight from L2: 8 KB, shared by every workgroup) x tile (B operand from LDS);
        // four 32x32 blocks, one per wave; same K order and epilogue arithmetic as the hmmr_conv_gemm launch
        __syncthreads();                              // tile complete; every read of s_c is done
        const int lr = lane & 31, lh = lane >> 5, fsw = (lr >> 1) & 7;
        const int wn = wave >> 1, wm = wave & 1;
        f32x16 acc1;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc1[r] = 0.f;
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) {
            const bf16x8 a = *(const bf16x8*)(w1 + (long long)(wn * 32 + lr) * 64 + kc * 16 + lh * 8);
            const bf16x8 b = *(const bf16x8*)(s_patch + (wm * 32 + lr) * 128 + (((2 * kc + lh) ^ fsw) << 4));
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc1, 0, 0, 0);
        }
        char* s_o = smem + LD::PATCH;                 // output tile [64 pixels][64 ch] in the (dead) conv staging region
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int cl = wn * 32 + 8 * g + 4 * lh;
            const f32x4 s4 = *(const f32x4*)(s1 + cl), b4 = *(const f32x4*)(b1 + cl);
            typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
            bf16x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = (bf16_t)fmaxf(fmaf(acc1[4 * g + j], s4[j], b4[j]), 0.f);
            *(bf16x4*)(s_o + (wm * 32 + lr) * 128 + (((cl >> 3) ^ fsw) << 4) + 8 * lh) = o;
        }
        __syncthreads();
        for (int it = tid; it < PT * PT * (CO / 8); it += 256) {
            const int ps = it & 7, pp = it >> 3, ls = ps ^ ((pp >> 1) & 7);
            const int py = pp / PT, px = pp - py * PT;
            *(u32x4*)(out_h1 + (((long long)n * POOL + PT * ty + py) * POOL + PT * tx + px) * CO + ls * 8) =
                *(const u32x4*)(s_o + pp * 128 + ps * 16);
        }
    }
}
}  // namespace


// ---------------------------------------------------------------------------------------------------------------
// f16x3 (split) operands: the same tile geometry with hi and lo PLANES of the patch and of the filters in LDS and
// three MFMAs per product (act.lo*w.hi, act.hi*w.lo, act.hi*w.hi -- the order of gemm_conv.hip's mma(split, split)).
// A workgroup (5 waves, two MFMA row tiles each) computes 32 of the 64 output channels (blockIdx.z), which keeps
// patch planes + filter planes / fp32 pooling stage at 66 KB: two workgroups per CU.  Rounding points are those of
// the three-kernel route (conv + bias -> split storage -> max pool -> fma + ReLU -> split storage), K order is tap
// by tap, 16 elements at a time, so the result is that route's, bit for bit (tests/test_gpu_sizes.py).
namespace {
constexpr int X3_NT = 320, X3_CO = 32;
constexpr int X3_PLANE = IP * IPW * 8;                       // one 16-bit RGBX plane of the patch
constexpr int X3_WROW = 7 * TAPK * 2 + 16;                   // padded filter row of one plane (464 B: conflict-free b128 reads)
constexpr int X3_WPLANE = X3_CO * X3_WROW;
constexpr int X3_STAGE = MT * 32 * X3_CO * 4;                // conv + bias as fp32 [pixel][32] (overlaps the filters)
constexpr int X3_PTILE = PT * PT * 128;                      // the pooled, pre-activated tile of one channel half as split rows: 64 pixels x 128 B
constexpr int X3_LDS0 = 2 * X3_PLANE + (2 * X3_WPLANE > X3_STAGE ? 2 * X3_WPLANE : X3_STAGE);
constexpr int X3_LDS = X3_LDS0 + X3_PTILE;                   // (the second half's tile goes over the patch planes, dead by then)
static_assert(2 * X3_PTILE <= 2 * X3_PLANE + X3_PTILE && 2 * X3_LDS <= 160 * 1024, "two workgroups per CU");

__global__ __launch_bounds__(X3_NT) void stem_fused_split_kernel(const float* __restrict__ img, const bsplit_t* __restrict__ wts,
                                                                 const float* __restrict__ wscale,
                                                                 const float* __restrict__ bias, const float* __restrict__ pscale,
                                                                 const float* __restrict__ pshift, bsplit_t* __restrict__ out,
                                                                 int n_real, const char* __restrict__ w1f, const float* __restrict__ s1,
                                                                 const float* __restrict__ b1, bsplit_t* __restrict__ out_h1) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* s_ph = smem;                            // patch, hi plane
    char* s_pl = smem + X3_PLANE;                 // patch, lo plane
    char* s_wh = smem + 2 * X3_PLANE;             // filters, hi plane
    char* s_wl = s_wh + X3_WPLANE;
    float* s_c = (float*)(smem + 2 * X3_PLANE);   // reuses the filter region after the MFMAs

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ty = blockIdx.x / (POOL / PT), tx = blockIdx.x % (POOL / PT);
    const int n = blockIdx.y;
    const int cy0 = 2 * PT * ty, cx0 = 2 * PT * tx;
    const int iy0 = 2 * cy0 - 3, ix0 = 2 * cx0 - 3;

    // ---- 1. input patch -> two fp16 RGBX planes (hi = fp16(x), lo = fp16(x - hi): stem_repack_split_kernel's values)
    const float* im = img + (long long)n * IMG * IMG * 3;
    float satin = 0.f;                            // (round 6) a NaN / inf / out-of-range PIXEL: the clamp below would make it a finite number
    for (int i = tid; i < IP * 11; i += X3_NT) {
        const int py = i / 11, q = i - py * 11;
        const int gy = iy0 + py, gx = ix0 - 1 + 4 * q;
        float f[12];
#pragma unroll
        for (int e = 0; e < 12; ++e) f[e] = 0.f;
        if (n < n_real && (unsigned)gy < (unsigned)IMG && (unsigned)gx < (unsigned)IMG) {
            const f32x4* p = (const f32x4*)(im + (gy * IMG + gx) * 3);
            const f32x4 v0 = p[0], v1 = p[1], v2 = p[2];
#pragma unroll
            for (int e = 0; e < 4; ++e) { f[e] = v0[e]; f[4 + e] = v1[e]; f[8 + e] = v2[e]; }
#pragma unroll
            for (int e = 0; e < 12; e += 2) satin = sat_acc(satin, f[e], f[e + 1]);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int px = 4 * q - 1 + e;
            if ((unsigned)px < (unsigned)IPW) {
                typedef __attribute__((ext_vector_type(4))) shalf_t shalf4;
                shalf4 h, l;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const float cv = split_clamp(f[3 * e + c]);
                    h[c] = (shalf_t)cv;
                    l[c] = (shalf_t)(cv - (float)h[c]);
                }
                h[3] = (shalf_t)0.f; l[3] = (shalf_t)0.f;
                *(shalf4*)(s_ph + (py * IPW + px) * 8) = h;
                *(shalf4*)(s_pl + (py * IPW + px) * 8) = l;
            }
        }
    }
    split_flag_max(satin);
    // Round 5: ONE workgroup per tile computes both halves of the 64 output channels, one after the other, from the patch it built once
    // (two workgroups per tile -- blockIdx.z -- each loaded and split the same 39 x 39 pixels: 0.75 GB of traffic for a 0.36 GB layer).
    // Per half the arithmetic is untouched: same bits.
#pragma unroll 1
    for (int ch0 = 0; ch0 < CO; ch0 += X3_CO) {
    // ---- 2. this half's 32 filters -> LDS planes.  HBM rows are [32 groups][hi 16 B | lo 16 B]; groups 0..27 = taps 0..6
    if (ch0) __syncthreads();                     // (the pool of the first half is done with the staging region the filters share)
    for (int i = tid; i < X3_CO * 28; i += X3_NT) {
        const int row = i / 28, g = i - row * 28;
        const u32x4* src = (const u32x4*)((const char*)wts + (long long)(ch0 + row) * (WK * 4) + g * 32);
        *(u32x4*)(s_wh + row * X3_WROW + g * 16) = src[0];
        *(u32x4*)(s_wl + row * X3_WROW + g * 16) = src[1];
    }
    __syncthreads();

    // ---- 3. implicit GEMM: wave w owns row tiles w and w + 5
    const int lr = lane & 31, lh = lane >> 5;
    constexpr int MPW = MT / 5;
    static_assert(MPW * 5 == MT, "10 row tiles over 5 waves");
    f32x16 acc[MPW];
#pragma unroll
    for (int i = 0; i < MPW; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    int abase[MPW];
#pragma unroll
    for (int i = 0; i < MPW; ++i) {
        int p = (wave + 5 * i) * 32 + lr;
        if (p >= NPIX) p = 0;                     // padding rows of the last tile: computed, never used
        const int cy = p / CT, cx = p - cy * CT;
        abase[i] = ((2 * cy) * IPW + 2 * cx) * 8 + lh * 16;
    }
    const int bbase = lr * X3_WROW + lh * 16;
    // The 14 K steps (7 taps x 2 chunks of 16) with the six fragment reads of step s + 1 in flight under the six MFMAs of step s.
    // The reads are inline assembly: left to itself hipcc issues each step's reads right in front of their first use and waits for
    // them with the matrix pipe idle, 30 times per wave (round 4; per accumulator the order of the products is unchanged: same bits).
    const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) void*)smem;
    const unsigned aw = lds0 + 2 * X3_PLANE + bbase;            // filters: hi plane (+ X3_WPLANE: lo), + ky * 64 + c * 32
    unsigned ap[MPW];                                           // patch: hi plane (+ X3_PLANE: lo), + ky * IPW * 8 + c * 32
#pragma unroll
    for (int i = 0; i < MPW; ++i) ap[i] = lds0 + abase[i];
    shalf8 fb[2][2], fa[2][MPW][2];                             // [set][...][hi, lo]
#define STEM_RD(dst, addr, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF))
    auto load_step = [&](auto s_c, int set) {
        constexpr int S = decltype(s_c)::value, KY = S >> 1, C = S & 1;
        shalf8& bh = fb[set][0];
        shalf8& bl = fb[set][1];
        STEM_RD(bh, aw, KY * 64 + C * 32);
        STEM_RD(bl, aw, X3_WPLANE + KY * 64 + C * 32);
#pragma unroll
        for (int i = 0; i < MPW; ++i) {
            shalf8& ah = fa[set][i][0];
            shalf8& al = fa[set][i][1];
            const unsigned ad = ap[i];
            STEM_RD(ah, ad, KY * IPW * 8 + C * 32);
            STEM_RD(al, ad, X3_PLANE + KY * IPW * 8 + C * 32);
        }
    };
    auto k_step = [&](auto s_c) {
        constexpr int S = decltype(s_c)::value, CUR = S & 1;
        if constexpr (S + 1 < 14) load_step(std::integral_constant<int, (S + 1 < 14 ? S + 1 : 13)>{}, CUR ^ 1);
#pragma unroll
        for (int i = 0; i < MPW; ++i) {
            acc[i] = mfma_split(fa[CUR][i][1], fb[CUR][0], acc[i]);
            acc[i] = mfma_split(fa[CUR][i][0], fb[CUR][1], acc[i]);
            acc[i] = mfma_split(fa[CUR][i][0], fb[CUR][0], acc[i]);
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_waitcnt(63 | (7 << 4) | (0 << 8) | (3 << 14));      // lgkmcnt(0): the next step's fragments
        __builtin_amdgcn_sched_barrier(0);
    };
    load_step(std::integral_constant<int, 0>{}, 0);
    __builtin_amdgcn_s_waitcnt(63 | (7 << 4) | (0 << 8) | (3 << 14));
    __builtin_amdgcn_sched_barrier(0);
    auto k_steps = [&](auto... S) { (k_step(S), ...); };
    k_steps(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{}, std::integral_constant<int, 2>{}, std::integral_constant<int, 3>{},
            std::integral_constant<int, 4>{}, std::integral_constant<int, 5>{}, std::integral_constant<int, 6>{}, std::integral_constant<int, 7>{},
            std::integral_constant<int, 8>{}, std::integral_constant<int, 9>{}, std::integral_constant<int, 10>{}, std::integral_constant<int, 11>{},
            std::integral_constant<int, 12>{}, std::integral_constant<int, 13>{});
#undef STEM_RD
    __syncthreads();                              // every wave is done with the filters: reuse as staging

    // ---- 4a. conv + bias, rounded to what split storage holds, -> LDS fp32 [pixel][32]
    {
        const float bch = bias[ch0 + lr];
        const float sch = wscale ? wscale[ch0 + lr] : 1.0f;   // undoes the pack-time power-of-two scale of the filter row
#pragma unroll
        for (int i = 0; i < MPW; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int p = (wave + 5 * i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                // (one explicit fma, like conv_gemm's epilogue: fma(v, 1, b) == v + b when there is no scale)
                s_c[p * X3_CO + lr] = stored_value<bsplit_t>(fmaf(acc[i][r], sch, bch));
            }
    }
    __syncthreads();

    // ---- 4b. 3x3/2 max pool (TF SAME) + preact BN + ReLU -> split storage
    for (int it = tid; it < PT * PT * (X3_CO / 8); it += X3_NT) {
        const int v8 = it & 3, pp = it >> 2;
        const int py = pp / PT, px = pp - py * PT;
        float m[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) m[j] = -3.0e38f;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            if (cy0 + 2 * py + dy >= CONV) continue;
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                if (cx0 + 2 * px + dx >= CONV) continue;
                float v[8];
                load8(s_c + ((2 * py + dy) * CT + 2 * px + dx) * X3_CO + v8 * 8, v);
#pragma unroll
                for (int j = 0; j < 8; ++j) m[j] = fmaxf(m[j], v[j]);
            }
        }
        float sc[8], sh[8];
        load8(pscale + ch0 + v8 * 8, sc); load8(pshift + ch0 + v8 * 8, sh);
#pragma unroll
        for (int j = 0; j < 8; ++j) m[j] = fmaxf(fmaf(m[j], sc[j], sh[j]), 0.f);   // one explicit fma: = maxpool_bn_relu_kernel
        bsplit_t* o = out + (((long long)n * POOL + PT * ty + py) * POOL + PT * tx + px) * CO + ch0 + v8 * 8;
        store8(o, m);
        if (w1f) {
            // block1/unit_1's conv1 reads this tile as STORED: the same 32 bytes (hi 8 | lo 8) go into the pooled tile of this half,
            // the 32-byte group index XOR-swizzled with the pixel (rows are 128 B: the fragment reads below would hit two banks' worth)
            shalf8 hi, lo;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float cv = split_clamp(m[j]);
                hi[j] = (shalf_t)cv;
                lo[j] = (shalf_t)(cv - (float)hi[j]);
            }
            char* pt = (ch0 ? smem : smem + X3_LDS0) + pp * 128 + ((v8 ^ (pp & 3)) << 5);
            *(shalf8*)pt = hi;
            *(shalf8*)(pt + 16) = lo;
        }
    }
    }
    // ---- 5. (round 5) block1/unit_1's conv1 (1x1, 64 -> 64, folded BN + ReLU) on the pooled tile before it leaves the CU: the launch of
    // that layer and its 0.2 GB read of this kernel's own output are gone.  Weights = the MFMA A operand as fragment-major copies straight
    // from L2 (packing.pack_frag_major), the tile's pixels = B out of the two pooled half tiles; products and K order are those of
    // hmmr_conv_gemm on the stored tensor (w.hi x.lo, w.lo x.hi, w.hi x.hi per 16-wide chunk, chunks in order): the same bits.
    if (!w1f) return;
    __syncthreads();
    if (wave < 4) {
        const int lr = lane & 31, lh = lane >> 5;
        const int wn = wave >> 1, wm = wave & 1;
        const int p = wm * 32 + lr;                                // this lane's pixel of the 8 x 8 tile
        f32x16 acc1;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc1[r] = 0.f;
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) {
            const shalf8* wp = (const shalf8*)(w1f + ((long long)((wn * 4 + kc) * 64 + lane)) * 32);
            wfrag wf; wf.hi = wp[0]; wf.lo = wp[1];
            const int g = (2 * kc + lh) & 3;                       // 8-channel group inside its half (kc >= 2: the second half)
            const char* pt = (kc >= 2 ? smem : smem + X3_LDS0) + p * 128 + ((g ^ (p & 3)) << 5);
            acc1 = mma3(wf, *(const shalf8*)pt, *(const shalf8*)(pt + 16), acc1);
        }
        const int py = p >> 3, px = p & 7;
        bsplit_t* o = out_h1 + (((long long)n * POOL + PT * ty + py) * POOL + PT * tx + px) * CO;
        float satmax = 0.f;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int ch = wn * 32 + 8 * g + 4 * lh;
            const f32x4 s4 = *(const f32x4*)(s1 + ch), b4 = *(const f32x4*)(b1 + ch);
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaf(acc1[4 * g + e], s4[e], b4[e]);
            unsigned h2[2], l2[2];
            split4_mix(v, 0.f, h2, l2, satmax);                    // (ReLU + clamp in one v_med3_f32)
            // lane (pixel, half lh) holds channels 8 g + 4 lh .. + 3 of its row block: 8 bytes of the hi and of the lo half of group g
            char* og = (char*)o + (wn * 4 + g) * 32 + 8 * lh;
            *(unsigned long long*)og = (unsigned long long)h2[0] | ((unsigned long long)h2[1] << 32);
            *(unsigned long long*)(og + 16) = (unsigned long long)l2[0] | ((unsigned long long)l2[1] << 32);
        }
        split_flag_max(satmax);
    }
}
}  // namespace

// images [n_real,224,224,3] fp32 (+ n - n_real implicit zero images) -> out [n,56,56,64] (dtype)
// w1 / s1 / b1 / out_h1 (bf16 only, may be NULL): block1/unit_1's conv1 [64][64] + folded BN, computed on the
// pooled tile in the same launch -> out_h1 [n,56,56,64]
int hmmr_stem_fused(const float* images, int n_real, int n, const void* wts, const float* wscale, const float* bias,
                    const float* pscale, const float* pshift, void* out, int dtype, hipStream_t s,
                    const void* w1, const float* s1, const float* b1, void* out_h1) {
    if (dtype == HMMR_F16X3) {
        auto kern = stem_fused_split_kernel;
        static DeviceOnce oncex3;
        if (const unsigned long long bit = oncex3.due()) {
            HMMR_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, X3_LDS));
            oncex3.mark(bit);
        }
        hipLaunchKernelGGL(kern, dim3((POOL / PT) * (POOL / PT), n), dim3(X3_NT), X3_LDS, s, images,
                           (const bsplit_t*)wts, wscale, bias, pscale, pshift, (bsplit_t*)out, n_real, (const char*)w1, s1, b1,
                           (bsplit_t*)out_h1);
    } else if (dtype == HMMR_BF16) {
        auto kern = stem_fused_kernel<bf16_t>;
        static DeviceOnce once16;
        if (const unsigned long long bit = once16.due()) {
            HMMR_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, StemLds<bf16_t>::TOTAL));
            once16.mark(bit);
        }
        hipLaunchKernelGGL(kern, dim3((POOL / PT) * (POOL / PT), n), dim3(256), StemLds<bf16_t>::TOTAL, s, images,
                           (const bf16_t*)wts, bias, pscale, pshift, (bf16_t*)out, n_real, (const bf16_t*)w1, s1, b1,
                           (bf16_t*)out_h1);
    } else {
        auto kern = stem_fused_kernel<float>;
        static DeviceOnce once32;
        if (const unsigned long long bit = once32.due()) {
            HMMR_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, StemLds<float>::TOTAL));
            once32.mark(bit);
        }
        hipLaunchKernelGGL(kern, dim3((POOL / PT) * (POOL / PT), n), dim3(256), StemLds<float>::TOTAL, s, images,
                           (const float*)wts, bias, pscale, pshift, (float*)out, n_real, (const float*)nullptr,
                           (const float*)nullptr, (const float*)nullptr, (float*)nullptr);
    }
    HMMR_CHECK_HIP(hipGetLastError());
    return 0;
}
