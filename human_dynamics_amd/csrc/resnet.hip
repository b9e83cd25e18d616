// ResNet-v2-50 image encoder for gfx950: encoder_resnet (src/models.py:50-77)
// -> tf.contrib.slim.nets.resnet_v2.resnet_v2_50(num_classes=None,
// is_training=False).  Layer semantics restated in SURVEY.md App. A.
//
// The 52 convolutions after the stem run through the implicit-GEMM kernel
// (gemm_conv.hip); the stem + pool1 + first preact is one fused kernel
// (stem.hip).  This file holds the launch sequence, pool5, and the unfused stem
// route kept for A/B measurements:
//   stem_repack       fp32 RGB [n,224,224,3] -> zero-padded RGBX [n,230,232,4]
//                     in the operand dtype, so the 7x7/2 stem (explicit pad 3,
//                     VALID) becomes an 8-tap x 32-element implicit GEMM
//                     (tap = ky, 32 elements = 8 pixels x 4 channels)
//   maxpool_bn_relu   3x3/2 TF-SAME max pool (pad bottom/right only) fused with
//                     block1/unit_1's `preact` BN + ReLU (every later unit's preact is fused
//                     into the operand staging of its conv1 / shortcut GEMMs)
//   bn_relu_avgpool   postnorm BN + ReLU + spatial mean (pool5)
// Inference BN is folded to y = x*scale + shift on the host
// (scale = gamma*rsqrt(var+1e-5), shift = beta - mean*scale).
#include <stdlib.h>

#include <type_traits>

#include "common.h"
#include "hmmr_hip.h"

// csrc/stem.hip
int hmmr_stem_fused(const float* images, int n_real, int n, const void* wts, const float* wscale, const float* bias,
                    const float* pscale, const float* pshift, void* out, int dtype, hipStream_t s,
                    const void* w1, const float* s1, const float* b1, void* out_h1);

static constexpr int IMG = 224, PADH = 230, PADW = 232;

// split (f16x3) image: one 8-"channel" group = two RGBX pixels; PADW is even, so a pair never straddles a row
__global__ void stem_repack_split_kernel(const float* __restrict__ img, bsplit_t* __restrict__ out, long long npairs,
                                         long long n_real) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < npairs;
         i += (long long)gridDim.x * blockDim.x) {
        const int xp = (int)(i % (PADW / 2));
        const long long t = i / (PADW / 2);
        const int y = (int)(t % PADH);
        const long long n = t / PADH;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = 0.f;
        const int sy = y - 3;
        if (n < n_real && (unsigned)sy < (unsigned)IMG) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int sx = 2 * xp + q - 3;
                if ((unsigned)sx < (unsigned)IMG) {
                    const float* p = img + ((n * IMG + sy) * IMG + sx) * 3;
                    v[4 * q] = p[0]; v[4 * q + 1] = p[1]; v[4 * q + 2] = p[2];
                }
            }
        }
        store8(out + i * 8, v);
    }
}

template <typename T>
__global__ void stem_repack_kernel(const float* __restrict__ img, T* __restrict__ out, long long npix,
                                   long long n_real) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < npix;
         i += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(i % PADW);
        const long long t = i / PADW;
        const int y = (int)(t % PADH);
        const long long n = t / PADH;
        float r = 0.f, g = 0.f, b = 0.f;
        const int sy = y - 3, sx = x - 3;
        if (n < n_real && (unsigned)sy < (unsigned)IMG && (unsigned)sx < (unsigned)IMG) {
            const float* p = img + ((n * IMG + sy) * IMG + sx) * 3;
            r = p[0]; g = p[1]; b = p[2];
        }
        T* o = out + i * 4;
        o[0] = elem_traits<T>::from_f32(r);
        o[1] = elem_traits<T>::from_f32(g);
        o[2] = elem_traits<T>::from_f32(b);
        o[3] = elem_traits<T>::from_f32(0.f);
    }
}

// in [n,112,112,64] -> out [n,56,56,64]; window rows 2oy..2oy+2, cols 2ox..2ox+2,
// out-of-range taps ignored (TF SAME with pad_before = 0).
template <typename T>
__global__ void maxpool_bn_relu_kernel(const T* __restrict__ in, T* __restrict__ out,
                                       const float* __restrict__ scale, const float* __restrict__ shift,
                                       long long nvec) {
    constexpr int HI = 112, HO = 56, C = 64, CV = C / 8;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec;
         i += (long long)gridDim.x * blockDim.x) {
        const int cv = (int)(i % CV);
        long long t = i / CV;
        const int ox = (int)(t % HO); t /= HO;
        const int oy = (int)(t % HO);
        const long long n = t / HO;
        float m[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) m[j] = -3.0e38f;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            const int iy = 2 * oy + dy;
            if (iy >= HI) continue;
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int ix = 2 * ox + dx;
                if (ix >= HI) continue;
                float v[8];
                load8(in + ((n * HI + iy) * HI + ix) * C + cv * 8, v);
#pragma unroll
                for (int j = 0; j < 8; ++j) m[j] = fmaxf(m[j], v[j]);
            }
        }
        float s[8], b[8];
        load8(scale + cv * 8, s); load8(shift + cv * 8, b);
#pragma unroll
        for (int j = 0; j < 8; ++j) m[j] = fmaxf(fmaf(m[j], s[j], b[j]), 0.f);   // one explicit fma: = stem.hip, bit for bit
        store8(out + i * 8, m);
    }
}

// in [n,hw,c] -> phi [n,c] fp32: mean over hw of relu(x*scale+shift)
template <typename T>
__global__ void bn_relu_avgpool_kernel(const T* __restrict__ in, float* __restrict__ phi,
                                       const float* __restrict__ scale, const float* __restrict__ shift,
                                       int n, int hw, int c) {
    const int cv = c / 8;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)n * cv) return;
    const int v8 = (int)(i % cv);
    const long long img = i / cv;
    float s[8], b[8], acc[8];
    load8(scale + v8 * 8, s); load8(shift + v8 * 8, b);
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    for (int p = 0; p < hw; ++p) {
        float v[8];
        load8(in + (img * hw + p) * c + v8 * 8, v);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += fmaxf(v[j] * s[j] + b[j], 0.f);
    }
    const float inv = 1.0f / (float)hw;
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] *= inv;
    store8(phi + img * c + v8 * 8, acc);
}

// ------------------------------------------------------------------------- //
struct Prof {
    float* ms; int slot; hipStream_t s; hipEvent_t ev[HMMR_RESNET_PROF_SLOTS + 1];
    bool on() const { return ms != nullptr; }
};

static int prof_begin(Prof& p) {
    if (!p.on()) return 0;
    for (int i = 0; i <= HMMR_RESNET_PROF_SLOTS; ++i) HMMR_CHECK_HIP(hipEventCreate(&p.ev[i]));
    HMMR_CHECK_HIP(hipEventRecord(p.ev[0], p.s));
    p.slot = 0;
    return 0;
}
static int prof_mark(Prof& p) {
    if (!p.on() || p.slot >= HMMR_RESNET_PROF_SLOTS) return 0;
    ++p.slot;
    HMMR_CHECK_HIP(hipEventRecord(p.ev[p.slot], p.s));
    return 0;
}
static int prof_end(Prof& p) {
    if (!p.on()) return 0;
    HMMR_CHECK_HIP(hipStreamSynchronize(p.s));
    for (int i = 0; i < HMMR_RESNET_PROF_SLOTS; ++i) {
        p.ms[i] = 0.f;
        if (i < p.slot) HMMR_CHECK_HIP(hipEventElapsedTime(&p.ms[i], p.ev[i], p.ev[i + 1]));
    }
    for (int i = 0; i <= HMMR_RESNET_PROF_SLOTS; ++i) (void)hipEventDestroy(p.ev[i]);
    return 0;
}

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct ResnetBufs {
    size_t xpad, stem, x[2], p[2], t1, t2, total;
};
static ResnetBufs resnet_layout(int n, int dtype) {
    const size_t e = dtype == HMMR_BF16 ? 2 : 4;
    ResnetBufs b; size_t off = 0;
    auto take = [&](size_t elems) { size_t o = off; off = align_up(off + elems * e, 256); return o; };
    b.xpad = take((size_t)n * PADH * PADW * 4);
    b.stem = take((size_t)n * 112 * 112 * 64);
    for (int i = 0; i < 2; ++i) b.x[i] = take((size_t)n * 56 * 56 * 256);
    for (int i = 0; i < 2; ++i) b.p[i] = take((size_t)n * 56 * 56 * 256);
    b.t1 = take((size_t)n * 56 * 56 * 64);
    b.t2 = take((size_t)n * 56 * 56 * 64);
    b.total = off;
    return b;
}

extern "C" size_t hmmr_resnet50_workspace_bytes(int n, int dtype) {
    return n > 0 ? resnet_layout(n, dtype).total : 0;
}

template <typename T>
static int resnet_fwd_t(const hmmr_resnet_weights_t* w, const float* images, int n_real, int n, float* phi,
                        char* ws, hipStream_t s, float* prof_ms) {
    const ResnetBufs L = resnet_layout(n, w->dtype);
    T* xpad = (T*)(ws + L.xpad);
    T* stem = (T*)(ws + L.stem);
    T* X[2] = {(T*)(ws + L.x[0]), (T*)(ws + L.x[1])};
    T* P[2] = {(T*)(ws + L.p[0]), (T*)(ws + L.p[1])};
    T* T1 = (T*)(ws + L.t1);
    T* T2 = (T*)(ws + L.t2);
    Prof pf; pf.ms = prof_ms; pf.s = s; pf.slot = 0;
    if (prof_begin(pf)) return -2;

    // ---- stem: 7x7/2 conv (+bias, no BN/ReLU) -> pool1 -> preact of block1/unit_1.
    // Default: ONE fused kernel (csrc/stem.hip).  HMMR_STEM=unfused keeps the three-kernel route
    // (re-pack, implicit GEMM, pool) for A/B measurements.
    // In fp32-operand mode the fused kernel needs 104 KB of LDS (one workgroup per CU) and measures
    // ~1 % slower than the three-kernel route, which therefore stays the fp32 default.  f16x3 has its own
    // fused kernel (stem_fused_split_kernel: hi/lo planes, 32 output channels per workgroup).
    const hmmr_debug_t* dbg = hmmr_debug_state();
    const bool unfused = dbg->stem_route == 1 || (dbg->stem_route == 0 && w->dtype == HMMR_F32);
    bool stem_c1 = false;
    if (!unfused) {
        // bf16: block1/unit_1's conv1 is computed on each pooled tile inside the same launch (-> T1)
        const hmmr_resnet_unit_t& U0 = w->unit[0];
        // (f16x3, round 5: with the fragment-major copy of the conv1 filters, hmmr_resnet_unit_t.conv1_frag)
        stem_c1 = !dbg->stem_no_conv1 && (w->dtype == HMMR_BF16 || (w->dtype == HMMR_F16X3 && U0.conv1_frag)) && !U0.sc_c1.w && U0.c_in == 64 &&
                  U0.base == 64 && U0.conv1.scale && U0.conv1.shift;
        if (hmmr_stem_fused(images, n_real, n, w->stem.w, w->stem.scale, w->stem.shift, U0.pre_scale, U0.pre_shift, P[0], w->dtype, s,
                            stem_c1 ? (w->dtype == HMMR_F16X3 ? U0.conv1_frag : U0.conv1.w) : nullptr, U0.conv1.scale, U0.conv1.shift,
                            stem_c1 ? T1 : nullptr))
            return -2;
        if (prof_mark(pf)) return -2;
        if (prof_mark(pf)) return -2;     // (keeps the profile slot numbering of the 3-kernel route)
        if (prof_mark(pf)) return -2;
    } else {
        const long long npix = (long long)n * PADH * PADW;
        const int grid = (int)((npix + 255) / 256 < 8192 ? (npix + 255) / 256 : 8192);
        if constexpr (std::is_same<T, bsplit_t>::value)
            hipLaunchKernelGGL(stem_repack_split_kernel, dim3(grid), dim3(256), 0, s, images, xpad, npix / 2, (long long)n_real);
        else
            hipLaunchKernelGGL(stem_repack_kernel<T>, dim3(grid), dim3(256), 0, s, images, xpad, npix, (long long)n_real);
        HMMR_CHECK_HIP(hipGetLastError());
        if (prof_mark(pf)) return -2;
        hmmr_conv_desc_t d = {};
        d.in = xpad; d.w = w->stem.w; d.scale = w->stem.scale; d.shift = w->stem.shift;
        d.out = stem; d.in_dtype = d.out_dtype = w->dtype;
        d.n_img = n; d.hin = PADH; d.win = 2 * 111 + 1; d.cin = 32;
        d.in_img_stride = (int64_t)PADH * PADW * 4; d.in_row_stride = PADW * 4; d.in_px_stride = 4;
        d.kh = 8; d.kw = 1; d.sy = 2; d.sx = 2; d.py = 0; d.px = 0;
        d.ho = 112; d.wo = 112; d.cout = 64; d.ldo = 64;
        if (hmmr_conv_gemm(&d, s)) return -2;
        if (prof_mark(pf)) return -2;
        const long long nvec = (long long)n * 56 * 56 * 8;
        const int g2 = (int)((nvec + 255) / 256 < 16384 ? (nvec + 255) / 256 : 16384);
        hipLaunchKernelGGL(maxpool_bn_relu_kernel<T>, dim3(g2), dim3(256), 0, s, (const T*)stem, P[0],
                           w->unit[0].pre_scale, w->unit[0].pre_shift, nvec);
        HMMR_CHECK_HIP(hipGetLastError());
        if (prof_mark(pf)) return -2;
    }

    // Units.  A unit's pre-activation BN + ReLU (`preact`) reaches its 1x1 consumers (conv1 and the
    // conv shortcut) in one of two ways, chosen per unit by `fuse_preact`:
    //   1: the consumers read the RAW trunk and apply the preact while staging their A operand
    //      (the tensor never exists in HBM; pays in the HBM-bound early blocks);
    //   0: the previous unit's conv3 epilogue writes it as a second output (the consumers keep the
    //      pure LDS-DMA operand path; better for the MFMA-bound late blocks).
    // Measured at batch 256 (bf16): fusing blocks 1-2 cuts the ResNet pass by 4.5 %, blocks 3-4 are
    // neutral; the packer enables it everywhere.
    int H = 56, cur = 0, pcur = 0;
    bool have_raw = false;            // X[cur] holds the raw input of the unit
    bool h1_ready = stem_c1;          // T1 already holds this unit's conv1 output (previous unit's fused tail / the stem)
    for (int u = 0; u < HMMR_RESNET_UNITS; ++u) {
        const hmmr_resnet_unit_t& U = w->unit[u];
        const int Ho = H / U.stride;
        const bool last = (u == HMMR_RESNET_UNITS - 1);
        const bool fused = u > 0 && U.fuse_preact;
        const T* xin = fused ? (const T*)X[cur] : (const T*)P[pcur];
        const float* ps = fused ? U.pre_scale : nullptr;
        const float* pb = fused ? U.pre_shift : nullptr;
        HMMR_REQUIRE(!fused || (ps && pb && have_raw), "resnet: unit %d cannot fuse its preact", u);
        const bool identity = !U.shortcut.w;
        HMMR_REQUIRE(!identity || have_raw, "resnet: unit %d has no raw input for its identity shortcut", u);
        // what the NEXT unit needs from this one
        const bool next_fused = !last && w->unit[u + 1].fuse_preact;
        const bool next_identity = !last && !w->unit[u + 1].shortcut.w;
        const bool write_raw = last || next_fused || next_identity;
        const bool write_pre = !last && !next_fused;
        T* xn = X[cur ^ 1];
        T* pn = P[pcur ^ 1];
        hmmr_conv_desc_t d;
        const bool sc_c1 = U.shortcut.w && U.sc_c1.w && !h1_ready && U.stride == 1;
        // a register-resident unit pair (csrc/unit_pair.hip) is one round of 128-pixel workgroups with a ~20 k-cycle prologue however few
        // pixels there are: below ~12 k pixels (61 frames in block 3) the two launches it replaces are faster (profiles/r04_unit_pair_check.log:
        // 0.064 against 0.096 ms at 33 frames), and they produce the same bits, so a short batch simply takes them
        // (hmmr_debug_t.pair_min_pixels moves the switch: tests run one batch on either side of it)
        // round 6 (profiles/r06e_pair_ws_check.log): in block 3 the two launches still win at 12 544 pixels (64 frames, FeatureExtractor's batch:
        // 0.085 against 0.093 ms) -- its switch is at 14 000; a block-2 pair is a shorter tile and wins from ~12 000 (0.034 against 0.043 ms at 15 680)
        const long long pair_min = dbg->pair_min_pixels > 0 ? dbg->pair_min_pixels : (U.base >= 256 ? 14000 : 12000);
        const bool pair_off = w->dtype == HMMR_F16X3 && U.pair_stream && U.fuse_tail == 1 && (long long)n * Ho * Ho < pair_min;
        const int fuse_tail = pair_off ? 0 : U.fuse_tail;
        const bool sc_in_tail = fuse_tail == 3;          // the conv shortcut is computed inside the fused tail
        // the conv shortcut is folded into conv3: ONE GEMM over {h2, preact} with [W3 | Wsc] (hmmr_conv_desc_t.in2);
        // the shortcut tensor (the widest tensor of the unit) is neither written nor read back
        const bool sc_in_c3 = U.c3sc.w != nullptr;
        HMMR_REQUIRE(!sc_in_c3 || (U.shortcut.w && !fused && U.stride == 1 && fuse_tail <= 2 && !U.sc_c1.w),
                     "resnet: unit %d cannot fold its shortcut into conv3", u);
        if (sc_in_c3) {
            if (prof_mark(pf)) return -2;
        } else if (sc_in_tail) {
            HMMR_REQUIRE(U.shortcut.w && !U.shortcut.scale && !fused && U.c_in == 64 && U.stride == 1,
                         "resnet: unit %d cannot compute its shortcut inside the tail", u);
            if (prof_mark(pf)) return -2;
        } else if (U.shortcut.w) {    // 1x1 conv on preact, bias, no BN/ReLU (stride is 1 here)
            d = hmmr_conv_desc_t{};
            d.in = xin; d.pro_scale = ps; d.pro_shift = pb;
            d.w = U.shortcut.w; d.scale = U.shortcut.scale; d.shift = U.shortcut.shift; d.tile = U.shortcut.tile;
            d.out = xn; d.in_dtype = d.out_dtype = w->dtype;
            d.n_img = n; d.hin = H; d.win = H; d.cin = U.c_in;
            d.in_img_stride = (int64_t)H * H * U.c_in; d.in_row_stride = H * U.c_in; d.in_px_stride = U.c_in;
            d.kh = d.kw = 1; d.sy = d.sx = U.stride; d.ho = d.wo = Ho; d.cout = U.depth; d.ldo = U.depth;
            if (sc_c1) {              // ... and conv1 over the same operand as extra output columns (-> T1, BN + ReLU)
                d.w = U.sc_c1.w; d.scale = U.sc_c1.scale; d.shift = U.sc_c1.shift; d.k_order = U.sc_c1.k_order;
                d.cout = U.depth + U.base; d.out_b = T1; d.ldo_b = U.base; d.n_split = U.depth; d.relu_b = 1;
            }
            if (hmmr_conv_gemm(&d, s)) return -2;
            if (prof_mark(pf)) return -2;
            if (sc_c1) h1_ready = true;
        }
        // conv1: 1x1 on preact, BN + ReLU (already in T1 if the previous unit ended in a fused tail)
        if (!h1_ready) {
            d = hmmr_conv_desc_t{};
            d.in = xin; d.pro_scale = ps; d.pro_shift = pb;
            d.w = U.conv1.w; d.scale = U.conv1.scale; d.shift = U.conv1.shift; d.relu = 1; d.tile = U.conv1.tile;
            d.k_order = U.conv1.k_order;      // (2: the two-ring stream kernel of csrc/conv1x1_stream.hip; it takes the pre-activated tensor)
            d.out = T1; d.in_dtype = d.out_dtype = w->dtype;
            d.n_img = n; d.hin = H; d.win = H; d.cin = U.c_in;
            d.in_img_stride = (int64_t)H * H * U.c_in; d.in_row_stride = H * U.c_in; d.in_px_stride = U.c_in;
            d.kh = d.kw = 1; d.sy = d.sx = 1; d.ho = d.wo = H; d.cout = U.base; d.ldo = U.base;
            if (hmmr_conv_gemm(&d, s)) return -2;
        }
        if (prof_mark(pf)) return -2;
        // conv2: 3x3 conv2d_same(stride): pad 1/1 both for stride 1 (SAME) and stride 2 (explicit pad + VALID);
        // with fuse_tail == 2 it runs inside the fused tail below
        const bool conv2_in_tail = fuse_tail >= 2;       // (4: the single-phase tail of a stride-2 unit)
        if (!conv2_in_tail) {
            d = hmmr_conv_desc_t{};
            d.in = T1; d.w = U.conv2.w; d.scale = U.conv2.scale; d.shift = U.conv2.shift; d.relu = 1; d.tile = U.conv2.tile;
            d.k_order = U.conv2.k_order;
            d.out = T2; d.in_dtype = d.out_dtype = w->dtype;
            d.n_img = n; d.hin = H; d.win = H; d.cin = U.base;
            d.in_img_stride = (int64_t)H * H * U.base; d.in_row_stride = H * U.base; d.in_px_stride = U.base;
            d.kh = d.kw = 3; d.sy = d.sx = U.stride; d.py = d.px = 1; d.ho = d.wo = Ho; d.cout = U.base; d.ldo = U.base;
            if (hmmr_conv_gemm(&d, s)) return -2;
        }
        if (prof_mark(pf)) return -2;
        // conv3: 1x1 + bias, + shortcut (no ReLU after the add)
        d = hmmr_conv_desc_t{};
        d.in = T2; d.w = U.conv3.w; d.scale = U.conv3.scale; d.shift = U.conv3.shift; d.tile = U.conv3.tile;
        d.in_dtype = d.out_dtype = w->dtype;
        d.n_img = n; d.hin = Ho; d.win = Ho; d.cin = U.base;
        d.in_img_stride = (int64_t)Ho * Ho * U.base; d.in_row_stride = Ho * U.base; d.in_px_stride = U.base;
        d.kh = d.kw = 1; d.sy = d.sx = 1; d.ho = d.wo = Ho; d.cout = U.depth; d.ldo = U.depth;
        d.k_order = U.conv3.k_order;         // (2: the conv3 form of csrc/conv1x1_stream.hip, block 4)
        if (sc_in_c3) { d.w = U.c3sc.w; d.scale = U.c3sc.scale; d.shift = U.c3sc.shift; d.tile = U.c3sc.tile; d.k_order = U.c3sc.k_order; d.in2 = xin; d.cin2 = U.c_in; }
        else if (U.shortcut.w) { d.res = xn; d.ldr = U.depth; }
        else if (U.stride == 1) { d.res = X[cur]; d.ldr = U.depth; }
        else {                        // max_pool2d(x, [1,1], stride) = x[:, ::s, ::s] of the RAW input
            d.res = X[cur]; d.res_strided = 1;
            d.res_img_stride = (int64_t)H * H * U.depth; d.res_row_stride = U.stride * H * U.depth;
            d.res_px_stride = U.stride * U.depth;
        }
        d.out = write_raw ? xn : nullptr;
        if (write_pre) {
            d.out2 = pn; d.scale2 = w->unit[u + 1].pre_scale; d.shift2 = w->unit[u + 1].pre_shift;
            HMMR_REQUIRE(d.scale2 && d.shift2, "resnet: unit %d lacks its preact BN", u + 1);
        }
        h1_ready = false;
        if (fuse_tail == 4) {         // stride-2 last unit of a block: conv2 + conv3 + add in one launch, no next conv1
            HMMR_REQUIRE(!last && w->dtype == HMMR_BF16 && U.conv2.scale && U.conv2.shift && !U.shortcut.w &&
                         ((U.base == 64 && U.depth == 256) || (U.base == 128 && U.depth == 512)),
                         "resnet: unit %d cannot run as a single-phase tail", u);
            hmmr_tail_desc_t t = {};
            t.dtype = w->dtype; t.m = n * Ho * Ho; t.c_mid = U.base; t.depth = U.depth;
            t.h1 = T1; t.hin = H; t.win = H; t.conv2_stride = U.stride; t.ho = Ho; t.wo = Ho;
            t.w2 = U.conv2.w; t.scale2 = U.conv2.scale; t.shift2 = U.conv2.shift;
            t.w3 = U.conv3.w; t.scale3 = U.conv3.scale; t.shift3 = U.conv3.shift;
            t.res = d.res; t.ldr = d.ldr; t.res_strided = d.res_strided; t.res_img_stride = d.res_img_stride;
            t.res_row_stride = d.res_row_stride; t.res_px_stride = d.res_px_stride;
            t.out = d.out; t.out_pre = d.out2; t.pre_scale = d.scale2; t.pre_shift = d.shift2;
            if (hmmr_bottleneck_tail(&t, s)) return -2;
        } else if (fuse_tail) {       // conv3 + add + the next unit's preact + conv1 in one launch (csrc/bottleneck.hip)
            const bool pair = w->dtype == HMMR_F16X3 && U.pair_stream && fuse_tail == 1;        // csrc/unit_pair.hip
            HMMR_REQUIRE(!last && (w->dtype == HMMR_BF16 || pair || (w->dtype == HMMR_F16X3 && ((U.w3_frag && U.w1n_frag) || U.unit_stream) && fuse_tail <= 2)) &&
                         U.stride == 1 && write_raw && !write_pre && next_fused &&
                         next_identity && w->unit[u + 1].base == U.base && w->unit[u + 1].c_in == U.depth &&
                         ((U.base == 64 && U.depth == 256) || (U.base == 128 && U.depth == 512) || (pair && U.base == 256 && U.depth == 1024)),
                         "resnet: unit %d cannot fuse its tail", u);
            const hmmr_resnet_unit_t& N = w->unit[u + 1];
            hmmr_tail_desc_t t = {};
            t.dtype = w->dtype; t.m = n * Ho * Ho; t.c_mid = U.base; t.depth = U.depth;
            if (conv2_in_tail) {      // h2 never exists in HBM; the conv1' output goes to T2 (T1 is still being read
                                      // by neighbouring tiles' halos), and the two buffers swap roles afterwards
                HMMR_REQUIRE(U.conv2.scale && U.conv2.shift, "resnet: unit %d cannot fuse its conv2", u);
                HMMR_REQUIRE((U.conv2.k_order == 2) == (w->dtype == HMMR_F16X3 && U.unit_stream != nullptr),
                             "resnet: unit %d: a k_order 2 conv2 runs inside the unit only as part of its unit_stream (csrc/b1_unit.hip)", u);
                t.h1 = T1; t.hin = H; t.win = H; t.w2 = U.conv2.w; t.scale2 = U.conv2.scale; t.shift2 = U.conv2.shift;
                if (w->dtype == HMMR_F16X3) t.unit_stream = U.unit_stream;
            } else {
                t.h2 = T2;
            }
            t.w3 = U.conv3.w; t.scale3 = U.conv3.scale; t.shift3 = U.conv3.shift;
            if (w->dtype == HMMR_F16X3) {            // fragment-major filters (or one fragment stream); a folded shortcut rides in conv3's K
                t.w3 = U.w3_frag;
                if (pair) t.pair_stream = U.pair_stream;
                if (sc_in_c3) { t.scale3 = U.c3sc.scale; t.shift3 = U.c3sc.shift; t.xp = xin; t.c_xp = U.c_in; }
            }
            if (sc_in_c3) {
            } else if (sc_in_tail) {
                t.xp = xin; t.wsc = U.shortcut.w; t.shift_sc = U.shortcut.shift;
            } else {
                t.res = d.res; t.ldr = d.ldr; t.res_strided = d.res_strided; t.res_img_stride = d.res_img_stride;
                t.res_row_stride = d.res_row_stride; t.res_px_stride = d.res_px_stride;
            }
            t.ho = Ho; t.wo = Ho;
            t.out = xn; t.pre_scale = N.pre_scale; t.pre_shift = N.pre_shift;
            t.w1 = w->dtype == HMMR_F16X3 ? U.w1n_frag : N.conv1.w;      // (not read with pair_stream)
            t.scale1 = N.conv1.scale; t.shift1 = N.conv1.shift; t.relu1 = 1; t.n2 = N.base;
            t.out_h1 = conv2_in_tail ? T2 : T1;
            if (hmmr_bottleneck_tail(&t, s)) return -2;
            if (conv2_in_tail) { T* tmp = T1; T1 = T2; T2 = tmp; }
            h1_ready = true;
        } else if (hmmr_conv_gemm(&d, s)) return -2;
        if (prof_mark(pf)) return -2;
        have_raw = write_raw;
        cur ^= 1; pcur ^= 1; H = Ho;
    }
    const T* xraw = X[cur];
    // ---- postnorm BN + ReLU + mean over 7x7
    {
        const long long nth = (long long)n * (2048 / 8);
        hipLaunchKernelGGL(bn_relu_avgpool_kernel<T>, dim3((unsigned)((nth + 255) / 256)), dim3(256), 0, s,
                           (const T*)xraw, phi, w->post_scale, w->post_shift, n, H * H, 2048);
        HMMR_CHECK_HIP(hipGetLastError());
        if (prof_mark(pf)) return -2;
    }
    return prof_end(pf);
}

extern "C" int hmmr_resnet50_fwd(const hmmr_resnet_weights_t* w, const float* images, int n, int n_zero,
                                 float* phi, void* ws, size_t ws_bytes, void* stream, float* prof_ms) {
    HMMR_REQUIRE(w && phi && ws && (images || n == 0), "hmmr_resnet50_fwd: null argument");
    HMMR_REQUIRE(n >= 0 && n_zero >= 0 && n + n_zero > 0, "hmmr_resnet50_fwd: need at least one image");
    const int nt = n + n_zero;
    HMMR_REQUIRE(ws_bytes >= hmmr_resnet50_workspace_bytes(nt, w->dtype),
                 "hmmr_resnet50_fwd: workspace too small (%zu < %zu)", ws_bytes,
                 hmmr_resnet50_workspace_bytes(nt, w->dtype));
    HMMR_REQUIRE(w->unit[0].c_in == 64 && w->unit[15].depth == 2048, "hmmr_resnet50_fwd: bad unit table");
    if (w->dtype == HMMR_BF16) return resnet_fwd_t<bf16_t>(w, images, n, nt, phi, (char*)ws, (hipStream_t)stream, prof_ms);
    if (w->dtype == HMMR_F32) return resnet_fwd_t<float>(w, images, n, nt, phi, (char*)ws, (hipStream_t)stream, prof_ms);
    if (w->dtype == HMMR_F16X3) return resnet_fwd_t<bsplit_t>(w, images, n, nt, phi, (char*)ws, (hipStream_t)stream, prof_ms);
    hmmr_set_error("hmmr_resnet50_fwd: bad dtype %d", w->dtype);
    return -1;
}
