// Host-side packers (include/hmmr_hip.h, "Packers"; ABI 18): checkpoint-named fp32 arrays (src/evaluation/tester.py:92-116
// restores exactly these variables) -> the device layouts libhmmr_hip.so consumes, and the hmmr_*_weights_t structs filled with pointers into
// ONE blob per stage.  No HIP call in here: the caller owns the device allocation (device_base), gets the bytes in host memory and copies
// them with a single hipMemcpy -- the boundary's "never allocate" rule.  This is what human_dynamics_amd/packing.py calls for the shipped
// configuration (its Python forms remain for the development switches of devflags.py and are pinned to these bytes by
// tests/test_packers.py), so a binder in any language can fill the structs without Python.
//
// Every fold happens once, in double: inference BN -> (scale, shift) (slim batch_norm), HWIO filters -> [cout_pad][K] rows,
// split (f16x3) banks scaled per row by a power of two, the fragment streams of the one-wave-per-SIMD kernels
// (DESIGN section 3), fc1 of the IEF regressors split into phi / theta rows, SMPL's planar blend basis, folded joint regressor, ELL
// skinning weights and CSR keypoint regressor (src/tf_smpl/batch_smpl.py:35-80).
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <string>
#include <vector>

#include "hmmr_hip.h"

void hmmr_set_error(const char* fmt, ...);

namespace {

struct PackError { std::string msg; };
[[noreturn]] void fail(const std::string& m) { throw PackError{m}; }

// ---- the blob: 256-byte aligned pieces; host == NULL counts only
struct Blob {
    char* host; size_t cap; size_t off; const char* dev; bool overflow;
    Blob(void* h, size_t c, const void* d) : host((char*)h), cap(c), off(256), dev((const char*)d), overflow(false) {}
    void* alloc(size_t bytes, const void** devp) {
        off = (off + 255) & ~(size_t)255;
        void* h = nullptr;
        if (host) {
            if (off + bytes > cap) overflow = true;
            else { h = host + off; memset(h, 0, bytes); }
        }
        *devp = dev + off;
        off += bytes;
        return h;
    }
};

struct Vars {
    const hmmr_var_t* v; int n;
    const hmmr_var_t* find(const std::string& name) const {
        for (int i = 0; i < n; ++i)
            if (v[i].name && name == v[i].name) return &v[i];
        return nullptr;
    }
    bool has(const std::string& name) const { return find(name) != nullptr; }
    const float* get(const std::string& name, int64_t numel) const {
        const hmmr_var_t* e = find(name);
        if (!e || !e->data) fail("variable '" + name + "' is missing");
        if (e->numel != numel) fail("variable '" + name + "' has " + std::to_string((long long)e->numel) + " values, expected " + std::to_string((long long)numel));
        return e->data;
    }
    int64_t numel(const std::string& name) const {
        const hmmr_var_t* e = find(name);
        if (!e) fail("variable '" + name + "' is missing");
        return e->numel;
    }
};

// ---- number formats
// fp32 -> fp16, round to nearest even, subnormals kept, overflow to inf (IEEE 754 binary16: torch's and the device's conversion), by hand:
// no dependency on the host compiler's _Float16 runtime
inline uint16_t f2h(float x) {
    uint32_t u; memcpy(&u, &x, 4);
    const uint32_t sign = (u >> 16) & 0x8000u;
    u &= 0x7fffffffu;
    if (u >= 0x7f800000u) return (uint16_t)(sign | (u > 0x7f800000u ? 0x7e00u : 0x7c00u));       // NaN / inf
    if (u >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);                                       // >= 65520 rounds to inf
    if (u < 0x38800000u) {                                                                         // below 2^-14: subnormal half (or zero)
        if (u < 0x33000000u) return (uint16_t)sign;                                                // < 2^-25 rounds to zero
        const int e = (int)(u >> 23);                                                              // biased fp32 exponent, 102 .. 112
        const uint32_t m = (u & 0x7fffffu) | 0x800000u;                                            // 24-bit significand
        const int shift = 126 - e;                                                                 // 14 .. 24: value = m 2^(e - 150) = q 2^-24
        const uint32_t q = m >> shift, rem = m & ((1u << shift) - 1u), halfway = 1u << (shift - 1);
        return (uint16_t)(sign | (q + ((rem > halfway || (rem == halfway && (q & 1u))) ? 1u : 0u)));
    }
    const uint32_t v = u - 0x38000000u;                                                            // rebias 127 -> 15
    const uint32_t q = v >> 13, rem = v & 0x1fffu;
    return (uint16_t)(sign | (q + ((rem > 0x1000u || (rem == 0x1000u && (q & 1u))) ? 1u : 0u)));   // (a carry into the exponent is the right value)
}
inline float h2f(uint16_t h) {
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16, e = (h >> 10) & 0x1fu, m = h & 0x3ffu;
    uint32_t u;
    if (e == 0) {
        if (m == 0) u = sign;
        else { const float f = (float)m * 5.9604644775390625e-08f; memcpy(&u, &f, 4); u |= sign; }  // m 2^-24, exact
    } else if (e == 31) u = sign | 0x7f800000u | (m << 13);
    else u = sign | ((e + 112u) << 23) | (m << 13);
    float f; memcpy(&f, &u, 4); return f;
}
inline uint16_t f2bf(float x) {                                                                                // bf16, round to nearest even (torch's cast)
    uint32_t u; memcpy(&u, &x, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);                                  // NaN stays a NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
inline float clamp_split(float x) { return x < -65504.f ? -65504.f : (x > 65504.f ? 65504.f : x); }
inline int pad_to(int n, int m) { return (n + m - 1) / m * m; }

const float* put_f32(Blob& b, const float* src, size_t n, size_t n_pad) {
    const void* d;
    float* h = (float*)b.alloc(n_pad * 4, &d);
    if (h && src) memcpy(h, src, n * 4);
    return (const float*)d;
}
const int32_t* put_i32(Blob& b, const int32_t* src, size_t n) {
    const void* d;
    int32_t* h = (int32_t*)b.alloc(n * 4, &d);
    if (h && src) memcpy(h, src, n * 4);
    return (const int32_t*)d;
}
// a [R][K] fp32 matrix in a stage's storage type: fp32, bf16, or split (every 8 consecutive values of a row as [hi x 8][lo x 8] fp16)
const void* put_mat(Blob& b, const std::vector<float>& m, int R, int K, int dtype) {
    const void* d;
    if (dtype == HMMR_F32) {
        float* h = (float*)b.alloc((size_t)R * K * 4, &d);
        if (h) memcpy(h, m.data(), (size_t)R * K * 4);
    } else if (dtype == HMMR_BF16) {
        uint16_t* h = (uint16_t*)b.alloc((size_t)R * K * 2, &d);
        if (h) for (size_t i = 0; i < (size_t)R * K; ++i) h[i] = f2bf(m[i]);
    } else {
        if (K % 8) fail("split tensors need a channel count that is a multiple of 8");
        uint16_t* h = (uint16_t*)b.alloc((size_t)R * K * 4, &d);
        if (h) for (size_t g = 0; g < (size_t)R * K / 8; ++g)
            for (int e = 0; e < 8; ++e) {
                const float t = clamp_split(m[g * 8 + e]);
                const uint16_t hi = f2h(t);
                h[g * 16 + e] = hi;
                h[g * 16 + 8 + e] = f2h(t - h2f(hi));
            }
    }
    return d;
}

// inference BN -> (scale, shift), in double (packing.fold_bn)
void fold_bn(const Vars& v, const std::string& p, int n, std::vector<float>& s, std::vector<float>& sh) {
    const float *g = v.get(p + "/gamma", n), *be = v.get(p + "/beta", n), *m = v.get(p + "/moving_mean", n), *va = v.get(p + "/moving_variance", n);
    s.resize(n); sh.resize(n);
    for (int i = 0; i < n; ++i) {
        const double sc = (double)g[i] / sqrt((double)va[i] + 1e-5);
        s[i] = (float)sc;
        const double prod = (double)m[i] * sc;            // (two roundings, like numpy: never contracted into one fma)
        sh[i] = (float)((double)be[i] - prod);
    }
}

// HWIO [T][cin][cout] -> rows [cout_pad][T cin], K contiguous; k_order 1: chunk-major K (packing.pack_conv_weight)
void conv_rows(const float* w, int T, int cin, int cout, int k_order, int chunk, std::vector<float>& rows, int& R, int& K) {
    K = T * cin; R = pad_to(cout, 128);
    rows.assign((size_t)R * K, 0.f);
    if (k_order && cin % chunk) fail("chunk-major filters need cin % chunk == 0");
    for (int t = 0; t < T; ++t)
        for (int ci = 0; ci < cin; ++ci) {
            const int k = k_order ? ((ci / chunk) * T + t) * chunk + ci % chunk : t * cin + ci;
            const float* src = w + ((size_t)t * cin + ci) * cout;
            for (int n = 0; n < cout; ++n) rows[(size_t)n * K + k] = src[n];
        }
}
// [in][out] fully-connected weights -> rows [out_pad][in]
void fc_rows(const float* w, int in, int out, std::vector<float>& rows, int& R, int& K) { conv_rows(w, 1, in, out, 0, 32, rows, R, K); }

// per-row exponent of the power-of-two scaling of a split filter bank: max |row| * 2^k in [2^13, 2^14); zero rows: 0 (packing.row_pow2)
void row_pow2(const float* rows, int R, int K, std::vector<int>& k) {
    k.assign(R, 0);
    for (int r = 0; r < R; ++r) {
        double m = 0.0;
        for (int j = 0; j < K; ++j) { const double a = fabs((double)rows[(size_t)r * K + j]); if (a > m) m = a; }
        if (m > 0.0) { int e; frexp(m, &e); k[r] = 14 - e; }            // m = f 2^e, f in [0.5, 1): floor(log2 m) = e - 1
    }
}
inline float scale_p2(float x, int k) { return (float)ldexp((double)x, k); }

struct Vec { const float* p; };

// one hmmr_layer_t from rows [R][K] (R a multiple of 128) + optional scale[ns] / shift[nsh] (packing._layer)
hmmr_layer_t layer(Blob& b, std::vector<float>& rows, int R, int K, int dtype, const float* scale, int ns, const float* shift, int nsh) {
    hmmr_layer_t l = {};
    std::vector<float> sc;
    if (dtype == HMMR_F16X3) {
        std::vector<int> k;
        row_pow2(rows.data(), R, K, k);
        for (int r = 0; r < R; ++r)
            if (k[r]) for (int j = 0; j < K; ++j) rows[(size_t)r * K + j] = scale_p2(rows[(size_t)r * K + j], k[r]);
        sc.assign(R, 1.f);
        for (int r = 0; r < R; ++r) {
            const double s0 = (scale && r < ns) ? (double)scale[r] : 1.0;
            sc[r] = (float)(s0 * ldexp(1.0, -k[r]));
        }
        scale = sc.data(); ns = R;
    }
    l.w = put_mat(b, rows, R, K, dtype);
    l.scale = scale ? put_f32(b, scale, ns, pad_to(ns, 128)) : nullptr;
    l.shift = shift ? put_f32(b, shift, nsh, pad_to(nsh, 128)) : nullptr;
    return l;
}

// the scaled value of W[t][ci][n] as fp16 hi / lo halves
struct HL { uint16_t hi, lo; };
inline HL split_w(float w, int k) {
    const float t = scale_p2(w, k);
    const uint16_t hi = f2h(t);
    return HL{hi, f2h(t - h2f(hi))};
}

// the filter stream of a k_order 2 layer: [cout / tw][T cin / 16][tw / 32][2 planes][64 lanes][8] fp16 (packing.pack_conv3x3_stream)
const void* stream_f16(Blob& b, const float* w, int T, int cin, int cout, const std::vector<int>& k) {
    if (cin % 16 || !(cout % 128 == 0 || cout == 64)) fail("filter stream: cin % 16, cout % 128 (or 64)");
    const int tw = cout % 128 == 0 ? 128 : 64, nrb = tw / 32, nkt = T * (cin / 16);
    const void* d;
    uint16_t* h = (uint16_t*)b.alloc((size_t)T * cin * cout * 4, &d);
    if (h)
        for (int tile = 0; tile < cout / tw; ++tile)
            for (int c16 = 0; c16 < cin / 16; ++c16)
                for (int t = 0; t < T; ++t)
                    for (int rb = 0; rb < nrb; ++rb) {
                        uint16_t* f = h + ((((size_t)tile * nkt + c16 * T + t) * nrb + rb) * 2) * 512;        // hi plane; lo plane + 512
                        for (int half = 0; half < 2; ++half)
                            for (int row = 0; row < 32; ++row) {
                                const int n = tile * tw + rb * 32 + row;
                                for (int e = 0; e < 8; ++e) {
                                    const int ci = 16 * c16 + 8 * half + e;
                                    const HL v = split_w(w[((size_t)t * cin + ci) * cout + n], k[n]);
                                    f[(half * 32 + row) * 8 + e] = v.hi;
                                    f[512 + (half * 32 + row) * 8 + e] = v.lo;
                                }
                            }
                    }
    return d;
}
// bf16 form: [cout / tw][9 cin / 32][tw / 32][2 (16-wide chunks)][64][8] bfloat16, no row scaling
const void* stream_bf16(Blob& b, const float* w, int cin, int cout) {
    if (cin % 32 || !(cout % 128 == 0 || cout == 64)) fail("bf16 filter stream: cin % 32, cout % 128 (or 64)");
    const int tw = cout % 128 == 0 ? 128 : 64, nrb = tw / 32, nkt = 9 * (cin / 32);
    const void* d;
    uint16_t* h = (uint16_t*)b.alloc((size_t)9 * cin * cout * 2, &d);
    if (h)
        for (int tile = 0; tile < cout / tw; ++tile)
            for (int c32 = 0; c32 < cin / 32; ++c32)
                for (int t = 0; t < 9; ++t)
                    for (int rb = 0; rb < nrb; ++rb)
                        for (int plane = 0; plane < 2; ++plane) {
                            uint16_t* f = h + ((((size_t)tile * nkt + c32 * 9 + t) * nrb + rb) * 2 + plane) * 512;
                            for (int half = 0; half < 2; ++half)
                                for (int row = 0; row < 32; ++row)
                                    for (int e = 0; e < 8; ++e) {
                                        const int ci = 32 * c32 + 16 * plane + 8 * half + e, n = tile * tw + rb * 32 + row;
                                        f[(half * 32 + row) * 8 + e] = f2bf(w[((size_t)t * cin + ci) * cout + n]);
                                    }
                        }
    return d;
}
// exponents of the rows of an HWIO bank (row n = output channel n over all of K)
void hwio_pow2(const float* w, int T, int cin, int cout, std::vector<int>& k) {
    k.assign(cout, 0);
    std::vector<double> m(cout, 0.0);
    for (size_t i = 0; i < (size_t)T * cin; ++i)
        for (int n = 0; n < cout; ++n) { const double a = fabs((double)w[i * cout + n]); if (a > m[n]) m[n] = a; }
    for (int n = 0; n < cout; ++n)
        if (m[n] > 0.0) { int e; frexp(m[n], &e); k[n] = 14 - e; }
}
// hmmr_layer_t of a k_order 2 layer (packing._layer_stream3x3 / _layer_stream1x1): scale NULL = ones
hmmr_layer_t layer_stream(Blob& b, const float* w, int T, int cin, int cout, const float* scale, const float* shift, bool bf16) {
    hmmr_layer_t l = {};
    std::vector<float> sc(cout);
    if (bf16) {
        l.w = stream_bf16(b, w, cin, cout);
        for (int n = 0; n < cout; ++n) sc[n] = scale ? scale[n] : 1.f;
    } else {
        std::vector<int> k;
        hwio_pow2(w, T, cin, cout, k);
        l.w = stream_f16(b, w, T, cin, cout, k);
        for (int n = 0; n < cout; ++n) sc[n] = (float)((scale ? (double)scale[n] : 1.0) * ldexp(1.0, -k[n]));
    }
    l.scale = put_f32(b, sc.data(), cout, pad_to(cout, 128));
    l.shift = put_f32(b, shift, cout, pad_to(cout, 128));
    l.k_order = 2;
    return l;
}

// fragments of a split filter bank w_nk [n][K]: value(rb, kc, plane, lane = 32 half + row, e) = plane(2^k w[32 rb + row][16 kc + 8 half + e])
struct Frags {
    int n, K; const float* w; std::vector<int> k;
    Frags(const float* w_, int n_, int K_) : n(n_), K(K_), w(w_) { row_pow2(w, n, K, k); }
    void write(uint16_t* dst_hi, uint16_t* dst_lo, int rb, int kc) const {       // 512 halves per plane, lane-linear
        for (int half = 0; half < 2; ++half)
            for (int row = 0; row < 32; ++row)
                for (int e = 0; e < 8; ++e) {
                    const int r = 32 * rb + row;
                    const HL v = split_w(w[(size_t)r * K + 16 * kc + 8 * half + e], k[r]);
                    dst_hi[(half * 32 + row) * 8 + e] = v.hi;
                    dst_lo[(half * 32 + row) * 8 + e] = v.lo;
                }
    }
};
// [n / 32][K / 16][64 lanes][2 (hi, lo)][8]: one coalesced 2 KB read per fragment (packing.pack_frag_major)
const void* frag_major(Blob& b, const float* w_nk, int n, int K) {
    if (n % 32 || K % 16) fail("fragment-major filters: n % 32, K % 16");
    const void* d;
    uint16_t* h = (uint16_t*)b.alloc((size_t)n * K * 4, &d);
    if (h) {
        Frags f(w_nk, n, K);
        std::vector<uint16_t> hi(512), lo(512);
        for (int rb = 0; rb < n / 32; ++rb)
            for (int kc = 0; kc < K / 16; ++kc) {
                f.write(hi.data(), lo.data(), rb, kc);
                uint16_t* o = h + ((size_t)rb * (K / 16) + kc) * 1024;
                for (int lane = 0; lane < 64; ++lane) { memcpy(o + lane * 16, hi.data() + lane * 8, 16); memcpy(o + lane * 16 + 8, lo.data() + lane * 8, 16); }
            }
    }
    return d;
}
bool pair_is_a(int i, int na, int ft) { return ((i + 1) * na) / ft > (i * na) / ft; }
// the filter stream of a unit pair (packing.pack_pair_stream): [depth / 32 + 2][ft][2 planes][64][8]
const void* pair_stream(Blob& b, const float* w3_nk, int depth, int K3, const float* w1_nk, int n2) {
    const int nch = depth / 32, na = K3 / 16, nf2 = n2 / 32, nb = 2 * nf2, ft = na + nb;
    if (depth % 32 || K3 % 16 || n2 % 32 || ft % 8) fail("pair stream: shapes");
    const void* d;
    uint16_t* h = (uint16_t*)b.alloc((size_t)(nch + 2) * ft * 2048, &d);
    if (h) {
        Frags f3(w3_nk, depth, K3), f1(w1_nk, n2, depth);
        for (int it = 0; it < nch + 2; ++it) {
            int a = 0, kb = 0;
            for (int i = 0; i < ft; ++i) {
                uint16_t* o = h + ((size_t)it * ft + i) * 1024;
                if (pair_is_a(i, na, ft)) { if (it < nch) f3.write(o, o + 512, it, a); ++a; }
                else { if (it >= 2) f1.write(o, o + 512, kb % nf2, (kb / nf2) + 2 * (it - 2)); ++kb; }
            }
        }
    }
    return d;
}
// the filter stream of a whole block-1 unit (packing.pack_b1_unit_stream)
const void* b1_unit_stream(Blob& b, const float* w2_hwio, const float* w3_nk, int K3, const float* w1_nk) {
    const int depth = 256, nch = 8;
    const void* d;
    const size_t halves = (size_t)72 * 1024 + (size_t)nch * (K3 / 16 + 4) * 1024;
    uint16_t* h = (uint16_t*)b.alloc(halves * 2, &d);
    if (h) {
        std::vector<int> k2;
        hwio_pow2(w2_hwio, 9, 64, 64, k2);
        // conv2's stream exactly as stream_f16 lays it out for cout = 64 (36 K steps x two row blocks)
        for (int c16 = 0; c16 < 4; ++c16)
            for (int t = 0; t < 9; ++t)
                for (int rb = 0; rb < 2; ++rb) {
                    uint16_t* f = h + (((size_t)(c16 * 9 + t) * 2 + rb) * 2) * 512;
                    for (int half = 0; half < 2; ++half)
                        for (int row = 0; row < 32; ++row)
                            for (int e = 0; e < 8; ++e) {
                                const int ci = 16 * c16 + 8 * half + e, n = rb * 32 + row;
                                const HL v = split_w(w2_hwio[((size_t)t * 64 + ci) * 64 + n], k2[n]);
                                f[(half * 32 + row) * 8 + e] = v.hi;
                                f[512 + (half * 32 + row) * 8 + e] = v.lo;
                            }
                }
        Frags f3(w3_nk, depth, K3), f1(w1_nk, 64, depth);
        uint16_t* o = h + (size_t)72 * 1024;
        auto put_a = [&](int c) { for (int kc = 0; kc < K3 / 16; ++kc) { f3.write(o, o + 512, c, kc); o += 1024; } };
        put_a(0);
        for (int c = 0; c < nch; ++c) {
            if (c + 1 < nch) put_a(c + 1);
            for (int kcl = 0; kcl < 2; ++kcl)
                for (int j = 0; j < 2; ++j) { f1.write(o, o + 512, j, 2 * c + kcl); o += 1024; }
        }
    }
    return d;
}

// transpose of a 1x1 HWIO bank [cin][cout] -> [cout][cin]
std::vector<float> transpose(const float* w, int rows, int cols) {
    std::vector<float> t((size_t)rows * cols);
    for (int r = 0; r < rows; ++r)
        for (int c = 0; c < cols; ++c) t[(size_t)c * rows + r] = w[(size_t)r * cols + c];
    return t;
}

struct UnitShape { std::string scope; int c_in, base, depth, stride; bool has_sc; int block; };
std::vector<UnitShape> resnet_units() {
    static const int bases[4] = {64, 128, 256, 512}, counts[4] = {3, 4, 6, 3}, bstride[4] = {2, 2, 2, 1};
    std::vector<UnitShape> u;
    int c_in = 64;
    for (int b = 0; b < 4; ++b)
        for (int i = 1; i <= counts[b]; ++i) {
            UnitShape s;
            s.scope = "resnet_v2_50/block" + std::to_string(b + 1) + "/unit_" + std::to_string(i) + "/bottleneck_v2";
            s.c_in = c_in; s.base = bases[b]; s.depth = 4 * bases[b]; s.stride = (i == counts[b]) ? bstride[b] : 1; s.has_sc = c_in != s.depth; s.block = b + 1;
            u.push_back(s);
            c_in = s.depth;
        }
    return u;
}

// ------------------------------------------------------------------------------------------------------------------ ResNet-v2-50
void pack_resnet(const Vars& v, int dtype, Blob& b, hmmr_resnet_weights_t* rw) {
    memset(rw, 0, sizeof(*rw));
    rw->dtype = dtype;
    const bool x3 = dtype == HMMR_F16X3, b16 = dtype == HMMR_BF16;
    const int bke = b16 ? 64 : 32;
    std::vector<float> rows, s, sh;
    int R, K;
    {   // stem: [7,7,3,64] -> [128][8 taps x 32]: k = ky 32 + kx 4 + c (packing.pack_stem_weight)
        const float* w = v.get("resnet_v2_50/conv1/weights", 7 * 7 * 3 * 64);
        rows.assign((size_t)128 * 256, 0.f);
        for (int ky = 0; ky < 7; ++ky)
            for (int kx = 0; kx < 7; ++kx)
                for (int c = 0; c < 3; ++c)
                    for (int n = 0; n < 64; ++n) rows[(size_t)n * 256 + ky * 32 + kx * 4 + c] = w[((ky * 7 + kx) * 3 + c) * 64 + n];
        rw->stem = layer(b, rows, 128, 256, dtype, nullptr, 0, v.get("resnet_v2_50/conv1/biases", 64), 64);
    }
    const std::vector<UnitShape> units = resnet_units();
    for (int i = 0; i < HMMR_RESNET_UNITS; ++i) {
        const UnitShape& U = units[i];
        hmmr_resnet_unit_t& u = rw->unit[i];
        const std::string& sc = U.scope;
        u.c_in = U.c_in; u.base = U.base; u.depth = U.depth; u.stride = U.stride;
        u.fuse_preact = (i > 0 && !U.has_sc) ? 1 : 0;                // a block's first unit reads the tensor the previous conv3 wrote
        const float* w1 = v.get(sc + "/conv1/weights", (int64_t)U.c_in * U.base);
        fold_bn(v, sc + "/conv1/BatchNorm", U.base, s, sh);
        const bool s1x1 = x3 && U.stride == 1;
        if (s1x1 && i > 0 && U.base % 128 == 0 && (U.base == 512 || !u.fuse_preact)) {
            u.fuse_preact = 0;
            u.conv1 = layer_stream(b, w1, 1, U.c_in, U.base, s.data(), sh.data(), false);
        } else {
            conv_rows(w1, 1, U.c_in, U.base, 0, bke, rows, R, K);
            u.conv1 = layer(b, rows, R, K, dtype, s.data(), U.base, sh.data(), U.base);
        }
        if (i == 0 && x3) {
            const std::vector<float> t = transpose(w1, U.c_in, U.base);
            u.conv1_frag = frag_major(b, t.data(), U.base, U.c_in);
        }
        const float* w2 = v.get(sc + "/conv2/weights", (int64_t)9 * U.base * U.base);
        fold_bn(v, sc + "/conv2/BatchNorm", U.base, s, sh);
        const bool stream2 = U.stride == 1 && (x3 || (b16 && U.base >= 256));
        if (stream2) u.conv2 = layer_stream(b, w2, 9, U.base, U.base, s.data(), sh.data(), b16);
        else {
            conv_rows(w2, 9, U.base, U.base, 0, bke, rows, R, K);
            u.conv2 = layer(b, rows, R, K, dtype, s.data(), U.base, sh.data(), U.base);
        }
        const float* w3 = v.get(sc + "/conv3/weights", (int64_t)U.base * U.depth);
        const float* b3 = v.get(sc + "/conv3/biases", U.depth);
        const bool s3x = s1x1 && U.base == 512;
        if (s3x) u.conv3 = layer_stream(b, w3, 1, U.base, U.depth, nullptr, b3, false);
        else {
            conv_rows(w3, 1, U.base, U.depth, 0, bke, rows, R, K);
            u.conv3 = layer(b, rows, R, K, dtype, nullptr, 0, b3, U.depth);
        }
        if (U.has_sc) {
            const float* wsc = v.get(sc + "/shortcut/weights", (int64_t)U.c_in * U.depth);
            const float* bsc = v.get(sc + "/shortcut/biases", U.depth);
            conv_rows(wsc, 1, U.c_in, U.depth, 0, bke, rows, R, K);
            u.shortcut = layer(b, rows, R, K, dtype, nullptr, 0, bsc, U.depth);
            bool folded = x3 && U.stride == 1 && !u.fuse_preact && U.base % bke == 0 && U.c_in % bke == 0;
            if (x3 && U.base == 256) folded = false;                 // the unit pair keeps 32 px x K in registers: K = 256 + 512 does not fit
            if (folded) {
                std::vector<float> both((size_t)(U.base + U.c_in) * U.depth), bias(U.depth);
                memcpy(both.data(), w3, (size_t)U.base * U.depth * 4);
                memcpy(both.data() + (size_t)U.base * U.depth, wsc, (size_t)U.c_in * U.depth * 4);
                for (int n = 0; n < U.depth; ++n) bias[n] = (float)((double)b3[n] + (double)bsc[n]);
                if (s3x) u.c3sc = layer_stream(b, both.data(), 1, U.base + U.c_in, U.depth, nullptr, bias.data(), false);
                else {
                    conv_rows(both.data(), 1, U.base + U.c_in, U.depth, 0, bke, rows, R, K);
                    u.c3sc = layer(b, rows, R, K, dtype, nullptr, 0, bias.data(), U.depth);
                }
            }
            if (!folded && U.stride == 1 && U.c_in >= 512) {         // shortcut + conv1 as one column-split GEMM (measured: pays in blocks 3-4)
                const int nc = U.depth + U.base;
                std::vector<float> both((size_t)U.c_in * nc), ss(nc), sb(nc), s1, b1;
                for (int ci = 0; ci < U.c_in; ++ci) {
                    memcpy(both.data() + (size_t)ci * nc, wsc + (size_t)ci * U.depth, (size_t)U.depth * 4);
                    memcpy(both.data() + (size_t)ci * nc + U.depth, w1 + (size_t)ci * U.base, (size_t)U.base * 4);
                }
                fold_bn(v, sc + "/conv1/BatchNorm", U.base, s1, b1);
                for (int n = 0; n < U.depth; ++n) { ss[n] = 1.f; sb[n] = bsc[n]; }
                for (int n = 0; n < U.base; ++n) { ss[U.depth + n] = s1[n]; sb[U.depth + n] = b1[n]; }
                if (s1x1 && !u.fuse_preact && U.depth % 128 == 0 && U.base % 128 == 0) {
                    u.sc_c1 = layer_stream(b, both.data(), 1, U.c_in, nc, ss.data(), sb.data(), false);
                    u.shortcut.k_order = 2;
                } else {
                    conv_rows(both.data(), 1, U.c_in, nc, 0, bke, rows, R, K);
                    u.sc_c1 = layer(b, rows, R, K, dtype, ss.data(), nc, sb.data(), nc);
                }
            }
        }
        fold_bn(v, sc + "/preact", U.c_in, s, sh);
        u.pre_scale = put_f32(b, s.data(), U.c_in, pad_to(U.c_in, 128));
        u.pre_shift = put_f32(b, sh.data(), U.c_in, pad_to(U.c_in, 128));
    }
    if (x3) {
        // conv3 + add + the next unit's preact + conv1 as one launch: unit pairs (blocks 2-3), whole units (block 1)
        for (int i = 0; i + 1 < HMMR_RESNET_UNITS; ++i) {
            const UnitShape &U = units[i], &N = units[i + 1];
            hmmr_resnet_unit_t &u = rw->unit[i], &nx = rw->unit[i + 1];
            const bool chain = U.stride == 1 && N.c_in == U.depth && N.base == U.base && nx.fuse_preact == 1 && !nx.shortcut.w;
            const bool pair_shape = (U.base == 128 && U.depth == 512) || (U.base == 256 && U.depth == 1024);
            const bool pair = chain && pair_shape && (!u.c3sc.w || (U.base == 128 && U.c_in == 256));
            const bool ok = chain && U.base == 64 && (!U.has_sc || (u.c3sc.w && U.c_in == 64));
            if (!(ok || pair)) continue;
            const float* w3 = v.get(U.scope + "/conv3/weights", (int64_t)U.base * U.depth);
            int K3 = U.base;
            std::vector<float> w3k((size_t)(U.base + (u.c3sc.w ? U.c_in : 0)) * U.depth);
            memcpy(w3k.data(), w3, (size_t)U.base * U.depth * 4);
            if (u.c3sc.w) {
                memcpy(w3k.data() + (size_t)U.base * U.depth, v.get(U.scope + "/shortcut/weights", (int64_t)U.c_in * U.depth), (size_t)U.c_in * U.depth * 4);
                K3 += U.c_in;
            }
            const std::vector<float> w3_nk = transpose(w3k.data(), K3, U.depth);                                               // [depth][K3]
            const std::vector<float> w1_nk = transpose(v.get(N.scope + "/conv1/weights", (int64_t)U.depth * U.base), U.depth, U.base);   // [base][depth]
            if (pair) {
                u.pair_stream = pair_stream(b, w3_nk.data(), U.depth, K3, w1_nk.data(), U.base);
                u.fuse_tail = 1;
            } else {
                u.unit_stream = b1_unit_stream(b, v.get(U.scope + "/conv2/weights", 9 * 64 * 64), w3_nk.data(), K3, w1_nk.data());
                u.fuse_tail = 2;
            }
        }
    } else if (b16) {
        for (int i = 0; i + 1 < HMMR_RESNET_UNITS; ++i) {
            hmmr_resnet_unit_t &u = rw->unit[i], &nx = rw->unit[i + 1];
            const bool shape = (u.base == 64 && u.depth == 256) || (u.base == 128 && u.depth == 512);
            if (u.stride == 1 && shape && nx.c_in == u.depth && nx.base == u.base && nx.fuse_preact == 1 && !nx.shortcut.w) {
                u.fuse_tail = 2;                                      // the unit's 3x3 conv2 inside the same launch
                if (u.shortcut.w && u.c_in == 64 && !u.fuse_preact) u.fuse_tail = 3;        // ... and its conv shortcut (block1/unit_1)
            }
        }
        for (int i = 0; i + 1 < HMMR_RESNET_UNITS; ++i) {
            hmmr_resnet_unit_t& u = rw->unit[i];
            const bool shape = (u.base == 64 && u.depth == 256) || (u.base == 128 && u.depth == 512);
            if (u.stride == 2 && !u.shortcut.w && shape) u.fuse_tail = 4;                    // a block's stride-2 last unit
        }
    }
    fold_bn(v, "resnet_v2_50/postnorm", 2048, s, sh);
    rw->post_scale = put_f32(b, s.data(), 2048, 2048);
    rw->post_shift = put_f32(b, sh.data(), 2048, 2048);
}

// ------------------------------------------------------------------------------------------------------------------ f_movie
void pack_temporal(const Vars& v, int dtype, int num_blocks, Blob& b, hmmr_temporal_weights_t* tw) {
    memset(tw, 0, sizeof(*tw));
    if (num_blocks < 1 || num_blocks > HMMR_MAX_TEMPORAL_BLOCKS) fail("num_blocks out of range");
    tw->dtype = dtype; tw->num_blocks = num_blocks;
    std::vector<float> rows;
    int R, K;
    for (int i = 0; i < num_blocks; ++i) {
        const std::string n = "block_" + std::to_string(i);         // (scope strings are concatenated without a separator: src/models.py:159,182)
        const std::string gn1 = "AZ_FC_block_preact_gn1" + n, c1 = "AZ_FC_block2_conv1" + n, gn2 = "AZ_FC_block_preact_gn2" + n, c2 = "AZ_FC_block2_conv2" + n;
        hmmr_temporal_block_t& t = tw->block[i];
        t.gn1_gamma = put_f32(b, v.get(gn1 + "/gamma", 2048), 2048, 2048);
        t.gn1_beta = put_f32(b, v.get(gn1 + "/beta", 2048), 2048, 2048);
        t.gn2_gamma = put_f32(b, v.get(gn2 + "/gamma", 2048), 2048, 2048);
        t.gn2_beta = put_f32(b, v.get(gn2 + "/beta", 2048), 2048, 2048);
        conv_rows(v.get(c1 + "/weights", (int64_t)3 * 2048 * 2048), 3, 2048, 2048, 0, 32, rows, R, K);
        t.conv1 = layer(b, rows, R, K, dtype, nullptr, 0, v.get(c1 + "/biases", 2048), 2048);
        conv_rows(v.get(c2 + "/weights", (int64_t)3 * 2048 * 2048), 3, 2048, 2048, 0, 32, rows, R, K);
        t.conv2 = layer(b, rows, R, K, dtype, nullptr, 0, v.get(c2 + "/biases", 2048), 2048);
        if (dtype == HMMR_F16X3) t.conv1.tile = t.conv2.tile = 8;   // the 128x256 ping-pong tile with 5 K slices (csrc/temporal.hip)
    }
}

void pack_hallucinator(const Vars& v, int dtype, Blob& b, hmmr_hallucinator_weights_t* hw) {
    memset(hw, 0, sizeof(*hw));
    hw->dtype = dtype;
    std::vector<float> rows;
    int R, K;
    hmmr_layer_t* l[3] = {&hw->fc1, &hw->fc2, &hw->fc3};
    for (int i = 0; i < 3; ++i) {
        const std::string p = "fc2_res/fc" + std::to_string(i + 1);
        fc_rows(v.get(p + "/weights", (int64_t)2048 * 2048), 2048, 2048, rows, R, K);
        *l[i] = layer(b, rows, R, K, dtype, nullptr, 0, v.get(p + "/biases", 2048), 2048);
    }
}

// ------------------------------------------------------------------------------------------------------------------ IEF regressors
void pack_ief(const Vars& v, int dtype, const int* delta_t, int n_delta, int num_stages, Blob& b, hmmr_ief_weights_t* iw) {
    memset(iw, 0, sizeof(*iw));
    if (n_delta < 0 || n_delta + 1 > HMMR_MAX_REGRESSORS) fail("too many delta regressors");
    std::vector<int> keys(1, 0);
    {
        std::vector<int> ds(delta_t, delta_t + n_delta);
        for (size_t i = 0; i < ds.size(); ++i)
            for (size_t j = i + 1; j < ds.size(); ++j) if (ds[j] < ds[i]) { int t = ds[i]; ds[i] = ds[j]; ds[j] = t; }      // sorted order (tester.py:245)
        keys.insert(keys.end(), ds.begin(), ds.end());
    }
    iw->dtype = dtype; iw->num_regressors = (int)keys.size(); iw->num_stages = num_stages;
    std::vector<float> rows;
    int R, K, nd_delta = -1;
    for (size_t r = 0; r < keys.size(); ++r) {
        const int key = keys[r];
        const std::string scope = key == 0 ? "single_view_ief" : (key > 0 ? "single_view_ief_future" + std::to_string(key) : "single_view_ief_past" + std::to_string(-key));
        const std::string p = scope + "/3D_module";
        int nd = 85;
        if (key != 0) {
            nd = (int)(v.numel(p + "/fc1/weights") / 1024) - 2048;  // 72 (use_optcam) or 75 (models.py:333-336): the checkpoint decides
            if (nd != 72 && nd != 75) fail(p + "/fc1/weights: a delta regressor takes 2048 + 72 or 2048 + 75 rows");
            if (nd_delta >= 0 && nd != nd_delta) fail("the delta regressors disagree on use_optcam");
            nd_delta = nd;
        }
        const float* W1 = v.get(p + "/fc1/weights", (int64_t)(2048 + nd) * 1024);
        hmmr_ief_regressor_t& g = iw->reg[r];
        g.nd = nd;
        fc_rows(W1, 2048, 1024, rows, R, K);                          // the phi rows of fc1
        g.fc1_phi = layer(b, rows, R, K, dtype, nullptr, 0, v.get(p + "/fc1/biases", 1024), 1024);
        rows.assign((size_t)1024 * 128, 0.f);                         // the theta rows, zero padded to K = 128; this path stays fp32
        for (int j = 0; j < nd; ++j)
            for (int n = 0; n < 1024; ++n) rows[(size_t)n * 128 + j] = W1[(size_t)(2048 + j) * 1024 + n];
        g.fc1_theta = layer(b, rows, 1024, 128, HMMR_F32, nullptr, 0, nullptr, 0);
        fc_rows(v.get(p + "/fc2/weights", (int64_t)1024 * 1024), 1024, 1024, rows, R, K);
        g.fc2 = layer(b, rows, R, K, dtype, nullptr, 0, v.get(p + "/fc2/biases", 1024), 1024);
        fc_rows(v.get(p + "/fc3/weights", (int64_t)1024 * nd), 1024, nd, rows, R, K);
        g.fc3 = layer(b, rows, R, K, dtype, nullptr, 0, v.get(p + "/fc3/biases", nd), nd);
    }
    iw->mean_theta = put_f32(b, v.get("mean_param", 85), 85, 85);
    iw->no_optcam = nd_delta == 75 ? 1 : 0;
}

// ------------------------------------------------------------------------------------------------------------------ SMPL
void pack_smpl(const hmmr_smpl_source_t* s, int lsp, int split, Blob& b, hmmr_smpl_consts_t* sc) {
    memset(sc, 0, sizeof(*sc));
    const int nv = s->num_verts, vpad = pad_to(nv, 256), nkall = s->num_kps;
    if (nv < 1 || nkall < 1 || !s->v_template || !s->shapedirs || !s->posedirs || !s->J_regressor || !s->lbs_weights || !s->kp_regressor || !s->parents)
        fail("hmmr_pack_smpl: incomplete source");
    // dirs [224][3][vpad]: row 0 v_template, 1..10 shapedirs, 11..217 posedirs; dirs[k][c][v] = basis[k][3 v + c]
    const void* d;
    float* dirs = (float*)b.alloc((size_t)224 * 3 * vpad * 4, &d);
    sc->dirs = (const float*)d;
    std::vector<float> dl;
    if (!dirs) { dl.assign(0, 0.f); }
    double amax = 0.0;
    auto at = [&](int k, int c, int vv) -> float {
        return k == 0 ? s->v_template[3 * vv + c] : (k <= 10 ? s->shapedirs[(size_t)(k - 1) * 3 * nv + 3 * vv + c] : s->posedirs[(size_t)(k - 11) * 3 * nv + 3 * vv + c]);
    };
    for (int k = 0; k < 218; ++k)
        for (int c = 0; c < 3; ++c)
            for (int vv = 0; vv < nv; ++vv) {
                const float x = at(k, c, vv);
                if (dirs) dirs[((size_t)k * 3 + c) * vpad + vv] = x;
                if (fabs((double)x) > amax) amax = fabs((double)x);
            }
    // folded joint regressor, in double: j_template = Jreg^T v_template [24][3]; j_shapedirs[b] = Jreg^T S_b
    std::vector<float> jt(72), js(720);
    for (int j = 0; j < 24; ++j)
        for (int c = 0; c < 3; ++c) {
            double a = 0.0;
            for (int vv = 0; vv < nv; ++vv) a += (double)s->J_regressor[(size_t)vv * 24 + j] * (double)s->v_template[3 * vv + c];
            jt[j * 3 + c] = (float)a;
            for (int k = 0; k < 10; ++k) {
                double q = 0.0;
                for (int vv = 0; vv < nv; ++vv) q += (double)s->J_regressor[(size_t)vv * 24 + j] * (double)s->shapedirs[(size_t)k * 3 * nv + 3 * vv + c];
                js[k * 72 + j * 3 + c] = (float)q;
            }
        }
    // ELL skinning weights: ascending joint order, like the dense sum
    int nnz = 1;
    for (int vv = 0; vv < nv; ++vv) { int c = 0; for (int j = 0; j < 24; ++j) c += s->lbs_weights[(size_t)vv * 24 + j] != 0.f; if (c > nnz) nnz = c; }
    std::vector<int32_t> idx((size_t)nv * nnz, 0);
    std::vector<float> val((size_t)nv * nnz, 0.f);
    for (int vv = 0; vv < nv; ++vv) {
        int c = 0;
        for (int j = 0; j < 24; ++j)
            if (s->lbs_weights[(size_t)vv * 24 + j] != 0.f) { idx[(size_t)vv * nnz + c] = j; val[(size_t)vv * nnz + c] = s->lbs_weights[(size_t)vv * 24 + j]; ++c; }
    }
    // keypoint regressor [nv][K] -> CSR by keypoint
    const int nk = lsp ? 14 : nkall;                                  // batch_smpl.py:81-82
    std::vector<int32_t> kptr(1, 0), kidx;
    std::vector<float> kval;
    for (int k = 0; k < nk; ++k) {
        for (int vv = 0; vv < nv; ++vv) {
            const float x = s->kp_regressor[(size_t)vv * nkall + k];
            if (x != 0.f) { kidx.push_back(vv); kval.push_back(x); }
        }
        kptr.push_back((int32_t)kidx.size());
    }
    if (kidx.empty()) { kidx.push_back(0); kval.push_back(0.f); }
    sc->num_verts = nv; sc->num_kps = nk; sc->lbs_nnz = nnz; sc->vpad = vpad;
    if (split && amax * 8192.0 < 65504.0) {
        // [14][3][hi, lo][k half][vpad][8]: plane(2^13 dirs[16 kc + 8 h + e][c][v])
        uint16_t* ds = (uint16_t*)b.alloc((size_t)14 * 3 * 2 * 2 * vpad * 8 * 2, &d);
        sc->dirs_split = d;
        if (ds && dirs)
            for (int kc = 0; kc < 14; ++kc)
                for (int c = 0; c < 3; ++c)
                    for (int h = 0; h < 2; ++h)
                        for (int vv = 0; vv < vpad; ++vv)
                            for (int e = 0; e < 8; ++e) {
                                const int k = 16 * kc + 8 * h + e;
                                const float x = (float)((double)dirs[((size_t)k * 3 + c) * vpad + vv] * 8192.0);
                                const uint16_t hi = f2h(x);
                                const size_t o = (((((size_t)kc * 3 + c) * 2 + 0) * 2 + h) * vpad + vv) * 8 + e;
                                ds[o] = hi;
                                ds[o + (size_t)2 * vpad * 8] = f2h(x - h2f(hi));
                            }
    }
    sc->j_template = put_f32(b, jt.data(), 72, 72);
    sc->j_shapedirs = put_f32(b, js.data(), 720, 720);
    sc->parents = put_i32(b, s->parents, 24);
    sc->lbs_idx = put_i32(b, idx.data(), idx.size());
    sc->lbs_w = put_f32(b, val.data(), val.size(), val.size());
    sc->kreg_ptr = put_i32(b, kptr.data(), kptr.size());
    sc->kreg_idx = put_i32(b, kidx.data(), kidx.size());
    sc->kreg_val = put_f32(b, kval.data(), kval.size(), kval.size());
}

template <typename F> int guarded(const char* what, Blob& b, F&& f) {
    try { f(); }
    catch (const PackError& e) { hmmr_set_error("%s: %s", what, e.msg.c_str()); return -1; }
    catch (const std::exception& e) { hmmr_set_error("%s: %s", what, e.what()); return -1; }
    if (b.overflow) { hmmr_set_error("%s: the blob is too small (%zu bytes needed)", what, b.off); return -1; }
    return 0;
}
template <typename F> size_t counted(const char* what, F&& f) {
    Blob b(nullptr, 0, nullptr);
    try { f(b); }
    catch (const PackError& e) { hmmr_set_error("%s: %s", what, e.msg.c_str()); return 0; }
    catch (const std::exception& e) { hmmr_set_error("%s: %s", what, e.what()); return 0; }
    return (b.off + 255) & ~(size_t)255;
}

}  // namespace

extern "C" size_t hmmr_pack_resnet_bytes(const hmmr_var_t* vars, int n_vars, int dtype) {
    return counted("hmmr_pack_resnet_bytes", [&](Blob& b) { hmmr_resnet_weights_t rw; pack_resnet(Vars{vars, n_vars}, dtype, b, &rw); });
}
extern "C" int hmmr_pack_resnet(const hmmr_var_t* vars, int n_vars, int dtype, void* host_blob, size_t blob_bytes, const void* device_base,
                                hmmr_resnet_weights_t* out) {
    if (!vars || !host_blob || !out) { hmmr_set_error("hmmr_pack_resnet: null argument"); return -1; }
    Blob b(host_blob, blob_bytes, device_base);
    return guarded("hmmr_pack_resnet", b, [&] { pack_resnet(Vars{vars, n_vars}, dtype, b, out); });
}
extern "C" size_t hmmr_pack_temporal_bytes(const hmmr_var_t* vars, int n_vars, int dtype, int num_blocks) {
    return counted("hmmr_pack_temporal_bytes", [&](Blob& b) { hmmr_temporal_weights_t t; pack_temporal(Vars{vars, n_vars}, dtype, num_blocks, b, &t); });
}
extern "C" int hmmr_pack_temporal(const hmmr_var_t* vars, int n_vars, int dtype, int num_blocks, void* host_blob, size_t blob_bytes, const void* device_base,
                                  hmmr_temporal_weights_t* out) {
    if (!vars || !host_blob || !out) { hmmr_set_error("hmmr_pack_temporal: null argument"); return -1; }
    Blob b(host_blob, blob_bytes, device_base);
    return guarded("hmmr_pack_temporal", b, [&] { pack_temporal(Vars{vars, n_vars}, dtype, num_blocks, b, out); });
}
extern "C" size_t hmmr_pack_hallucinator_bytes(const hmmr_var_t* vars, int n_vars, int dtype) {
    return counted("hmmr_pack_hallucinator_bytes", [&](Blob& b) { hmmr_hallucinator_weights_t t; pack_hallucinator(Vars{vars, n_vars}, dtype, b, &t); });
}
extern "C" int hmmr_pack_hallucinator(const hmmr_var_t* vars, int n_vars, int dtype, void* host_blob, size_t blob_bytes, const void* device_base,
                                      hmmr_hallucinator_weights_t* out) {
    if (!vars || !host_blob || !out) { hmmr_set_error("hmmr_pack_hallucinator: null argument"); return -1; }
    Blob b(host_blob, blob_bytes, device_base);
    return guarded("hmmr_pack_hallucinator", b, [&] { pack_hallucinator(Vars{vars, n_vars}, dtype, b, out); });
}
extern "C" size_t hmmr_pack_ief_bytes(const hmmr_var_t* vars, int n_vars, int dtype, const int* delta_t, int n_delta) {
    return counted("hmmr_pack_ief_bytes", [&](Blob& b) { hmmr_ief_weights_t t; pack_ief(Vars{vars, n_vars}, dtype, delta_t, n_delta, 3, b, &t); });
}
extern "C" int hmmr_pack_ief(const hmmr_var_t* vars, int n_vars, int dtype, const int* delta_t, int n_delta, int num_stages, void* host_blob, size_t blob_bytes,
                             const void* device_base, hmmr_ief_weights_t* out) {
    if (!vars || !host_blob || !out || (n_delta > 0 && !delta_t)) { hmmr_set_error("hmmr_pack_ief: null argument"); return -1; }
    Blob b(host_blob, blob_bytes, device_base);
    return guarded("hmmr_pack_ief", b, [&] { pack_ief(Vars{vars, n_vars}, dtype, delta_t, n_delta, num_stages, b, out); });
}
extern "C" size_t hmmr_pack_smpl_bytes(const hmmr_smpl_source_t* src, int lsp, int split) {
    if (!src) return 0;
    return counted("hmmr_pack_smpl_bytes", [&](Blob& b) { hmmr_smpl_consts_t t; pack_smpl(src, lsp, split, b, &t); });
}
extern "C" int hmmr_pack_smpl(const hmmr_smpl_source_t* src, int lsp, int split, void* host_blob, size_t blob_bytes, const void* device_base,
                              hmmr_smpl_consts_t* out) {
    if (!src || !host_blob || !out) { hmmr_set_error("hmmr_pack_smpl: null argument"); return -1; }
    Blob b(host_blob, blob_bytes, device_base);
    return guarded("hmmr_pack_smpl", b, [&] { pack_smpl(src, lsp, split, b, out); });
}
