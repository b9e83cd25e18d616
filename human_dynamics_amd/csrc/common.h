// Shared device/host helpers for the HMMR gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef _Float16 f16_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

#define HMMR_WAVE 64

// Error plumbing shared by all translation units (defined in api.cpp).
void hmmr_set_error(const char* fmt, ...);

#define HMMR_CHECK_HIP(expr)                                                         \
    do {                                                                             \
        hipError_t _e = (expr);                                                      \
        if (_e != hipSuccess) {                                                      \
            hmmr_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),    \
                           __FILE__, __LINE__);                                      \
            return -2;                                                               \
        }                                                                            \
    } while (0)

#define HMMR_REQUIRE(cond, ...)                                                      \
    do {                                                                             \
        if (!(cond)) {                                                               \
            hmmr_set_error(__VA_ARGS__);                                             \
            return -1;                                                               \
        }                                                                            \
    } while (0)

// Development switches (include/hmmr_hip.h: hmmr_debug_t), owned by api.cpp
struct hmmr_debug_s;
const struct hmmr_debug_s* hmmr_debug_state();
// launch counters (include/hmmr_hip.h: hmmr_launch_counts_t, field index), owned by api.cpp
enum { HMMR_COUNT_UNIT_PAIR = 0, HMMR_COUNT_B1_UNIT = 1, HMMR_COUNT_TAIL_SPLIT = 2, HMMR_COUNT_CONV3X3_STREAM = 3, HMMR_COUNT_CONV1X1_STREAM = 4 };
void hmmr_count_launch(int which);

// "has this (kernel, device) pair had its one-time hipFuncSetAttribute?"  One bit per device; a redundant call
// from a racing thread is harmless, so relaxed atomics are enough.
struct DeviceOnce {
    std::atomic<unsigned long long> done{0ull};
    // returns the bit to publish with mark() when the one-time work is still due on the current device, else 0
    unsigned long long due() const {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) dev = 0;
        const unsigned long long bit = 1ull << (dev & 63);
        return (done.load(std::memory_order_relaxed) & bit) ? 0ull : bit;
    }
    void mark(unsigned long long bit) { done.fetch_or(bit, std::memory_order_relaxed); }
};

// compute units of the device `stream` belongs to (the current device for the null stream), asked once per device (api.cpp)
int hmmr_cu_count(hipStream_t stream);

// XCD-aware, bijective remap of a 1-D block id: the hardware dispatches block
// b to XCD b % 8; give each XCD a contiguous range of logical ids so tiles
// that share an operand panel hit the same private L2 (speed only).
__device__ __forceinline__ int xcd_remap(int b, int nblk) {
    const int q = nblk >> 3, r = nblk & 7;
    const int xcd = b & 7, i = b >> 3;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + i;
}

template <typename T> struct elem_traits;
template <> struct elem_traits<float> {
    static constexpr int EPS = 4;      // elements per 16-byte slot
    __device__ static __forceinline__ float to_f32(float v) { return v; }
    __device__ static __forceinline__ float from_f32(float v) { return v; }
};
template <> struct elem_traits<bf16_t> {
    static constexpr int EPS = 8;
    __device__ static __forceinline__ float to_f32(bf16_t v) { return (float)v; }
    __device__ static __forceinline__ bf16_t from_f32(float v) { return (bf16_t)v; }
};

// HMMR_F16X3 storage ("split" tensors): every group of 8 consecutive channels is 32 bytes,
// [hi0..hi7][lo0..lo7] with hi = fp16(x), lo = fp16(x - hi): x ~ hi + lo carries 22 mantissa bits for values whose lo
// half stays a normal fp16 (|x| > ~0.12; below that lo is an fp16 subnormal with an ABSOLUTE resolution of 6e-8 -- the
// matrix pipe keeps subnormals, tools/probes/mfma_f16_denorm.hip).  Filter banks are scaled per output channel by a
// power of two to a maximum of 2^14 at pack time so that their lo halves are normal numbers too (undone exactly by
// the epilogue's scale, packing.py).  Values are clamped to the fp16 range (+-65504) where they are split.  The element
// is 4 bytes wide for all address arithmetic (offsets are multiples of 8 elements); a 16-byte slot holds the hi OR the
// lo half of one group.  The GEMM kernels multiply two split operands with three fp16 MFMAs (lo*hi + hi*lo + hi*hi,
// fp32 accumulate; the dropped lo*lo term is <= 2^-22 of the product): fp32-class operands at a third of the 16-bit
// MFMA rate, 5x the fp32-MFMA rate.  (Rounds 1-2 used bf16 halves: 16-17 bits, "bf16x3".)
struct bsplit_t { unsigned int raw; };
static_assert(sizeof(bsplit_t) == 4, "bsplit_t is addressed as a 4-byte element");
template <> struct elem_traits<bsplit_t> {
    static constexpr int EPS = 4;      // "elements" of address arithmetic per 16-byte slot
};
typedef f16_t shalf_t;                  // the half type of a split tensor
typedef f16x8 shalf8;
typedef f16x2 shalf2;
#define HMMR_SPLIT_MAX 65504.0f
__device__ __forceinline__ float split_clamp(float v) { return __builtin_amdgcn_fmed3f(v, -HMMR_SPLIT_MAX, HMMR_SPLIT_MAX); }

// ---- run flags (include/hmmr_hip.h: hmmr_run_flags).  The clamp above is silent by itself: a value beyond the fp16 range becomes
// +-65504 and the network goes on.  Every place that splits a value checks it against the range first and raises a sticky flag word
// (one per translation unit: no relocatable device code in this build; api.cpp ORs them).  The normal case costs the comparison;
// only a saturating lane issues the atomic.
#ifndef HMMR_FLAG_SATURATED
#define HMMR_FLAG_SATURATED 1u      /* include/hmmr_hip.h */
#endif
#ifndef HMMR_FLAG_NAN
#define HMMR_FLAG_NAN 2u            /* include/hmmr_hip.h */
#endif
static __device__ unsigned g_split_flags __attribute__((unused));
__device__ __forceinline__ void split_flag(bool bad) {
#ifndef HMMR_NO_SATURATION_CHECK      // (development A/B only: what the checks cost)
    if (bad) atomicOr(&g_split_flags, HMMR_FLAG_SATURATED);
#endif
}
// Round 6: a NaN must not slip through.  v_med3_f32 (the clamp) turns a NaN into a FINITE value (it returns min3 of the other two
// operands) and v_max_f32 / v_max3_f32 (IEEE maxNum) drop a quiet NaN, so `fmaxf(...) > MAX` was false for it: a NaN activation became
// a plausible number with hmmr_run_flags == 0.  gfx950 has the IEEE-754-2019 `maximum` as v_maximum3_f32 (NaN-propagating, |x| as a
// source modifier): the running maximum of the hot epilogues costs the same instruction and BECOMES NaN, and the final test is
// !(m <= MAX) -- true for NaN, +-inf and anything beyond the fp16 range.
__device__ __forceinline__ float sat_acc(float m, float a, float b) {        // max(m, |a|, |b|), NaN-propagating
    float r; asm("v_maximum3_f32 %0, %1, |%2|, |%3|" : "=v"(r) : "v"(m), "v"(a), "v"(b)); return r;
}
__device__ __forceinline__ float sat_acc(float m, float a) {                 // max(m, |a|)
    float r; asm("v_maximum3_f32 %0, %1, |%2|, |%2|" : "=v"(r) : "v"(m), "v"(a)); return r;
}
__device__ __forceinline__ float sat_acc_signed(float m, float a) {          // max(m, a): in front of a ReLU, where a large negative value is not a clamp
    float r; asm("v_maximum3_f32 %0, %1, %2, %2" : "=v"(r) : "v"(m), "v"(a)); return r;
}
__device__ __forceinline__ bool sat_bad(float m) { return !(m <= HMMR_SPLIT_MAX); }
// raise the flag(s) from a running maximum: SATURATED for anything the clamp changed, + NAN when the maximum is a NaN
__device__ __forceinline__ void split_flag_max(float m) {
#ifndef HMMR_NO_SATURATION_CHECK
    if (sat_bad(m)) atomicOr(&g_split_flags, m != m ? (HMMR_FLAG_SATURATED | HMMR_FLAG_NAN) : HMMR_FLAG_SATURATED);
#endif
}
__device__ __forceinline__ bool split_overflows(float v) { return !(__builtin_fabsf(v) <= HMMR_SPLIT_MAX); }     // (+-inf and NaN included)
typedef int (*hmmr_flag_reader_t)(unsigned* flags, int clear);
void hmmr_register_flag_reader(hmmr_flag_reader_t fn);          // api.cpp
namespace {
int hmmr_tu_flags(unsigned* flags, int clear) {
    unsigned v = 0;
    if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_split_flags), sizeof(v)) != hipSuccess) return -2;
    if (clear && v) {
        const unsigned z = 0;
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_split_flags), &z, sizeof(z)) != hipSuccess) return -2;
    }
    *flags = v;
    return 0;
}
struct HmmrFlagRegistration { HmmrFlagRegistration() { hmmr_register_flag_reader(&hmmr_tu_flags); } };
static HmmrFlagRegistration g_flag_registration __attribute__((unused));
}  // namespace
// relu + clamp in one instruction
__device__ __forceinline__ float split_relu(float v) { return __builtin_amdgcn_fmed3f(v, 0.f, HMMR_SPLIT_MAX); }
// one 32x32x16 MFMA on split halves
__device__ __forceinline__ f32x16 mfma_split(const shalf8& a, const shalf8& b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
// the two halves packed in one dword <-> floats
__device__ __forceinline__ float shalf_lo(unsigned v) { return (float)__builtin_bit_cast(shalf2, v)[0]; }
__device__ __forceinline__ float shalf_hi(unsigned v) { return (float)__builtin_bit_cast(shalf2, v)[1]; }
__device__ __forceinline__ unsigned shalf_pack(shalf_t a, shalf_t b) { return __builtin_bit_cast(unsigned, shalf2{a, b}); }


// ---- mixed-precision FMA forms of the split arithmetic (round 5).  hi + lo of a packed pair as ONE v_fma_mix_f32 (exact before its
// single rounding to fp32: the value of (float)hi + (float)lo); two clamped values -> packed hi halves (one v_cvt_pk_f16_f32) and packed lo
// halves, lo = fp16(c - hi): c - hi is exact in fp32, so v_fma_mixlo/hi_f16 rounds once, like the casts they replace -- the same bits in
// half the instructions
__device__ __forceinline__ float split_sum_lo(unsigned h, unsigned l) {
    float v; asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel_hi:[1,0,1]" : "=v"(v) : "v"(h), "v"(l)); return v;
}
__device__ __forceinline__ float split_sum_hi(unsigned h, unsigned l) {
    float v; asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[1,0,1] op_sel_hi:[1,0,1]" : "=v"(v) : "v"(h), "v"(l)); return v;
}
__device__ __forceinline__ void split2_mix(float c0, float c1, unsigned& h, unsigned& l) {
    h = __builtin_bit_cast(unsigned, shalf2{(shalf_t)c0, (shalf_t)c1});
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l) : "v"(h), "v"(c0));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l) : "v"(h), "v"(c1));
}

// the value an output element has after being stored in type T and read back
template <typename T> __device__ __forceinline__ float stored_value(float v);
template <> __device__ __forceinline__ float stored_value<float>(float v) { return v; }
template <> __device__ __forceinline__ float stored_value<bf16_t>(float v) { return (float)(bf16_t)v; }
template <> __device__ __forceinline__ float stored_value<bsplit_t>(float v) {
    v = split_clamp(v);
    const float hi = (float)(shalf_t)v;
    return hi + (float)(shalf_t)(v - hi);
}

// 8 consecutive elements <-> 8 floats
__device__ __forceinline__ void load8(const float* p, float (&v)[8]) {
    const f32x4 a = *(const f32x4*)p, b = *(const f32x4*)(p + 4);
    v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3];
    v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
}
__device__ __forceinline__ void load8(const bf16_t* p, float (&v)[8]) {
    const bf16x8 a = *(const bf16x8*)p;
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (float)a[i];
}
__device__ __forceinline__ void load8(const bsplit_t* p, float (&v)[8]) {      // p: start of an 8-channel group
    const shalf8 hi = *(const shalf8*)p, lo = *((const shalf8*)p + 1);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (float)hi[i] + (float)lo[i];
}
__device__ __forceinline__ void store8(float* p, const float (&v)[8]) {
    f32x4 a = {v[0], v[1], v[2], v[3]}, b = {v[4], v[5], v[6], v[7]};
    *(f32x4*)p = a;
    *(f32x4*)(p + 4) = b;
}
__device__ __forceinline__ void store8(bf16_t* p, const float (&v)[8]) {
    bf16x8 a;
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = (bf16_t)v[i];
    *(bf16x8*)p = a;
}

// the same with the range check left to the caller: satmax = the running maximum of |v| (four v_max3_f32 per call), for
// split_flag(satmax > HMMR_SPLIT_MAX) once at the end of the kernel (the hot GEMM epilogues)
template <typename T> __device__ __forceinline__ void store8(T* p, const float (&v)[8], float& satmax);
template <> __device__ __forceinline__ void store8<float>(float* p, const float (&v)[8], float&) { store8(p, v); }
template <> __device__ __forceinline__ void store8<bf16_t>(bf16_t* p, const float (&v)[8], float&) { store8(p, v); }
template <> __device__ __forceinline__ void store8<bsplit_t>(bsplit_t* p, const float (&v)[8], float& satmax) {
    satmax = sat_acc(sat_acc(sat_acc(sat_acc(satmax, v[0], v[1]), v[2], v[3]), v[4], v[5]), v[6], v[7]);      // (four v_maximum3_f32)
    unsigned h[4], l[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) split2_mix(split_clamp(v[2 * i]), split_clamp(v[2 * i + 1]), h[i], l[i]);
    *(u32x4*)p = u32x4{h[0], h[1], h[2], h[3]};
    *((u32x4*)p + 1) = u32x4{l[0], l[1], l[2], l[3]};
}
__device__ __forceinline__ void store8(bsplit_t* p, const float (&v)[8]) {
    split_flag_max(sat_acc(sat_acc(sat_acc(sat_acc(0.f, v[0], v[1]), v[2], v[3]), v[4], v[5]), v[6], v[7]));
    shalf8 hi, lo;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float c = split_clamp(v[i]);
        hi[i] = (shalf_t)c;
        lo[i] = (shalf_t)(c - (float)hi[i]);
    }
    *(shalf8*)p = hi;
    *((shalf8*)p + 1) = lo;
}

// 8 consecutive elements held as raw 16-byte pieces -> 8 floats
template <typename T> __device__ __forceinline__ void unpack8(const u32x4 (&r)[(int)(8 * sizeof(T) / 16)], float (&v)[8]);
template <> __device__ __forceinline__ void unpack8<float>(const u32x4 (&r)[2], float (&v)[8]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[i] = __uint_as_float(r[0][i]); v[4 + i] = __uint_as_float(r[1][i]); }
}
template <> __device__ __forceinline__ void unpack8<bf16_t>(const u32x4 (&r)[1], float (&v)[8]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        v[2 * i] = __uint_as_float(r[0][i] << 16);            // bf16 -> f32 is a 16-bit shift
        v[2 * i + 1] = __uint_as_float(r[0][i] & 0xffff0000u);
    }
}
template <> __device__ __forceinline__ void unpack8<bsplit_t>(const u32x4 (&r)[2], float (&v)[8]) {   // r[0] = hi, r[1] = lo
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        v[2 * i] = shalf_lo(r[0][i]) + shalf_lo(r[1][i]);
        v[2 * i + 1] = shalf_hi(r[0][i]) + shalf_hi(r[1][i]);
    }
}

// split tensors keep the hi and the lo half of an 8-channel group in neighbouring slots, i.e. in neighbouring lanes of
// the staging geometry (lslot ^ 1 <-> lane ^ 1).  The two lanes share the work: the lane holding the hi slot
// pre-activates channels 0-3 of the group, the lane holding the lo slot channels 4-7 (each needs two dwords of its
// partner: DPP quad_perm [1,0,3,2]), both split their four results, and a second exchange completes each lane's own
// slot.  sc / sh: scale and shift of THIS lane's four channels.  Value for value this is
// store8<bsplit_t>(relu(fma(load8<bsplit_t>(x), s, b))), the producer-side `out2` of the epilogue.
__device__ __forceinline__ unsigned dpp_swap(unsigned v) {
    return (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xF, 0xF, true);
}
__device__ __forceinline__ u32x4 preact_slot_split(const u32x4& v, const f32x4& sc, const f32x4& sh, bool is_lo) {
    // what the partner needs from this lane: hi lane -> its hi[4..7] (dwords 2, 3); lo lane -> its lo[0..3] (dwords 0, 1)
    const unsigned r0 = dpp_swap(is_lo ? v[0] : v[2]), r1 = dpp_swap(is_lo ? v[1] : v[3]);
    const unsigned h0 = is_lo ? r0 : v[0], h1 = is_lo ? r1 : v[1];       // hi halves of this lane's four channels
    const unsigned l0 = is_lo ? v[2] : r0, l1 = is_lo ? v[3] : r1;       // lo halves
    float y[4];       // relu and the clamp to the fp16 range are one v_med3_f32 (= store8's split_clamp of a ReLU'd value)
    y[0] = split_relu(fmaf(split_sum_lo(h0, l0), sc[0], sh[0]));      // (hi + lo as one v_fma_mix_f32: round 5)
    y[1] = split_relu(fmaf(split_sum_hi(h0, l0), sc[1], sh[1]));
    y[2] = split_relu(fmaf(split_sum_lo(h1, l1), sc[2], sh[2]));
    y[3] = split_relu(fmaf(split_sum_hi(h1, l1), sc[3], sh[3]));
    unsigned ph[2], pl[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) split2_mix(y[2 * i], y[2 * i + 1], ph[i], pl[i]);
    // the partner's slot needs this lane's OTHER half: hi lane sends its lo[0..3], lo lane sends its hi[4..7]
    const unsigned s0 = dpp_swap(is_lo ? ph[0] : pl[0]), s1 = dpp_swap(is_lo ? ph[1] : pl[1]);
    return is_lo ? u32x4{s0, s1, pl[0], pl[1]} : u32x4{ph[0], ph[1], s0, s1};
}

// ---- helpers shared by the fused unit kernels for split operands (bottleneck_split.hip, unit_pair.hip)
// one MFMA A-operand fragment of a split filter bank: 32 rows x 16 K, hi and lo halves
struct wfrag { shalf8 hi, lo; };
// weights as the A operand, activations as B: gemm_conv.hip's product order (x.lo*w.hi, x.hi*w.lo, x.hi*w.hi) with the operands swapped
__device__ __forceinline__ f32x16 mma3(const wfrag& w, const shalf8& xh, const shalf8& xl, f32x16 c) {
    c = mfma_split(w.hi, xl, c);
    c = mfma_split(w.lo, xh, c);
    return mfma_split(w.hi, xh, c);
}
// four fp32 values -> their split halves, 4 halves (8 bytes) each; clamped to the fp16 range like store8<bsplit_t>
// the same with the check accumulated in a SCALAR register pair (a wave-wide mask of the lanes that saw a value beyond the range;
// v_cmp + s_or, no vector register stays live: the block-1 tails have none to spare) -- raise with split_flag(satmask != 0)
__device__ __forceinline__ void split4(const float (&v)[4], unsigned long long& hi, unsigned long long& lo, unsigned long long& satmask);
// satmax: running maximum of |v| over everything this thread has split (two v_max3_f32 per call; the caller raises the flag once,
// with split_flag(satmax > HMMR_SPLIT_MAX), when it is done: these kernels are bound by their instruction count)
__device__ __forceinline__ void split4(const float (&v)[4], unsigned long long& hi, unsigned long long& lo, float& satmax) {
    satmax = sat_acc(sat_acc(satmax, v[0], v[1]), v[2], v[3]);
    unsigned h[2], l[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const float c0 = split_clamp(v[2 * i]), c1 = split_clamp(v[2 * i + 1]);
        const shalf_t a = (shalf_t)c0, b = (shalf_t)c1;
        h[i] = shalf_pack(a, b);
        l[i] = shalf_pack((shalf_t)(c0 - (float)a), (shalf_t)(c1 - (float)b));
    }
    hi = (unsigned long long)h[0] | ((unsigned long long)h[1] << 32);
    lo = (unsigned long long)l[0] | ((unsigned long long)l[1] << 32);
}

// four values of one lane: clamp to [lo_clamp, 65504] (lo_clamp = 0: a ReLU in the same v_med3_f32), split; satmax as in split4
__device__ __forceinline__ void split4_mix(const float (&v)[4], float lo_clamp, unsigned (&h)[2], unsigned (&l)[2], float& satmax) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        satmax = sat_acc(satmax, v[2 * i], v[2 * i + 1]);
        split2_mix(__builtin_amdgcn_fmed3f(v[2 * i], lo_clamp, HMMR_SPLIT_MAX), __builtin_amdgcn_fmed3f(v[2 * i + 1], lo_clamp, HMMR_SPLIT_MAX), h[i], l[i]);
    }
}

__device__ __forceinline__ void split4(const float (&v)[4], unsigned long long& hi, unsigned long long& lo, unsigned long long& satmask) {
    const float m = sat_acc(sat_acc(0.f, v[0], v[1]), v[2], v[3]);
    satmask |= __builtin_amdgcn_ballot_w64(sat_bad(m));
    float unused = 0.f;
    split4(v, hi, lo, unused);
}
