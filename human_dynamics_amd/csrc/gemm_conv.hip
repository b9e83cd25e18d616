// Implicit-GEMM convolution / fully-connected kernel for gfx950 (MI355X).
//
//   out[m, n] = epi( sum_k A[m, k] * W[n, k] )
//
// A is gathered on the fly from an NHWC activation tensor (m = (img, oy, ox),
// k = (ky, kx, ci)); W is the filter bank packed K-contiguous per output
// channel.  One kernel serves every conv of the slim ResNet-v2-50 (1x1, 3x3
// stride 1/2, and the 7x7 stem re-expressed as an 8-tap x 32-"channel" conv on
// a zero-padded RGBX image), the [3,1] temporal convs of f_movie and the IEF
// fully-connected layers (reference: src/models.py:65-74, 102-113, 173-184).
//
// gfx950 mapping
//   * workgroups of 8 waves (512 threads; 128x128 or 128x64 block tile, wave tile
//     32x64 / 32x32) or 4 waves (64x64 for short-M GEMMs); K advances 128 BYTES
//     per row per step (64 bf16 or 32 fp32);
//   * operands go HBM -> LDS directly (global_load_lds_dwordx4, 1 KiB per wave
//     instruction, two LDS stages): the LDS image is lane-linear per wave and the
//     bank swizzle is applied to each lane's SOURCE slot; zero-padding taps and
//     M tails read a 64-byte zero page.  With a fused pre-activation (PRO) the A
//     operand takes the register route instead (HBM -> VGPR -> relu(x*s+b) ->
//     ds_write_b128), one K step ahead of its use;
//   * LDS rows are 128 B, slot index XOR-swizzled with (row>>1)&7 so that the
//     16 lanes serviced together by ds_read_b128 hit 16 distinct 16-B slots of
//     the 256-B bank row (conflict-free fragment reads);
//   * MFMA 32x32 tiles: v_mfma_f32_32x32x16_bf16 (bf16 in, fp32 accumulate) or
//     v_mfma_f32_32x32x2_f32 (exact fp32: bitwise an fmaf chain);
//   * epilogue: accumulators -> LDS (fp32) -> each lane owns 8 consecutive
//     output channels of one row: folded-BN scale/shift, residual add (vectors
//     prefetched before the K loop), ReLU, optional second output
//     relu(bn_next(v)), all as 16-byte coalesced accesses;
//   * split-K over blockIdx.y: fixed slice count per layer, fp32 partial planes,
//     ordered reduction in splitk_reduce_kernel (no atomics => batch-independent);
//   * 1-D grid remapped so that each XCD (private 4 MiB L2) walks a contiguous
//     range of tiles that share A rows.
#include <stdlib.h>
#include <string.h>
#include <type_traits>

#include "common.h"
#include "hmmr_hip.h"

struct ConvArgs {
    const void* in; const void* w; const float* scale; const float* shift;
    const void* res; void* out; void* out2; const float* scale2; const float* shift2;
    const float* pro_scale; const float* pro_shift;     // fused pre-activation of the A operand (PRO)
    void* out_b; int ldo_b, n_split, relu_b;            // column split: channels >= n_split go to out_b
    int M, K, cout, ldo, ldr;
    int Wo, HoWo, Hin, Win;
    long long in_img_stride; int in_row_stride, in_px_stride;
    int cin_log2, KH, KW, SY, SX, PY, PX;
    int res_strided; long long res_img_stride; int res_row_stride, res_px_stride;
    int relu, tiles_n, n_tiles;
    int kt_per_slice;                 // K steps per split-K slice (blockIdx.y); == all of them without split-K
    long long out_slice_stride;       // elements between the partial-sum planes of consecutive slices
    int probe;                        // always 0 in the product build (see HMMR_GEMM_PROBE below)
    const void* in2; int cin2, kt_split;   // second operand source: K steps >= kt_split read rows of in2 [M][cin2]
    // grouped launch: `batch` problems of one shape, problem z (blockIdx.z) at these BYTE offsets from problem 0
    int batch; long long bz_in, bz_w, bz_out, bz_res, bz_scale, bz_shift;
};

// The arguments of problem z of a grouped launch (hmmr_conv_desc_t.batch): every operand pointer moved by its stride.
__device__ __forceinline__ ConvArgs batch_problem(const ConvArgs& a0, int z) {
    ConvArgs a = a0;
    if (a0.batch > 1) {
        const long long zz = z;
        a.in = (const char*)a0.in + zz * a0.bz_in;
        a.w = (const char*)a0.w + zz * a0.bz_w;
        if (a0.out) a.out = (char*)a0.out + zz * a0.bz_out;
        if (a0.out2) a.out2 = (char*)a0.out2 + zz * a0.bz_out;
        if (a0.res) a.res = (const char*)a0.res + zz * a0.bz_res;
        if (a0.scale) a.scale = (const float*)((const char*)a0.scale + zz * a0.bz_scale);
        if (a0.shift) a.shift = (const float*)((const char*)a0.shift + zz * a0.bz_shift);
        if (a0.scale2) a.scale2 = (const float*)((const char*)a0.scale2 + zz * a0.bz_scale);
        if (a0.shift2) a.shift2 = (const float*)((const char*)a0.shift2 + zz * a0.bz_shift);
    }
    return a;
}

// Development build only (-DHMMR_GEMM_PROBE, tools/probe_build.sh -> libhmmr_hip_probe.so): the K loop can drop its
// MFMAs (1), its operand loads after the first stage (2) or its barriers (4), to measure which of the three bounds
// a shape.  Results are garbage in those modes.  The product build compiles the switches away.
#ifdef HMMR_GEMM_PROBE
#define HMMR_PROBE(a, bit) (((a).probe & (bit)) != 0)
#else
#define HMMR_PROBE(a, bit) false
#endif

template <typename TA> struct Frag;
template <> struct Frag<float>  { typedef f32x4 type; };
template <> struct Frag<bf16_t> { typedef bf16x8 type; };
struct split_frag { shalf8 hi, lo; };
template <> struct Frag<bsplit_t> { typedef split_frag type; };

__device__ __forceinline__ f32x16 mma(const f32x4& a, const f32x4& b, f32x16 c) {
#pragma unroll
    for (int j = 0; j < 4; ++j) c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[j], c, 0, 0, 0);
    return c;
}
__device__ __forceinline__ f32x16 mma(const bf16x8& a, const bf16x8& b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
// split operands: (ah + al)(bh + bl) ~ ah*bh + ah*bl + al*bh; every fp16 product is exact in fp32, the
// dropped al*bl term is <= 2^-22 of |a*b|.  The two small terms go first.
__device__ __forceinline__ f32x16 mma(const split_frag& a, const split_frag& b, f32x16 c) {
    c = mfma_split(a.lo, b.hi, c);
    c = mfma_split(a.hi, b.lo, c);
    return mfma_split(a.hi, b.hi, c);
}
// fragment of MFMA chunk c out of a 128-byte LDS row (fsw = the row's slot swizzle, lh = lane half).
// bf16 / fp32: 4 chunks of two 16-byte slots (one per lane half).  split: 2 chunks of two 8-channel
// groups (one per lane half), each group = a hi slot followed by a lo slot.
template <typename TA> struct FragIO {
    static constexpr int CHUNKS = 4;
    __device__ static __forceinline__ typename Frag<TA>::type read(const char* row, int c, int lh, int fsw) {
        return *(const typename Frag<TA>::type*)(row + (((2 * c + lh) ^ fsw) << 4));
    }
};
template <> struct FragIO<bsplit_t> {
    static constexpr int CHUNKS = 2;
    __device__ static __forceinline__ split_frag read(const char* row, int c, int lh, int fsw) {
        const int g = 2 * (2 * c + lh);
        split_frag f;
        f.hi = *(const shalf8*)(row + ((g ^ fsw) << 4));
        f.lo = *(const shalf8*)(row + (((g + 1) ^ fsw) << 4));
        return f;
    }
};
// ragged tail of an output row (cout % 8 != 0): element-wise stores; split rows are whole groups (host-checked)
template <typename TO> __device__ __forceinline__ void store_tail(TO* p, const float (&v)[8], int cnt) {
    if constexpr (!std::is_same<TO, bsplit_t>::value)
        for (int j = 0; j < cnt; ++j) p[j] = elem_traits<TO>::from_f32(v[j]);
}

// 64 zero bytes in HBM: out-of-bounds gather slots of the LDS-DMA path read from here
__device__ const u32x4 g_zero_page[4] = {};

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// pre-activation of one 16-byte slot: relu(x * scale + shift) per element, back to the operand type
template <typename TA> __device__ __forceinline__ u32x4 preact_slot(const u32x4& v, const f32x4* sc, const f32x4* sh);
template <> __device__ __forceinline__ u32x4 preact_slot<float>(const u32x4& v, const f32x4* sc, const f32x4* sh) {
    u32x4 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = __float_as_uint(fmaxf(fmaf(__uint_as_float(v[i]), sc[0][i], sh[0][i]), 0.f));
    return o;
}
template <> __device__ __forceinline__ u32x4 preact_slot<bf16_t>(const u32x4& v, const f32x4* sc, const f32x4* sh) {
    u32x4 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float lo = __uint_as_float(v[i] << 16), hi = __uint_as_float(v[i] & 0xffff0000u);
        const int e = 2 * i;
        const float a = fmaxf(fmaf(lo, sc[e >> 2][e & 3], sh[e >> 2][e & 3]), 0.f);
        const float b = fmaxf(fmaf(hi, sc[(e + 1) >> 2][(e + 1) & 3], sh[(e + 1) >> 2][(e + 1) & 3]), 0.f);
        typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
        bf16x2 pk = {(bf16_t)a, (bf16_t)b};
        o[i] = __builtin_bit_cast(unsigned, pk);
    }
    return o;
}

// Operands go HBM -> LDS directly (global_load_lds_dwordx4, 1 KiB per wave instruction, LDS image
// lane-linear, swizzle applied to the per-lane SOURCE slot).
// PRO  = true : the A operand instead takes the register route HBM -> VGPR -> (x*scale[ci] +
//               shift[ci], ReLU) -> ds_write_b128: the consumer-side fusion of the pre-activation
//               BN + ReLU of a ResNet-v2 unit (`preact`), so that tensor never exists in HBM.
//               Each element is transformed once per workgroup, while the MFMAs of the previous
//               K step run.  Only for un-padded gathers (1x1 convs): padding must stay zero.
// UTAP = true : every 128-byte K step lies inside one filter tap (cin*sizeof >= 128), so the
//               tap decode is wave-uniform scalar arithmetic.
// NSTAGE: LDS stages (2 = next tile in flight while the current one is consumed).  A 3-stage ring
//         (two tiles ahead, counted vmcnt + raw s_barrier) was measured 10-40 % SLOWER on every
//         ResNet shape, with 4-wave 128x128, 8-wave 128x128 and 8-wave 256x128 tiles alike: it
//         halves the resident workgroups per CU, and resident waves are what hides latency in this
//         structure.
// s_waitcnt with vmcnt = N, lgkmcnt = 0 (gfx9 encoding: vmcnt[3:0] | expcnt[6:4] | lgkmcnt[11:8] | vmcnt_hi[15:14])
template <int N> __device__ __forceinline__ void wait_vm_lgkm0() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
    __builtin_amdgcn_s_waitcnt((N & 15) | (7 << 4) | (0 << 8) | ((N >> 4) << 14));
}

// (A chunk-major K order for the 3x3 layers -- k' = (channel chunk, tap), so that the nine re-reads of a pixel's line fall
//  into consecutive K steps -- was written in round 2 and measured in round 3: equal to rounding, 0 ... 3 % SLOWER on
//  every shape and tile, profiles/r03a_chunk_major_k.log; removed.)
template <typename TA, typename TO, int BM, int BN, int WGM, int WGN, bool PRO, bool UTAP, int NSTAGE>
__global__ __launch_bounds__(WGM * WGN * 64, NSTAGE >= 3 ? (WGM * WGN) / 4 : ((WGM * WGN == 8) ? 4 : 1))
void conv_gemm_kernel(const ConvArgs a0) {
    const ConvArgs a = batch_problem(a0, blockIdx.z);
    static_assert(NSTAGE >= 2 && NSTAGE <= 4, "2 stages (two workgroups per CU) or a 3/4-stage ring (one workgroup per CU)");
    static_assert(NSTAGE == 2 || WGM * WGN == 8, "the deep ring is written for 8-wave workgroups");
    static_assert(!PRO || UTAP, "the fused pre-activation needs one tap per K step");
    constexpr int NT = WGM * WGN * 64;            // threads per workgroup (4 or 8 waves)
    constexpr int RPP = NT / 8;                   // tile rows staged per pass (8 lanes per row)
    constexpr int EPS = elem_traits<TA>::EPS;     // elements per 16-B slot
    constexpr int BKE = 8 * EPS;                  // elements per 128-B K step
    constexpr int TM = BM / WGM, TN = BN / WGN;   // wave tile
    constexpr int FM = TM / 32, FN = TN / 32;     // 32x32 fragments per wave
    constexpr int PA = BM / RPP, PB = BN / RPP;   // staging passes
    static_assert(PA >= 1 && PB >= 1, "tile smaller than one staging pass");
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
    typedef typename Frag<TA>::type frag_t;

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int L = xcd_remap(blockIdx.x, a.n_tiles);
    const int m0 = (L / a.tiles_n) * BM;
    const int n0 = (L % a.tiles_n) * BN;

    // ---- staging geometry: 8 lanes per row (one 128-B line), 32 rows per pass.
    // Lane (r0, pslot) fills PHYSICAL slot pslot of row r0 with LOGICAL slot lslot.
    const int pslot = tid & 7, r0 = tid >> 3;
    const int lslot = pslot ^ ((r0 >> 1) & 7);
    const TA* __restrict__ in = (const TA*)a.in;
    const TA* __restrict__ wt = (const TA*)a.w;

    const TA* aptr[PA]; unsigned amask[PA];
#pragma unroll
    for (int p = 0; p < PA; ++p) {
        const int m = m0 + r0 + RPP * p;
        aptr[p] = in; amask[p] = 0u;
        if (m < a.M) {
            const int img = m / a.HoWo, rem = m - img * a.HoWo;
            const int oy = rem / a.Wo, ox = rem - oy * a.Wo;
            const int iy0 = oy * a.SY - a.PY, ix0 = ox * a.SX - a.PX;
            aptr[p] = in + (long long)img * a.in_img_stride + (long long)iy0 * a.in_row_stride +
                      (long long)ix0 * a.in_px_stride + lslot * EPS;
            unsigned mk = 0u;
            for (int t = 0; t < a.KH * a.KW; ++t) {       // <= 32 taps (host-checked)
                const int ky = t / a.KW, kx = t - ky * a.KW;
                if ((unsigned)(iy0 + ky) < (unsigned)a.Hin && (unsigned)(ix0 + kx) < (unsigned)a.Win) mk |= 1u << t;
            }
            amask[p] = mk;
        }
    }
    const TA* wptr = wt + (long long)(n0 + r0) * a.K + lslot * EPS;
    const int cin_mask = (1 << a.cin_log2) - 1;

    // element offset of K step kt inside the gathered row + its tap index
    auto tap_of = [&](int k, int& tap) -> int {
        tap = k >> a.cin_log2;
        const int ci = k & cin_mask;
        int ky, kx;
        if (a.KW == 1) { ky = tap; kx = 0; }
        else if (a.KW == 3) { ky = (int)(((unsigned)tap * 43691u) >> 17); kx = tap - 3 * ky; }
        else { ky = tap / a.KW; kx = tap - ky * a.KW; }
        return ky * a.in_row_stride + kx * a.in_px_stride + ci;
    };

    constexpr bool SPLIT = std::is_same<TA, bsplit_t>::value;
    constexpr int PCH = EPS;                                  // channels whose preact constants this lane needs (split: 4 of the group's 8)
    u32x4 ra[PRO ? PA : 1];
    f32x4 psc[PRO ? PCH / 4 : 1], psh[PRO ? PCH / 4 : 1];     // scale/shift of this lane's channels
    // A operand of K step kt, LDS-DMA route (non-PRO)
    auto glds_a = [&](int kt, int buf) {
        int tap;
        const int koff = UTAP ? tap_of(kt * BKE, tap) : tap_of(kt * BKE + lslot * EPS, tap) - lslot * EPS;
        char* sa = smem + buf * STAGE + wave * 1024;
        if (a.in2 && kt >= a.kt_split) {
            // second source (dense rows of cin2 elements): the address is rebuilt per instruction, nothing stays live
            const TA* in2 = (const TA*)a.in2 + (kt - a.kt_split) * BKE + lslot * EPS;
#pragma unroll
            for (int p = 0; p < PA; ++p) {
                const void* src = (amask[p] & 1u) ? (const void*)(in2 + (long long)(m0 + r0 + RPP * p) * a.cin2)
                                                  : (const void*)g_zero_page;
                __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(sa + p * (RPP * 128)), 16, 0, 0);
            }
            return;
        }
#pragma unroll
        for (int p = 0; p < PA; ++p) {
            const bool ok = (amask[p] >> tap) & 1u;
            const void* src = ok ? (const void*)(aptr[p] + koff) : (const void*)g_zero_page;
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(sa + p * (RPP * 128)), 16, 0, 0);
        }
    };
    // B operand of K step kt, always LDS-DMA
    auto glds_b = [&](int kt, int buf) {
        char* sb = smem + buf * STAGE + A_BYTES + wave * 1024;
#pragma unroll
        for (int p = 0; p < PB; ++p)
            __builtin_amdgcn_global_load_lds((gptr_t)(wptr + (long long)(RPP * p) * a.K + kt * BKE),
                                             (lptr_t)(sb + p * (RPP * 128)), 16, 0, 0);
    };
    // PRO: raw A slots and B slots of K step kt (and the pre-activation constants of the A
    // channels) -> VGPRs.  BOTH operands take the register route here: with an LDS-DMA in flight
    // hipcc waits vmcnt(0) before every ordinary load's use and at every barrier, which serialised
    // the B fetch against the A prefetch; an all-register pipeline gets counted waits and a bare
    // s_barrier.  PRO gathers are un-padded 1x1 taps, so the only invalid rows are the M tail:
    // their pointer is clamped to row 0 (aptr = in) and their outputs are never stored.
    u32x4 rb[PRO ? PB : 1];
    auto load_regs = [&](int kt) {
        if constexpr (PRO) {
            int tap;
            const int koff = tap_of(kt * BKE, tap);
            // (split: channels 0-3 of the group for the lane holding its hi slot, 4-7 for the lane holding the lo slot)
            const int ci = ((kt * BKE) & cin_mask) + lslot * EPS;
#pragma unroll
            for (int q = 0; q < PCH / 4; ++q) {
                psc[q] = *(const f32x4*)(a.pro_scale + ci + 4 * q);
                psh[q] = *(const f32x4*)(a.pro_shift + ci + 4 * q);
            }
#pragma unroll
            for (int p = 0; p < PA; ++p) ra[p] = *(const u32x4*)(aptr[p] + koff);
#pragma unroll
            for (int p = 0; p < PB; ++p) rb[p] = *(const u32x4*)(wptr + (long long)(RPP * p) * a.K + kt * BKE);
        }
    };
    // PRO: pre-activate the A slots held in VGPRs and write both operands to LDS stage `buf`
    auto store_regs = [&](int buf) {
        if constexpr (PRO) {
            char* sa = smem + buf * STAGE + r0 * 128 + pslot * 16;
#pragma unroll
            for (int p = 0; p < PA; ++p) {
                if constexpr (SPLIT) *(u32x4*)(sa + p * (RPP * 128)) = preact_slot_split(ra[p], psc[0], psh[0], lslot & 1);
                else *(u32x4*)(sa + p * (RPP * 128)) = preact_slot<TA>(ra[p], psc, psh);
            }
#pragma unroll
            for (int p = 0; p < PB; ++p) *(u32x4*)(sa + A_BYTES + p * (RPP * 128)) = rb[p];
        }
    };

    // ---- fragment geometry
    const int wm = wave / WGN, wn = wave % WGN;
    const int lr = lane & 31, lh = lane >> 5;
    const int fsw = (lr >> 1) & 7;
    const int a_row_off = (wm * TM + lr) * 128;
    const int b_row_off = A_BYTES + (wn * TN + lr) * 128;

    f32x16 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // ---- epilogue geometry: each lane owns 8 consecutive output channels of NIT rows.
    // The residual vectors do not depend on the GEMM: issue their loads NOW so their HBM
    // latency overlaps the operand loads and the K loop instead of serialising after it.
    constexpr int VPR = BN / 8;                   // 8-channel vectors per row
    constexpr int NIT = (BM * VPR) / NT;
    constexpr int RV = (int)(8 * sizeof(TO) / 16); // 16-byte pieces per residual vector
    // (a PRO launch never carries a residual -- host-checked -- which keeps its A/B register
    //  pipeline inside the 128-VGPR budget of two 8-wave workgroups per CU)
    const TO* __restrict__ res = PRO ? nullptr : (const TO*)a.res;
    u32x4 rres[PRO ? 1 : NIT][RV];
    if (!PRO && res) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = it * NT + tid;
            const int m = m0 + idx / VPR, n = n0 + (idx % VPR) * 8;
#pragma unroll
            for (int q = 0; q < RV; ++q) rres[it][q] = u32x4{0u, 0u, 0u, 0u};
            if (m < a.M && n < a.cout) {
                long long ro;
                if (a.res_strided) {
                    const int img = m / a.HoWo, rem = m - img * a.HoWo;
                    const int oy = rem / a.Wo, ox = rem - oy * a.Wo;
                    ro = (long long)img * a.res_img_stride + (long long)oy * a.res_row_stride +
                         (long long)ox * a.res_px_stride + n;
                } else {
                    ro = (long long)m * a.ldr + n;
                }
#pragma unroll
                for (int q = 0; q < RV; ++q) rres[it][q] = *((const u32x4*)(res + ro) + q);
            }
        }
    }

    const int nk = a.K / BKE;
    // one K step of MFMAs out of LDS stage `sbuf`; fragments of 32-B chunk c+1 are fetched before
    // the MFMAs of chunk c issue
    auto compute_stage = [&](const char* sbuf) {
        constexpr int NCHK = FragIO<TA>::CHUNKS;
        // split fragments are twice as wide: 8-wave 128x128 tiles keep ONE fragment set (the 128-VGPR budget of
        // two workgroups per CU has no room for a second one next to the prefetched residual)
        constexpr bool DBUF = NSTAGE >= 3 || !(std::is_same<TA, bsplit_t>::value && WGM * WGN == 8 && FM * FN >= 2);
        frag_t fa[DBUF ? 2 : 1][FM], fb[DBUF ? 2 : 1][FN];
        auto read_frags = [&](int c, int slot_) {
#pragma unroll
            for (int i = 0; i < FM; ++i) fa[slot_][i] = FragIO<TA>::read(sbuf + a_row_off + i * 32 * 128, c, lh, fsw);
#pragma unroll
            for (int j = 0; j < FN; ++j) fb[slot_][j] = FragIO<TA>::read(sbuf + b_row_off + j * 32 * 128, c, lh, fsw);
        };
        if constexpr (DBUF) {
            read_frags(0, 0);
#pragma unroll
            for (int c = 0; c < NCHK; ++c) {
                if (c + 1 < NCHK) read_frags(c + 1, (c + 1) & 1);
                __builtin_amdgcn_sched_barrier(0);     // keep the prefetch ahead of this chunk's MFMAs
#pragma unroll
                for (int i = 0; i < FM; ++i)
#pragma unroll
                    for (int j = 0; j < FN; ++j) acc[i][j] = mma(fa[c & 1][i], fb[c & 1][j], acc[i][j]);
            }
        } else {
#pragma unroll
            for (int c = 0; c < NCHK; ++c) {
                read_frags(c, 0);
#pragma unroll
                for (int i = 0; i < FM; ++i)
#pragma unroll
                    for (int j = 0; j < FN; ++j) acc[i][j] = mma(fa[0][i], fb[0][j], acc[i][j]);
            }
        }
    };
    // split-K: slice blockIdx.y owns K steps [kt0, kt1) and writes a raw fp32 partial plane
    const int kt0 = blockIdx.y * a.kt_per_slice;
    const int kt1 = min(nk, kt0 + a.kt_per_slice);
    if constexpr (NSTAGE >= 3) {
        // ---- deep ring, ONE workgroup per CU: the LDS-DMA of K step kt + NSTAGE - 1 is issued during step kt,
        // so NSTAGE - 1 stages (the stage being awaited and NSTAGE - 2 behind it) are in flight while step kt's MFMAs
        // run.  Waits are counted (the newest stages stay outstanding) and the barrier is a bare s_barrier: a
        // __syncthreads() fence would drain vmcnt(0) every step, which is what serialises loads against MFMAs in
        // the 2-stage loop (measured with the HMMR_GEMM_PROBE build: loads-only + MFMA-only ~ the full kernel).
        // PRO: the pre-activation is applied IN LDS by the lane that DMA'd the slot, after its own wait and before
        // the barrier that publishes the stage (constants live in LDS behind the ring).
        constexpr int LPW = PA + PB;                              // LDS-DMA instructions per wave per stage
        static_assert(LPW * (NSTAGE - 2) < 64, "vmcnt range");
        [[maybe_unused]] float* s_pro = (float*)(smem + NSTAGE * STAGE);      // [2][K]: scale, shift
        if constexpr (PRO) {
            for (int i = tid; i < a.K; i += NT) { s_pro[i] = a.pro_scale[i]; s_pro[a.K + i] = a.pro_shift[i]; }
        }
        auto transform = [&](int kt, int buf) {
            if constexpr (PRO) {
                const int ci = ((kt * BKE) & cin_mask) + lslot * EPS;
                f32x4 sc_[PCH / 4], sh_[PCH / 4];
#pragma unroll
                for (int q = 0; q < PCH / 4; ++q) {
                    sc_[q] = *(const f32x4*)(s_pro + ci + 4 * q);
                    sh_[q] = *(const f32x4*)(s_pro + a.K + ci + 4 * q);
                }
                char* sa = smem + buf * STAGE + r0 * 128 + pslot * 16;
#pragma unroll
                for (int p = 0; p < PA; ++p) {
                    u32x4* q = (u32x4*)(sa + p * (RPP * 128));
                    if constexpr (SPLIT) *q = preact_slot_split(*q, sc_[0], sh_[0], lslot & 1);
                    else *q = preact_slot<TA>(*q, sc_, sh_);
                }
            }
        };
        // wait until at most `ahead` whole stages of this wave's DMA are outstanding (+ all LDS traffic), then barrier
        auto wait_ahead = [&](int ahead) {
            __builtin_amdgcn_sched_barrier(0);
            if (ahead >= 2) { if constexpr (NSTAGE >= 4) wait_vm_lgkm0<2 * LPW>(); }
            else if (ahead == 1) wait_vm_lgkm0<LPW>();
            else wait_vm_lgkm0<0>();
        };
        const int nloc = kt1 - kt0;
#pragma unroll
        for (int s_ = 0; s_ < NSTAGE - 1; ++s_)
            if (s_ < nloc && !HMMR_PROBE(a, 8)) { glds_a(kt0 + s_, s_); glds_b(kt0 + s_, s_); }
        if constexpr (PRO) {                                      // the constants are visible to every wave (no vmcnt drain)
            wait_vm_lgkm0<63>();
            __builtin_amdgcn_s_barrier();
        }
        wait_ahead(min(NSTAGE - 2, nloc - 1));
        transform(kt0, 0);
        wait_vm_lgkm0<63>();                                      // (lgkmcnt(0): the transformed slots are written)
        __builtin_amdgcn_s_barrier();
        // ---- ping-pong: waves 0-3 (group 0) and waves 4-7 (group 1) share the four SIMDs pairwise.  Each group
        // alternates a LOAD segment (all fragments of its stage -> VGPRs, the DMA of stage kt + NSTAGE - 1) with a
        // COMPUTE segment (the stage's MFMAs), one bare s_barrier between segments, and group 1 runs ONE segment
        // behind group 0: while one wave of a SIMD feeds the matrix pipe its partner is in LDS / the address queue.
        //   interval 2k  : group 0 LOAD(k)     | group 1 COMPUTE(k-1)
        //   interval 2k+1: group 0 COMPUTE(k)  | group 1 LOAD(k)
        // RAW: stage k is read first in interval 2k; every wave retires its own DMA of stage k (counted vmcnt) before
        // the barrier that ends interval 2k-1 (group 0 at the end of COMPUTE(k-1), group 1 at the end of LOAD(k-1)).
        // WAR: the DMA of stage k + NSTAGE - 1 lands in the buffer of stage k-1, whose last reads (group 1, LOAD(k-1),
        // interval 2k-1) are retired by the lgkmcnt(0) in front of that interval's barrier.
        constexpr int NCHK = FragIO<TA>::CHUNKS;
        const int grp = wave >> 2;
        int cur = 0, nxt = NSTAGE - 1;                            // ring positions of step kt and of the stage to fill
        if (grp == 1) __builtin_amdgcn_s_barrier();
        for (int kt = kt0; kt < kt1; ++kt) {
            const char* sbuf = smem + cur * STAGE;
            const int c1 = (cur + 1 == NSTAGE) ? 0 : cur + 1;
            const bool fill = kt + NSTAGE - 1 < kt1 && !HMMR_PROBE(a, 2);
            // ---- LOAD segment
            frag_t fa[NCHK][FM], fb[NCHK][FN];
#pragma unroll
            for (int c = 0; c < NCHK; ++c) {
                if (c > 0 && HMMR_PROBE(a, 16)) {                  // probe: half the fragment reads
#pragma unroll
                    for (int i = 0; i < FM; ++i) fa[c][i] = fa[0][i];
#pragma unroll
                    for (int j = 0; j < FN; ++j) fb[c][j] = fb[0][j];
                    continue;
                }
#pragma unroll
                for (int i = 0; i < FM; ++i) fa[c][i] = FragIO<TA>::read(sbuf + a_row_off + i * 32 * 128, c, lh, fsw);
#pragma unroll
                for (int j = 0; j < FN; ++j) fb[c][j] = FragIO<TA>::read(sbuf + b_row_off + j * 32 * 128, c, lh, fsw);
            }
            if (fill) {
                if (!HMMR_PROBE(a, 64)) glds_a(kt + NSTAGE - 1, nxt);
                if (!HMMR_PROBE(a, 32)) glds_b(kt + NSTAGE - 1, nxt);
            }
            if (grp == 1 && kt + 1 < kt1) {
                wait_ahead(min(NSTAGE - 2, kt1 - 2 - kt));
                transform(kt + 1, c1);
            }
            wait_vm_lgkm0<63>();
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            // ---- COMPUTE segment
            if (!HMMR_PROBE(a, 1)) {
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int c = 0; c < NCHK; ++c)
#pragma unroll
                    for (int i = 0; i < FM; ++i)
#pragma unroll
                        for (int j = 0; j < FN; ++j) acc[i][j] = mma(fa[c][i], fb[c][j], acc[i][j]);
                __builtin_amdgcn_s_setprio(0);
            }
            if (grp == 0 && kt + 1 < kt1) {
                wait_ahead(min(NSTAGE - 2, kt1 - 2 - kt));
                transform(kt + 1, c1);
                wait_vm_lgkm0<63>();
            }
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            nxt = cur; cur = c1;
        }
        if (grp == 0) __builtin_amdgcn_s_barrier();
    } else if constexpr (PRO) {
        // tile kt+1 was fetched into VGPRs a whole K step earlier, so pre-activating and writing it
        // at the TOP of step kt never waits for HBM; its registers are then free for tile kt+2.
        load_regs(kt0);
        store_regs(0);
        if (kt0 + 1 < kt1) load_regs(kt0 + 1);
        __syncthreads();
        for (int kt = kt0; kt < kt1; ++kt) {
            const int cur = (kt - kt0) & 1;
            if (kt + 1 < kt1 && !HMMR_PROBE(a, 2)) store_regs(cur ^ 1);
            if (kt + 2 < kt1 && !HMMR_PROBE(a, 2)) load_regs(kt + 2);
            if (!HMMR_PROBE(a, 1)) compute_stage(smem + cur * STAGE);
            if (!HMMR_PROBE(a, 4)) __syncthreads();
        }
    } else {
        glds_a(kt0, 0);
        glds_b(kt0, 0);
        __syncthreads();
        for (int kt = kt0; kt < kt1; ++kt) {
            const int cur = (kt - kt0) & 1;
            if (kt + 1 < kt1 && !HMMR_PROBE(a, 2)) { glds_a(kt + 1, cur ^ 1); glds_b(kt + 1, cur ^ 1); }
            if (!HMMR_PROBE(a, 1)) compute_stage(smem + cur * STAGE);
            if (!HMMR_PROBE(a, 4)) __syncthreads();
        }
    }

    // ---- epilogue: accumulators -> LDS as fp32 [BM][BN]
    float* sc = (float*)smem;
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wm * TM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                const int col = wn * TN + j * 32 + lr;
                sc[row * BN + col] = acc[i][j][r];
            }
    __syncthreads();

    TO* __restrict__ out = a.out ? (TO*)a.out + (long long)blockIdx.y * a.out_slice_stride : nullptr;
    TO* __restrict__ out2 = (TO*)a.out2;
    float satmax = 0.f;                           // split outputs: the largest |value| stored (common.h: hmmr_run_flags)
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int idx = it * NT + tid;
        const int row = idx / VPR, col = (idx % VPR) * 8;
        const int m = m0 + row, n = n0 + col;
        if (m >= a.M || n >= a.cout) continue;
        float v[8];
        load8(sc + row * BN + col, v);
        // scale and shift together are ONE fused multiply-add, spelled out so that no -ffp-contract mood can
        // make two builds (or this kernel and bottleneck.hip) round differently
        if (a.scale && a.shift) {
            float s[8], b[8]; load8(a.scale + n, s); load8(a.shift + n, b);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = fmaf(v[j], s[j], b[j]);
        } else if (a.scale) {
            float s[8]; load8(a.scale + n, s);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] *= s[j];
        } else if (a.shift) {
            float b[8]; load8(a.shift + n, b);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] += b[j];
        }
        const bool full = (n + 8 <= a.cout);
        if (!PRO && res) {
            float rr[8];
            unpack8<TO>(rres[PRO ? 0 : it], rr);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] += rr[j];
        }
        const bool second = a.out_b && n >= a.n_split;      // column split: the second convolution's channels
        if (second ? a.relu_b : a.relu) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = fmaxf(v[j], 0.f);
        }
        if (second) {
            TO* ob = (TO*)a.out_b + (long long)m * a.ldo_b + (n - a.n_split);
            if (full) store8<TO>(ob, v, satmax);
            else store_tail(ob, v, a.cout - n);
            continue;
        }
        const long long oo = (long long)m * a.ldo + n;
        if (out) {
            if (full) store8<TO>(out + oo, v, satmax);
            else store_tail(out + oo, v, a.cout - n);
        }
        if (out2) {
            float s2[8], b2[8], u[8];
            load8(a.scale2 + n, s2); load8(a.shift2 + n, b2);
#pragma unroll
            for (int j = 0; j < 8; ++j) {       // the preact of the STORED value: = the consumer-side preact_slot, bit for bit
                const float vr = stored_value<TO>(v[j]);
                u[j] = fmaxf(fmaf(vr, s2[j], b2[j]), 0.f);
            }
            if (full) store8<TO>(out2 + oo, u, satmax);
            else store_tail(out2 + oo, u, a.cout - n);
        }
    }
    if constexpr (std::is_same<TO, bsplit_t>::value) split_flag_max(satmax);
}

// Shared by the two 3x3 patch kernels below.
// bit t of the result: tap t = ky*3 + kx of output pixel m (flattened [img][y][x]) lies inside its image
__device__ __forceinline__ unsigned patch_tap_mask(int m, const ConvArgs& a) {
    if (m >= a.M) return 0u;
    const int rem = m % a.HoWo, y = rem / a.Wo, x = rem - y * a.Wo;
    unsigned mk = 0u;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
        if ((unsigned)yy < (unsigned)a.Hin && (unsigned)xx < (unsigned)a.Win) mk |= 1u << t;
    }
    return mk;
}
// accumulators of 64x64 wave tiles -> LDS as fp32 [BM][BN] -> folded BN (scale / shift), ReLU, 8 channels per lane
template <typename TO, int BM, int BN, int WGM, int WGN>
__device__ __forceinline__ void patch_epilogue(const ConvArgs& a, char* smem, const f32x16 (&acc)[2][2], int m0, int n0,
                                               int tid, int wm, int wn, int lr, int lh) {
    constexpr int NT = 512, TM = BM / WGM, TN = BN / WGN;
    static_assert(TM == 64 && TN == 64, "64x64 wave tiles");
    float* sc = (float*)smem;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wm * TM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                const int col = wn * TN + j * 32 + lr;
                sc[row * BN + col] = acc[i][j][r];
            }
    __syncthreads();
    constexpr int VPR = BN / 8, NIT = (BM * VPR) / NT;
    TO* __restrict__ out = (TO*)a.out;
    float satmax = 0.f;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int idx = it * NT + tid;
        const int row = idx / VPR, col = (idx % VPR) * 8;
        const int m = m0 + row, n = n0 + col;
        if (m >= a.M || n >= a.cout) continue;
        float v[8];
        load8(sc + row * BN + col, v);
        if (a.scale && a.shift) {
            float s[8], b[8]; load8(a.scale + n, s); load8(a.shift + n, b);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = fmaf(v[j], s[j], b[j]);
        } else if (a.scale) {
            float s[8]; load8(a.scale + n, s);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] *= s[j];
        } else if (a.shift) {
            float b[8]; load8(a.shift + n, b);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] += b[j];
        }
        if (a.relu) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = fmaxf(v[j], 0.f);
        }
        store8<TO>(out + (long long)m * a.ldo + n, v, satmax);
    }
    if constexpr (std::is_same<TO, bsplit_t>::value) split_flag_max(satmax);
}

// ------------------------------------------------------------------------- //
// 3x3 / stride 1 / SAME convolutions out of an LDS-resident input PATCH (hmmr_conv_desc_t.k_order = 1; tiles 9, 10)
// ------------------------------------------------------------------------- //
// What the probe build measured on the ring tiles (profiles/r03j_ring_probe.log): the im2col gather of the A operand
// -- one 128-byte line per output row per K step, the same input line nine times over a launch -- is the most
// expensive single piece of a 3x3 launch (15-24 % of it), more than the B stream that moves twice the bytes; halving
// the fragment reads out of LDS changes nothing.  So the A operand does not come through the ring here: with K in
// chunk-major order (K step kt = chunk kt / 9, tap kt % 9) a workgroup loads the 128-byte chunk of every input pixel
// its tile can touch ONCE per chunk into a PATCH and reads the nine taps as nine shifted fragment sets.
//   * patch geometry: patch row r holds input pixel m0 - (W+1) + r of the flattened [img][y][x] order (zero page beyond
//     the tensor), so tap (ky, kx) of tile row i is patch row i + ky*W + kx for EVERY pixel: the 32 lanes of a
//     fragment read consecutive rows, exactly the conflict-free geometry of the ring tiles (a first version numbered
//     the pixels with in-line zero columns / rows instead: no masks, but the skipped rows put two lanes of a
//     ds_read_b128 group on one bank -- SQ_LDS_BANK_CONFLICT 33 % of the LDS cycles, profiles/r03l).  The SAME
//     padding is an ADDRESS select: a lane whose tap falls outside its image (a 9-bit mask per fragment row,
//     computed once) reads one of the patch's last two rows instead, which the DMA keeps zero -- the one of its
//     own row's parity, at its own row's slot, so it stays on its own bank;
//   * the patch of chunk c+1 (NPP x 64 rows) is DMA'd during tap 0 of chunk c into the other of two patch buffers;
//     the B operand keeps the 3-stage ring and the ping-pong schedule of tiles 7 / 8 (see conv_gemm_kernel);
//   * LDS rows keep the 128-byte / XOR-swizzled geometry, the swizzle of a fragment row now follows the patch row.
template <typename TA, typename TO, int BM, int BN, int WGM, int WGN, int NPP>
__global__ __launch_bounds__(512, 1) void conv3x3_patch_kernel(const ConvArgs a) {
    static_assert(WGM * WGN == 8, "8-wave workgroups");
    constexpr int RPP = 64, NSB = 3;
    constexpr int EPS = elem_traits<TA>::EPS, BKE = 8 * EPS;
    constexpr int TM = BM / WGM, TN = BN / WGN, FM = TM / 32, FN = TN / 32;
    constexpr int PB = BN / RPP;
    constexpr int B_BYTES = BN * 128, P_BYTES = NPP * RPP * 128, OFF_P = NSB * B_BYTES;
    constexpr int NCHK = FragIO<TA>::CHUNKS;
    typedef typename Frag<TA>::type frag_t;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int L = xcd_remap(blockIdx.x, a.n_tiles);
    const int m0 = (L / a.tiles_n) * BM;
    const int n0 = (L % a.tiles_n) * BN;
    const int W = a.Win, C = 1 << a.cin_log2;
    const int need = BM + 2 * W + 2;                    // patch rows the tile reads; the rows behind them stay zero
    constexpr int ZROW = NPP * RPP - 2;                 // ... and the last two (one per row parity) stand in for every out-of-image tap
    const int base = m0 - W - 1;                        // input pixel of patch row 0

    // ---- staging geometry (as conv_gemm_kernel): 8 lanes per 128-byte row, 64 rows per pass
    const int pslot = tid & 7, r0 = tid >> 3;
    const int lslot = pslot ^ ((r0 >> 1) & 7);
    const TA* __restrict__ in = (const TA*)a.in;
    const TA* pptr[NPP];                                // source line of this lane's slot of patch row 64 p + r0; NULL: zero
#pragma unroll
    for (int p = 0; p < NPP; ++p) {
        const int r = RPP * p + r0, px = base + r;
        pptr[p] = (r < need && px >= 0 && px < a.M) ? in + (long long)px * C + lslot * EPS : nullptr;
    }
    const TA* wptr = (const TA*)a.w + (long long)(n0 + r0) * a.K + lslot * EPS;
    auto glds_patch = [&](int c, int buf) {
        char* sp = smem + OFF_P + buf * P_BYTES + wave * 1024;
#pragma unroll
        for (int p = 0; p < NPP; ++p) {
            const void* src = pptr[p] ? (const void*)(pptr[p] + c * BKE) : (const void*)g_zero_page;
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(sp + p * (RPP * 128)), 16, 0, 0);
        }
    };
    auto glds_b = [&](int kt, int buf) {
        char* sb = smem + buf * B_BYTES + wave * 1024;
#pragma unroll
        for (int p = 0; p < PB; ++p)
            __builtin_amdgcn_global_load_lds((gptr_t)(wptr + (long long)(RPP * p) * a.K + kt * BKE),
                                             (lptr_t)(sb + p * (RPP * 128)), 16, 0, 0);
    };

    // ---- fragment geometry
    const int wm = wave / WGN, wn = wave % WGN;
    const int lr = lane & 31, lh = lane >> 5;
    const int fswb = (lr >> 1) & 7;
    const int b_row_off = (wn * TN + lr) * 128;
    int prow[FM];                                       // patch row of tap (0, 0) of this lane's output pixels
    unsigned pmask[FM];                                 // bit t: tap t of that pixel lies inside its image
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        prow[i] = wm * TM + i * 32 + lr;
        pmask[i] = patch_tap_mask(m0 + prow[i], a);
    }

    f32x16 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nc = C / BKE, nk = 9 * nc;
    // counted waits: `pend` = the patch of the next chunk was issued (AFTER the B stage of that step) during tap 0 of
    // this chunk and may stay outstanding through taps 0 and 1; vmcnt retires in order, so the wait of tap 2 covers it
    auto wait_stage = [&](int kt, int t, int c) {
        __builtin_amdgcn_sched_barrier(0);
        if (kt + 2 < nk) {
            if (t <= 1 && c + 1 < nc) wait_vm_lgkm0<PB + NPP>();
            else wait_vm_lgkm0<PB>();
        } else wait_vm_lgkm0<0>();
    };
    glds_patch(0, 0);
    glds_b(0, 0);
    glds_b(1, 1);
    __builtin_amdgcn_sched_barrier(0);
    wait_vm_lgkm0<PB>();
    __builtin_amdgcn_s_barrier();
    const int grp = wave >> 2;
    int cur = 0, nxt = NSB - 1, t = 0, c = 0;
    if (grp == 1) __builtin_amdgcn_s_barrier();
    for (int kt = 0; kt < nk; ++kt) {
        const char* sb = smem + cur * B_BYTES;
        const char* sp = smem + OFF_P + (c & 1) * P_BYTES;
        const int c1 = (cur + 1 == NSB) ? 0 : cur + 1;
        const int ky = (t * 11) >> 5;                   // t / 3 for t < 9
        int ts = ky * W + (t - 3 * ky);
        if (HMMR_PROBE(a, 128)) ts = 0;                 // probes: which tap shifts make the fragment reads collide
        if (HMMR_PROBE(a, 256)) ts &= ~1;
        if (HMMR_PROBE(a, 512)) ts &= ~15;
        // ---- LOAD segment
        frag_t fa[NCHK][FM], fb[NCHK][FN];
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            const int tap_row = prow[i] + ts;
            // an out-of-image tap reads the zero row of the same parity AT THE SLOT the tap's own row would use: the
            // lane stays on the bank it would have had, so the 16 lanes of a ds_read_b128 group stay conflict-free
            const int row = ((pmask[i] >> t) & 1u) ? tap_row : ZROW + (tap_row & 1);
            const char* rp = sp + row * 128;
            const int fsw = (tap_row >> 1) & 7;
#pragma unroll
            for (int ch = 0; ch < NCHK; ++ch) {
                if (ch > 0 && HMMR_PROBE(a, 16)) { fa[ch][i] = fa[0][i]; continue; }
                fa[ch][i] = FragIO<TA>::read(rp, ch, lh, fsw);
            }
        }
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int ch = 0; ch < NCHK; ++ch) {
                if (ch > 0 && HMMR_PROBE(a, 16)) { fb[ch][j] = fb[0][j]; continue; }
                fb[ch][j] = FragIO<TA>::read(sb + b_row_off + j * 32 * 128, ch, lh, fswb);
            }
        if (kt + 2 < nk && !HMMR_PROBE(a, 32)) glds_b(kt + 2, nxt);
        if (t == 0 && c + 1 < nc && !HMMR_PROBE(a, 64)) glds_patch(c + 1, (c + 1) & 1);
        if (grp == 1 && kt + 1 < nk) wait_stage(kt, t, c);
        wait_vm_lgkm0<63>();
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        // ---- COMPUTE segment
        if (!HMMR_PROBE(a, 1)) {
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int ch = 0; ch < NCHK; ++ch)
#pragma unroll
                for (int i = 0; i < FM; ++i)
#pragma unroll
                    for (int j = 0; j < FN; ++j) acc[i][j] = mma(fa[ch][i], fb[ch][j], acc[i][j]);
            __builtin_amdgcn_s_setprio(0);
        }
        if (grp == 0 && kt + 1 < nk) wait_stage(kt, t, c);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        nxt = cur; cur = c1;
        if (++t == 9) { t = 0; ++c; }
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();

    patch_epilogue<TO, BM, BN, WGM, WGN>(a, smem, acc, m0, n0, tid, wm, wn, lr, lh);
}

// ------------------------------------------------------------------------- //
// The same layers WITHOUT a load segment (tile 11, 256x128): what profiles/r03m measured on the ping-pong schedule is that
// an interval is as long as ONE wave needs to issue its ~100 address / ds_read / DMA instructions in program order
// (~1040 cycles), not as long as the 24 MFMAs of its partner (768).  Here every wave keeps TWO fragment sets and reads
// the fragments of K step kt+1 between the MFMAs of step kt, all eight waves run the same stream (the two waves of a
// SIMD share its matrix pipe), and a K step is one barrier: the stage being read next was awaited at the end of the
// previous step.  Filter ring of 4 stages (stage kt+3 is issued during step kt), two patches, K chunk-major, the patch
// geometry and the epilogue of conv3x3_patch_kernel; same products in the same order per output element: same bits.
template <typename TA, typename TO, int NPP>
__global__ __launch_bounds__(512, 1) void conv3x3_pipe_kernel(const ConvArgs a) {
    constexpr int BM = 256, BN = 128, WGM = 4, WGN = 2;
    constexpr int RPP = 64, NSB = 4;
    constexpr int EPS = elem_traits<TA>::EPS, BKE = 8 * EPS;
    constexpr int TM = BM / WGM, TN = BN / WGN, FM = TM / 32, FN = TN / 32;
    constexpr int PB = BN / RPP;
    constexpr int B_BYTES = BN * 128, P_BYTES = NPP * RPP * 128, OFF_P = NSB * B_BYTES;
    constexpr int NCHK = FragIO<TA>::CHUNKS;
    static_assert(FM == 2 && FN == 2 && NCHK == 2, "written for 64x64 wave tiles of split operands");
    typedef typename Frag<TA>::type frag_t;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int L = xcd_remap(blockIdx.x, a.n_tiles);
    const int m0 = (L / a.tiles_n) * BM;
    const int n0 = (L % a.tiles_n) * BN;
    const int W = a.Win, C = 1 << a.cin_log2;
    const int need = BM + 2 * W + 2;
    constexpr int ZROW = NPP * RPP - 2;
    const int base = m0 - W - 1;

    const int pslot = tid & 7, r0 = tid >> 3;
    const int lslot = pslot ^ ((r0 >> 1) & 7);
    const TA* __restrict__ in = (const TA*)a.in;
    const TA* pptr[NPP];
#pragma unroll
    for (int p = 0; p < NPP; ++p) {
        const int r = RPP * p + r0, px = base + r;
        pptr[p] = (r < need && px >= 0 && px < a.M) ? in + (long long)px * C + lslot * EPS : nullptr;
    }
    const TA* wptr = (const TA*)a.w + (long long)(n0 + r0) * a.K + lslot * EPS;
    auto glds_patch = [&](int c, int buf) {
        char* sp = smem + OFF_P + buf * P_BYTES + wave * 1024;
#pragma unroll
        for (int p = 0; p < NPP; ++p) {
            const void* src = pptr[p] ? (const void*)(pptr[p] + c * BKE) : (const void*)g_zero_page;
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(sp + p * (RPP * 128)), 16, 0, 0);
        }
    };
    auto glds_b = [&](int kt, int buf) {
        char* sb = smem + buf * B_BYTES + wave * 1024;
#pragma unroll
        for (int p = 0; p < PB; ++p)
            __builtin_amdgcn_global_load_lds((gptr_t)(wptr + (long long)(RPP * p) * a.K + kt * BKE),
                                             (lptr_t)(sb + p * (RPP * 128)), 16, 0, 0);
    };

    const int wm = wave / WGN, wn = wave % WGN;
    const int lr = lane & 31, lh = lane >> 5;
    const int fswb = (lr >> 1) & 7;
    const int b_row_off = (wn * TN + lr) * 128;
    int prow[FM];
    unsigned pmask[FM];
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        prow[i] = wm * TM + i * 32 + lr;
        pmask[i] = patch_tap_mask(m0 + prow[i], a);
    }

    f32x16 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nc = C / BKE, nk = 9 * nc;
    frag_t fa[2][NCHK][FM], fb[2][NCHK][FN];
    // one fragment (hi + lo: two ds_read_b128) of K step (tap t_, patch buffer pb_, filter stage sb_): piece q = 0 ... 7
    auto read_piece = [&](frag_t (&da)[NCHK][FM], frag_t (&db)[NCHK][FN], int q, int t_, const char* sp_, const char* sb_) {
        const int ch = q >> 2, w_ = q & 3;                  // per chunk: A row block 0, B 0, B 1, A row block 1
        if (w_ == 0 || w_ == 3) {
            const int i = w_ == 0 ? 0 : 1;
            const int ky = (t_ * 11) >> 5;
            const int tap_row = prow[i] + ky * W + (t_ - 3 * ky);
            const int row = ((pmask[i] >> t_) & 1u) ? tap_row : ZROW + (tap_row & 1);
            da[ch][i] = FragIO<TA>::read(sp_ + row * 128, ch, lh, (tap_row >> 1) & 7);
        } else {
            const int j = w_ - 1;
            db[ch][j] = FragIO<TA>::read(sb_ + b_row_off + j * 32 * 128, ch, lh, fswb);
        }
    };
    glds_patch(0, 0);
#pragma unroll
    for (int s_ = 0; s_ < NSB - 1; ++s_) glds_b(s_, s_);
    __builtin_amdgcn_sched_barrier(0);
    wait_vm_lgkm0<PB>();                                   // patch 0, stages 0 and 1 (only stage 2 may still be in flight)
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int q = 0; q < 8; ++q) read_piece(fa[0], fb[0], q, 0, smem + OFF_P, smem);
    int t = 0, c = 0;
    auto step = [&](auto cur_c, int kt) {
        constexpr int CUR = decltype(cur_c)::value, NXT = CUR ^ 1;
        // (the DMA issue in front of the MFMAs: in the middle of them it measured 1-2 % slower, profiles/r03n)
        if (kt + 3 < nk) glds_b(kt + 3, (kt + 3) & 3);
        if (t == 0 && c + 1 < nc) glds_patch(c + 1, (c + 1) & 1);
        const int t1 = (t == 8) ? 0 : t + 1, cn = (t == 8) ? c + 1 : c;
        const char* sp1 = smem + OFF_P + (cn & 1) * P_BYTES;
        const char* sb1 = smem + ((kt + 1) & 3) * B_BYTES;
        // fragments of step kt+1 (past the last step: a harmless read of stale LDS) between the MFMAs of step kt
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            read_piece(fa[NXT], fb[NXT], q, t1, sp1, sb1);
            const int ch = q >> 2, i = (q >> 1) & 1, j = q & 1;
            acc[i][j] = mma(fa[CUR][ch][i], fb[CUR][ch][j], acc[i][j]);
        }
        __builtin_amdgcn_sched_barrier(0);
        // stage kt+2 (read during step kt+1) has landed: younger requests may stay in flight -- stage kt+3 and, through
        // taps 0 and 1, the patch of the next chunk (issued during tap 0 behind that step's filter stage; in-order vmcnt)
        if (kt + 2 < nk) {
            const bool pend = t <= 1 && c + 1 < nc;
            if (kt + 3 < nk) { if (pend) wait_vm_lgkm0<PB + NPP>(); else wait_vm_lgkm0<PB>(); }
            else wait_vm_lgkm0<0>();
        } else wait_vm_lgkm0<63>();
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        t = t1; c = cn;
    };
    for (int kt = 0; kt < nk; kt += 2) {
        step(std::integral_constant<int, 0>{}, kt);
        if (kt + 1 < nk) step(std::integral_constant<int, 1>{}, kt + 1);
    }

    patch_epilogue<TO, BM, BN, WGM, WGN>(a, smem, acc, m0, n0, tid, wm, wn, lr, lh);
}

template <typename TA, typename TO, int NPP>
static int launch_pipe(const ConvArgs& base, hipStream_t stream) {
    ConvArgs a = base;
    const int tiles_m = (a.M + 255) / 256;
    a.tiles_n = (a.cout + 127) / 128;
    a.n_tiles = tiles_m * a.tiles_n;
    constexpr int kloop = 4 * 128 * 128 + 2 * NPP * 64 * 128, epi = 256 * 128 * 4;
    constexpr int lds = kloop > epi ? kloop : epi;
    static_assert(lds <= 160 * 1024, "LDS");
    auto kern = conv3x3_pipe_kernel<TA, TO, NPP>;
    static DeviceOnce once;
    if (const unsigned long long bit = once.due()) {
        HMMR_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        once.mark(bit);
    }
    hipLaunchKernelGGL(kern, dim3(a.n_tiles), dim3(512), lds, stream, a);
    HMMR_CHECK_HIP(hipGetLastError());
    return 0;
}

// rows of the patch of a BM-row tile: its pixels, a halo of W + 1 on either side, and the two zero rows
static int patch_rows_bound(int bm, int h, int w) { (void)h; return bm + 2 * w + 2 + 2; }

template <typename TA, typename TO, int BM, int BN, int WGM, int WGN, int NPP>
static int launch_patch(const ConvArgs& base, hipStream_t stream) {
    ConvArgs a = base;
    const int tiles_m = (a.M + BM - 1) / BM;
    a.tiles_n = (a.cout + BN - 1) / BN;
    a.n_tiles = tiles_m * a.tiles_n;
    constexpr int kloop = 3 * BN * 128 + 2 * NPP * 64 * 128, epi = BM * BN * 4;
    constexpr int lds = kloop > epi ? kloop : epi;
    static_assert(lds <= 160 * 1024, "LDS");
    auto kern = conv3x3_patch_kernel<TA, TO, BM, BN, WGM, WGN, NPP>;
    static DeviceOnce once;
    if (const unsigned long long bit = once.due()) {
        HMMR_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        once.mark(bit);
    }
    hipLaunchKernelGGL(kern, dim3(a.n_tiles), dim3(512), lds, stream, a);
    HMMR_CHECK_HIP(hipGetLastError());
    return 0;
}

template <typename TA, typename TO>
static int launch_patch_tiled(const ConvArgs& a, int tile, hipStream_t stream) {
    const int bm = tile == 10 ? 128 : 256;
    const int npp = (patch_rows_bound(bm, a.Hin, a.Win) + 63) / 64;
    if (tile == 9) {
        if (npp <= 5) return launch_patch<TA, TO, 256, 128, 4, 2, 5>(a, stream);
        if (npp <= 7) return launch_patch<TA, TO, 256, 128, 4, 2, 7>(a, stream);
    } else if (tile == 10) {
        if (npp <= 3) return launch_patch<TA, TO, 128, 256, 2, 4, 3>(a, stream);
        if (npp <= 4) return launch_patch<TA, TO, 128, 256, 2, 4, 4>(a, stream);
    } else if (tile == 11) {
        if constexpr (std::is_same<TA, bsplit_t>::value) {
            if (npp <= 5) return launch_pipe<TA, TO, 5>(a, stream);
            if (npp <= 6) return launch_pipe<TA, TO, 6>(a, stream);
        }
    } else {
        hmmr_set_error("hmmr_conv_gemm: k_order 1 runs tiles 9 / 11 (256x128) and 10 (128x256), not %d", tile);
        return -1;
    }
    hmmr_set_error("hmmr_conv_gemm: k_order 1, tile %d: a %d x %d image needs a patch of %d x 64 rows, more than LDS holds",
                   tile, a.Hin, a.Win, npp);
    return -1;
}

// ------------------------------------------------------------------------- //
// Host side
// ------------------------------------------------------------------------- //
template <typename TA, typename TO, int BM, int BN, int WGM, int WGN, bool PRO, bool UTAP, int NSTAGE>
static int launch_cfg(const ConvArgs& base, int slices, hipStream_t stream) {
    ConvArgs a = base;
    const int tiles_m = (a.M + BM - 1) / BM;
    a.tiles_n = (a.cout + BN - 1) / BN;
    a.n_tiles = tiles_m * a.tiles_n;
    constexpr int kring = NSTAGE * (BM + BN) * 128, epi = BM * BN * 4;
    const int kloop = kring + ((PRO && NSTAGE >= 3) ? 8 * a.K : 0);      // deep PRO: [2][K] fp32 constants behind the ring
    const int lds = kloop > epi ? kloop : epi;
    if (lds > 160 * 1024) { hmmr_set_error("hmmr_conv_gemm: tile needs %d B of LDS (K = %d)", lds, a.K); return -1; }
    auto kern = conv_gemm_kernel<TA, TO, BM, BN, WGM, WGN, PRO, UTAP, NSTAGE>;
    static DeviceOnce once;              // per kernel instantiation, per device
    if (const unsigned long long bit = once.due()) {
        HMMR_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           NSTAGE >= 3 ? 160 * 1024 : lds));
        once.mark(bit);
    }
    constexpr int BKE_ = 8 * elem_traits<TA>::EPS;
    const int nk = a.K / BKE_;
    a.kt_per_slice = (nk + slices - 1) / slices;
    hipLaunchKernelGGL(kern, dim3(a.n_tiles, slices, a.batch), dim3(WGM * WGN * 64), lds, stream, a);
    HMMR_CHECK_HIP(hipGetLastError());
    return 0;
}

template <typename TA, typename TO, bool PRO, bool UTAP>
static int launch_tiled(const ConvArgs& a, int tile, int slices, hipStream_t stream) {
    switch (tile) {   // BM, BN, waves along M, waves along N
        case 1: return launch_cfg<TA, TO, 128, 128, 2, 2, PRO, UTAP, 2>(a, slices, stream);   // 4 waves, 64x64 each
        case 2: return launch_cfg<TA, TO, 128, 64, 2, 2, PRO, UTAP, 2>(a, slices, stream);    // 4 waves, 64x32 each
        case 3: return launch_cfg<TA, TO, 64, 64, 2, 2, PRO, UTAP, 2>(a, slices, stream);     // 4 waves, 32x32 each
        case 5: return launch_cfg<TA, TO, 128, 128, 4, 2, PRO, UTAP, 2>(a, slices, stream);   // 8 waves, 32x64 each
        case 6: return launch_cfg<TA, TO, 128, 64, 4, 2, PRO, UTAP, 2>(a, slices, stream);    // 8 waves, 32x32 each
        // deep ring + ping-pong wave groups, one 8-wave workgroup per CU (256 VGPRs per lane).  128x128 and 256x64 variants
        // of this structure were measured 15-40 % slower than tiles 5 / 6 on every ResNet shape and are not instantiated; 128x256 (tile 8) is the faster 3x3 tile.
        case 7: return launch_cfg<TA, TO, 256, 128, 4, 2, PRO, UTAP, 3>(a, slices, stream);   // 64x64 each, 3 x 48 KB
        case 8: return launch_cfg<TA, TO, 128, 256, 2, 4, PRO, UTAP, 3>(a, slices, stream);   // 64x64 each; half the A rows: half the fused-preact work per MFMA
        default: hmmr_set_error("hmmr_conv_gemm: bad tile %d", tile); return -1;
    }
}

template <typename TA, typename TO>
static int launch_typed(const ConvArgs& a, int tile, int slices, hipStream_t stream) {
    if (tile == 0) {
        // Measured on the ResNet-50 shapes at batch 256 (tools/conv_bench.py): 8-wave workgroups
        // (4 waves per SIMD at 2 workgroups per CU) beat 4-wave ones by 5-20 %, and a tile count of
        // ~1.5x the CU count with 128x128 tiles beats twice as many 128x64 tiles.
        const long long t128 = (long long)((a.M + 127) / 128) * ((a.cout + 127) / 128);
        const long long t12864 = (long long)((a.M + 127) / 128) * ((a.cout + 63) / 64);
        if (a.cout >= 128 && a.cout % 128 == 0 && t128 * slices >= 192) tile = 5;
        else if (t12864 * slices >= 192) tile = 6;
        else tile = 3;
    }
    const bool utap = ((size_t)1 << a.cin_log2) * sizeof(TA) >= 128;
    if (a.pro_scale) {
        if (!utap) { hmmr_set_error("hmmr_conv_gemm: fused pre-activation needs cin*sizeof >= 128"); return -1; }
        return launch_tiled<TA, TO, true, true>(a, tile, slices, stream);
    }
    return utap ? launch_tiled<TA, TO, false, true>(a, tile, slices, stream)
                : launch_tiled<TA, TO, false, false>(a, tile, slices, stream);
}

// Split-K second pass: sum the S fp32 partial planes in slice order, then the same epilogue as
// the GEMM kernel (scale/shift, residual, ReLU, optional second output).
template <typename TO>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const ConvArgs a0, const float* __restrict__ part,
                                                            int slices, int ldw, long long plane) {
    const ConvArgs a = batch_problem(a0, blockIdx.y);      // grouped launch: problem blockIdx.y, its planes behind the previous problem's
    part += (long long)blockIdx.y * slices * plane;
    const int vpr = (a.cout + 7) / 8;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)a.M * vpr) return;
    const int m = (int)(i / vpr), n = (int)(i % vpr) * 8;
    float v[8];
    load8(part + (long long)m * ldw + n, v);
    for (int s = 1; s < slices; ++s) {
        float p[8];
        load8(part + s * plane + (long long)m * ldw + n, p);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] += p[j];
    }
    if (a.scale && a.shift) {
        float s[8], b[8]; load8(a.scale + n, s); load8(a.shift + n, b);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = fmaf(v[j], s[j], b[j]);
    } else if (a.scale) {
        float s[8]; load8(a.scale + n, s);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] *= s[j];
    } else if (a.shift) {
        float b[8]; load8(a.shift + n, b);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] += b[j];
    }
    const bool full = (n + 8 <= a.cout);
    const TO* __restrict__ res = (const TO*)a.res;
    if (res) {
        long long ro;
        if (a.res_strided) {
            const int img = m / a.HoWo, rem = m - img * a.HoWo;
            const int oy = rem / a.Wo, ox = rem - oy * a.Wo;
            ro = (long long)img * a.res_img_stride + (long long)oy * a.res_row_stride + (long long)ox * a.res_px_stride + n;
        } else {
            ro = (long long)m * a.ldr + n;
        }
        float rr[8];
        load8(res + ro, rr);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] += rr[j];
    }
    if (a.relu) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = fmaxf(v[j], 0.f);
    }
    TO* __restrict__ out = (TO*)a.out;
    TO* __restrict__ out2 = (TO*)a.out2;
    const long long oo = (long long)m * a.ldo + n;
    if (out) {
        if (full) store8(out + oo, v);
        else store_tail(out + oo, v, a.cout - n);
    }
    if (out2) {
        float s2[8], b2[8], u[8];
        load8(a.scale2 + n, s2); load8(a.shift2 + n, b2);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float vr = stored_value<TO>(v[j]);
            u[j] = fmaxf(fmaf(vr, s2[j], b2[j]), 0.f);
        }
        if (full) store8(out2 + oo, u);
        else store_tail(out2 + oo, u, a.cout - n);
    }
}

extern "C" size_t hmmr_conv_splitk_workspace_bytes(int m, int cout, int split_k) {
    if (split_k <= 1 || m <= 0) return 0;
    return (size_t)split_k * (size_t)m * (size_t)((cout + 127) / 128 * 128) * sizeof(float);
}

static int ilog2_exact(int v) {
    int l = 0;
    while ((1 << l) < v) ++l;
    return ((1 << l) == v) ? l : -1;
}

int hmmr_conv3x3_stream(const hmmr_conv_desc_t* d, hipStream_t stream);     // conv3x3_stream.hip
int hmmr_conv1x1_stream(const hmmr_conv_desc_t* d, hipStream_t stream);     // conv1x1_stream.hip

extern "C" int hmmr_conv_gemm(const hmmr_conv_desc_t* d, void* stream) {
    HMMR_REQUIRE(d && d->in && d->w && (d->out || d->out2), "hmmr_conv_gemm: null operand");
    const int esz = d->in_dtype == HMMR_BF16 ? 2 : 4;
    // alignment unit of the gather in elements: one 16-byte slot, or one 32-byte hi/lo group of a split tensor
    const int eps = d->in_dtype == HMMR_F16X3 ? 8 : 16 / esz;
    const int cl2 = ilog2_exact(d->cin);
    HMMR_REQUIRE(cl2 >= 0 && d->cin % eps == 0, "hmmr_conv_gemm: cin=%d must be a power of two >= %d", d->cin, eps);
    const int K = d->kh * d->kw * d->cin + (d->in2 ? d->cin2 : 0);      // (in2: a second 1x1 source appended along K)
    HMMR_REQUIRE(d->kh * d->kw <= 32, "hmmr_conv_gemm: at most 32 filter taps");
    const int bke = 128 / esz;             // elements per 128-byte K step
    HMMR_REQUIRE(K % bke == 0, "hmmr_conv_gemm: K=%d must be a multiple of %d", K, bke);
    HMMR_REQUIRE(d->out_dtype != HMMR_F16X3 || d->cout % 8 == 0,
                 "hmmr_conv_gemm: a split (f16x3) output needs cout %% 8 == 0 (rows are whole hi/lo groups)");
    HMMR_REQUIRE(d->ldo % 8 == 0, "hmmr_conv_gemm: ldo=%d must be a multiple of 8", d->ldo);
    // every gathered 16-byte slot must stay aligned: ix = ox*sx + kx - px
    const bool px_ok = d->in_px_stride % eps == 0 ||
                       (d->kw == 1 && d->px == 0 && (d->in_px_stride * d->sx) % eps == 0);
    HMMR_REQUIRE(px_ok && d->in_row_stride % eps == 0 && d->in_img_stride % eps == 0,
                 "hmmr_conv_gemm: input strides must keep 16-byte alignment");
    HMMR_REQUIRE(!d->res || d->res_strided || d->ldr % 8 == 0, "hmmr_conv_gemm: ldr must be a multiple of 8");
    // residual rows are read as whole 8-channel vectors: a ragged cout needs the padding to exist
    HMMR_REQUIRE(!d->res || d->cout % 8 == 0 || (!d->res_strided && d->ldr >= (d->cout + 7) / 8 * 8),
                 "hmmr_conv_gemm: residual rows must be readable up to cout rounded up to 8");
    HMMR_REQUIRE(!d->out2 || (d->scale2 && d->shift2), "hmmr_conv_gemm: out2 needs scale2/shift2");
    HMMR_REQUIRE(!d->pro_scale == !d->pro_shift, "hmmr_conv_gemm: pro_scale and pro_shift go together");
    HMMR_REQUIRE(!d->out_b || (d->out && !d->out2 && !d->res && d->split_k <= 1 && d->n_split > 0 && d->n_split < d->cout &&
                               d->n_split % 8 == 0 && d->ldo_b % 8 == 0 && d->ldo_b >= d->cout - d->n_split),
                 "hmmr_conv_gemm: bad column split (out_b needs out, n_split %% 8 == 0, no out2 / res / split_k)");
    HMMR_REQUIRE(!d->pro_scale || !d->res, "hmmr_conv_gemm: a fused pre-activation (pro_*) cannot be combined with a residual");
    HMMR_REQUIRE(!d->pro_scale || (d->py == 0 && d->px == 0 && d->kh == 1 && d->kw == 1),
                 "hmmr_conv_gemm: the fused pre-activation is for un-padded 1x1 gathers (padding must stay zero)");
    HMMR_REQUIRE(d->tile != 8 || d->cout % 256 == 0, "hmmr_conv_gemm: tile 8 (128x256) needs cout %% 256 == 0 (filter rows are padded to 128)");
    HMMR_REQUIRE(!d->in2 || (d->kh == 1 && d->kw == 1 && d->py == 0 && d->px == 0 && d->sy == 1 && d->sx == 1 && !d->pro_scale &&
                             d->split_k <= 1 && d->cin2 > 0 && d->cin % bke == 0 && d->cin2 % bke == 0 &&
                             d->in_px_stride == d->cin && d->in_row_stride == d->win * d->cin &&
                             d->in_img_stride == (int64_t)d->hin * d->win * d->cin && d->ho == d->hin && d->wo == d->win),
                 "hmmr_conv_gemm: a second operand source (in2) needs a dense 1x1 stride-1 un-padded GEMM, cin and cin2 "
                 "multiples of the 128-byte K step, no pro_scale, no split_k");
    ConvArgs a;
    a.in = d->in; a.w = d->w; a.scale = d->scale; a.shift = d->shift; a.res = d->res;
    a.out = d->out; a.out2 = d->out2; a.scale2 = d->scale2; a.shift2 = d->shift2;
    a.pro_scale = d->pro_scale; a.pro_shift = d->pro_shift;
    a.out_b = d->out_b; a.ldo_b = d->ldo_b; a.n_split = d->n_split; a.relu_b = d->relu_b;
    a.M = d->n_img * d->ho * d->wo; a.K = K; a.cout = d->cout; a.ldo = d->ldo; a.ldr = d->ldr;
    a.Wo = d->wo; a.HoWo = d->ho * d->wo; a.Hin = d->hin; a.Win = d->win;
    a.in_img_stride = d->in_img_stride; a.in_row_stride = d->in_row_stride; a.in_px_stride = d->in_px_stride;
    a.cin_log2 = cl2; a.KH = d->kh; a.KW = d->kw; a.SY = d->sy; a.SX = d->sx; a.PY = d->py; a.PX = d->px;
    a.res_strided = d->res_strided; a.res_img_stride = d->res_img_stride;
    a.res_row_stride = d->res_row_stride; a.res_px_stride = d->res_px_stride;
    a.relu = d->relu; a.tiles_n = 0; a.n_tiles = 0;
    a.kt_per_slice = 0; a.out_slice_stride = 0; a.probe = 0;
    a.in2 = d->in2; a.cin2 = d->in2 ? d->cin2 : 0; a.kt_split = d->cin / bke;
    a.batch = d->batch > 1 ? d->batch : 1;
    a.bz_in = d->batch_in_bytes; a.bz_w = d->batch_w_bytes; a.bz_out = d->batch_out_bytes;
    a.bz_res = d->batch_res_bytes; a.bz_scale = d->batch_scale_bytes; a.bz_shift = d->batch_shift_bytes;
    HMMR_REQUIRE(a.batch == 1 || (a.batch <= 64 && !d->k_order && !d->out_b && !d->in2 && !d->pro_scale &&
                                  ((d->batch_in_bytes | d->batch_w_bytes | d->batch_out_bytes | d->batch_res_bytes | d->batch_scale_bytes | d->batch_shift_bytes) & 15) == 0),
                 "hmmr_conv_gemm: a grouped launch (batch > 1) takes up to 64 problems, strides that are multiples of 16 bytes, "
                 "and no k_order / out_b / in2 / pro_scale");
#ifdef HMMR_GEMM_PROBE
    a.probe = hmmr_debug_state()->gemm_probe;
#endif
    if (a.M <= 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    if (d->k_order == 2 && d->kh == 1 && d->kw == 1) return hmmr_conv1x1_stream(d, s);      // (checks its own geometry)
    if (d->k_order) {
        HMMR_REQUIRE((d->k_order == 1 || d->k_order == 2) && d->kh == 3 && d->kw == 3 && d->sy == 1 && d->sx == 1 && d->py == 1 && d->px == 1 &&
                     d->ho == d->hin && d->wo == d->win && d->in_px_stride == d->cin && d->in_row_stride == d->win * d->cin &&
                     d->in_img_stride == (int64_t)d->hin * d->win * d->cin && (d->cin * esz) % 128 == 0 &&
                     !d->res && !d->out2 && !d->out_b && !d->pro_scale && !d->in2 && d->split_k <= 1 && d->out && d->cout % 8 == 0,
                     "hmmr_conv_gemm: k_order 1 is for 3x3 / stride 1 / pad 1 convolutions over a dense NHWC tensor with "
                     "cin a multiple of the 128-byte K step and a scale/shift/relu epilogue (no res, out2, out_b, pro_scale, in2, split_k)");
        if (d->k_order == 2) return hmmr_conv3x3_stream(d, s);
        const bool px3 = d->in_dtype == HMMR_F16X3 && d->out_dtype == HMMR_F16X3, pbf = d->in_dtype == HMMR_BF16 && d->out_dtype == HMMR_BF16;
        HMMR_REQUIRE(px3 || pbf, "hmmr_conv_gemm: k_order 1 is built for split (f16x3) and bf16 tensors");
        // library's choice: the 256x128 ping-pong tile (inside the network the tuner prefers it to tile 11 on ten layers of eleven,
        // profiles/r03p; tile 10 needs 256 output columns and a narrower image)
        const int ptile = d->tile ? d->tile : 9;
        HMMR_REQUIRE((ptile == 10 ? d->cout % 256 : d->cout % 128) == 0, "hmmr_conv_gemm: k_order 1: cout must fill the tile's columns (filter rows are padded to 128)");
        if (pbf) {
            HMMR_REQUIRE(ptile != 11, "hmmr_conv_gemm: k_order 1, tile 11 (no load segment) is written for split operands; bf16 runs tiles 9 / 10");
            return launch_patch_tiled<bf16_t, bf16_t>(a, ptile, s);
        }
        return launch_patch_tiled<bsplit_t, bsplit_t>(a, ptile, s);
    }
    const bool in16 = d->in_dtype == HMMR_BF16, in32 = d->in_dtype == HMMR_F32, inx3 = d->in_dtype == HMMR_F16X3;
    const bool out16 = d->out_dtype == HMMR_BF16, out32 = d->out_dtype == HMMR_F32, outx3 = d->out_dtype == HMMR_F16X3;
    HMMR_REQUIRE((in16 || in32 || inx3) && (out16 || out32 || outx3), "hmmr_conv_gemm: unsupported dtypes %d -> %d", d->in_dtype, d->out_dtype);
    const int nk = K / bke;
    int slices = d->split_k > 1 ? d->split_k : 1;
    if (slices > nk) slices = nk;
    slices = (nk + ((nk + slices - 1) / slices) - 1) / ((nk + slices - 1) / slices);   // no empty slice
    if (slices > 1) {
        // pass 1: raw fp32 partial planes [slice][M][ldw]; pass 2: ordered sum + epilogue
        const int ldw = (d->cout + 127) / 128 * 128;
        const long long plane = (long long)a.M * ldw;
        HMMR_REQUIRE(d->ws && d->ws_bytes >= a.batch * hmmr_conv_splitk_workspace_bytes(a.M, d->cout, slices),
                     "hmmr_conv_gemm: split-K workspace missing or too small (a grouped launch needs batch x the planes)");
        ConvArgs p = a;
        p.bz_out = (long long)slices * plane * (long long)sizeof(float);      // problem z's planes behind problem z-1's
        p.scale = p.shift = nullptr; p.res = nullptr; p.out2 = nullptr; p.scale2 = p.shift2 = nullptr;
        // (a fused pre-activation, if any, stays: it acts on the A operand)
        p.relu = 0; p.out = d->ws; p.ldo = ldw; p.out_slice_stride = plane;
        const int rc = in16 ? launch_typed<bf16_t, float>(p, d->tile, slices, s)
                     : inx3 ? launch_typed<bsplit_t, float>(p, d->tile, slices, s)
                            : launch_typed<float, float>(p, d->tile, slices, s);
        if (rc) return rc;
        const long long nvec = (long long)a.M * ((a.cout + 7) / 8);
        const unsigned grid = (unsigned)((nvec + 255) / 256);
        const dim3 rgrid(grid, a.batch);
        if (out16) hipLaunchKernelGGL(splitk_reduce_kernel<bf16_t>, rgrid, dim3(256), 0, s, a, (const float*)d->ws, slices, ldw, plane);
        else if (outx3) hipLaunchKernelGGL(splitk_reduce_kernel<bsplit_t>, rgrid, dim3(256), 0, s, a, (const float*)d->ws, slices, ldw, plane);
        else hipLaunchKernelGGL(splitk_reduce_kernel<float>, rgrid, dim3(256), 0, s, a, (const float*)d->ws, slices, ldw, plane);
        HMMR_CHECK_HIP(hipGetLastError());
        return 0;
    }
    if (in16 && out16) return launch_typed<bf16_t, bf16_t>(a, d->tile, 1, s);
    if (in16 && out32) return launch_typed<bf16_t, float>(a, d->tile, 1, s);
    if (in32 && out32) return launch_typed<float, float>(a, d->tile, 1, s);
    if (in32 && out16) return launch_typed<float, bf16_t>(a, d->tile, 1, s);
    if (inx3 && outx3) return launch_typed<bsplit_t, bsplit_t>(a, d->tile, 1, s);
    if (inx3 && out32) return launch_typed<bsplit_t, float>(a, d->tile, 1, s);
    if (in32 && outx3) return launch_typed<float, bsplit_t>(a, d->tile, 1, s);
    hmmr_set_error("hmmr_conv_gemm: unsupported dtypes %d -> %d", d->in_dtype, d->out_dtype);
    return -1;
}
