// SMPL forward for gfx950: SMPL.__call__ (src/tf_smpl/batch_smpl.py:89-162),
// batch_rodrigues / batch_global_rigid_transformation
// (src/tf_smpl/batch_lbs.py:42-60, 133-194) and batch_orth_proj_idrot
// (src/tf_smpl/projection.py:16-29).  Always fp32.
//
// Three launches per call, nothing but the outputs and a 2 KB/instance
// scratch record ever touches HBM (the reference materialises the tiled
// skinning weights [m,6890,24] and T [m,6890,4,4]):
//   smpl_pose_kernel   per instance: Rodrigues (24 joints), shape-dependent
//                      joints, forward kinematics in the reference's
//                      multiplication order, relative transforms A, and the
//                      blend feature row [1, beta, (R_1..23 - I)]
//   smpl_verts_kernel  per (128-vertex tile, 16 instances): blend shapes as a
//                      218-deep FMA chain whose per-instance coefficients are
//                      wave-uniform (scalar registers), then linear-blend
//                      skinning over the vertex's non-zero weights (ELL) with
//                      A staged in LDS; T and W are never materialised
//   smpl_joints_kernel per instance: keypoints = sparse regressor x verts
//                      (CSR), then kps = s * (xy + t)
// Layout note: `dirs` is the tf_smpl basis re-packed on the host as
// [224][3][VPAD] (planar x/y/z, VPAD = 6912, rows >= 218 zero) so that a wave's 64 lanes read
// 256 contiguous bytes per coordinate.
#include "common.h"
#include "hmmr_hip.h"

static constexpr int NJ = 24;
static constexpr int NFEAT = 218;        // 1 + 10 + 207
static constexpr int LDF = 224;          // feature row stride (floats)
static constexpr int LDA = NJ * 12;      // A record: 24 x (3x4) floats
static constexpr int NFEAT_PAD = 220;    // blend loop length (multiple of 4; rows >= 218 of `dirs` are zero)
static constexpr int VT = 128;           // vertices per workgroup
static constexpr int IB = 16;            // instances per workgroup

// Record mode (hmmr_smpl_fwd_records): the m = R x n instances are R containers of n frames; instance r*n + i writes
// field f of frame i's packed record at rec + i*ld + off[r][f].  n_per == 0: plain mode, instance i writes base + i*ld.
enum { F_CAMS = 0, F_JOINTS, F_KPS, F_POSES, F_SHAPES, F_VERTS, F_OMEGAS, F_COUNT };
struct RecMap { int n_per; int off[HMMR_MAX_REGRESSORS][F_COUNT]; };
__device__ __forceinline__ float* rec_ptr(float* base, int inst, long long ld, const RecMap& rm, int field) {
    if (rm.n_per == 0) return base + (long long)inst * ld;
    const int r = inst / rm.n_per, i = inst - r * rm.n_per;
    return base + (long long)i * ld + rm.off[r][field];
}

// ---- kernel 1 ------------------------------------------------------------ //
__global__ __launch_bounds__(256) void smpl_pose_kernel(
    const float* __restrict__ theta, int ld_theta, const float* __restrict__ beta, int ld_beta,
    const float* __restrict__ j_template, const float* __restrict__ j_shapedirs,
    const int* __restrict__ parents, int m, float* __restrict__ feat, float* __restrict__ Aout,
    float* __restrict__ rs, long long ld_rs, const RecMap rm) {
    // per instance slot: local transform (R 9, t 3) and global (R 9, t 3) per joint
    __shared__ float sLoc[8][NJ][12];
    __shared__ float sGlb[8][NJ][12];
    __shared__ float sJ[8][NJ][3];
    const int slot = threadIdx.x >> 5, j = threadIdx.x & 31;
    const int inst = blockIdx.x * 8 + slot;
    const bool live = inst < m && j < NJ;
    float R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    if (live) {
        const float* th = theta + (long long)inst * ld_theta + 3 * j;
        const float x = th[0], y = th[1], z = th[2];
        // batch_lbs.py:48-50: angle = ||theta + 1e-8||, r = theta / angle
        const float ex = x + 1e-8f, ey = y + 1e-8f, ez = z + 1e-8f;
        const float angle = sqrtf(ex * ex + ey * ey + ez * ez);
        const float rx = x / angle, ry = y / angle, rz = z / angle;
        const float c = cosf(angle), s = sinf(angle), oc = 1.0f - c;
        // R = cos*I + (1-cos)*r r^T + sin*skew(r)      (batch_lbs.py:56-59, :24-36)
        R[0] = c + oc * rx * rx;      R[1] = oc * rx * ry - s * rz; R[2] = oc * rx * rz + s * ry;
        R[3] = oc * ry * rx + s * rz; R[4] = c + oc * ry * ry;      R[5] = oc * ry * rz - s * rx;
        R[6] = oc * rz * rx - s * ry; R[7] = oc * rz * ry + s * rx; R[8] = c + oc * rz * rz;
        if (rs) {
            float* o = rec_ptr(rs, inst, ld_rs, rm, F_POSES) + j * 9;
#pragma unroll
            for (int e = 0; e < 9; ++e) o[e] = R[e];
        }
        float* f = feat + (long long)inst * LDF;
        if (j >= 1) {
#pragma unroll
            for (int e = 0; e < 9; ++e) f[11 + (j - 1) * 9 + e] = R[e] - ((e == 0 || e == 4 || e == 8) ? 1.0f : 0.0f);
        }
        const float* bt = beta + (long long)inst * ld_beta;
        if (j < 10) f[1 + j] = bt[j];
        if (j == 0) f[0] = 1.0f;
        if (j < LDF - NFEAT) f[NFEAT + j] = 0.0f;
        // joints of the shaped template: J = J_regressor^T (v_template + S beta), folded on the host
        float J[3];
#pragma unroll
        for (int c3 = 0; c3 < 3; ++c3) {
            float acc = j_template[j * 3 + c3];
#pragma unroll
            for (int b = 0; b < 10; ++b) acc += bt[b] * j_shapedirs[b * (NJ * 3) + j * 3 + c3];
            J[c3] = acc;
            sJ[slot][j][c3] = acc;
        }
#pragma unroll
        for (int e = 0; e < 9; ++e) sLoc[slot][j][e] = R[e];
        (void)J;
    }
    __syncthreads();
    if (live) {
        // local translation: J_i - J_parent(i); root keeps J_0   (batch_lbs.py:163-173)
        const int p = j == 0 ? -1 : parents[j];
#pragma unroll
        for (int c3 = 0; c3 < 3; ++c3)
            sLoc[slot][j][9 + c3] = sJ[slot][j][c3] - (p >= 0 ? sJ[slot][p][c3] : 0.0f);
        if (j == 0) {
#pragma unroll
            for (int e = 0; e < 12; ++e) sGlb[slot][0][e] = sLoc[slot][0][e];
        }
    }
    __syncthreads();
    // G_i = G_parent(i) . [R_i | t_i], in index order like the reference loop (parents[i] < i)
    for (int i = 1; i < NJ; ++i) {
        if (live && j == i) {
            const int p = parents[i];
            const float* G = sGlb[slot][p];
            const float* Lc = sLoc[slot][i];
            float o[12];
#pragma unroll
            for (int r = 0; r < 3; ++r) {
#pragma unroll
                for (int cc = 0; cc < 3; ++cc)
                    o[r * 3 + cc] = G[r * 3 + 0] * Lc[0 * 3 + cc] + G[r * 3 + 1] * Lc[1 * 3 + cc] + G[r * 3 + 2] * Lc[2 * 3 + cc];
                o[9 + r] = G[r * 3 + 0] * Lc[9] + G[r * 3 + 1] * Lc[10] + G[r * 3 + 2] * Lc[11] + G[9 + r];
            }
#pragma unroll
            for (int e = 0; e < 12; ++e) sGlb[slot][i][e] = o[e];
        }
        __syncthreads();
    }
    if (live) {
        // A = G - [0 | G . [J; 0]]  ->  rows [R_g | t_g - R_g J]   (batch_lbs.py:188-192)
        const float* G = sGlb[slot][j];
        float* o = Aout + (long long)inst * LDA + j * 12;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const float bone = G[r * 3 + 0] * sJ[slot][j][0] + G[r * 3 + 1] * sJ[slot][j][1] + G[r * 3 + 2] * sJ[slot][j][2];
            o[r * 4 + 0] = G[r * 3 + 0]; o[r * 4 + 1] = G[r * 3 + 1]; o[r * 4 + 2] = G[r * 3 + 2];
            o[r * 4 + 3] = G[9 + r] - bone;
        }
    }
}

// ---- kernel 2 ------------------------------------------------------------ //
// 128 vertices x 16 instances per workgroup.  The blend loop walks k four at a time: the basis
// values of step k+4 are requested before the FMAs of step k issue (two register sets), the 16
// instances' coefficient rows sit in LDS and are read as wave-wide broadcasts, every basis value
// feeds 16 FMAs (packed: v_pk_fma_f32), and the basis rows beyond 218 are zero (host-padded to 224).
__global__ __launch_bounds__(VT) void smpl_verts_kernel(
    const float* __restrict__ dirs, int vpad, const float* __restrict__ feat, const float* __restrict__ A,
    const int* __restrict__ lbs_idx, const float* __restrict__ lbs_w, int nnz, int nv, int m,
    float* __restrict__ verts, long long ld_verts, const RecMap rm) {
    __shared__ __attribute__((aligned(16))) float smem[IB * LDA + NFEAT_PAD * IB];
    float (*sA)[LDA] = (float (*)[LDA])smem;
    float (*sF)[IB] = (float (*)[IB])(smem + IB * LDA);          // [k][instance]: 4 instances per ds_read_b128
    // 1-D grid, instance blocks fastest, XCD-aware: the workgroups that share a vertex tile's basis slice (224 x 3 x 128 floats =
    // 344 KB) are consecutive on ONE XCD and read it out of that XCD's L2.  (With the vertex tile as the fast index every instance
    // block streamed the whole 17.9 MB basis again: 505 MB of fetch per 768 instances against 20 MB of constants, profiles/r03s.)
    const int nib = (m + IB - 1) / IB;
    const int L = xcd_remap(blockIdx.x, gridDim.x);
    const int v = (L / nib) * VT + threadIdx.x;
    const int i0 = (L % nib) * IB;
    for (int e = threadIdx.x; e < IB * LDA; e += VT) {
        const int ii = e / LDA;
        sA[ii][e % LDA] = (i0 + ii < m) ? A[(long long)(i0 + ii) * LDA + (e % LDA)] : 0.f;
    }
    // feature rows of the 16 instances, transposed (the scratch buffer is sized for m rounded up to
    // IB, so tail blocks read -- and ignore -- valid memory)
    for (int e = threadIdx.x; e < IB * NFEAT_PAD; e += VT) {
        const int ii = e / NFEAT_PAD, k = e % NFEAT_PAD;
        sF[k][ii] = feat[(long long)(i0 + ii) * LDF + k];
    }
    // accumulators paired over INSTANCES: one v_pk_fma_f32 = two instances' coefficients (adjacent in
    // LDS) times one broadcast basis value
    f32x2 acc[IB / 2][3];
#pragma unroll
    for (int ip = 0; ip < IB / 2; ++ip) acc[ip][0] = acc[ip][1] = acc[ip][2] = f32x2{0.f, 0.f};
    // v_posed = v_template + beta.S + pose_feature.P   (batch_smpl.py:110-112, 131-133)
    // basis row (k, c) starts at the wave-uniform address dirs + (3k + c) * vpad: scalar base + lane offset
    float dv[2][4][3];
    auto fetch = [&](int k, int set) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int c = 0; c < 3; ++c) dv[set][q][c] = (dirs + (long long)((k + q) * 3 + c) * vpad)[v];   // v < vpad always
    };
    auto blend = [&](int k, int set) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
            for (int i4 = 0; i4 < IB / 4; ++i4) {
                const f32x4 c4 = *(const f32x4*)&sF[k + q][4 * i4];
                const f32x2 lo = {c4[0], c4[1]}, hi = {c4[2], c4[3]};
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const f32x2 b = {dv[set][q][c], dv[set][q][c]};
                    acc[2 * i4][c] = __builtin_elementwise_fma(lo, b, acc[2 * i4][c]);
                    acc[2 * i4 + 1][c] = __builtin_elementwise_fma(hi, b, acc[2 * i4 + 1][c]);
                }
            }
        }
    };
    fetch(0, 0);
    __syncthreads();
    static_assert(NFEAT_PAD % 8 == 4, "the two-set loop below peels one step");
    for (int k = 0; k < NFEAT_PAD - 4; k += 8) {
        fetch(k + 4, 1);
        blend(k, 0);
        fetch(k + 8, 0);
        blend(k + 4, 1);
    }
    blend(NFEAT_PAD - 4, 0);
    __syncthreads();
    if (v >= nv) return;
    // skinning: T = sum_j W[v,j] A_j over the non-zero weights; v' = T [v_posed; 1]   (batch_smpl.py:141-151)
    const int* vidx = lbs_idx + (long long)v * nnz;
    const float* vw = lbs_w + (long long)v * nnz;
#pragma unroll
    for (int g = 0; g < IB; g += 4) {               // 4 instances at a time: T stays in registers
        if (i0 + g >= m) continue;
        float T[4][12];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int e = 0; e < 12; ++e) T[q][e] = 0.f;
        for (int z = 0; z < nnz; ++z) {
            const int jj = vidx[z] * 12;
            const float wv = vw[z];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4* a4 = (const f32x4*)&sA[g + q][jj];
                const f32x4 r0 = a4[0], r1 = a4[1], r2 = a4[2];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    T[q][e] = fmaf(wv, r0[e], T[q][e]);
                    T[q][4 + e] = fmaf(wv, r1[e], T[q][4 + e]);
                    T[q][8 + e] = fmaf(wv, r2[e], T[q][8 + e]);
                }
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (i0 + g + q < m) {
                const float x = acc[(g + q) >> 1][0][(g + q) & 1], y = acc[(g + q) >> 1][1][(g + q) & 1], z = acc[(g + q) >> 1][2][(g + q) & 1];
                float* o = rec_ptr(verts, i0 + g + q, ld_verts, rm, F_VERTS) + v * 3;
                o[0] = T[q][0] * x + T[q][1] * y + T[q][2] * z + T[q][3];
                o[1] = T[q][4] * x + T[q][5] * y + T[q][6] * z + T[q][7];
                o[2] = T[q][8] * x + T[q][9] * y + T[q][10] * z + T[q][11];
            }
        }
    }
}

// ---- kernel 2, matrix-core form ------------------------------------------- //
// The blend-shape product v_posed = [1, beta, pose_feature] . [v_template; shapedirs; posedirs] (batch_smpl.py:110-112,
// 131-133) is the dense contraction of the stage: [m, 218] x [218, 3 x 6890].  Here it runs on the matrix cores in EXACT
// fp32 (v_mfma_f32_32x32x2_f32: exact fp32 products and fp32 accumulation; measured within one ulp of the fmaf chains
// smpl_verts_kernel forms on the vector units, not bit-identical to them), D[instance][vertex] per coordinate: a wave owns 32 vertices x 32
// instances, the A operand (instance rows of the feature matrix) comes from LDS, the B operand (one basis row of the
// planar `dirs` layout) is a coalesced 128-byte read per half wave, requested four k-pairs ahead.  A lane ends up with
// its vertex's blended position for 16 instances, which is exactly what the skinning loop (unchanged: ELL weights, A in
// LDS) consumes.  The skinning sum itself stays on the vector units: with SMPL's weights it is 4-sparse per vertex.
static constexpr int IBM = 32;           // instances per workgroup (= MFMA rows)
__global__ __launch_bounds__(256, 2) void smpl_verts_mfma_kernel(
    const float* __restrict__ dirs, int vpad, const float* __restrict__ feat, const float* __restrict__ A,
    const int* __restrict__ lbs_idx, const float* __restrict__ lbs_w, int nnz, int nv, int m,
    float* __restrict__ verts, long long ld_verts, const RecMap rm) {
    __shared__ __attribute__((aligned(16))) float smem[IBM * LDA + NFEAT_PAD * IBM];
    float (*sA)[LDA] = (float (*)[LDA])smem;
    float (*sF)[IBM] = (float (*)[IBM])(smem + IBM * LDA);       // [k][instance]
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int lc = lane & 31, lh = lane >> 5;
    const int v = blockIdx.x * VT + wave * 32 + lc;              // this lane's vertex = its column of D (v < vpad always)
    const int i0 = blockIdx.y * IBM;
    for (int e = threadIdx.x; e < IBM * LDA; e += 256) {
        const int ii = e / LDA;
        sA[ii][e % LDA] = (i0 + ii < m) ? A[(long long)(i0 + ii) * LDA + (e % LDA)] : 0.f;
    }
    for (int e = threadIdx.x; e < IBM * NFEAT_PAD; e += 256) {   // (the scratch rows are sized for m rounded up to IBM)
        const int ii = e / NFEAT_PAD, k = e % NFEAT_PAD;
        sF[k][ii] = feat[(long long)(i0 + ii) * LDF + k];
    }
    f32x16 acc[3];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    // K step kk covers k = 2 kk (lanes 0-31) and 2 kk + 1 (lanes 32-63)
    constexpr int NKK = NFEAT_PAD / 2, PF = 4;
    static_assert(NKK % PF == 2, "the two-set loop below peels two steps");
    float dv[2][PF][3];
    auto fetch = [&](int kk, int set) {
#pragma unroll
        for (int q = 0; q < PF; ++q)
#pragma unroll
            for (int c = 0; c < 3; ++c)
                dv[set][q][c] = (kk + q < NKK) ? (dirs + (long long)((2 * (kk + q) + lh) * 3 + c) * vpad)[v] : 0.f;
    };
    auto blend = [&](int kk, int set) {
#pragma unroll
        for (int q = 0; q < PF; ++q) {
            if (kk + q < NKK) {
                const float a = sF[2 * (kk + q) + lh][lc];
#pragma unroll
                for (int c = 0; c < 3; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, dv[set][q][c], acc[c], 0, 0, 0);
            }
        }
    };
    fetch(0, 0);
    __syncthreads();
    for (int kk = 0; kk < NKK; kk += 2 * PF) {
        fetch(kk + PF, 1);
        blend(kk, 0);
        fetch(kk + 2 * PF, 0);
        blend(kk + PF, 1);
    }
    if (v >= nv) return;
    // skinning, as in smpl_verts_kernel: the lane holds instances i0 + 8 g + 4 lh + {0..3} in accumulator rows 4 g .. 4 g + 3
    const int* vidx = lbs_idx + (long long)v * nnz;
    const float* vw = lbs_w + (long long)v * nnz;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int ib = 8 * g + 4 * lh;
        if (i0 + ib >= m) continue;
        float T[4][12];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int e = 0; e < 12; ++e) T[q][e] = 0.f;
        for (int z = 0; z < nnz; ++z) {
            const int jj = vidx[z] * 12;
            const float wv = vw[z];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4* a4 = (const f32x4*)&sA[ib + q][jj];
                const f32x4 r0 = a4[0], r1 = a4[1], r2 = a4[2];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    T[q][e] = fmaf(wv, r0[e], T[q][e]);
                    T[q][4 + e] = fmaf(wv, r1[e], T[q][4 + e]);
                    T[q][8 + e] = fmaf(wv, r2[e], T[q][8 + e]);
                }
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (i0 + ib + q < m) {
                const float x = acc[0][4 * g + q], y = acc[1][4 * g + q], z = acc[2][4 * g + q];
                float* o = rec_ptr(verts, i0 + ib + q, ld_verts, rm, F_VERTS) + v * 3;
                o[0] = T[q][0] * x + T[q][1] * y + T[q][2] * z + T[q][3];
                o[1] = T[q][4] * x + T[q][5] * y + T[q][6] * z + T[q][7];
                o[2] = T[q][8] * x + T[q][9] * y + T[q][10] * z + T[q][11];
            }
        }
    }
}

// ---- kernel 2, split-fp16 matrix-core form (the default since round 4) --------------------------------------------- //
// The same contraction [m, 218] x [218, 3 x 6890] with BOTH operands as fp16 hi/lo pairs and three v_mfma_f32_32x32x16_f16 per
// product (x.lo*b.hi + x.hi*b.lo + x.hi*b.hi, fp32 accumulate) -- the ResNet's operand format (csrc/common.h) applied to the
// dense part of the SMPL stage: 16 K per instruction instead of 2, i.e. ~5x the rate of the exact-fp32 MFMA / packed-FMA forms
// at 22 operand bits (vertices within 1e-6 of the fp32 forms; the tolerance of the path is 1e-4).  The host packs the basis as
// B-operand fragments (`dirs_split`: [K / 16][coordinate][hi, lo][k half][vertex][8 halves], pre-scaled by 2^13 so that the lo
// halves of millimetre-sized blend-shape entries are normal fp16 numbers); the features are scaled by 2^8 and split into LDS here
// (a feature beyond +-255 raises the saturation flag); the accumulator is scaled back by 2^-21 (exact).  D[instance][vertex]
// per coordinate as in the exact-fp32 form: a lane ends up with its vertex's blended position for 16 instances and runs the
// (4-sparse, vector-unit) skinning sum unchanged.  Same 1-D, XCD-aware grid as smpl_verts_kernel.
static constexpr int KCH = LDF / 16;     // 14 K chunks of 16 (rows >= 218 of the basis are zero)
#ifndef SMPL_PROBE_BITS                  // development builds: drop the stores (1), the skinning sum (2), the matrix-core loop (4)
#define SMPL_PROBE_BITS 0
#endif
#define SMPL_PROBE(bit) (((SMPL_PROBE_BITS) & (bit)) != 0)
static constexpr float DIRS_SPLIT_SCALE = 8192.f, FEAT_SPLIT_SCALE = 256.f;
// Round 5 (profiles/r05u: a launch without its stores / skinning sum / matrix loop is 14 / 10 / 35 us shorter, and 68 us of the 126-us call
// remain when all three are gone): (i) a workgroup keeps its 32 instances' records and features in LDS and walks `vpw` vertex tiles, so
// the 65-KB prologue is paid once per ~3 tiles instead of once per tile; (ii) the basis fragments are requested two K chunks ahead (three
// register sets) -- one chunk ahead made the 14-step loop 14 L2 round trips;
// (iii) a wave's 32 vertices x 8 instances go through a wave-private 3-KB LDS tile and leave as 16-byte stores of whole 384-byte rows
// (one instance's 32 vertices) instead of 4-byte stores 12 bytes apart.  Same arithmetic per vertex: the same bits as before.
typedef float f32x4_u __attribute__((ext_vector_type(4), aligned(4)));      // a record's vertex field is 4-byte aligned, no more
__global__ __launch_bounds__(256, 2) void smpl_verts_split_kernel(
    const shalf8* __restrict__ dsplit, int vpad, const float* __restrict__ feat, const float* __restrict__ A,
    const int* __restrict__ lbs_idx, const float* __restrict__ lbs_w, int nnz, int nv, int m, int vpw,
    float* __restrict__ verts, long long ld_verts, const RecMap rm) {
    __shared__ __attribute__((aligned(16))) float smem[IBM * LDA + KCH * 2 * 2 * IBM * 4 + 4 * 768];
    float (*sA)[LDA] = (float (*)[LDA])smem;
    shalf8* sF = (shalf8*)(smem + IBM * LDA);                    // [kc][plane][k half][instance]: the A-operand fragments
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float* stg = smem + IBM * LDA + KCH * 2 * 2 * IBM * 4 + wave * 768;      // this wave's [8 instances][32 vertices x 3] tile
    const int lc = lane & 31, lh = lane >> 5;
    const int nib = (m + IBM - 1) / IBM;
    const int L = xcd_remap(blockIdx.x, gridDim.x);
    const int vtiles = vpad / VT;
    const int t0 = (L / nib) * vpw, t1 = t0 + vpw < vtiles ? t0 + vpw : vtiles;
    const int i0 = (L % nib) * IBM;
    for (int e = threadIdx.x; e < IBM * LDA; e += 256) {
        const int ii = e / LDA;
        sA[ii][e % LDA] = (i0 + ii < m) ? A[(long long)(i0 + ii) * LDA + (e % LDA)] : 0.f;
    }
    bool sat = false;
    for (int e = threadIdx.x; e < IBM * KCH * 2; e += 256) {     // one 8-wide K group of one instance per step
        const int ii = e % IBM, g8 = e / IBM;                    // g8 = 2 kc + k half
        const float* f = feat + (long long)(i0 + ii) * LDF + g8 * 8;        // (the scratch rows are sized for m rounded up to IBM)
        shalf8 hi, lo;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float x = f[j] * FEAT_SPLIT_SCALE;
            sat = sat || (split_overflows(x) && i0 + ii < m);          // (rows beyond m are scratch)
            const float c = split_clamp(x);
            hi[j] = (shalf_t)c;
            lo[j] = (shalf_t)(c - (float)hi[j]);
        }
        sF[((g8 >> 1) * 2 + 0) * 2 * IBM + (g8 & 1) * IBM + ii] = hi;
        sF[((g8 >> 1) * 2 + 1) * 2 * IBM + (g8 & 1) * IBM + ii] = lo;
    }
    split_flag(sat);
    // B fragments of chunk kc: [kc][c][plane][lh][vertex]; this lane's vertex = its column of D (v < vpad always)
    int v = t0 * VT + wave * 32 + lc;
    constexpr int PF = 2;                                        // K chunks requested ahead (PF + 1 register sets of 24)
    shalf8 bh[PF + 1][3], bl[PF + 1][3];
    // (address = a uniform plane base + a 32-bit lane offset: the scalar unit walks the planes, a lane keeps ONE offset per tile)
    const unsigned long long plane = (unsigned long long)(2 * vpad) * 16ull;
    unsigned voff = 0;
    auto fetch = [&](int kc, int set) {                          // kc, set are constants after unrolling
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const char* ph = (const char*)dsplit + (unsigned long long)((kc * 3 + c) * 2 + 0) * plane;
            const char* pl = (const char*)dsplit + (unsigned long long)((kc * 3 + c) * 2 + 1) * plane;
            bh[set][c] = *(const shalf8*)(ph + voff);
            bl[set][c] = *(const shalf8*)(pl + voff);
        }
    };
    __syncthreads();
    constexpr float UNSCALE = 1.0f / (DIRS_SPLIT_SCALE * FEAT_SPLIT_SCALE);
    for (int t = t0; t < t1; ++t) {
        // (opaque per tile: as induction variables of this loop the 84 fragment addresses of a tile are 168 registers, and the feature
        //  fragments hoisted out of it another 112)
        asm volatile("" : "+v"(v) :: "memory");
        voff = (unsigned)(lh * vpad + v) * 16u;
        int lane_o = lane;                                       // (likewise: the 12 row addresses of a lane's stores are recomputed per tile, not kept)
        asm volatile("" : "+v"(lane_o));
#pragma unroll
        for (int kc = 0; kc < PF; ++kc) fetch(kc, kc);
        f32x16 acc[3];
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
#pragma unroll
        for (int kc = 0; kc < (SMPL_PROBE(4) ? 1 : KCH); ++kc) {
            if (kc + PF < KCH) fetch(kc + PF, (kc + PF) % (PF + 1));
            const shalf8 ah = sF[(kc * 2 + 0) * 2 * IBM + lh * IBM + lc], al = sF[(kc * 2 + 1) * 2 * IBM + lh * IBM + lc];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                acc[c] = mfma_split(al, bh[kc % (PF + 1)][c], acc[c]);
                acc[c] = mfma_split(ah, bl[kc % (PF + 1)][c], acc[c]);
                acc[c] = mfma_split(ah, bh[kc % (PF + 1)][c], acc[c]);
            }
        }
        const int vc = v, v0 = t * VT + wave * 32;               // this tile's vertex of the lane / first vertex of the wave
        v += VT;
        if (v0 >= nv) continue;                                  // (the padding of the last tile: the last tile of the last group)
        const bool whole = v0 + 32 <= nv;                        // every vertex of the wave exists: whole 384-byte rows
        // skinning, as in smpl_verts_kernel: the lane holds instances i0 + 8 g + 4 lh + {0..3} in accumulator rows 4 g .. 4 g + 3
        const int* vidx = lbs_idx + (long long)(vc < nv ? vc : nv - 1) * nnz;
        const float* vw = lbs_w + (long long)(vc < nv ? vc : nv - 1) * nnz;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int ib = 8 * g + 4 * lh;
            if (i0 + 8 * g >= m) break;
            if (i0 + ib < m && vc < nv) {
                float T[4][12];
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int e = 0; e < 12; ++e) T[q][e] = 0.f;
                for (int z = 0; z < (SMPL_PROBE(2) ? 1 : nnz); ++z) {
                    const int jj = vidx[z] * 12;
                    const float wv = vw[z];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f32x4* a4 = (const f32x4*)&sA[ib + q][jj];
                        const f32x4 r0 = a4[0], r1 = a4[1], r2 = a4[2];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            T[q][e] = fmaf(wv, r0[e], T[q][e]);
                            T[q][4 + e] = fmaf(wv, r1[e], T[q][4 + e]);
                            T[q][8 + e] = fmaf(wv, r2[e], T[q][8 + e]);
                        }
                    }
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (i0 + ib + q < m) {
                        const float x = acc[0][4 * g + q] * UNSCALE, y = acc[1][4 * g + q] * UNSCALE, z = acc[2][4 * g + q] * UNSCALE;
                        const float ox = T[q][0] * x + T[q][1] * y + T[q][2] * z + T[q][3];
                        const float oy = T[q][4] * x + T[q][5] * y + T[q][6] * z + T[q][7];
                        const float oz = T[q][8] * x + T[q][9] * y + T[q][10] * z + T[q][11];
                        if (whole) {
                            float* o = stg + (4 * lh + q) * 96 + lc * 3;
                            o[0] = ox; o[1] = oy; o[2] = oz;
                        } else {
                            if (SMPL_PROBE(1) && !(ox == 1234.5f && oy == 2.f)) continue;      // (probe: the values stay live, nothing is written)
                            float* o = rec_ptr(verts, i0 + ib + q, ld_verts, rm, F_VERTS) + vc * 3;
                            o[0] = ox; o[1] = oy; o[2] = oz;
                        }
                    }
                }
            }
            if (whole) {                                         // the tile's 8 rows of 384 bytes as 3 x 16 bytes per lane
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    const int o4 = (p * 64 + lane_o) * 4;        // float index inside the tile
                    const int il = o4 / 96, inst = i0 + 8 * g + il;
                    const f32x4 val = *(const f32x4*)(stg + o4);
                    if (inst < m && !(SMPL_PROBE(1) && val[0] != 1234.5f))
                        *(f32x4_u*)(rec_ptr(verts, inst, ld_verts, rm, F_VERTS) + v0 * 3 + (o4 - il * 96)) = val;
                }
            }
        }
    }
}

// ---- kernel 3 ------------------------------------------------------------ //
// joints = cocoplus_regressor^T verts (batch_smpl.py:154-157); kps = s*(xy + t) (projection.py:25-29)
__global__ __launch_bounds__(256) void smpl_joints_kernel(
    const float* __restrict__ verts, const int* __restrict__ kptr, const int* __restrict__ kidx,
    const float* __restrict__ kval, const float* __restrict__ cams, int ld_cam, int nv, int nk,
    float* __restrict__ joints, float* __restrict__ kps, long long ld_verts, long long ld_joints,
    long long ld_kps, const RecMap rm, const float* __restrict__ omegas) {
    const int inst = blockIdx.x;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const float* vb = rec_ptr(const_cast<float*>(verts), inst, ld_verts, rm, F_VERTS);
    const int cam_inst = rm.n_per ? inst % rm.n_per : inst;     // record mode: every container projects with omega_0's camera (tester.py:211-213)
    if (rm.n_per) {
        // the container's raw omega, its shape and the camera it was projected with: the record's remaining fields
        // (make_fetch_dict, tester.py:217-227), written here instead of by three copies per container
        const float* om = omegas + (long long)inst * 85;
        const float* cm = cams + (long long)cam_inst * ld_cam;
        const int t = threadIdx.x;
        if (t < 85) rec_ptr(joints, inst, ld_joints, rm, F_OMEGAS)[t] = om[t];
        else if (t < 95) rec_ptr(joints, inst, ld_joints, rm, F_SHAPES)[t - 85] = om[75 + (t - 85)];
        else if (t < 98) rec_ptr(joints, inst, ld_joints, rm, F_CAMS)[t - 95] = cm[t - 95];
    }
    for (int k = wave; k < nk; k += 4) {
        float sx = 0.f, sy = 0.f, sz = 0.f;
        for (int e = kptr[k] + lane; e < kptr[k + 1]; e += 64) {
            const float w = kval[e];
            const float* p = vb + (long long)kidx[e] * 3;
            sx = fmaf(w, p[0], sx); sy = fmaf(w, p[1], sy); sz = fmaf(w, p[2], sz);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            sx += __shfl_xor(sx, o); sy += __shfl_xor(sy, o); sz += __shfl_xor(sz, o);
        }
        if (lane == 0) {
            float* jo = rec_ptr(joints, inst, ld_joints, rm, F_JOINTS) + k * 3;
            jo[0] = sx; jo[1] = sy; jo[2] = sz;
            if (kps && cams) {
                const float* cm = cams + (long long)cam_inst * ld_cam;
                float* ko = rec_ptr(kps, inst, ld_kps, rm, F_KPS) + k * 2;
                ko[0] = cm[0] * (sx + cm[1]);
                ko[1] = cm[0] * (sy + cm[2]);
            }
        }
    }
}

// ---- batch_global_rigid_transformation on its own (src/tf_smpl/batch_lbs.py:133-194) ---- //
// rotate_base (batch_lbs.py:151-158): root rotation = R_0 . diag(1, -1, -1), i.e. columns 1 and 2 of R_0 negated (exact).
// One thread per instance walks the kinematic chain in index order (parents[i] < i), with the same expression
// order as smpl_pose_kernel: results[i] = results[parent] . [[R_i, J_i - J_parent], [0, 1]]; new_J = its
// translation; A = results - [0 | results . [J; 0]].
__global__ __launch_bounds__(64) void smpl_fk_kernel(const float* __restrict__ Rs, const float* __restrict__ Js,
                                                     const int* __restrict__ parents, int m, float* __restrict__ new_j,
                                                     float* __restrict__ A44, int rotate_base) {
    const int inst = blockIdx.x * 64 + threadIdx.x;
    if (inst >= m) return;
    const float* R = Rs + (long long)inst * NJ * 9;
    const float* J = Js + (long long)inst * NJ * 3;
    float G[NJ][12];
#pragma unroll 1
    for (int i = 0; i < NJ; ++i) {
        const int p = i == 0 ? -1 : parents[i];
        const float* Lc = R + i * 9;
        float t[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) t[c] = J[i * 3 + c] - (p >= 0 ? J[p * 3 + c] : 0.0f);
        if (p < 0) {
#pragma unroll
            for (int e = 0; e < 9; ++e) G[i][e] = (rotate_base && e % 3 != 0) ? -Lc[e] : Lc[e];
#pragma unroll
            for (int c = 0; c < 3; ++c) G[i][9 + c] = t[c];
        } else {
#pragma unroll
            for (int r = 0; r < 3; ++r) {
#pragma unroll
                for (int cc = 0; cc < 3; ++cc)
                    G[i][r * 3 + cc] = G[p][r * 3 + 0] * Lc[0 * 3 + cc] + G[p][r * 3 + 1] * Lc[1 * 3 + cc] + G[p][r * 3 + 2] * Lc[2 * 3 + cc];
                G[i][9 + r] = G[p][r * 3 + 0] * t[0] + G[p][r * 3 + 1] * t[1] + G[p][r * 3 + 2] * t[2] + G[p][9 + r];
            }
        }
        float* nj = new_j + ((long long)inst * NJ + i) * 3;
        float* o = A44 + ((long long)inst * NJ + i) * 16;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const float bone = G[i][r * 3 + 0] * J[i * 3 + 0] + G[i][r * 3 + 1] * J[i * 3 + 1] + G[i][r * 3 + 2] * J[i * 3 + 2];
            nj[r] = G[i][9 + r];
            o[r * 4 + 0] = G[i][r * 3 + 0]; o[r * 4 + 1] = G[i][r * 3 + 1]; o[r * 4 + 2] = G[i][r * 3 + 2];
            o[r * 4 + 3] = G[i][9 + r] - bone;
        }
        // bottom row of results - init_bone: [0, 0, 0, 1 - 0] (the homogeneous row; init_bone's w component is 0)
        o[12] = 0.f; o[13] = 0.f; o[14] = 0.f; o[15] = 1.f;
    }
}

extern "C" int hmmr_global_rigid_transformation(const float* Rs, const float* Js, const int32_t* parents, int m,
                                                float* new_j, float* A, int rotate_base, void* stream) {
    HMMR_REQUIRE(Rs && Js && parents && new_j && A && m > 0, "hmmr_global_rigid_transformation: bad arguments");
    hipLaunchKernelGGL(smpl_fk_kernel, dim3((m + 63) / 64), dim3(64), 0, (hipStream_t)stream, Rs, Js, (const int*)parents, m, new_j, A, rotate_base);
    HMMR_CHECK_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------- //
static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

extern "C" size_t hmmr_smpl_workspace_bytes(int m) {
    if (m <= 0) return 0;
    const size_t mp = ((size_t)m + IBM - 1) / IBM * IBM;       // the verts kernels read whole instance groups
    return align_up(mp * LDF * 4, 256) + align_up(mp * LDA * 4, 256);
}

static int smpl_launch(const hmmr_smpl_consts_t* c, const float* theta, int ld_theta, const float* beta,
                       int ld_beta, const float* cams, int ld_cam, int m, float* verts, float* joints,
                       float* kps, float* rs, long long ld_verts, long long ld_joints, long long ld_kps,
                       long long ld_rs, void* ws, size_t ws_bytes, void* stream, const RecMap& rm = RecMap{},
                       const float* omegas = nullptr) {
    HMMR_REQUIRE(c && theta && beta && verts && joints && ws, "hmmr_smpl_fwd: null argument");
    HMMR_REQUIRE(m > 0, "hmmr_smpl_fwd: m must be positive");
    HMMR_REQUIRE(c->lbs_nnz >= 1 && c->lbs_nnz <= NJ, "hmmr_smpl_fwd: lbs_nnz=%d out of range", c->lbs_nnz);
    HMMR_REQUIRE(ws_bytes >= hmmr_smpl_workspace_bytes(m), "hmmr_smpl_fwd: workspace too small");
    HMMR_REQUIRE(!kps || cams, "hmmr_smpl_fwd: kps requested without cams");
    HMMR_REQUIRE(c->vpad >= (c->num_verts + VT - 1) / VT * VT && c->vpad % VT == 0,
                 "hmmr_smpl_fwd: dirs row stride vpad=%d must be a multiple of %d covering num_verts=%d", c->vpad, VT, c->num_verts);
    hipStream_t s = (hipStream_t)stream;
    float* feat = (float*)ws;
    float* A = (float*)((char*)ws + align_up(((size_t)m + IBM - 1) / IBM * IBM * LDF * 4, 256));
    const int vtiles = (c->num_verts + VT - 1) / VT;
    hipLaunchKernelGGL(smpl_pose_kernel, dim3((m + 7) / 8), dim3(256), 0, s, theta, ld_theta, beta, ld_beta,
                       c->j_template, c->j_shapedirs, c->parents, m, feat, A, rs, ld_rs, rm);
    HMMR_CHECK_HIP(hipGetLastError());
    // Default: the vector-unit form.  Measured on MI355X (tools/smpl_bench.py, profiles/r03b): 86.9 us per 256 instances
    // against 106.7 us for the matrix-core form below -- v_mfma_f32_32x32x2_f32 runs at the vector units' own fp32 rate,
    // so the MFMA form can only win on operand delivery, and the packed-FMA kernel (two instances per v_pk_fma_f32, the
    // coefficients as LDS broadcasts) already has the cheaper one.  hmmr_debug_t.smpl_blend_mfma selects the MFMA form.
    // Round 4: with `dirs_split` packed, the blend product runs on the matrix cores with split-fp16 operands (three fp16 MFMAs per
    // product, 16 K per instruction): smpl_verts_split_kernel.  hmmr_debug_t.smpl_blend_mfma: 0 = that default (the packed-FMA
    // vector form when dirs_split is NULL), 1 = the exact-fp32 MFMA form, 2 = the packed-FMA vector form.
    const int form = hmmr_debug_state()->smpl_blend_mfma;
    if (form == 0 && c->dirs_split)
    {
        // vertex tiles per workgroup: as many as it takes for the grid to be one resident round (two workgroups per CU)
        const int nib = (m + IBM - 1) / IBM, vt = c->vpad / VT;
        int vpw = (vt * nib + 511) / 512;
        vpw = vpw < 1 ? 1 : (vpw > vt ? vt : vpw);
        hipLaunchKernelGGL(smpl_verts_split_kernel, dim3(((vt + vpw - 1) / vpw) * nib), dim3(256), 0, s, (const shalf8*)c->dirs_split,
                           c->vpad, (const float*)feat, (const float*)A, c->lbs_idx, c->lbs_w, c->lbs_nnz,
                           c->num_verts, m, vpw, verts, ld_verts, rm);
    }
    else if (form != 1)
        hipLaunchKernelGGL(smpl_verts_kernel, dim3(vtiles * ((m + IB - 1) / IB)), dim3(VT), 0, s, c->dirs,
                           c->vpad, (const float*)feat, (const float*)A, c->lbs_idx, c->lbs_w, c->lbs_nnz,
                           c->num_verts, m, verts, ld_verts, rm);
    else
        hipLaunchKernelGGL(smpl_verts_mfma_kernel, dim3(vtiles, (m + IBM - 1) / IBM), dim3(256), 0, s, c->dirs,
                           c->vpad, (const float*)feat, (const float*)A, c->lbs_idx, c->lbs_w, c->lbs_nnz,
                           c->num_verts, m, verts, ld_verts, rm);
    HMMR_CHECK_HIP(hipGetLastError());
    hipLaunchKernelGGL(smpl_joints_kernel, dim3(m), dim3(256), 0, s, (const float*)verts, c->kreg_ptr,
                       c->kreg_idx, c->kreg_val, cams, ld_cam, c->num_verts, c->num_kps, joints, kps,
                       ld_verts, ld_joints, ld_kps, rm, omegas);
    HMMR_CHECK_HIP(hipGetLastError());
    return 0;
}

extern "C" int hmmr_smpl_fwd(const hmmr_smpl_consts_t* c, const float* theta, int ld_theta,
                             const float* beta, int ld_beta, const float* cams, int ld_cam, int m,
                             float* verts, float* joints, float* kps, float* rs,
                             void* ws, size_t ws_bytes, void* stream) {
    HMMR_REQUIRE(c, "hmmr_smpl_fwd: null argument");
    return smpl_launch(c, theta, ld_theta, beta, ld_beta, cams, ld_cam, m, verts, joints, kps, rs,
                       (long long)c->num_verts * 3, (long long)c->num_kps * 3, (long long)c->num_kps * 2,
                       NJ * 9, ws, ws_bytes, stream);
}

extern "C" int hmmr_smpl_fwd_strided(const hmmr_smpl_consts_t* c, const float* theta, int ld_theta,
                                     const float* beta, int ld_beta, const float* cams, int ld_cam, int m,
                                     float* verts, float* joints, float* kps, float* rs, int64_t ld_out,
                                     void* ws, size_t ws_bytes, void* stream) {
    return smpl_launch(c, theta, ld_theta, beta, ld_beta, cams, ld_cam, m, verts, joints, kps, rs,
                       ld_out, ld_out, ld_out, ld_out, ws, ws_bytes, stream);
}

// R containers of n frames in ONE launch set, every field of the packed per-frame record written in place.
extern "C" int hmmr_smpl_fwd_records(const hmmr_smpl_consts_t* c, const float* omegas, int num_containers, int n,
                                     float* rec, int64_t ld_rec, const int32_t* field_offsets, void* ws, size_t ws_bytes,
                                     void* stream) {
    HMMR_REQUIRE(c && omegas && rec && field_offsets, "hmmr_smpl_fwd_records: null argument");
    HMMR_REQUIRE(num_containers >= 1 && num_containers <= HMMR_MAX_REGRESSORS && n > 0, "hmmr_smpl_fwd_records: bad container count / frames");
    RecMap rm;
    rm.n_per = n;
    for (int r = 0; r < HMMR_MAX_REGRESSORS; ++r)
        for (int f = 0; f < F_COUNT; ++f) {
            rm.off[r][f] = r < num_containers ? field_offsets[r * F_COUNT + f] : 0;
            HMMR_REQUIRE(rm.off[r][f] >= 0 && rm.off[r][f] < ld_rec, "hmmr_smpl_fwd_records: field offset outside the record");
        }
    // instance r*n + i: theta / beta = columns 3..74 / 75..84 of omegas row r*n + i; camera = omegas row i (container 0)
    return smpl_launch(c, omegas + 3, 85, omegas + 75, 85, omegas, 85, num_containers * n, rec, rec, rec, rec,
                       ld_rec, ld_rec, ld_rec, ld_rec, ws, ws_bytes, stream, rm, omegas);
}
