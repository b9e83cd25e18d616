// On-device evaluation metrics of src/evaluation/eval_util.py:
//   compute_error_3d     per frame: MPJPE after pelvis alignment and after Procrustes alignment
//                        (eval_util.py:30-60, align_by_pelvis :158-174, compute_similarity_transform :177-232)
//   compute_error_accel  per interior frame: mean || (X[i-1]-2X[i]+X[i+1])_pred - (...)_gt ||   (:63-94)
//   compute_accel        per interior frame: mean || X[i-1]-2X[i]+X[i+1] ||                     (:14-27)
//   compute_error_verts  per frame: mean vertex distance                                        (:140-155)
// so the joints / vertices that the SMPL stage leaves in HBM can be scored without a PCIe trip.
// Inputs fp32, arithmetic fp64 (one lane per frame for the Procrustes problem: a 3x3 SVD by cyclic
// Jacobi on K^T K; the work is a few hundred flops per frame, the kernel is latency-bound).
#include "common.h"
#include "hmmr_hip.h"

namespace {
constexpr int MAXK = 32;

__device__ void jacobi_eig3(double A[3][3], double V[3][3]) {
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) V[i][j] = i == j ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 30; ++sweep) {
        const double off = A[0][1] * A[0][1] + A[0][2] * A[0][2] + A[1][2] * A[1][2];
        if (off < 1e-300) break;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                if (fabs(A[p][q]) < 1e-300) continue;
                const double theta = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < 3; ++k) {              // A <- A J
                    const double akp = A[k][p], akq = A[k][q];
                    A[k][p] = c * akp - s * akq; A[k][q] = s * akp + c * akq;
                }
                for (int k = 0; k < 3; ++k) {              // A <- J^T A
                    const double apk = A[p][k], aqk = A[q][k];
                    A[p][k] = c * apk - s * aqk; A[q][k] = s * apk + c * aqk;
                }
                for (int k = 0; k < 3; ++k) {              // V <- V J
                    const double vkp = V[k][p], vkq = V[k][q];
                    V[k][p] = c * vkp - s * vkq; V[k][q] = s * vkp + c * vkq;
                }
            }
    }
}

__device__ double det3(const double M[3][3]) {
    return M[0][0] * (M[1][1] * M[2][2] - M[1][2] * M[2][1]) - M[0][1] * (M[1][0] * M[2][2] - M[1][2] * M[2][0]) +
           M[0][2] * (M[1][0] * M[2][1] - M[1][1] * M[2][0]);
}

// one lane per frame
__global__ void eval_joints_kernel(const float* __restrict__ gt, const float* __restrict__ pred, int n, int k,
                                   int left_id, int right_id, float* __restrict__ mpjpe,
                                   float* __restrict__ pa_mpjpe) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= n) return;
    double G[MAXK][3], P[MAXK][3];
    const float* g = gt + (long long)f * k * 3;
    const float* p = pred + (long long)f * k * 3;
    double pg[3], pp[3];
    for (int c = 0; c < 3; ++c) {
        pg[c] = ((double)g[left_id * 3 + c] + (double)g[right_id * 3 + c]) / 2.0;
        pp[c] = ((double)p[left_id * 3 + c] + (double)p[right_id * 3 + c]) / 2.0;
    }
    double err = 0.0;
    for (int j = 0; j < k; ++j) {
        double d2 = 0.0;
        for (int c = 0; c < 3; ++c) {
            G[j][c] = (double)g[j * 3 + c] - pg[c];
            P[j][c] = (double)p[j * 3 + c] - pp[c];
            const double d = G[j][c] - P[j][c];
            d2 += d * d;
        }
        err += sqrt(d2);
    }
    mpjpe[f] = (float)(err / k);
    // ---- compute_similarity_transform(S1 = pred, S2 = gt)
    double mu1[3] = {0, 0, 0}, mu2[3] = {0, 0, 0};
    for (int j = 0; j < k; ++j)
        for (int c = 0; c < 3; ++c) { mu1[c] += P[j][c]; mu2[c] += G[j][c]; }
    for (int c = 0; c < 3; ++c) { mu1[c] /= k; mu2[c] /= k; }
    double K[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}}, var1 = 0.0;
    for (int j = 0; j < k; ++j) {
        double x1[3], x2[3];
        for (int c = 0; c < 3; ++c) { x1[c] = P[j][c] - mu1[c]; x2[c] = G[j][c] - mu2[c]; var1 += x1[c] * x1[c]; }
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b) K[a][b] += x1[a] * x2[b];              // K = X1 X2^T
    }
    // K = U S V^T.  Eigen-decompose K^T K = V S^2 V^T, then U = K V S^-1 (third column by cross
    // product when the smallest singular value vanishes).
    double KtK[3][3], V[3][3];
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) KtK[a][b] = K[0][a] * K[0][b] + K[1][a] * K[1][b] + K[2][a] * K[2][b];
    jacobi_eig3(KtK, V);
    int ord[3] = {0, 1, 2};                                                      // descending singular values
    for (int a = 0; a < 2; ++a)
        for (int b = a + 1; b < 3; ++b)
            if (KtK[ord[b]][ord[b]] > KtK[ord[a]][ord[a]]) { const int t = ord[a]; ord[a] = ord[b]; ord[b] = t; }
    double Vs[3][3], U[3][3], sv[3];
    for (int i = 0; i < 3; ++i) {
        sv[i] = sqrt(fmax(KtK[ord[i]][ord[i]], 0.0));
        for (int r = 0; r < 3; ++r) Vs[r][i] = V[r][ord[i]];
    }
    for (int i = 0; i < 3; ++i)
        for (int r = 0; r < 3; ++r)
            U[r][i] = sv[i] > 1e-12 * (sv[0] + 1e-300)
                          ? (K[r][0] * Vs[0][i] + K[r][1] * Vs[1][i] + K[r][2] * Vs[2][i]) / sv[i] : 0.0;
    if (!(sv[2] > 1e-12 * (sv[0] + 1e-300))) {                                   // rank-deficient: complete U
        U[0][2] = U[1][0] * U[2][1] - U[2][0] * U[1][1];
        U[1][2] = U[2][0] * U[0][1] - U[0][0] * U[2][1];
        U[2][2] = U[0][0] * U[1][1] - U[1][0] * U[0][1];
    }
    // R = V Z U^T with Z = diag(1, 1, sign(det(U V^T)))
    double UVt[3][3];
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) UVt[a][b] = U[a][0] * Vs[b][0] + U[a][1] * Vs[b][1] + U[a][2] * Vs[b][2];
    const double dz = det3(UVt) < 0 ? -1.0 : 1.0;
    double R[3][3];
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) R[a][b] = Vs[a][0] * U[b][0] + Vs[a][1] * U[b][1] + dz * Vs[a][2] * U[b][2];
    double trRK = 0.0;
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) trRK += R[a][b] * K[b][a];
    const double scale = trRK / var1;
    double t[3];
    for (int a = 0; a < 3; ++a) t[a] = mu2[a] - scale * (R[a][0] * mu1[0] + R[a][1] * mu1[1] + R[a][2] * mu1[2]);
    double epa = 0.0;
    for (int j = 0; j < k; ++j) {
        double d2 = 0.0;
        for (int a = 0; a < 3; ++a) {
            const double h = scale * (R[a][0] * P[j][0] + R[a][1] * P[j][1] + R[a][2] * P[j][2]) + t[a];
            const double d = G[j][a] - h;
            d2 += d * d;
        }
        epa += sqrt(d2);
    }
    pa_mpjpe[f] = (float)(epa / k);
}

// one lane per interior frame i in [0, n-2): second difference centred on frame i+1
__global__ void eval_accel_kernel(const float* __restrict__ gt, const float* __restrict__ pred, int n, int k,
                                  float* __restrict__ accel_pred, float* __restrict__ accel_err) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n - 2) return;
    double sa = 0.0, se = 0.0;
    for (int j = 0; j < k; ++j) {
        double na = 0.0, ne = 0.0;
        for (int c = 0; c < 3; ++c) {
            const long long o = ((long long)i * k + j) * 3 + c, s = (long long)k * 3;
            const double ap = (double)pred[o] - 2.0 * (double)pred[o + s] + (double)pred[o + 2 * s];
            na += ap * ap;
            if (gt) {
                const double ag = (double)gt[o] - 2.0 * (double)gt[o + s] + (double)gt[o + 2 * s];
                ne += (ap - ag) * (ap - ag);
            }
        }
        sa += sqrt(na); se += sqrt(ne);
    }
    if (accel_pred) accel_pred[i] = (float)(sa / k);
    if (accel_err && gt) accel_err[i] = (float)(se / k);
}

// one workgroup per frame: mean over vertices of || gt - pred ||
__global__ __launch_bounds__(256) void eval_verts_kernel(const float* __restrict__ gt, const float* __restrict__ pred,
                                                         int nv, long long ld_gt, long long ld_pred,
                                                         float* __restrict__ out) {
    __shared__ double red[4];
    const float* g = gt + (long long)blockIdx.x * ld_gt;
    const float* p = pred + (long long)blockIdx.x * ld_pred;
    double s = 0.0;
    for (int v = threadIdx.x; v < nv; v += 256) {
        const double dx = (double)g[v * 3] - (double)p[v * 3], dy = (double)g[v * 3 + 1] - (double)p[v * 3 + 1],
                     dz = (double)g[v * 3 + 2] - (double)p[v * 3 + 2];
        s += sqrt(dx * dx + dy * dy + dz * dz);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = (float)((red[0] + red[1] + red[2] + red[3]) / nv);
}
}  // namespace

extern "C" int hmmr_eval_joints(const float* gt, const float* pred, int n, int k, int left_id, int right_id,
                                float* mpjpe, float* pa_mpjpe, float* accel_pred, float* accel_err, void* stream) {
    HMMR_REQUIRE(pred && n > 0 && k > 0 && k <= MAXK, "hmmr_eval_joints: bad arguments (k <= %d)", MAXK);
    HMMR_REQUIRE(left_id >= 0 && left_id < k && right_id >= 0 && right_id < k, "hmmr_eval_joints: bad hip ids");
    hipStream_t s = (hipStream_t)stream;
    if (mpjpe || pa_mpjpe) {
        HMMR_REQUIRE(gt && mpjpe && pa_mpjpe, "hmmr_eval_joints: MPJPE needs gt and both outputs");
        hipLaunchKernelGGL(eval_joints_kernel, dim3((n + 63) / 64), dim3(64), 0, s, gt, pred, n, k, left_id, right_id,
                           mpjpe, pa_mpjpe);
        HMMR_CHECK_HIP(hipGetLastError());
    }
    if ((accel_pred || accel_err) && n > 2) {
        hipLaunchKernelGGL(eval_accel_kernel, dim3((n - 2 + 63) / 64), dim3(64), 0, s, gt, pred, n, k, accel_pred, accel_err);
        HMMR_CHECK_HIP(hipGetLastError());
    }
    return 0;
}

extern "C" int hmmr_eval_verts(const float* gt, int64_t ld_gt, const float* pred, int64_t ld_pred, int n, int nv,
                               float* err, void* stream) {
    HMMR_REQUIRE(gt && pred && err && n > 0 && nv > 0, "hmmr_eval_verts: bad arguments");
    hipLaunchKernelGGL(eval_verts_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, gt, pred, nv, (long long)ld_gt,
                       (long long)ld_pred, err);
    HMMR_CHECK_HIP(hipGetLastError());
    return 0;
}
