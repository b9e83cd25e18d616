// f_movie temporal encoder for gfx950: az_fc2_groupnorm / az_fc_block2
// (src/models.py:121-228).  Per residual block, on a [b, t, 1, 2048] tensor:
//   GN(32 groups, stats over time x 64 channels, eps 1e-6) -> ReLU ->
//   conv [3,1] SAME + bias -> GN -> ReLU -> conv [3,1] SAME + bias -> + input
// The two convolutions run through the implicit-GEMM kernel (M = b*t,
// K = 3*2048, N = 2048; zero rows outside the window come from its bounds
// check); the residual trunk and the GN statistics stay fp32 in every mode,
// only the GEMM operands take the struct's dtype.
#include "common.h"
#include "hmmr_hip.h"

#define GN_EPS 1e-6f
// K = 3*2048 is long and M = b*t is short (640 rows for 32 windows): 4 K-slices quadruple the
// number of workgroups.  Fixed per layer and operand mode, never derived from b, so a window's result does not
// depend on how many windows share the launch.  Split operands: 5 slices of the 128x256 ring tile (40 tiles x 5 = 200
// workgroups on 256 CUs, 39 K steps each) measured 0.427 ms per f_movie pass of 32 windows against 0.477 for 4 slices
// of the 256x128 tile (tools/stage_bench.py, round 3; 6 / 8 slices and the two-stage tiles are slower).
#define TEMPORAL_SPLIT_K_MAX 5
static inline int temporal_split_k(int dtype) { return dtype == HMMR_F16X3 ? 5 : 4; }

__device__ __forceinline__ float block_sum_256(float v, float* red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();                       // protect `red` from the previous use
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// One workgroup per (window b, group g).  tf.contrib.layers.group_norm with
// reduction_axes=(-3,-2) on [b,t,1,c]: mean and POPULATION variance over
// (t, c/groups); gain = rsqrt(var+eps)*gamma; offset = -mean*gain + beta.
template <typename TO>
__global__ __launch_bounds__(256) void groupnorm_relu_kernel(const float* __restrict__ x,
                                                             const float* __restrict__ gamma,
                                                             const float* __restrict__ beta,
                                                             TO* __restrict__ out, int t, int c, int groups) {
    __shared__ float red[4];
    const int cpg = c / groups;
    const int b = blockIdx.x / groups, g = blockIdx.x % groups;
    const int cnt = t * cpg;
    const float* xb = x + (long long)b * t * c + g * cpg;
    float s = 0.f;
    for (int i = threadIdx.x; i < cnt; i += 256) s += xb[(i / cpg) * c + (i % cpg)];
    const float mean = block_sum_256(s, red) / (float)cnt;
    float q = 0.f;
    for (int i = threadIdx.x; i < cnt; i += 256) {
        const float d = xb[(i / cpg) * c + (i % cpg)] - mean;
        q += d * d;
    }
    const float var = block_sum_256(q, red) / (float)cnt;
    const float rstd = rsqrtf(var + GN_EPS);
    TO* ob = out + (long long)b * t * c + g * cpg;
    const int v8 = cpg / 8;                // 8-channel vectors per row of the group (cpg % 8 == 0, host-checked)
    for (int i = threadIdx.x; i < t * v8; i += 256) {
        const int tt = i / v8, j = (i % v8) * 8;
        float xv[8], gm[8], bt[8], o[8];
        load8(xb + tt * c + j, xv); load8(gamma + g * cpg + j, gm); load8(beta + g * cpg + j, bt);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            // x*gain + (beta - mean*gain) evaluated as (x - mean)*gain + beta: same value,
            // without the cancellation of two large products when var -> 0
            const float gain = rstd * gm[e];
            o[e] = fmaxf((xv[e] - mean) * gain + bt[e], 0.f);
        }
        store8(ob + tt * c + j, o);
    }
}

extern "C" int hmmr_groupnorm_relu(const float* x, const float* gamma, const float* beta, int b, int t,
                                   int c, int groups, void* out, int out_dtype, void* stream) {
    HMMR_REQUIRE(x && gamma && beta && out, "hmmr_groupnorm_relu: null argument");
    HMMR_REQUIRE(b > 0 && t > 0 && groups > 0 && c % groups == 0 && (c / groups) % 8 == 0,
                 "hmmr_groupnorm_relu: bad shape (channels per group must be a multiple of 8)");
    hipStream_t s = (hipStream_t)stream;
    if (out_dtype == HMMR_BF16)
        hipLaunchKernelGGL(groupnorm_relu_kernel<bf16_t>, dim3(b * groups), dim3(256), 0, s, x, gamma, beta,
                           (bf16_t*)out, t, c, groups);
    else if (out_dtype == HMMR_F32)
        hipLaunchKernelGGL(groupnorm_relu_kernel<float>, dim3(b * groups), dim3(256), 0, s, x, gamma, beta,
                           (float*)out, t, c, groups);
    else if (out_dtype == HMMR_F16X3)
        hipLaunchKernelGGL(groupnorm_relu_kernel<bsplit_t>, dim3(b * groups), dim3(256), 0, s, x, gamma, beta,
                           (bsplit_t*)out, t, c, groups);
    else { hmmr_set_error("hmmr_groupnorm_relu: bad dtype %d", out_dtype); return -1; }
    HMMR_CHECK_HIP(hipGetLastError());
    return 0;
}

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

extern "C" size_t hmmr_temporal_workspace_bytes(int b, int t, int dtype) {
    if (b <= 0 || t <= 0) return 0;
    const size_t m = (size_t)b * t, e = dtype == HMMR_BF16 ? 2 : 4;
    // h (operand dtype) + h1 (fp32) + two fp32 trunk buffers + split-K partial planes
    return align_up(m * 2048 * e, 256) + 3 * align_up(m * 2048 * 4, 256) +
           align_up(hmmr_conv_splitk_workspace_bytes((int)m, 2048, TEMPORAL_SPLIT_K_MAX), 256);
}

extern "C" int hmmr_temporal_fwd(const hmmr_temporal_weights_t* w, const float* phi, int b, int t,
                                 float* strips, void* ws, size_t ws_bytes, void* stream) {
    HMMR_REQUIRE(w && phi && strips && ws, "hmmr_temporal_fwd: null argument");
    HMMR_REQUIRE(b > 0 && t > 0, "hmmr_temporal_fwd: bad shape");
    HMMR_REQUIRE(w->num_blocks >= 1 && w->num_blocks <= HMMR_MAX_TEMPORAL_BLOCKS, "hmmr_temporal_fwd: bad num_blocks");
    HMMR_REQUIRE(ws_bytes >= hmmr_temporal_workspace_bytes(b, t, w->dtype), "hmmr_temporal_fwd: workspace too small");
    const int C = 2048;
    const size_t m = (size_t)b * t, e = w->dtype == HMMR_BF16 ? 2 : 4;
    char* p = (char*)ws;
    void* h = p;              p += align_up(m * C * e, 256);
    float* h1 = (float*)p;    p += align_up(m * C * 4, 256);
    float* net[2];
    net[0] = (float*)p;       p += align_up(m * C * 4, 256);
    net[1] = (float*)p;       p += align_up(m * C * 4, 256);
    void* skws = p;
    const size_t skbytes = hmmr_conv_splitk_workspace_bytes((int)m, C, TEMPORAL_SPLIT_K_MAX);
    const float* cur = phi;
    for (int i = 0; i < w->num_blocks; ++i) {
        const hmmr_temporal_block_t& B = w->block[i];
        float* dst = (i == w->num_blocks - 1) ? strips : net[i & 1];
        if (hmmr_groupnorm_relu(cur, B.gn1_gamma, B.gn1_beta, b, t, C, 32, h, w->dtype, stream)) return -2;
        hmmr_conv_desc_t d = {};
        d.in = h; d.w = B.conv1.w; d.scale = B.conv1.scale; d.shift = B.conv1.shift; d.tile = B.conv1.tile;
        d.out = h1; d.in_dtype = w->dtype; d.out_dtype = HMMR_F32;
        d.n_img = b; d.hin = t; d.win = 1; d.cin = C;
        d.in_img_stride = (int64_t)t * C; d.in_row_stride = C; d.in_px_stride = C;
        d.kh = 3; d.kw = 1; d.sy = d.sx = 1; d.py = 1; d.px = 0; d.ho = t; d.wo = 1; d.cout = C; d.ldo = C;
        d.split_k = temporal_split_k(w->dtype); d.ws = skws; d.ws_bytes = skbytes;
        if (hmmr_conv_gemm(&d, stream)) return -2;
        if (hmmr_groupnorm_relu(h1, B.gn2_gamma, B.gn2_beta, b, t, C, 32, h, w->dtype, stream)) return -2;
        d.w = B.conv2.w; d.scale = B.conv2.scale; d.shift = B.conv2.shift; d.tile = B.conv2.tile;
        d.res = cur; d.ldr = C; d.out = dst;            // residual adds the BLOCK INPUT (models.py:226)
        if (hmmr_conv_gemm(&d, stream)) return -2;
        cur = dst;
    }
    return 0;
}

// ------------------------------------------------------------------------- //
// Hallucinator fc2_res (src/models.py:270-296, pred_mode == 'hal'):
//   phi + fc3(relu(fc2(relu(fc1(phi)))))     three 2048x2048 fully-connected layers.
extern "C" size_t hmmr_hallucinator_workspace_bytes(int m, int dtype) {
    if (m <= 0) return 0;
    const size_t e = dtype == HMMR_BF16 ? 2 : 4;
    return 3 * align_up((size_t)m * 2048 * e, 256) +
           align_up(hmmr_conv_splitk_workspace_bytes(m, 2048, TEMPORAL_SPLIT_K_MAX), 256);
}

template <typename TO>
__global__ void hal_cast_kernel(const float* __restrict__ in, TO* __restrict__ out, long long n8) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n8) return;
    float v[8];
    load8(in + i * 8, v);
    store8(out + i * 8, v);
}

extern "C" int hmmr_hallucinator_fwd(const hmmr_hallucinator_weights_t* w, const float* phi, int m,
                                     float* out, void* ws, size_t ws_bytes, void* stream) {
    HMMR_REQUIRE(w && phi && out && ws, "hmmr_hallucinator_fwd: null argument");
    HMMR_REQUIRE(m > 0, "hmmr_hallucinator_fwd: m must be positive");
    HMMR_REQUIRE(ws_bytes >= hmmr_hallucinator_workspace_bytes(m, w->dtype), "hmmr_hallucinator_fwd: workspace too small");
    const int C = 2048;
    const size_t e = w->dtype == HMMR_BF16 ? 2 : 4;
    char* p = (char*)ws;
    void* x = p;  p += align_up((size_t)m * C * e, 256);
    void* h1 = p; p += align_up((size_t)m * C * e, 256);
    void* h2 = p; p += align_up((size_t)m * C * e, 256);
    void* sk = p;
    const size_t skb = hmmr_conv_splitk_workspace_bytes(m, C, TEMPORAL_SPLIT_K_MAX);
    hipStream_t s = (hipStream_t)stream;
    const void* xin = phi;
    if (w->dtype != HMMR_F32) {
        const long long n8 = (long long)m * C / 8;
        if (w->dtype == HMMR_BF16)
            hipLaunchKernelGGL(hal_cast_kernel<bf16_t>, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, s, phi, (bf16_t*)x, n8);
        else
            hipLaunchKernelGGL(hal_cast_kernel<bsplit_t>, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, s, phi, (bsplit_t*)x, n8);
        HMMR_CHECK_HIP(hipGetLastError());
        xin = x;
    }
    auto fc = [&](const void* in, const hmmr_layer_t& l, void* o, int odt, int relu, const float* res) {
        hmmr_conv_desc_t d = {};
        d.in = in; d.w = l.w; d.scale = l.scale; d.shift = l.shift; d.out = o;
        d.in_dtype = w->dtype; d.out_dtype = odt;
        d.n_img = m; d.hin = d.win = 1; d.cin = C;
        d.in_img_stride = C; d.in_row_stride = C; d.in_px_stride = C;
        d.kh = d.kw = 1; d.sy = d.sx = 1; d.ho = d.wo = 1; d.cout = C; d.ldo = C;
        d.relu = relu; d.res = res; d.ldr = C;
        d.split_k = 4; d.ws = sk; d.ws_bytes = skb;            // (the hallucinator keeps 4 slices in every mode)
        return hmmr_conv_gemm(&d, s);
    };
    if (fc(xin, w->fc1, h1, w->dtype, 1, nullptr)) return -2;
    if (fc(h1, w->fc2, h2, w->dtype, 1, nullptr)) return -2;
    if (fc(h2, w->fc3, out, HMMR_F32, 0, phi)) return -2;      // + phi (models.py:295)
    return 0;
}
