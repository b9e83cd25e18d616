// IEF regressors for gfx950: batch_pred_omega / call_hmr_ief / hmr_ief /
// encoder_fc3_dropout (src/models.py:80-116, 233-267, 299-415), inference mode
// (dropout = identity).  Default: use_optcam=True, use_delta_from_pred=True (tester.py:196-207); the other three
// combinations through hmmr_ief_weights_t.no_optcam / delta_from_start.
//
//   theta <- theta + fc3(relu(fc2(relu(fc1([phi, theta])))))      x num_stages
//
// fc1 acts on concat([phi, theta]) (models.py:402).  It is evaluated as
// phi.W1[:2048] + b1 (ONCE per regressor: phi does not change over the
// stages) plus theta.W1[2048:] per stage (K padded to 128).  The theta state
// and its small GEMM stay fp32 in every mode; the K=2048/1024 GEMMs take the
// struct's dtype.  All GEMMs go through the implicit-GEMM kernel.
#include "common.h"
#include "hmmr_hip.h"

static constexpr int LDT = 128;      // row stride of the zero-padded theta state
// The K = 2048 / 1024 GEMMs have at most m/64 x 16 output tiles (m = kept frames of a step, a few
// hundred): 4 K-slices fill the chip.  Fixed per layer so a row's result is batch-independent.
static constexpr int IEF_SPLIT_K = 4;

template <typename TO>
__global__ void cast_rows_kernel(const float* __restrict__ in, TO* __restrict__ out, long long n8) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n8) return;
    float v[8];
    load8(in + i * 8, v);
    store8(out + i * 8, v);
}

// theta0[m, :] = [src(row m or broadcast)[off : off+nd], 0...]; the second state buffer starts as zeros (its padding
// columns are operands of the K = 128 theta GEMM and are never written by fc3).  Grid y = problem of a grouped launch
// (the same start for every problem; its state buffers `th_stride` floats behind the previous problem's).
__global__ void ief_init_theta_kernel(const float* __restrict__ src, int ld_src, int off, int nd,
                                      float* __restrict__ theta, float* __restrict__ theta_other, int m, long long th_stride) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)m * LDT) return;
    const int row = (int)(i / LDT), col = (int)(i % LDT);
    const long long z = (long long)blockIdx.y * th_stride;
    theta[z + i] = col < nd ? src[(long long)row * ld_src + off + col] : 0.f;
    theta_other[z + i] = 0.f;
}

// omega_out[m, 85]: mode 0 (present regressor) -> theta[:, :85];
// mode 1 (delta, use_optcam)   -> [1, 0, 0, theta[:, :72], beta]     (models.py:367-371)
// mode 2 (delta, no optcam)    -> [theta[:, :75], beta]              (models.py:372-373)
// beta = columns 75..84 of the delta's starting omega: row `row` of `bsrc` (ld_b = 85) or one broadcast row (ld_b = 0).
// Grid y = problem of a grouped launch: its theta `th_stride` floats, its output m * 85 floats behind the previous one's.
__global__ void ief_finalize_kernel(const float* __restrict__ theta, const float* __restrict__ bsrc, int ld_b,
                                    int mode, float* __restrict__ out, int m, long long th_stride) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)m * 85) return;
    const int row = (int)(i / 85), col = (int)(i % 85);
    theta += (long long)blockIdx.y * th_stride;
    out += (long long)blockIdx.y * m * 85;
    float v;
    if (mode == 0) v = theta[(long long)row * LDT + col];
    else if (col >= 75) v = bsrc[(long long)row * ld_b + col];
    else if (mode == 2) v = theta[(long long)row * LDT + col];
    else if (col == 0) v = 1.f;
    else if (col < 3) v = 0.f;
    else v = theta[(long long)row * LDT + (col - 3)];
    out[i] = v;
}

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// One set of (pre, h1, h2, theta x 2, split-K planes) per problem of the grouped launches: the delta regressors run as ONE
// launch per layer (hmmr_conv_desc_t.batch), each on its own set, `*_s` bytes apart; regressor 0 runs alone on set 0.
struct IefBufs { size_t xin, pre, h1, h2, th[2], sk, skbytes, pre_s, th_s, total; };
static IefBufs ief_layout(int m, int dtype, int group) {
    const size_t e = dtype == HMMR_BF16 ? 2 : 4;
    const size_t g = group > 1 ? group : 1;
    IefBufs b; size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
    b.xin = take((size_t)m * 2048 * e);
    b.pre_s = align_up((size_t)m * 1024 * e, 256);
    b.th_s = align_up((size_t)m * LDT * 4, 256);
    b.pre = take(g * b.pre_s);
    b.h1 = take(g * b.pre_s);
    b.h2 = take(g * b.pre_s);
    b.th[0] = take(g * b.th_s);
    b.th[1] = take(g * b.th_s);
    b.skbytes = g * hmmr_conv_splitk_workspace_bytes(m, 1024, IEF_SPLIT_K);
    b.sk = take(b.skbytes);
    b.total = off;
    return b;
}

extern "C" size_t hmmr_ief_workspace_bytes(int m, int num_regressors, int dtype) {
    return m > 0 ? ief_layout(m, dtype, num_regressors - 1).total : 0;
}

static hmmr_conv_desc_t fc_desc(const void* in, int in_dtype, int m, int k, const hmmr_layer_t& l,
                                void* out, int out_dtype, int cout, int ldo, void* sk = nullptr,
                                size_t skbytes = 0) {
    hmmr_conv_desc_t d = {};
    d.in = in; d.w = l.w; d.scale = l.scale; d.shift = l.shift; d.out = out;
    d.in_dtype = in_dtype; d.out_dtype = out_dtype;
    d.n_img = m; d.hin = d.win = 1; d.cin = k;
    d.in_img_stride = k; d.in_row_stride = k; d.in_px_stride = k;
    d.kh = d.kw = 1; d.sy = d.sx = 1; d.ho = d.wo = 1; d.cout = cout; d.ldo = ldo;
    if (sk) { d.split_k = IEF_SPLIT_K; d.ws = sk; d.ws_bytes = skbytes; }
    return d;
}

extern "C" int hmmr_ief_fwd(const hmmr_ief_weights_t* w, const float* strips, int m, float* omegas,
                            void* ws, size_t ws_bytes, void* stream) {
    return hmmr_ief_fwd_from(w, strips, nullptr, m, omegas, ws, ws_bytes, stream);
}

extern "C" int hmmr_ief_fwd_from(const hmmr_ief_weights_t* w, const float* strips, const float* omega_start, int m,
                                 float* omegas, void* ws, size_t ws_bytes, void* stream) {
    HMMR_REQUIRE(w && strips && omegas && ws, "hmmr_ief_fwd: null argument");
    HMMR_REQUIRE(m > 0, "hmmr_ief_fwd: m must be positive");
    HMMR_REQUIRE(w->num_regressors >= 1 && w->num_regressors <= HMMR_MAX_REGRESSORS, "hmmr_ief_fwd: bad num_regressors");
    HMMR_REQUIRE(w->reg[0].nd == 85, "hmmr_ief_fwd: regressor 0 must predict 85-D omega");
    HMMR_REQUIRE(ws_bytes >= hmmr_ief_workspace_bytes(m, w->num_regressors, w->dtype), "hmmr_ief_fwd: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    const IefBufs L = ief_layout(m, w->dtype, w->num_regressors - 1);
    char* base = (char*)ws;
    const void* xin = strips;
    if (w->dtype != HMMR_F32) {
        const long long n8 = (long long)m * 2048 / 8;
        if (w->dtype == HMMR_BF16)
            hipLaunchKernelGGL(cast_rows_kernel<bf16_t>, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, s, strips,
                               (bf16_t*)(base + L.xin), n8);
        else
            hipLaunchKernelGGL(cast_rows_kernel<bsplit_t>, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, s, strips,
                               (bsplit_t*)(base + L.xin), n8);
        HMMR_CHECK_HIP(hipGetLastError());
        xin = base + L.xin;
    }
    void* pre = base + L.pre; void* h1 = base + L.h1; void* h2 = base + L.h2;
    float* th[2] = {(float*)(base + L.th[0]), (float*)(base + L.th[1])};
    const unsigned gth = (unsigned)(((long long)m * LDT + 255) / 256);
    const unsigned gfin = (unsigned)(((long long)m * 85 + 255) / 256);
    const int nd_delta = w->no_optcam ? 75 : 72;               // models.py:333-336
    for (int r = 0; r < w->num_regressors; ++r)
        HMMR_REQUIRE(w->reg[r].nd == (r == 0 ? 85 : nd_delta), "hmmr_ief_fwd: regressor %d has nd=%d (expected %d)", r,
                     w->reg[r].nd, r == 0 ? 85 : nd_delta);
    // The delta regressors are independent of each other (each starts from the present prediction, models.py:343-361)
    // and of one shape: they run as ONE grouped launch per layer when their operands lie at one common stride (always
    // true for two of them; the packer allocates more of them evenly).  Same kernels on the same operands: same bits.
    const int nd_group = w->num_regressors - 1;
    struct Strides { long long w, scale, shift; };
    auto strides_of = [&](const hmmr_layer_t hmmr_ief_regressor_t::*lay, Strides& st) -> bool {
        st = Strides{0, 0, 0};
        for (int r = 2; r < w->num_regressors; ++r) {
            const hmmr_layer_t& a = w->reg[r - 1].*lay; const hmmr_layer_t& b = w->reg[r].*lay;
            const Strides cur = {(const char*)b.w - (const char*)a.w,
                                 a.scale ? (const char*)b.scale - (const char*)a.scale : 0,
                                 a.shift ? (const char*)b.shift - (const char*)a.shift : 0};
            if (!a.scale != !b.scale || !a.shift != !b.shift) return false;
            if (r > 2 && (cur.w != st.w || cur.scale != st.scale || cur.shift != st.shift)) return false;
            st = cur;
        }
        return ((st.w | st.scale | st.shift) & 15) == 0;
    };
    Strides s_phi, s_th, s_fc2, s_fc3;
    const bool grouped = nd_group >= 2 && !hmmr_debug_state()->ief_no_group &&
                         strides_of(&hmmr_ief_regressor_t::fc1_phi, s_phi) && strides_of(&hmmr_ief_regressor_t::fc1_theta, s_th) &&
                         strides_of(&hmmr_ief_regressor_t::fc2, s_fc2) && strides_of(&hmmr_ief_regressor_t::fc3, s_fc3);
    // the IEF's own starting point: omega_start rows, or the mean theta in every row (tester.py:181)
    const float* start = omega_start ? omega_start : w->mean_theta;
    const int ld_start = omega_start ? 85 : 0;
    // a delta regressor starts from the present prediction omega0 (use_delta_from_pred) or from `start`
    // (models.py:349), trimmed to [3:75] (use_optcam) or [:75] (models.py:353-357); beta = its last 10 columns
    const float* dsrc = w->delta_from_start ? start : (const float*)omegas;
    const int ld_d = w->delta_from_start ? ld_start : 85;
    // regressor r alone (nb = 1), or regressors r ... r + nb - 1 as grouped launches
    auto run = [&](int r, int nb, const Strides& z_phi, const Strides& z_th, const Strides& z_fc2, const Strides& z_fc3) -> int {
        const hmmr_ief_regressor_t& R = w->reg[r];
        const long long ths = (long long)(L.th_s / 4);
        auto group = [&](hmmr_conv_desc_t& d, const Strides& z, long long in_b, long long out_b, long long res_b) {
            if (nb < 2) return;
            d.batch = nb; d.batch_in_bytes = in_b; d.batch_w_bytes = z.w; d.batch_out_bytes = out_b; d.batch_res_bytes = res_b;
            d.batch_scale_bytes = z.scale; d.batch_shift_bytes = z.shift;
        };
        if (r == 0)
            hipLaunchKernelGGL(ief_init_theta_kernel, dim3(gth), dim3(256), 0, s, start, ld_start, 0, 85, th[0], th[1], m, ths);
        else
            hipLaunchKernelGGL(ief_init_theta_kernel, dim3(gth, nb), dim3(256), 0, s, dsrc, ld_d, w->no_optcam ? 0 : 3, nd_delta,
                               th[0], th[1], m, ths);
        HMMR_CHECK_HIP(hipGetLastError());
        // pre = phi . W1[:2048] + b1
        hmmr_conv_desc_t d = fc_desc(xin, w->dtype, m, 2048, R.fc1_phi, pre, w->dtype, 1024, 1024, base + L.sk, L.skbytes);
        group(d, z_phi, 0, (long long)L.pre_s, 0);
        if (hmmr_conv_gemm(&d, s)) return -2;
        int cur = 0;
        for (int st = 0; st < w->num_stages; ++st) {
            // h1 = relu(pre + theta . W1[2048:])
            d = fc_desc(th[cur], HMMR_F32, m, LDT, R.fc1_theta, h1, w->dtype, 1024, 1024);
            d.res = pre; d.ldr = 1024; d.relu = 1;
            group(d, z_th, (long long)L.th_s, (long long)L.pre_s, (long long)L.pre_s);
            if (hmmr_conv_gemm(&d, s)) return -2;
            // h2 = relu(h1 . W2 + b2)
            d = fc_desc(h1, w->dtype, m, 1024, R.fc2, h2, w->dtype, 1024, 1024, base + L.sk, L.skbytes);
            d.relu = 1;
            group(d, z_fc2, (long long)L.pre_s, (long long)L.pre_s, 0);
            if (hmmr_conv_gemm(&d, s)) return -2;
            // theta' = theta + h2 . W3 + b3
            d = fc_desc(h2, w->dtype, m, 1024, R.fc3, th[cur ^ 1], HMMR_F32, R.nd, LDT, base + L.sk, L.skbytes);
            d.res = th[cur]; d.ldr = LDT;
            group(d, z_fc3, (long long)L.pre_s, (long long)L.th_s, (long long)L.th_s);
            if (hmmr_conv_gemm(&d, s)) return -2;
            cur ^= 1;
        }
        hipLaunchKernelGGL(ief_finalize_kernel, dim3(gfin, nb), dim3(256), 0, s, (const float*)th[cur],
                           dsrc, ld_d, r == 0 ? 0 : (w->no_optcam ? 2 : 1), omegas + (size_t)r * m * 85, m, ths);
        HMMR_CHECK_HIP(hipGetLastError());
        return 0;
    };
    const Strides none = {0, 0, 0};
    if (run(0, 1, none, none, none, none)) return -2;
    if (grouped) return run(1, nd_group, s_phi, s_th, s_fc2, s_fc3);
    for (int r = 1; r < w->num_regressors; ++r)
        if (run(r, 1, none, none, none, none)) return -2;
    return 0;
}
