// Hand-off of the packed per-frame records to a rasteriser.
//
// The reference renders with pytorch-NMR through src/util/render/nmr_renderer.py: per frame it
//   (1) moves the weak-perspective camera [s, tx, ty] and the 2-D keypoints from the 224x224 crop to
//       the (squared, possibly down-scaled) original image (visualize_img_orig, nmr_renderer.py:368-401),
//   (2) projects the vertices with that camera, keeps z and flips y
//       (VisRenderer.__call__, nmr_renderer.py:139-144 -> torch_utils.py:11-29),
// all on the host in NumPy, one frame at a time, after the whole prediction dict crossed PCIe.
// Here both steps run in ONE launch that reads cams / verts / kps in place inside the packed
// records (row strides) and writes exactly what nr.Renderer.render(proj_verts, faces, texture)
// consumes.  HBM-bound: 12 B read + 12 B written per vertex.
//
// Camera algebra in fp64 (the reference's NumPy promotes to float64), new_cam rounded to fp32 as
// the reference does (.astype(np.float32)); the vertex projection is fp32 mul(add) with NO fma
// contraction, so proj_verts is bit-identical to torch's `scale * (X + trans)`.
#include "common.h"
#include "hmmr_hip.h"

namespace {

struct FrameCam { float s, tx, ty; };

// geom row = {undo_scale, start_x, start_y, proc_size, img_size}; geom == NULL: stay in the crop
__device__ __forceinline__ FrameCam frame_camera(const float* cam, const float* g) {
    FrameCam c = {cam[0], cam[1], cam[2]};
    if (!g) return c;
    const double undo = g[0], sx = g[1], sy = g[2], proc = g[3], size = g[4];
    const double crop_s = proc * (double)cam[0] * 0.5;                 // camera in crop pixels
    const double half = (2.0 / (double)cam[0]) * 0.5;
    const double crop_tx = (double)cam[1] + half, crop_ty = (double)cam[2] + half;
    const double orig_s = crop_s * undo;                               // camera in original pixels
    const double orig_tx = crop_tx + (sx - proc) / crop_s, orig_ty = crop_ty + (sy - proc) / crop_s;
    const double k = 2.0 / size;                                       // normalised original image
    c.s = (float)(orig_s * k);
    c.tx = (float)(orig_tx - 1.0 / (k * orig_s));
    c.ty = (float)(orig_ty - 1.0 / (k * orig_s));
    return c;
}

__global__ void render_handoff_kernel(const float* __restrict__ cams, long long ld_cam,
                                      const float* __restrict__ verts, long long ld_verts,
                                      const float* __restrict__ kps, long long ld_kps,
                                      const float* __restrict__ geom, int nv, int nk,
                                      float* __restrict__ new_cam, float* __restrict__ proj,
                                      float* __restrict__ kp_out, int f0) {
    const int f = f0 + blockIdx.y;
    const float* g = geom ? geom + (long long)f * 5 : nullptr;
    const FrameCam c = frame_camera(cams + (long long)f * ld_cam, g);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0 && new_cam) { new_cam[f * 3 + 0] = c.s; new_cam[f * 3 + 1] = c.tx; new_cam[f * 3 + 2] = c.ty; }
    if (i < nv) {
        const float* v = verts + (long long)f * ld_verts + 3 * i;
        float* o = proj + ((long long)f * nv + i) * 3;
        o[0] = __fmul_rn(c.s, __fadd_rn(v[0], c.tx));
        o[1] = -__fmul_rn(c.s, __fadd_rn(v[1], c.ty));                 // image y points down
        o[2] = v[2];                                                   // offset_z = 0
    }
    if (kp_out && kps && i < nk * 2) {
        const float p = kps[(long long)f * ld_kps + i];
        float r = p;
        if (g) {
            const double start = (i & 1) ? g[2] : g[1];
            const double px = (((double)p + 1.0) * 0.5) * (double)g[3];            // crop pixels
            const double po = (px + start - (double)g[3]) * (double)g[0];          // original pixels
            r = (float)(2.0 * (po / (double)g[4]) - 1.0);
        }
        kp_out[(long long)f * nk * 2 + i] = r;
    }
}
}  // namespace

extern "C" int hmmr_render_handoff(const float* cams, int64_t ld_cam, const float* verts, int64_t ld_verts,
                                   const float* kps, int64_t ld_kps, const float* geom, int n, int nv, int nk,
                                   float* new_cam, float* proj_verts, float* kp_orig, void* stream) {
    HMMR_REQUIRE(cams && verts && proj_verts && n > 0 && nv > 0 && nk >= 0, "hmmr_render_handoff: bad arguments");
    HMMR_REQUIRE(ld_cam >= 3 && ld_verts >= 3LL * nv && (!kps || ld_kps >= 2LL * nk),
                 "hmmr_render_handoff: row strides smaller than the rows");
    HMMR_REQUIRE(!kp_orig || kps, "hmmr_render_handoff: kp_orig requested without kps");
    const int work = nv > 2 * nk ? nv : 2 * nk;
    for (int f0 = 0; f0 < n; f0 += 32768) {            // grid.y is limited to 65535: long videos go in slabs of frames
        const int nf = n - f0 < 32768 ? n - f0 : 32768;
        hipLaunchKernelGGL(render_handoff_kernel, dim3((unsigned)((work + 255) / 256), (unsigned)nf), dim3(256), 0,
                           (hipStream_t)stream, cams, (long long)ld_cam, verts, (long long)ld_verts, kps,
                           (long long)ld_kps, geom, nv, nk, new_cam, proj_verts, kp_orig, f0);
        HMMR_CHECK_HIP(hipGetLastError());
    }
    return 0;
}
