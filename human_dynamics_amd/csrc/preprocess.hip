// Crop that precedes the hot path, on the device:
// `process_image` (src/evaluation/run_video.py:56-107) = uint8 frame -> [-1,1] -> cv2.resize
// (bilinear, to floor(shape*scale)) -> edge pad 224 -> 224x224 crop around the scaled bbox centre.
// The scaled image is never materialised: every output pixel clamps its coordinates into the scaled
// image (that IS the edge padding) and evaluates the four bilinear taps of OpenCV's INTER_LINEAR
// directly on the uint8 frame (pixel-centre alignment in float32, float32 weights, fp64
// accumulation -- the arithmetic of cv2's double-image path).  The per-frame integers (scaled size,
// crop origin) come from the host, where the reference's float64 rounding is reproduced exactly.
// HBM-bound: 3 B read (cached, ~4 taps) and 12 B written per output element.
#include "common.h"
#include "hmmr_hip.h"

namespace {
constexpr int S = 224;

__device__ __forceinline__ void taps(int d, int src, int dst, int& s0, int& s1, double& w0, double& w1) {
    float f = (float)(((double)d + 0.5) * ((double)src / (double)dst) - 0.5);
    int s = (int)floorf(f);
    f -= (float)s;
    if (s < 0) { f = 0.f; s = 0; }
    if (s >= src - 1) { f = 0.f; s = src - 1; }
    s0 = s; s1 = min(s + 1, src - 1);
    w0 = (double)(1.0f - f); w1 = (double)f;
}

// geom[n] = {Hs, Ws, u0, v0}: scaled image size and the scaled-image coordinates of crop pixel (0,0)
__global__ void crop_frames_kernel(const unsigned char* __restrict__ frames, const int4* __restrict__ geom,
                                   int n, int H, int W, float* __restrict__ out) {
    // ((b / 255) - 0.5) * 2 in float64 (run_video.py:73) takes 256 values: one division per thread instead of twelve
    __shared__ double lut[256];
    lut[threadIdx.x] = ((double)threadIdx.x / 255.0 - 0.5) * 2.0;
    __syncthreads();
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)n * S * S) return;
    const int x = (int)(i % S), y = (int)((i / S) % S), f = (int)(i / (S * S));
    const int4 g = geom[f];
    const int u = min(max(g.z + x, 0), g.y - 1);          // clamp = np.pad(mode='edge')
    const int v = min(max(g.w + y, 0), g.x - 1);
    int x0, x1, y0, y1; double a0, a1, b0, b1;
    taps(u, W, g.y, x0, x1, a0, a1);
    taps(v, H, g.x, y0, y1, b0, b1);
    const unsigned char* fr = frames + (long long)f * H * W * 3;
    float* o = out + i * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        auto px = [&](int yy, int xx) { return lut[fr[(yy * W + xx) * 3 + c]]; };
        const double r0 = px(y0, x0) * a0 + px(y0, x1) * a1;
        const double r1 = px(y1, x0) * a0 + px(y1, x1) * a1;
        o[c] = (float)(r0 * b0 + r1 * b1);
    }
}
}  // namespace

extern "C" int hmmr_crop_frames(const unsigned char* frames, const int32_t* geom, int n, int h, int w,
                                float* out, void* stream) {
    HMMR_REQUIRE(frames && geom && out && n > 0 && h > 0 && w > 0, "hmmr_crop_frames: bad arguments");
    const long long tot = (long long)n * S * S;
    hipLaunchKernelGGL(crop_frames_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       frames, (const int4*)geom, n, h, w, out);
    HMMR_CHECK_HIP(hipGetLastError());
    return 0;
}
