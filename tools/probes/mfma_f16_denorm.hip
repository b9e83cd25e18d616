// Does v_mfma_f32_32x32x16_f16 keep fp16 SUBNORMAL inputs (dev probe for the "fp16x3" split-operand idea)?
//   hipcc --offload-arch=gfx950 -O2 tools/probes/mfma_f16_denorm.hip -o /tmp/mfma_f16_denorm && /tmp/mfma_f16_denorm
// D = A x B with A[i][k] = a (one value everywhere), B[k][j] = 1: every D element = 16 * a.  For a below the smallest normal
// fp16 (6.1035e-5) a flush-to-zero matrix pipe returns 0.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__global__ void probe(const float* vals, float* out, int n) {
    for (int t = 0; t < n; ++t) {
        f16x8 a, b;
        for (int i = 0; i < 8; ++i) { a[i] = (_Float16)vals[t]; b[i] = (_Float16)1.0f; }
        f32x16 c;
        for (int i = 0; i < 16; ++i) c[i] = 0.f;
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
        if (threadIdx.x == 0) out[t] = c[0];
    }
}

int main() {
    const int n = 8;
    float h[n] = {1.0f, 1e-3f, 6.2e-5f, 6.0e-5f, 3.0e-5f, 1.0e-6f, 6.0e-8f, 5.96e-8f};
    float *dv, *dout, r[n];
    hipMalloc(&dv, sizeof(h)); hipMalloc(&dout, sizeof(h));
    hipMemcpy(dv, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dv, dout, n);
    hipMemcpy(r, dout, sizeof(r), hipMemcpyDeviceToHost);
    for (int t = 0; t < n; ++t) {
        const float q = (float)(_Float16)h[t];
        printf("a = %.4e (as fp16 %.6e%s): D = %.6e, expected 16 a = %.6e  %s\n", h[t], q, q != 0 && q < 6.1035e-5f ? ", SUBNORMAL" : "",
               r[t], 16 * q, r[t] == 16 * q ? "ok" : "DIFFERENT");
    }
    return 0;
}
