/* The Python-free path of the boundary (`hmmr_resnet50_fwd(imgs, weights blob, ...)`), as a C program: read checkpoint
 * variables (a flat dump: count, then per variable name length, name, numel, fp32 data), fill the weight struct with hmmr_pack_resnet, copy
 * the blob to the device with one hipMemcpy, run hmmr_resnet50_fwd on frames read from a file, write phi.  tests/test_gpu_c_abi.py compiles it
 * with hipcc, runs it and compares phi with the Python mirror's, bit for bit.
 *   pack_and_run <vars.bin> <frames.bin> <n_frames> <dtype> <phi_out.bin> */
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "hmmr_hip.h"

#define CHECK(x) do { if ((x) != hipSuccess) { fprintf(stderr, "HIP error at %s:%d\n", __FILE__, __LINE__); return 2; } } while (0)

int main(int argc, char** argv) {
    if (argc != 6) { fprintf(stderr, "usage: pack_and_run vars.bin frames.bin n dtype phi.bin\n"); return 1; }
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 1;
    int n_vars = 0;
    if (fread(&n_vars, 4, 1, f) != 1) return 1;
    hmmr_var_t* vars = (hmmr_var_t*)calloc((size_t)n_vars, sizeof(hmmr_var_t));
    for (int i = 0; i < n_vars; ++i) {
        int len = 0; long long numel = 0;
        if (fread(&len, 4, 1, f) != 1) return 1;
        char* name = (char*)calloc((size_t)len + 1, 1);
        if (fread(name, 1, (size_t)len, f) != (size_t)len || fread(&numel, 8, 1, f) != 1) return 1;
        float* data = (float*)malloc((size_t)numel * 4);
        if (fread(data, 4, (size_t)numel, f) != (size_t)numel) return 1;
        vars[i].name = name; vars[i].data = data; vars[i].numel = numel;
    }
    fclose(f);
    const int n = atoi(argv[3]), dtype = atoi(argv[4]);
    if (hmmr_abi_version() != HMMR_ABI_VERSION) { fprintf(stderr, "ABI mismatch\n"); return 1; }

    const size_t nb = hmmr_pack_resnet_bytes(vars, n_vars, dtype);
    if (!nb) { fprintf(stderr, "hmmr_pack_resnet_bytes: %s\n", hmmr_last_error()); return 1; }
    void *host = malloc(nb), *dev = NULL;
    CHECK(hipMalloc(&dev, nb));
    hmmr_resnet_weights_t w;
    if (hmmr_pack_resnet(vars, n_vars, dtype, host, nb, dev, &w)) { fprintf(stderr, "hmmr_pack_resnet: %s\n", hmmr_last_error()); return 1; }
    CHECK(hipMemcpy(dev, host, nb, hipMemcpyHostToDevice));
    free(host);

    const size_t fbytes = (size_t)n * 224 * 224 * 3 * 4;
    float* frames_h = (float*)malloc(fbytes);
    f = fopen(argv[2], "rb");
    if (!f || fread(frames_h, 1, fbytes, f) != fbytes) return 1;
    fclose(f);
    float *frames = NULL, *phi = NULL; void* ws = NULL;
    const size_t wsb = hmmr_resnet50_workspace_bytes(n, dtype);
    CHECK(hipMalloc((void**)&frames, fbytes)); CHECK(hipMalloc((void**)&phi, (size_t)n * 2048 * 4)); CHECK(hipMalloc(&ws, wsb));
    CHECK(hipMemcpy(frames, frames_h, fbytes, hipMemcpyHostToDevice));
    if (hmmr_resnet50_fwd(&w, frames, n, 0, phi, ws, wsb, NULL, NULL)) { fprintf(stderr, "hmmr_resnet50_fwd: %s\n", hmmr_last_error()); return 1; }
    CHECK(hipDeviceSynchronize());
    unsigned flags = 0;
    if (hmmr_run_flags(&flags, 1)) return 1;
    float* phi_h = (float*)malloc((size_t)n * 2048 * 4);
    CHECK(hipMemcpy(phi_h, phi, (size_t)n * 2048 * 4, hipMemcpyDeviceToHost));
    f = fopen(argv[5], "wb");
    fwrite(phi_h, 4, (size_t)n * 2048, f);
    fclose(f);
    printf("packed %zu bytes, %d frames, run flags %u\n", nb, n, flags);
    return 0;
}
