/* A plain-C host of libcrfp_hip.so: no Python, no torch -- what a C / cgo / JNI caller of the drop-in boundary looks like.
 *
 *   gcc -std=c99 -I include -I /opt/rocm/include -D__HIP_PLATFORM_AMD__ examples/c_host_smoke.c \
 *       -L crfp_amd -lcrfp_hip -L /opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/crfp_amd -Wl,-rpath,/opt/rocm/lib -lm -o /tmp/c_host_smoke
 *
 * 1. crfp_flow_warp_f32 (model/CRFP.py:90-130) with a constant flow of (+1, 0): out[y][x] = x[y][x + 1] (to 1e-5), zero in the last column.
 * 2. crfp_upsample_bilinear_f32 x2 of a constant plane stays constant.
 * 3. the argument-error path (null tensors) returns CRFP_E_BADARG and a message.
 * Exit code 0 and "c_host_smoke: OK" when all three hold. */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "crfp_hip.h"

#define CHECK_HIP(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { fprintf(stderr, "HIP error %d at line %d\n", (int)r_, __LINE__); return 2; } } while (0)

int main(void) {
    const int n = 1, c = 8, h = 20, w = 36;
    const size_t nx = (size_t)n * c * h * w, nf = (size_t)n * h * w * 2;
    float* hx = (float*)malloc(nx * sizeof(float));
    float* hf = (float*)malloc(nf * sizeof(float));
    float* ho = (float*)malloc(nx * sizeof(float));
    size_t i;
    int y, x, ch, bad = 0;
    float *dx, *df, *dout;
    void* dws;
    size_t wsb;
    int rc;
    hipStream_t stream;

    printf("libcrfp_hip version %d\n", crfp_version());
    for (i = 0; i < nx; ++i) hx[i] = (float)((i * 2654435761u) % 1000) / 1000.0f;
    for (i = 0; i < nf; i += 2) { hf[i] = 1.0f; hf[i + 1] = 0.0f; }   /* (dx, dy) = (+1, 0) */
    CHECK_HIP(hipStreamCreate(&stream));
    CHECK_HIP(hipMalloc((void**)&dx, nx * sizeof(float)));
    CHECK_HIP(hipMalloc((void**)&df, nf * sizeof(float)));
    CHECK_HIP(hipMalloc((void**)&dout, nx * sizeof(float)));
    wsb = crfp_flow_warp_workspace_bytes(n, c, h, w);
    CHECK_HIP(hipMalloc(&dws, wsb ? wsb : 16));
    CHECK_HIP(hipMemcpyAsync(dx, hx, nx * sizeof(float), hipMemcpyHostToDevice, stream));
    CHECK_HIP(hipMemcpyAsync(df, hf, nf * sizeof(float), hipMemcpyHostToDevice, stream));
    rc = crfp_flow_warp_f32(dx, df, dout, n, c, h, w, 0, dws, wsb, stream);
    if (rc) { fprintf(stderr, "crfp_flow_warp_f32 failed (%d): %s\n", rc, crfp_last_error_string()); return 1; }
    CHECK_HIP(hipMemcpyAsync(ho, dout, nx * sizeof(float), hipMemcpyDeviceToHost, stream));
    CHECK_HIP(hipStreamSynchronize(stream));
    for (ch = 0; ch < c; ++ch)
        for (y = 0; y < h; ++y)
            for (x = 0; x < w; ++x) {
                const float want = x + 1 < w ? hx[((size_t)ch * h + y) * w + x + 1] : 0.0f;
                if (fabsf(ho[((size_t)ch * h + y) * w + x] - want) > 1e-5f) ++bad;   /* the position goes through the [-1, 1] normalisation in fp32 */
            }
    printf("flow_warp with flow (+1, 0): %d mismatches of %zu\n", bad, nx);

    {   /* x2 bilinear of a constant plane */
        const int C = 3, H = 9, W = 14;
        const size_t ni = (size_t)C * H * W, no = ni * 4;
        float *di, *dou;
        float* hi = (float*)malloc(ni * sizeof(float));
        float* hou = (float*)malloc(no * sizeof(float));
        for (i = 0; i < ni; ++i) hi[i] = 0.625f;
        CHECK_HIP(hipMalloc((void**)&di, ni * sizeof(float)));
        CHECK_HIP(hipMalloc((void**)&dou, no * sizeof(float)));
        CHECK_HIP(hipMemcpyAsync(di, hi, ni * sizeof(float), hipMemcpyHostToDevice, stream));
        rc = crfp_upsample_bilinear_f32(di, dou, 1, C, H, W, 2 * H, 2 * W, 0.5f, 0.5f, 1.0f, stream);
        if (rc) { fprintf(stderr, "crfp_upsample_bilinear_f32 failed (%d): %s\n", rc, crfp_last_error_string()); return 1; }
        CHECK_HIP(hipMemcpyAsync(hou, dou, no * sizeof(float), hipMemcpyDeviceToHost, stream));
        CHECK_HIP(hipStreamSynchronize(stream));
        {
            int bad2 = 0;
            for (i = 0; i < no; ++i) if (fabsf(hou[i] - 0.625f) > 1e-6f) ++bad2;
            printf("bilinear x2 of a constant plane: %s\n", bad2 ? "MISMATCH" : "constant");
            bad += bad2;
        }
        hipFree(di); hipFree(dou); free(hi); free(hou);
    }

    rc = crfp_flow_warp_f32(NULL, NULL, NULL, n, c, h, w, 0, NULL, 0, stream);
    printf("null tensors -> rc %d (%s)\n", rc, crfp_last_error_string());
    if (rc != CRFP_E_BADARG) ++bad;

    crfp_shutdown();
    hipFree(dx); hipFree(df); hipFree(dout); hipFree(dws);
    hipStreamDestroy(stream);
    free(hx); free(hf); free(ho);
    if (bad) { printf("c_host_smoke: FAILED\n"); return 1; }
    printf("c_host_smoke: OK\n");
    return 0;
}
