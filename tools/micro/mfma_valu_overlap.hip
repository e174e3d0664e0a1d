// Does VALU work overlap with the 16-bit 32x32x16 MFMA on gfx950 -- inside one wave (shadow) and between the two waves of a SIMD?
// One workgroup of 512 threads per CU (2 waves per SIMD), every wave runs ITER rounds of {NM MFMAs, 8*NM VALU FMAs} arranged per mode.
//   hipcc --offload-arch=gfx950 -O3 mfma_valu_overlap.hip -o mfma_valu_overlap && ./mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define V8(x)  asm volatile("v_fma_f32 %0, %0, %1, %1\n v_fma_f32 %0, %0, %1, %1\n v_fma_f32 %0, %0, %1, %1\n v_fma_f32 %0, %0, %1, %1\n" \
                            "v_fma_f32 %0, %0, %1, %1\n v_fma_f32 %0, %0, %1, %1\n v_fma_f32 %0, %0, %1, %1\n v_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(k));
#define V8I(x, y) asm volatile("v_fma_f32 %0, %0, %2, %2\n v_fma_f32 %1, %1, %2, %2\n v_fma_f32 %0, %0, %2, %2\n v_fma_f32 %1, %1, %2, %2\n" \
                            "v_fma_f32 %0, %0, %2, %2\n v_fma_f32 %1, %1, %2, %2\n v_fma_f32 %0, %0, %2, %2\n v_fma_f32 %1, %1, %2, %2" : "+v"(x), "+v"(y) : "v"(k));

template <int MODE>
__global__ __launch_bounds__(512) void kern(float* out, int iters, float k) {
    const int wave = threadIdx.x >> 6;
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f); b[i] = (_Float16)(i * 0.01f); }
    f32x16 c0 = {0}, c1 = {0};
    float x = threadIdx.x, y = threadIdx.x * 0.5f;
    const bool domf = MODE == 0 || MODE == 3 || MODE == 4 || (MODE == 2 && wave < 4);
    const bool dova = MODE == 1 || MODE == 3 || MODE == 4 || (MODE == 2 && wave >= 4);
    for (int it = 0; it < iters; ++it) {
        if (MODE == 3) {   // interleaved: 1 MFMA, 8 VALU
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                if (m & 1) c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);
                else c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
                V8I(x, y)
            }
        } else {
            if (domf) {
#pragma unroll
                for (int m = 0; m < 8; ++m) {
                    if (m & 1) c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);
                    else c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
                }
            }
            if (dova) {
#pragma unroll
                for (int m = 0; m < 8; ++m) V8I(x, y)
            }
        }
    }
    float s = x + y;
    for (int i = 0; i < 16; ++i) s += c0[i] + c1[i];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int MODE>
float run(float* out, int iters, int blocks) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    kern<MODE><<<blocks, 512>>>(out, 10, 0.5f);
    hipEventRecord(e0);
    kern<MODE><<<blocks, 512>>>(out, iters, 0.5f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    float* out; hipMalloc(&out, 256 * 512 * 4);
    const int iters = 20000, blocks = 256;
    // per wave and iteration: 8 MFMAs (8 x 32 clk of the MFMA pipe) and / or 64 VALU FMAs (64 x 4 clk of VALU issue)
    const char* names[5] = {"MFMA only (both waves of a SIMD)", "VALU only (both waves)", "one wave MFMA, the other VALU",
                            "every wave: 1 MFMA + 8 VALU interleaved", "every wave: 8 MFMA then 64 VALU (phased)"};
    float t[5] = {run<0>(out, iters, blocks), run<1>(out, iters, blocks), run<2>(out, iters, blocks), run<3>(out, iters, blocks), run<4>(out, iters, blocks)};
    for (int i = 0; i < 5; ++i) printf("mode %d  %-48s %8.3f ms   %6.1f clk per wave-iteration at 2.4 GHz\n", i, names[i], t[i], t[i] * 1e-3 * 2.4e9 / iters);
    return 0;
}
