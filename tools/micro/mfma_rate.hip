// Micro-test: calibrate s_memtime and the issue rate of v_mfma_f32_32x32x16_bf16 (1 or 2 waves per SIMD,
// 1 or 2 independent accumulator chains), with and without LDS operand reads.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int CHAINS, int LDSREAD>
__global__ __launch_bounds__(512) void rate(long long* out, float* sink, int iters) {
    __shared__ bf16x8 lds[4096];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 4096; i += blockDim.x) { bf16x8 v; for (int e = 0; e < 8; ++e) v[e] = (__bf16)0.01f; lds[i] = v; }
    __syncthreads();
    f32x16 c0 = {}, c1 = {};
    bf16x8 a = lds[lane], b = lds[64 + lane];
    const long long t0 = __builtin_amdgcn_s_memtime();
    const long long r0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (LDSREAD) { a = lds[(it * 64 + lane) & 4095]; b = lds[(it * 64 + 2048 + lane) & 4095]; }
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
        if (CHAINS == 2) c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, c1, 0, 0, 0);
        else c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, c0, 0, 0, 0);
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    const long long r1 = __builtin_readcyclecounter();
    float s = 0; for (int e = 0; e < 16; ++e) s += c0[e] + c1[e];
    if (s == 1.2345f) sink[tid] = s;
    if (tid == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[1] = r1 - r0; }
}

template <int CHAINS, int LDSREAD>
void run(const char* name, int threads, int blocks, long long* out, float* sink) {
    const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    rate<CHAINS, LDSREAD><<<blocks, threads>>>(out, sink, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    rate<CHAINS, LDSREAD><<<blocks, threads>>>(out, sink, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h[2]; hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
    const double mfma_per_wave = 2.0 * iters;
    printf("%-44s %7.3f ms  s_memtime %9lld (%.1f MHz)  cyclecounter %9lld (%.1f MHz)  -> %.1f ns per MFMA per wave, %.2f TFLOP/s/CU\n", name, ms,
           h[0], h[0] / (ms * 1e3), h[1], h[1] / (ms * 1e3), ms * 1e6 / mfma_per_wave,
           (threads / 64) * mfma_per_wave * 32768.0 / (ms * 1e-3) / 1e12);
}

int main() {
    long long* out; float* sink;
    hipMalloc(&out, 16); hipMalloc(&sink, 4096);
    run<2, 0>("1 wave/SIMD, 2 chains, regs   (256 blocks)", 256, 256, out, sink);
    run<1, 0>("1 wave/SIMD, 1 chain,  regs   (256 blocks)", 256, 256, out, sink);
    run<2, 0>("2 waves/SIMD, 2 chains, regs  (256 blocks)", 512, 256, out, sink);
    run<2, 1>("1 wave/SIMD, 2 chains, LDS    (256 blocks)", 256, 256, out, sink);
    run<2, 1>("2 waves/SIMD, 2 chains, LDS   (256 blocks)", 512, 256, out, sink);
    run<2, 0>("1 wave/SIMD, 2 chains, regs   (1 block)", 256, 1, out, sink);
    return 0;
}
