// Micro-test: producer -> consumer kernel pairs on two independent HIP streams (own buffers each).
// The consumer checks every element the producer wrote (reading a shifted index so that most reads cross
// workgroups / XCDs) and checks a 2.5 KB by-value kernarg table read at a dynamic index.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

struct Desc { float4* base; long long stride; int it, a, b, c; };
struct Big { Desc d[80]; int it; int n; };

__global__ __launch_bounds__(256) void produce(Big k, int spin) {
    const Desc d = k.d[blockIdx.x % 80];
    float4* x = d.base;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= k.n) return;
    // scattered 16-B stores (stride 4 elements, like a pixel-shuffle store)
    const int j = (i & ~1023) + ((i & 255) << 2) + ((i >> 8) & 3);
    long long t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < spin) {}
    x[j] = make_float4((float)d.it, (float)j, (float)k.it, 1.0f);
}

__global__ __launch_bounds__(256) void consume(Big k, unsigned* errs, int shift) {
    const Desc d = k.d[(blockIdx.x * 7) % 80];
    const float4* x = d.base;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= k.n) return;
    const int j = (i + shift) % k.n;
    const float4 v = x[j];
    if (d.it != k.it) atomicAdd(errs + 1, 1u);
    if (v.x != (float)k.it || v.y != (float)j || v.z != (float)k.it) atomicAdd(errs, 1u);
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 300;
    const int n = 4 << 20;  // 64 MB per buffer
    hipStream_t st[2];
    float4* buf[2];
    unsigned* errs[2];
    for (int s = 0; s < 2; ++s) {
        hipStreamCreateWithFlags(&st[s], hipStreamNonBlocking);
        hipMalloc(&buf[s], (size_t)n * 16);
        hipMemset(buf[s], 0, (size_t)n * 16);
        hipMalloc(&errs[s], 8);
        hipMemset(errs[s], 0, 8);
    }
    hipDeviceSynchronize();
    for (int mode = 0; mode < 2; ++mode) {   // 0: both streams concurrently, 1: control, everything on stream 0
        for (int s = 0; s < 2; ++s) hipMemset(errs[s], 0, 8);
        hipDeviceSynchronize();
        for (int it = 1; it <= iters; ++it)
            for (int s = 0; s < 2; ++s) {
                Big k;
                for (int q = 0; q < 80; ++q) k.d[q] = Desc{buf[s], 0, it, q, s, 0};
                k.it = it; k.n = n;
                hipStream_t q = mode ? st[0] : st[s];
                produce<<<n / 256, 256, 0, q>>>(k, (it % 3) * 200);
                consume<<<n / 256, 256, 0, q>>>(k, errs[s], 4096 * 37 + 5);
            }
        hipDeviceSynchronize();
        for (int s = 0; s < 2; ++s) {
            unsigned h[2];
            hipMemcpy(h, errs[s], 8, hipMemcpyDeviceToHost);
            printf("%s stream %d: data errors %u, kernarg errors %u\n", mode ? "control" : "concurrent", s, h[0], h[1]);
        }
    }
    return 0;
}
