// Micro-test: do workgroups of two kernels running concurrently on different HIP streams keep private LDS?
// Each workgroup fills its static LDS with a workgroup-unique pattern, idles, then verifies it.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int WORDS, int TAG>
__global__ __launch_bounds__(256, 2) void lds_hold(unsigned* errs, int spin) {
    __shared__ unsigned buf[WORDS];
    const unsigned key = (TAG << 28) ^ (blockIdx.x * 2654435761u);
    for (int i = threadIdx.x; i < WORDS; i += 256) buf[i] = key + i;
    __syncthreads();
    long long t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < spin) __builtin_amdgcn_s_sleep(8);
    __syncthreads();
    unsigned bad = 0;
    for (int i = threadIdx.x; i < WORDS; i += 256) bad += buf[i] != key + i;
    if (bad) atomicAdd(errs + TAG, bad);
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 200;
    unsigned* errs;
    hipMalloc(&errs, 16);
    hipMemset(errs, 0, 16);
    hipStream_t a, b;
    hipStreamCreateWithFlags(&a, hipStreamNonBlocking);
    hipStreamCreateWithFlags(&b, hipStreamNonBlocking);
    unsigned h[4];
    for (int it = 0; it < iters; ++it) {
        lds_hold<16800, 1><<<1024, 256, 0, a>>>(errs, 4000);   // 65.6 KB (> 64 KB)
        lds_hold<14600, 2><<<1024, 256, 0, b>>>(errs, 3000);   // 57 KB
    }
    hipDeviceSynchronize();
    hipMemcpy(h, errs, 16, hipMemcpyDeviceToHost);
    printf("lds_overlap 65.6K+57K: big-kernel bad words %u, small-kernel bad words %u\n", h[1], h[2]);
    hipMemset(errs, 0, 16);
    for (int it = 0; it < iters; ++it) {
        lds_hold<16800, 1><<<1024, 256, 0, a>>>(errs, 4000);   // 65.6 KB (> 64 KB)
        lds_hold<5040, 2><<<4096, 256, 0, b>>>(errs, 1500);    // 19.7 KB
    }
    hipDeviceSynchronize();
    hipMemcpy(h, errs, 16, hipMemcpyDeviceToHost);
    printf("lds_overlap 65.6K+19.7K: big-kernel bad words %u, small-kernel bad words %u\n", h[1], h[2]);
    hipMemset(errs, 0, 16);
    for (int it = 0; it < iters; ++it) {
        lds_hold<16000, 1><<<1024, 256, 0, a>>>(errs, 4000);   // 62.5 KB (< 64 KB)
        lds_hold<5040, 2><<<4096, 256, 0, b>>>(errs, 1500);    // 19.7 KB
    }
    hipDeviceSynchronize();
    hipMemcpy(h, errs, 16, hipMemcpyDeviceToHost);
    printf("lds_overlap 62.5K+19.7K: big-kernel bad words %u, small-kernel bad words %u\n", h[1], h[2]);
    hipMemset(errs, 0, 16);
    // control: same two kernels serialised on one stream
    hipMemset(errs, 0, 16);
    for (int it = 0; it < iters; ++it) {
        lds_hold<16800, 1><<<1024, 256, 0, a>>>(errs, 4000);
        lds_hold<14600, 2><<<1024, 256, 0, a>>>(errs, 3000);
    }
    hipDeviceSynchronize();
    hipMemcpy(h, errs, 16, hipMemcpyDeviceToHost);
    printf("control (one stream): big %u small %u\n", h[1], h[2]);
    return 0;
}
