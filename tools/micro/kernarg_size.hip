// How large may a kernel's by-value arguments be on this stack?  (round 6: two ConvArgs in one launch = 6.4 KB)
#include <hip/hip_runtime.h>
#include <cstdio>
template <int N> struct Big { int v[N]; };
template <int N> __global__ void k(const Big<N> a, const Big<N> b, int* out) { out[threadIdx.x] = a.v[threadIdx.x % N] + b.v[N - 1]; }
template <int N> int run() {
    Big<N> a, b;
    for (int i = 0; i < N; ++i) { a.v[i] = i; b.v[i] = 2 * i; }
    int* d; hipMalloc(&d, 64 * sizeof(int));
    k<N><<<1, 64>>>(a, b, d);
    hipError_t e = hipDeviceSynchronize();
    int h[64]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("2 x %d bytes of kernel arguments: %s, out[5] = %d (expected %d)\n", (int)sizeof(a), hipGetErrorString(e == hipSuccess ? hipGetLastError() : e), h[5], 5 + 2 * (N - 1));
    hipFree(d);
    return 0;
}
int main() { run<256>(); run<800>(); run<1000>(); run<2000>(); return 0; }
