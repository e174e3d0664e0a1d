// Micro-test: how many 256-thread workgroups with a given static LDS size are resident per CU on MI355X?
// N workgroups per CU each spin for a fixed time; elapsed / spin = number of rounds.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int BYTES, int VG>
__global__ __launch_bounds__(256) void hold(float* sink, long long spin) {
    __shared__ char buf[BYTES];
    buf[threadIdx.x] = (char)threadIdx.x;
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < spin) __builtin_amdgcn_s_sleep(16);
    if (buf[(threadIdx.x + 1) & 255] == 77 && spin < 0) sink[threadIdx.x] = 1.0f;
}
template <int BYTES>
void run(float* sink, int per_cu) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const long long spin = 200000;  // ~100 us at 2 GHz
    hold<BYTES, 0><<<256, 256>>>(sink, 1000);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hold<BYTES, 0><<<256 * per_cu, 256>>>(sink, spin);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("LDS %6d B, %d WGs per CU launched: %.1f us  (one round ~ %.0f us)\n", BYTES, per_cu, ms * 1e3, spin / 2.1e3);
}
int main() {
    float* sink; hipMalloc(&sink, 4096);
    run<65664>(sink, 1); run<65664>(sink, 2); run<65664>(sink, 3);
    run<65536>(sink, 2); run<61440>(sink, 2); run<81920>(sink, 2); run<53000>(sink, 3); run<40000>(sink, 4); run<32768>(sink, 5);
    return 0;
}
