// Are the two codings of the exact fp16 two-term split (scalar: cvt / cvt back / sub / mul / cvt; packed: v_cvt_pk_f16_f32 + v_fma_mix_f32) the
// same function on gfx950?  Compares them bit for bit over random floats of every magnitude, and checks x0 + x1 / 2^11 == x.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off split_forms.hip -o split_forms && ./split_forms
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
typedef float f2_t __attribute__((ext_vector_type(2)));

__global__ void kern(const float* x, unsigned short* a0, unsigned short* a1, unsigned short* b0, unsigned short* b1, int n) {
    const int i = 2 * (blockIdx.x * blockDim.x + threadIdx.x);
    if (i + 1 >= n) return;
    const float v0 = x[i], v1 = x[i + 1];
    {   // scalar form
        const _Float16 h0 = (_Float16)v0, h1 = (_Float16)v1;
        const _Float16 l0 = (_Float16)((v0 - (float)h0) * 2048.0f), l1 = (_Float16)((v1 - (float)h1) * 2048.0f);
        a0[i] = __builtin_bit_cast(unsigned short, h0); a0[i + 1] = __builtin_bit_cast(unsigned short, h1);
        a1[i] = __builtin_bit_cast(unsigned short, l0); a1[i + 1] = __builtin_bit_cast(unsigned short, l1);
    }
    {   // packed form
        float negone = -1.0f;
        asm("" : "+v"(negone));
        const h2_t h2 = __builtin_convertvector(f2_t{v0, v1}, h2_t);
        const float r0 = __builtin_fmaf((float)h2[0], negone, v0), r1 = __builtin_fmaf((float)h2[1], negone, v1);
        const h2_t l2 = __builtin_convertvector(f2_t{r0 * 2048.0f, r1 * 2048.0f}, h2_t);
        // (whole-pair bit casts, as in the library: per-component casts of a converted pair are miscompiled by hipcc 7.2)
        *reinterpret_cast<unsigned*>(b0 + i) = __builtin_bit_cast(unsigned, h2);
        *reinterpret_cast<unsigned*>(b1 + i) = __builtin_bit_cast(unsigned, l2);
    }
}

static float h2f(unsigned short h) { _Float16 v; __builtin_memcpy(&v, &h, 2); return (float)v; }

int main() {
    const int n = 1 << 22;
    float* hx = (float*)malloc(n * 4);
    srand(1);
    for (int i = 0; i < n; ++i) {
        const float m = (float)rand() / RAND_MAX * 2.0f - 1.0f;
        const int e = rand() % 40 - 30;          // 2^-30 .. 2^9
        hx[i] = ldexpf(m, e);
    }
    float* dx; unsigned short *d[4], *h[4];
    (void)hipMalloc(&dx, n * 4); (void)hipMemcpy(dx, hx, n * 4, hipMemcpyHostToDevice);
    for (int k = 0; k < 4; ++k) { (void)hipMalloc(&d[k], n * 2); h[k] = (unsigned short*)malloc(n * 2); }
    kern<<<n / 512, 256>>>(dx, d[0], d[1], d[2], d[3], n);
    for (int k = 0; k < 4; ++k) (void)hipMemcpy(h[k], d[k], n * 2, hipMemcpyDeviceToHost);
    long diff_hi = 0, diff_lo = 0, bad_a = 0, bad_b = 0; int shown = 0;
    for (int i = 0; i < n; ++i) {
        if (h[0][i] != h[2][i]) ++diff_hi;
        if (h[1][i] != h[3][i]) {
            ++diff_lo;
            if (shown++ < 8) printf("x=%.9g scalar (%04x,%04x) packed (%04x,%04x)\n", hx[i], h[0][i], h[1][i], h[2][i], h[3][i]);
        }
        const double ra = (double)h2f(h[0][i]) + (double)h2f(h[1][i]) / 2048.0, rb = (double)h2f(h[2][i]) + (double)h2f(h[3][i]) / 2048.0;
        if (fabs(ra - hx[i]) > fabs(hx[i]) * 2.4e-7 + 3e-11) ++bad_a;
        if (fabs(rb - hx[i]) > fabs(hx[i]) * 2.4e-7 + 3e-11) ++bad_b;
    }
    printf("values %d: hi parts differ %ld, lo parts differ %ld; reconstruction off (rel 2^-22 + 3e-11): scalar %ld, packed %ld\n", n, diff_hi, diff_lo, bad_a, bad_b);
    return 0;
}
