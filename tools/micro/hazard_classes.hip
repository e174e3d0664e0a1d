// Micro-test: which VALU instruction classes are disturbed when a kernel on ANOTHER stream issues 16-bit 32x32x16 MFMAs?
// Each victim applies one instruction (inline asm, so the compiler cannot substitute it) to known inputs many times and
// compares with the same arithmetic done by scalar-FP32 instructions (which the first micro-test showed to be safe).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float gen(unsigned i, unsigned salt) {
    unsigned h = (i * 2654435761u) ^ (salt * 40503u); h ^= h >> 13; h *= 0x5bd1e995u; h ^= h >> 15;
    return (float)(h & 0xffffu) * (1.0f / 65536.0f) + 0.25f;
}

// CLS 0: v_pk_fma_f32   1: v_pk_mul_f32   2: v_pk_add_f32   3: v_cvt_pk_f16_f32 (pkrtz)   4: v_fma_f32 (control)   5: v_pk_fma_f16
// 6: v_pk_fma_f32 op_sel_hi:[1,0,1] (operand broadcast, as hipcc emits for stencils)   7: v_pk_fma_f32 on operands that just arrived from LDS
template <int CLS>
__global__ __launch_bounds__(256, 4) void victim(unsigned* errs, int reps) {
    const unsigned tid = blockIdx.x * 256 + threadIdx.x;
    __shared__ float4 stage[2][256];
    unsigned bad = 0;
    for (int r = 0; r < reps; ++r) {
        const float a0 = gen(tid, 3 * r), a1 = gen(tid, 3 * r + 1), b0 = gen(tid + 7, 3 * r), b1 = gen(tid + 7, 3 * r + 1);
        const float c0 = gen(tid + 13, 3 * r + 2), c1 = gen(tid + 29, 3 * r + 2);
        f32x2 a = {a0, a1}, b = {b0, b1}, c = {c0, c1}, d = {0, 0};
        float e0, e1;
        if (CLS == 0) {
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
            e0 = __builtin_fmaf(a0, b0, c0); e1 = __builtin_fmaf(a1, b1, c1);
        } else if (CLS == 1) {
            asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
            e0 = a0 * b0; e1 = a1 * b1;
        } else if (CLS == 2) {
            asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
            e0 = a0 + b0; e1 = a1 + b1;
        } else if (CLS == 3) {
            unsigned pk;
            asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(pk) : "v"(a0), "v"(a1));
            const f16x2 h = __builtin_bit_cast(f16x2, pk);
            d[0] = (float)h[0]; d[1] = (float)h[1];
            // round-toward-zero reference through integer masking of the fp32 mantissa (inputs are normal, in [0.25, 1.25))
            e0 = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, a0) & 0xffffe000u);
            e1 = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, a1) & 0xffffe000u);
        } else if (CLS == 4) {
            float t0, t1;
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(t0) : "v"(a0), "v"(b0), "v"(c0));
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(t1) : "v"(a1), "v"(b1), "v"(c1));
            d[0] = t0; d[1] = t1;
            e0 = __builtin_fmaf(a0, b0, c0); e1 = __builtin_fmaf(a1, b1, c1);
        } else if (CLS == 6) {
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(d) : "v"(a), "v"(b), "v"(c));
            e0 = __builtin_fmaf(a0, b0, c0); e1 = __builtin_fmaf(a1, b0, c1);
        } else if (CLS == 7) {
            stage[0][threadIdx.x] = make_float4(a0, a1, b0, b1);
            stage[1][threadIdx.x] = make_float4(c0, c1, 0.0f, 0.0f);
            __syncthreads();
            const unsigned o = (threadIdx.x ^ 37) & 255;   // somebody else's slot: a real LDS round trip
            f32x2 la, lb, lc;
            {
                const float4 u = stage[0][o], w = stage[1][o];
                la = f32x2{u.x, u.y}; lb = f32x2{u.z, u.w}; lc = f32x2{w.x, w.y};
            }
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(la), "v"(lb), "v"(lc));
            const unsigned ot = blockIdx.x * 256 + o;
            e0 = __builtin_fmaf(gen(ot, 3 * r), gen(ot + 7, 3 * r), gen(ot + 13, 3 * r + 2));
            e1 = __builtin_fmaf(gen(ot, 3 * r + 1), gen(ot + 7, 3 * r + 1), gen(ot + 29, 3 * r + 2));
            __syncthreads();
        } else {
            f16x2 ha = {(_Float16)a0, (_Float16)a1}, hb = {(_Float16)b0, (_Float16)b1}, hc = {(_Float16)c0, (_Float16)c1}, hd;
            asm volatile("v_pk_fma_f16 %0, %1, %2, %3" : "=v"(hd) : "v"(ha), "v"(hb), "v"(hc));
            d[0] = (float)hd[0]; d[1] = (float)hd[1];
            e0 = (float)(_Float16)__builtin_fmaf((float)ha[0], (float)hb[0], (float)hc[0]);
            e1 = (float)(_Float16)__builtin_fmaf((float)ha[1], (float)hb[1], (float)hc[1]);
        }
        bad += (d[0] != e0) + (d[1] != e1);
    }
    if (bad) atomicAdd(errs, bad);
}

template <int F16>
__global__ __launch_bounds__(256, 2) void aggressor(float* sink, int iters) {
    __shared__ f16x8 big[4100];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 4100; i += 256) { f16x8 v; for (int e = 0; e < 8; ++e) v[e] = (_Float16)(0.001f * ((i + e) & 31)); big[i] = v; }
    __syncthreads();
    f32x16 c0 = {}, c1 = {};
    for (int it = 0; it < iters; ++it) {
        const int base = ((it * 192) + (tid >> 6) * 64) % 3900;
        const f16x8 a = big[base + lane], b = big[base + 64 + lane];
        if (F16) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, c1, 0, 0, 0);
        } else {
            typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, b), __builtin_bit_cast(bf16x8, a), c1, 0, 0, 0);
        }
    }
    float s = 0; for (int e = 0; e < 16; ++e) s += c0[e] + c1[e];
    if (s == 123.456f) sink[tid] = s;
}

template <int CLS>
void run(const char* name, unsigned* errs, float* sink, hipStream_t a, hipStream_t b, int rounds) {
    unsigned h[2];
    for (int with = 0; with < 2; ++with) {
        hipMemsetAsync(errs, 0, 4, a);
        hipDeviceSynchronize();
        for (int r = 0; r < rounds; ++r) {
            victim<CLS><<<4096, 256, 0, a>>>(errs, 400);
            if (with) aggressor<1><<<1800, 256, 0, b>>>(sink, 300);
        }
        hipDeviceSynchronize();
        hipMemcpy(&h[with], errs, 4, hipMemcpyDeviceToHost);
    }
    printf("%-22s mismatches alone %u, next to fp16 32x32x16 MFMA on another stream %u  (of %.2e checks)\n", name, h[0], h[1],
           (double)rounds * 4096 * 256 * 400 * 2);
}

int main(int argc, char** argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 60;
    unsigned* errs; float* sink;
    hipMalloc(&errs, 4); hipMalloc(&sink, 4096);
    hipStream_t a, b;
    hipStreamCreateWithFlags(&a, hipStreamNonBlocking);
    hipStreamCreateWithFlags(&b, hipStreamNonBlocking);
    run<0>("v_pk_fma_f32", errs, sink, a, b, rounds);
    run<1>("v_pk_mul_f32", errs, sink, a, b, rounds);
    run<2>("v_pk_add_f32", errs, sink, a, b, rounds);
    run<3>("v_cvt_pkrtz_f16_f32", errs, sink, a, b, rounds);
    run<5>("v_pk_fma_f16", errs, sink, a, b, rounds);
    run<6>("v_pk_fma_f32 op_sel_hi", errs, sink, a, b, rounds);
    run<7>("v_pk_fma_f32 after LDS", errs, sink, a, b, rounds);
    run<4>("v_fma_f32 (control)", errs, sink, a, b, rounds);
    return 0;
}
