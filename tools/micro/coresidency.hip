// Micro-test: an LDS + VALU stencil kernel (self-checking) on stream A while stream B runs an MFMA kernel.
// Reproduces (or not) the 16-lane glitches seen in conv3x3_narrow_kernel outputs when a split-bf16 conv runs
// concurrently on another stream.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float pat(unsigned r, unsigned c, unsigned comp, unsigned salt) {
    unsigned h = (r * 73856093u) ^ (c * 19349663u) ^ (comp * 83492791u) ^ (salt * 2654435761u);
    h ^= h >> 13; h *= 0x5bd1e995u; h ^= h >> 15;
    return (float)(h & 0xffffu) * (1.0f / 65536.0f);
}

__global__ __launch_bounds__(256, 4) void victim(unsigned* errs, unsigned* where, int reps) {
    __shared__ float4 tile[18][66];
    __shared__ float4 wl[36];
    const int tid = threadIdx.x, tx = tid & 63, ty = tid >> 6;
    if (tid < 36) wl[tid] = make_float4(pat(tid, 0, 0, 7) - 0.5f, pat(tid, 1, 0, 7) - 0.5f, pat(tid, 2, 0, 7) - 0.5f, pat(tid, 3, 0, 7) - 0.5f);
    for (int idx = tid; idx < 18 * 66; idx += 256) {
        const int r = idx / 66, c = idx - r * 66;
        (&tile[0][0])[idx] = make_float4(pat(r, c, 0, blockIdx.x), pat(r, c, 1, blockIdx.x), pat(r, c, 2, blockIdx.x), pat(r, c, 3, blockIdx.x));
    }
    __syncthreads();
    for (int rep = 0; rep < reps; ++rep) {
        float acc[4][4], exp_[4][4];
        for (int i = 0; i < 4; ++i) for (int o = 0; o < 4; ++o) { acc[i][o] = 0.0f; exp_[i][o] = 0.0f; }
#pragma unroll 1
        for (int ky = 0; ky < 3; ++ky) {
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const float4* wq = &wl[(ky * 3 + kx) * 4];
                const float4 w0 = wq[0], w1 = wq[1], w2 = wq[2], w3 = wq[3];
                float4 uu[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) uu[i] = tile[4 * ty + ky + i][tx + kx];
#ifdef FULLWAIT
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // every LDS return complete before the first packed FMA
#endif
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float4 u = uu[i];
                    acc[i][0] = fmaf(w3.x, u.w, fmaf(w2.x, u.z, fmaf(w1.x, u.y, fmaf(w0.x, u.x, acc[i][0]))));
                    acc[i][1] = fmaf(w3.y, u.w, fmaf(w2.y, u.z, fmaf(w1.y, u.y, fmaf(w0.y, u.x, acc[i][1]))));
                    acc[i][2] = fmaf(w3.z, u.w, fmaf(w2.z, u.z, fmaf(w1.z, u.y, fmaf(w0.z, u.x, acc[i][2]))));
                    acc[i][3] = fmaf(w3.w, u.w, fmaf(w2.w, u.z, fmaf(w1.w, u.y, fmaf(w0.w, u.x, acc[i][3]))));
                }
            }
        }
        // expected: same arithmetic, operands regenerated without LDS
#pragma unroll 1
        for (int ky = 0; ky < 3; ++ky) {
#pragma unroll 1
            for (int kx = 0; kx < 3; ++kx) {
                const int t = (ky * 3 + kx) * 4;
                float4 w[4];
                for (int q = 0; q < 4; ++q)
                    w[q] = make_float4(pat(t + q, 0, 0, 7) - 0.5f, pat(t + q, 1, 0, 7) - 0.5f, pat(t + q, 2, 0, 7) - 0.5f, pat(t + q, 3, 0, 7) - 0.5f);
                for (int i = 0; i < 4; ++i) {
                    const int r = 4 * ty + ky + i, c = tx + kx;
                    const float4 u = make_float4(pat(r, c, 0, blockIdx.x), pat(r, c, 1, blockIdx.x), pat(r, c, 2, blockIdx.x), pat(r, c, 3, blockIdx.x));
                    exp_[i][0] = fmaf(w[3].x, u.w, fmaf(w[2].x, u.z, fmaf(w[1].x, u.y, fmaf(w[0].x, u.x, exp_[i][0]))));
                    exp_[i][1] = fmaf(w[3].y, u.w, fmaf(w[2].y, u.z, fmaf(w[1].y, u.y, fmaf(w[0].y, u.x, exp_[i][1]))));
                    exp_[i][2] = fmaf(w[3].z, u.w, fmaf(w[2].z, u.z, fmaf(w[1].z, u.y, fmaf(w[0].z, u.x, exp_[i][2]))));
                    exp_[i][3] = fmaf(w[3].w, u.w, fmaf(w[2].w, u.z, fmaf(w[1].w, u.y, fmaf(w[0].w, u.x, exp_[i][3]))));
                }
            }
        }
        unsigned bad = 0;
        for (int i = 0; i < 4; ++i) for (int o = 0; o < 4; ++o) bad += acc[i][o] != exp_[i][o];
        if (bad) {
            const unsigned k = atomicAdd(errs, bad);
            if (k < 64) where[k] = (blockIdx.x << 16) | (rep << 8) | tid;
        }
    }
}

// MODE 0: bf16 32x32x16 MFMA fed from LDS; 1: bf16 MFMA from registers only; 2: f32 32x32x2 MFMA fed from LDS; 3: LDS reads + VALU only;
// 4: f16 32x32x16 MFMA fed from LDS
template <int MODE>
__global__ __launch_bounds__(256, 2) void aggressor(float* sink, int iters) {
    __shared__ bf16x8 big[4100];   // 65.6 KB
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 4100; i += 256) {
        bf16x8 v;
        for (int e = 0; e < 8; ++e) v[e] = (__bf16)(0.001f * ((i + e) & 31));
        big[i] = v;
    }
    __syncthreads();
    f32x16 c0 = {}, c1 = {};
    bf16x8 ra = big[lane], rb = big[64 + lane];
    float vs = 0.0f;
    for (int it = 0; it < iters; ++it) {
        const int base = ((it * 192) + (tid >> 6) * 64) % 3900;
        if (MODE == 0) {
            const bf16x8 a = big[base + lane], b = big[base + 64 + lane], d = big[base + 128 + lane];
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, d, c1, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(d, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, d, c1, 0, 0, 0);
        } else if (MODE == 4) {
            const f16x8 a = __builtin_bit_cast(f16x8, big[base + lane]), b = __builtin_bit_cast(f16x8, big[base + 64 + lane]),
                        d = __builtin_bit_cast(f16x8, big[base + 128 + lane]);
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, d, c1, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(d, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, d, c1, 0, 0, 0);
        } else if (MODE == 1) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ra, rb, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rb, ra, c1, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rb, rb, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ra, ra, c1, 0, 0, 0);
        } else if (MODE == 2) {
            const float4 a = reinterpret_cast<const float4*>(big)[base + lane], b = reinterpret_cast<const float4*>(big)[base + 64 + lane];
            c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, c1, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, c1, 0, 0, 0);
        } else {
            const float4 a = reinterpret_cast<const float4*>(big)[base + lane], b = reinterpret_cast<const float4*>(big)[base + 64 + lane];
            vs = fmaf(a.x, b.x, fmaf(a.y, b.y, fmaf(a.z, b.z, fmaf(a.w, b.w, vs))));
        }
    }
    float s = vs;
    for (int e = 0; e < 16; ++e) s += c0[e] + c1[e];
    if (s == 123.456f) sink[tid] = s;
}

// Same-kernel variant: even workgroups run the victim stencil, odd workgroups the bf16-MFMA loop (one launch).
__global__ __launch_bounds__(256, 2) void mixed(unsigned* errs, unsigned* where, float* sink, int reps, int iters) {
    __shared__ float4 tile[18][66];
    __shared__ float4 wl[36];
    __shared__ bf16x8 big[2800];
    const int tid = threadIdx.x, tx = tid & 63, ty = tid >> 6, lane = tid & 63;
    if (blockIdx.x & 1) {
        for (int i = tid; i < 2800; i += 256) {
            bf16x8 v;
            for (int e = 0; e < 8; ++e) v[e] = (__bf16)(0.001f * ((i + e) & 31));
            big[i] = v;
        }
        __syncthreads();
        f32x16 c0 = {}, c1 = {};
        for (int it = 0; it < iters; ++it) {
            const int base = ((it * 192) + (tid >> 6) * 64) % 2600;
            const bf16x8 a = big[base + lane], b = big[base + 64 + lane], d = big[base + 128 + lane];
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, d, c1, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(d, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, d, c1, 0, 0, 0);
        }
        float s = 0.0f;
        for (int e = 0; e < 16; ++e) s += c0[e] + c1[e];
        if (s == 123.456f) sink[tid] = s;
        return;
    }
    if (tid < 36) wl[tid] = make_float4(pat(tid, 0, 0, 7) - 0.5f, pat(tid, 1, 0, 7) - 0.5f, pat(tid, 2, 0, 7) - 0.5f, pat(tid, 3, 0, 7) - 0.5f);
    for (int idx = tid; idx < 18 * 66; idx += 256) {
        const int r = idx / 66, c = idx - r * 66;
        (&tile[0][0])[idx] = make_float4(pat(r, c, 0, blockIdx.x), pat(r, c, 1, blockIdx.x), pat(r, c, 2, blockIdx.x), pat(r, c, 3, blockIdx.x));
    }
    __syncthreads();
    for (int rep = 0; rep < reps; ++rep) {
        float acc[4][4], exp_[4][4];
        for (int i = 0; i < 4; ++i) for (int o = 0; o < 4; ++o) { acc[i][o] = 0.0f; exp_[i][o] = 0.0f; }
#pragma unroll 1
        for (int ky = 0; ky < 3; ++ky) {
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const float4* wq = &wl[(ky * 3 + kx) * 4];
                const float4 w0 = wq[0], w1 = wq[1], w2 = wq[2], w3 = wq[3];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float4 u = tile[4 * ty + ky + i][tx + kx];
                    acc[i][0] = fmaf(w3.x, u.w, fmaf(w2.x, u.z, fmaf(w1.x, u.y, fmaf(w0.x, u.x, acc[i][0]))));
                    acc[i][1] = fmaf(w3.y, u.w, fmaf(w2.y, u.z, fmaf(w1.y, u.y, fmaf(w0.y, u.x, acc[i][1]))));
                    acc[i][2] = fmaf(w3.z, u.w, fmaf(w2.z, u.z, fmaf(w1.z, u.y, fmaf(w0.z, u.x, acc[i][2]))));
                    acc[i][3] = fmaf(w3.w, u.w, fmaf(w2.w, u.z, fmaf(w1.w, u.y, fmaf(w0.w, u.x, acc[i][3]))));
                }
            }
        }
#pragma unroll 1
        for (int ky = 0; ky < 3; ++ky) {
#pragma unroll 1
            for (int kx = 0; kx < 3; ++kx) {
                const int t = (ky * 3 + kx) * 4;
                float4 w[4];
                for (int q = 0; q < 4; ++q)
                    w[q] = make_float4(pat(t + q, 0, 0, 7) - 0.5f, pat(t + q, 1, 0, 7) - 0.5f, pat(t + q, 2, 0, 7) - 0.5f, pat(t + q, 3, 0, 7) - 0.5f);
                for (int i = 0; i < 4; ++i) {
                    const int r = 4 * ty + ky + i, c = tx + kx;
                    const float4 u = make_float4(pat(r, c, 0, blockIdx.x), pat(r, c, 1, blockIdx.x), pat(r, c, 2, blockIdx.x), pat(r, c, 3, blockIdx.x));
                    exp_[i][0] = fmaf(w[3].x, u.w, fmaf(w[2].x, u.z, fmaf(w[1].x, u.y, fmaf(w[0].x, u.x, exp_[i][0]))));
                    exp_[i][1] = fmaf(w[3].y, u.w, fmaf(w[2].y, u.z, fmaf(w[1].y, u.y, fmaf(w[0].y, u.x, exp_[i][1]))));
                    exp_[i][2] = fmaf(w[3].z, u.w, fmaf(w[2].z, u.z, fmaf(w[1].z, u.y, fmaf(w[0].z, u.x, exp_[i][2]))));
                    exp_[i][3] = fmaf(w[3].w, u.w, fmaf(w[2].w, u.z, fmaf(w[1].w, u.y, fmaf(w[0].w, u.x, exp_[i][3]))));
                }
            }
        }
        unsigned bad = 0;
        for (int i = 0; i < 4; ++i) for (int o = 0; o < 4; ++o) bad += acc[i][o] != exp_[i][o];
        if (bad) {
            const unsigned k = atomicAdd(errs, bad);
            if (k < 64) where[k] = (blockIdx.x << 16) | (rep << 8) | tid;
        }
    }
}

int main(int argc, char** argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 100;
    unsigned *errs, *where;
    float* sink;
    hipMalloc(&errs, 4); hipMalloc(&where, 256); hipMalloc(&sink, 4096);
    hipStream_t a, b;
    hipStreamCreateWithFlags(&a, hipStreamNonBlocking);
    hipStreamCreateWithFlags(&b, hipStreamNonBlocking);
    const char* names[6] = {"bf16 mfma + LDS", "bf16 mfma regs", "f32 mfma + LDS", "LDS + VALU", "f16 mfma + LDS", "none"};
    for (int mode = 0; mode < 6; ++mode) {
        hipMemset(errs, 0, 4);
        hipDeviceSynchronize();
        for (int r = 0; r < rounds; ++r) {
            victim<<<3600, 256, 0, a>>>(errs, where, 8);
            switch (mode) {
                case 0: aggressor<0><<<1800, 256, 0, b>>>(sink, 300); break;
                case 1: aggressor<1><<<1800, 256, 0, b>>>(sink, 300); break;
                case 2: aggressor<2><<<1800, 256, 0, b>>>(sink, 300); break;
                case 3: aggressor<3><<<1800, 256, 0, b>>>(sink, 300); break;
                case 4: aggressor<4><<<1800, 256, 0, b>>>(sink, 300); break;
                default: break;
            }
        }
        hipDeviceSynchronize();
        unsigned h, w[64];
        hipMemcpy(&h, errs, 4, hipMemcpyDeviceToHost);
        hipMemcpy(w, where, 256, hipMemcpyDeviceToHost);
        printf("aggressor %-16s: victim mismatches %u", names[mode], h);
        for (unsigned i = 0; i < h && i < 8; ++i) printf("  [blk %u rep %u tid %u]", w[i] >> 16, (w[i] >> 8) & 255, w[i] & 255);
        printf("\n");
    }
    hipMemset(errs, 0, 4);
    hipDeviceSynchronize();
    for (int r = 0; r < rounds; ++r) mixed<<<7200, 256, 0, a>>>(errs, where, sink, 8, 300);
    hipDeviceSynchronize();
    unsigned h;
    hipMemcpy(&h, errs, 4, hipMemcpyDeviceToHost);
    printf("same kernel, one stream (even WGs victim / odd WGs bf16 mfma): victim mismatches %u\n", h);
    return 0;
}
