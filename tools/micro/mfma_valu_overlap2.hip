// Round-3 redo of the MFMA / VALU co-issue measurement on gfx950 (Weak #6): how much vector-ALU work hides in the
// gaps of v_mfma_f32_32x32x16_f16, measured in SHADER CYCLES (s_memtime) with the achieved clock reported next to it
// (s_memrealtime, 100 MHz), with measured -- not derived -- single-role legs, at one and at two waves per SIMD, and for
// 0 / 2 / 4 / 6 / 8 / 12 single-issue fillers per MFMA gap.
//   hipcc --offload-arch=gfx950 -O3 mfma_valu_overlap2.hip -o mfma_valu_overlap2 && ./mfma_valu_overlap2
//   llvm-objdump -d (tools/micro/check_overlap2_isa.sh) confirms that the loop bodies carry no s_nop between the instructions.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// One iteration = 8 "slots"; a slot = the MFMA (if MF) followed by NF independent v_fma_f32 on four accumulators (no
// dependent-issue stalls: chain distance 4).  A whole iteration sits in ONE asm volatile statement, so the compiler can neither
// move an MFMA away from its fillers nor insert hazard s_nops anywhere in the measured stream (checked in the disassembly).
#define F1(r) "v_fma_f32 %" #r ", %" #r ", %6, %6\n"
#define FILL0 ""
#define FILL2 F1(2) F1(3)
#define FILL4 F1(2) F1(3) F1(4) F1(5)
#define FILL6 FILL4 FILL2
#define FILL8 FILL4 FILL4
#define FILL12 FILL4 FILL4 FILL4
#define MFMA0 "v_mfma_f32_32x32x16_f16 %0, %7, %8, %0\n"
#define MFMA1 "v_mfma_f32_32x32x16_f16 %1, %7, %8, %1\n"
#define ITER_MF(F) MFMA0 F MFMA1 F MFMA0 F MFMA1 F MFMA0 F MFMA1 F MFMA0 F MFMA1 F
#define ITER_VA(F) F F F F F F F F
#define ITER_ASM(body) asm volatile(body : "+v"(c0), "+v"(c1), "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(k), "v"(a), "v"(b))
template <bool MF, int NF>
__device__ __forceinline__ void iter8(f32x16& c0, f32x16& c1, float& x0, float& x1, float& x2, float& x3, float k, const f16x8& a, const f16x8& b) {
    if constexpr (MF) {
        if constexpr (NF == 0) ITER_ASM(ITER_MF(FILL0));
        if constexpr (NF == 2) ITER_ASM(ITER_MF(FILL2));
        if constexpr (NF == 4) ITER_ASM(ITER_MF(FILL4));
        if constexpr (NF == 6) ITER_ASM(ITER_MF(FILL6));
        if constexpr (NF == 8) ITER_ASM(ITER_MF(FILL8));
        if constexpr (NF == 12) ITER_ASM(ITER_MF(FILL12));
    } else {
        if constexpr (NF == 2) ITER_ASM(ITER_VA(FILL2));
        if constexpr (NF == 4) ITER_ASM(ITER_VA(FILL4));
        if constexpr (NF == 6) ITER_ASM(ITER_VA(FILL6));
        if constexpr (NF == 8) ITER_ASM(ITER_VA(FILL8));
        if constexpr (NF == 12) ITER_ASM(ITER_VA(FILL12));
    }
}

// ROLE 0: every wave {MFMA, NF fillers} x 8 per iteration          (interleaved)
// ROLE 1: MFMA only          ROLE 2: fillers only (8 x NF per iteration)
// ROLE 3: waves 0-3 MFMA only, waves 4-7 fillers only              (specialised partners; 512-thread launches)
// ROLE 4: every wave 8 MFMA then 8 x NF fillers                    (phased, partners in step)
// ROLE 5: as 4, but waves 4-7 run the filler phase FIRST           (phased, partners half a block apart = the guide's stagger)
template <int ROLE, int NF, int NT>
__global__ __launch_bounds__(NT) void kern(float* out, long long* cyc, int iters, float k) {
    const int wave = threadIdx.x >> 6;
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.01f + 0.5f); }
    f32x16 c0 = {0}, c1 = {0};
    float x0 = threadIdx.x, x1 = x0 * 0.5f, x2 = x0 * 0.25f, x3 = x0 * 0.125f;
    const bool mf = ROLE == 1 || ROLE == 4 || ROLE == 5 || (ROLE == 3 && wave < 4);
    const bool va = ROLE == 2 || ROLE == 4 || ROLE == 5 || (ROLE == 3 && wave >= 4);
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    const long long r0 = __builtin_amdgcn_s_memrealtime();
    if (ROLE == 5 && wave >= 4) iter8<false, NF>(c0, c1, x0, x1, x2, x3, k, a, b);
    for (int it = 0; it < iters; ++it) {
        if (ROLE == 0) {
            iter8<true, NF>(c0, c1, x0, x1, x2, x3, k, a, b);
        } else {
            if (mf) iter8<true, 0>(c0, c1, x0, x1, x2, x3, k, a, b);
            if (va) iter8<false, NF>(c0, c1, x0, x1, x2, x3, k, a, b);
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    const long long r1 = __builtin_amdgcn_s_memrealtime();
    float s = x0 + x1 + x2 + x3;
    for (int i = 0; i < 16; ++i) s += c0[i] + c1[i];
    out[blockIdx.x * NT + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) {
        cyc[(blockIdx.x * (NT / 64) + wave) * 2] = t1 - t0;
        cyc[(blockIdx.x * (NT / 64) + wave) * 2 + 1] = r1 - r0;
    }
}

struct Res { double cyc_per_gap, ghz; };

template <int ROLE, int NF, int NT>
Res run(float* out, long long* cyc, int iters, int blocks, bool second_half = false) {
    kern<ROLE, NF, NT><<<blocks, NT>>>(out, cyc, 200, 0.5f);
    hipDeviceSynchronize();
    kern<ROLE, NF, NT><<<blocks, NT>>>(out, cyc, iters, 0.5f);
    hipDeviceSynchronize();
    const int nw = NT / 64;
    std::vector<long long> h(blocks * nw * 2);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> c, g;
    for (int b = 0; b < blocks; ++b)
        for (int w = 0; w < nw; ++w) {
            if (ROLE == 3 && ((w >= 4) != second_half)) continue;
            c.push_back((double)h[(b * nw + w) * 2]);
            g.push_back((double)h[(b * nw + w) * 2] / ((double)h[(b * nw + w) * 2 + 1] * 10.0) );   // cycles per ns = GHz
        }
    std::sort(c.begin(), c.end()); std::sort(g.begin(), g.end());
    return {c[c.size() / 2] / (8.0 * iters), g[g.size() / 2]};
}

#define ROW(ROLE, NF, NT, label)                                                                                         \
    { Res r = run<ROLE, NF, NT>(out, cyc, iters, 256);                                                                     \
      printf("%-74s %7.2f cyc per MFMA slot   clock %.2f GHz\n", label, r.cyc_per_gap, r.ghz); }

int main() {
    float* out; long long* cyc;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8 * 2 * 8);
    const int iters = 20000;
    printf("per wave and iteration: 8 'slots' = 8 x {one v_mfma_f32_32x32x16_f16 and / or NF x v_fma_f32}; median over waves; random-ish operands\n\n");
    printf("--- one wave per SIMD (256-thread workgroups, one per CU)\n");
    ROW(1, 0, 256, "MFMA only");
    ROW(2, 2, 256, "VALU only, 2 per slot");
    ROW(2, 4, 256, "VALU only, 4 per slot");
    ROW(2, 6, 256, "VALU only, 6 per slot");
    ROW(2, 8, 256, "VALU only, 8 per slot");
    ROW(2, 12, 256, "VALU only, 12 per slot");
    ROW(0, 2, 256, "MFMA + 2 fillers per gap");
    ROW(0, 4, 256, "MFMA + 4 fillers per gap");
    ROW(0, 6, 256, "MFMA + 6 fillers per gap");
    ROW(0, 8, 256, "MFMA + 8 fillers per gap");
    ROW(0, 12, 256, "MFMA + 12 fillers per gap");
    printf("--- two waves per SIMD (512-thread workgroups, one per CU); per-wave figures\n");
    ROW(1, 0, 512, "MFMA only (both partners)");
    ROW(2, 4, 512, "VALU only, 4 per slot (both partners)");
    ROW(2, 8, 512, "VALU only, 8 per slot (both partners)");
    ROW(0, 2, 512, "both partners: MFMA + 2 fillers per gap");
    ROW(0, 4, 512, "both partners: MFMA + 4 fillers per gap");
    ROW(0, 6, 512, "both partners: MFMA + 6 fillers per gap");
    ROW(0, 8, 512, "both partners: MFMA + 8 fillers per gap");
    { Res a = run<3, 8, 512>(out, cyc, iters, 256, false), b = run<3, 8, 512>(out, cyc, iters, 256, true);
      printf("%-74s %7.2f / %7.2f cyc per slot (MFMA wave / VALU wave)   clock %.2f GHz\n", "specialised partners: waves 0-3 MFMA only, waves 4-7 8 VALU per slot",
             a.cyc_per_gap, b.cyc_per_gap, a.ghz); }
    { Res a = run<3, 4, 512>(out, cyc, iters, 256, false), b = run<3, 4, 512>(out, cyc, iters, 256, true);
      printf("%-74s %7.2f / %7.2f cyc per slot (MFMA wave / VALU wave)   clock %.2f GHz\n", "specialised partners: waves 0-3 MFMA only, waves 4-7 4 VALU per slot",
             a.cyc_per_gap, b.cyc_per_gap, a.ghz); }
    ROW(4, 8, 512, "phased, partners in step: 8 MFMA then 64 VALU");
    ROW(5, 8, 512, "phased, waves 4-7 half a block behind (stagger): 8 MFMA then 64 VALU");
    ROW(4, 4, 512, "phased, partners in step: 8 MFMA then 32 VALU");
    ROW(5, 4, 512, "phased, waves 4-7 half a block behind (stagger): 8 MFMA then 32 VALU");
    return 0;
}
