// Micro-test: what does ONE iteration of a persistent loader / consumer skeleton cost on MI355X, and what makes it grow
// with the number of workgroups on the chip?  profiles/r04_bf16_ring_ab.txt left "0.45-0.93 us per empty barrier iteration, growing with the
// workgroups on the chip -- suspects not separated" (instruction-cache sharing between neighbouring CUs against the 50 KB kernel image, the
// loader wave's instruction path, the kernarg scalar loads).  This file separates them with a workgroup of NCONS consumer waves + one loader
// wave that meet at ONE s_barrier per iteration and otherwise do only what the selected leg adds:
//
//   leg 0  barrier only (all waves)
//   leg 1  + loader: L dependent SALU operations per iteration                      (scalar bookkeeping: cursors, compares)
//   leg 2  + loader: L/4 x (two 64-bit selects + one 64-bit add) on the VALU        (the ring's per-DMA address arithmetic)
//   leg 3  leg 2 + 4 s_load_dwordx4 from the kernarg segment per iteration          (the quad descriptors)
//   leg 4  leg 2 + one global_load_dword (L2 hit) + s_waitcnt vmcnt(0) per iteration (a landing wait)
//   leg 5  leg 2 + 16 global_load_lds_dwordx4 (16 KiB ring slot, L2-resident source), counted vmcnt(16) (the real ring fill)
//   leg 6  leg 5, loader addresses kept incrementally (no selects: one 64-bit add per DMA)
//   leg 7  leg 6 with TWO loader waves (8 DMAs each), on different SIMDs
//
// IMG = 0: the kernel image is the loop (a few hundred bytes to ~3 KB).  IMG = 1: the same loop with ~38 KB of never-executed code in the
// same kernel (a switch over 12 bulky "epilogue modes", as the shipped conv kernels carry), placed BETWEEN the loader path and the consumer
// path so that the executed instructions straddle it.  IMG = 2: the bulk is EXECUTED once per iteration by the consumers (a 16 KB straight-line
// VALU body: the footprint a heavily unrolled tap loop has), i.e. the instruction cache has to hold loader + body for every resident workgroup.
//
// Per configuration the kernel runs with I1 and I2 iterations; (t(I2) - t(I1)) / (I2 - I1) is the cost of one iteration without launch and
// preamble.  Reported in ns (hipEvents) and in s_memtime ticks of wave 0 of the median workgroup (measured: 2.4 ticks per ns, i.e. the counter runs at the 2.4 GHz shader clock on this part).
// Build: make -C tools/micro barrier_skeleton ; run: tools/micro/barrier_skeleton > profiles/r05_barrier_skeleton.txt
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

struct Desc { const float* base; long long bstride; int rs, cs, mask, pad; };   // a conv quad descriptor look-alike (32 B)
struct Args {
    const float* src;      // 4 MiB L2-resident source for the DMA legs
    float* sink;
    long long* stamps;
    int iters, leg, L, cold;
    Desc qd[16];
};

typedef __attribute__((address_space(3))) void* lds_vp;
typedef const __attribute__((address_space(1))) void* glb_vp;

__device__ __forceinline__ void bar() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// ~4 KB of VALU code per instantiation that cannot be folded: used as never-executed bulk (IMG 1) or executed body (IMG 2)
template <int SALT>
__device__ __forceinline__ float bulk(float x, float y) {
#pragma unroll
    for (int i = 0; i < 160; ++i) {
        x = __builtin_fmaf(x, y, (float)(SALT * 131 + i));
        y = __builtin_fmaf(y, x, (float)(SALT * 17 + i) * 0.5f);
        asm volatile("" : "+v"(x), "+v"(y));
    }
    return x + y;
}

template <int NCONS, int NLOAD, int IMG>
__global__ __launch_bounds__(64 * (NCONS + NLOAD)) void skel(const Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    long long t0 = 0;
    if (wave == 0) t0 = __builtin_amdgcn_s_memtime();
    lds[tid] = (unsigned char)tid;
    __syncthreads();
    const int iters = a.iters, leg = a.leg, L = a.L;
    float keep = 0.0f;
    if (wave >= NCONS) {
        // ------------------------------------------------------------ loader wave(s)
        const int lw = wave - NCONS;
        unsigned sacc = (unsigned)blockIdx.x;
        const float* p0 = a.src + (size_t)(blockIdx.x & 255) * 4096 + lane * 4;     // 16 KiB per workgroup, L2-resident after the first pass
        const float* p1 = p0 + 1024 * 1024 / 4;
        const float* inc[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) inc[k] = p0 + k * 256;
        for (int u = 0; u < iters; ++u) {
            if (leg == 1) {
                for (int i = 0; i < L; ++i) { sacc = sacc * 5u + 1u; asm volatile("" : "+s"(sacc)); }
            }
            if (leg >= 2 && leg <= 5) {
                const int nd = L / 4;
                for (int k = 0; k < nd; ++k) {          // two 64-bit selects + one 64-bit add per "DMA address", dependent on the cursor
                    const float* qa = (lane + u + k) & 1 ? p0 : p1;
                    const float* qb = (lane + k) & 2 ? p1 : p0;
                    const float* g = (lane >= 34 ? qa : qb) + (size_t)((u * 7 + k) & 1023) * 4;
                    asm volatile("" :: "v"(g));
                    if (leg == 5 && k < 16)
                        __builtin_amdgcn_global_load_lds((glb_vp)(p0 + k * 256 + ((u & 3) << 12)), (lds_vp)(lds + 1024 + (u & 3) * 16384 + k * 1024), 16, 0, 0);
                }
                if (leg == 5) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            }
            if (leg == 3) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const Desc d = a.qd[(u + q) & 15];
                    sacc += (unsigned)d.rs + (unsigned)(size_t)d.base;
                }
                asm volatile("" : "+s"(sacc));
            }
            if (leg == 4) {
                const float v = __builtin_nontemporal_load(p0 + ((u & 63) << 6));
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                keep += v;
            }
            if (leg == 6 || leg == 7) {
                const int per = 16 / NLOAD;
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    if (k / per != lw && NLOAD > 1) continue;
                    __builtin_amdgcn_global_load_lds((glb_vp)inc[k], (lds_vp)(lds + 1024 + (u & 3) * 16384 + k * 1024), 16, 0, 0);
                    inc[k] += ((u & 3) == 3) ? -3 * 4096 : 4096;   // walk 4 slots of the source, incrementally
                }
                if (per == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            }
            bar();
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (sacc == 0x12345u || keep == 1.5f) a.sink[0] = 1.0f;
        return;
    }
    if (IMG == 1 && a.cold) {   // never taken (cold == 0): ~48 KB of code between the two roles' paths
        float x = (float)tid, y = 1.0f + lane;
        switch (a.cold) {
            case 1: x = bulk<1>(x, y); break;   case 2: x = bulk<2>(x, y); break;   case 3: x = bulk<3>(x, y); break;
            case 4: x = bulk<4>(x, y); break;   case 5: x = bulk<5>(x, y); break;   case 6: x = bulk<6>(x, y); break;
            case 7: x = bulk<7>(x, y); break;   case 8: x = bulk<8>(x, y); break;   case 9: x = bulk<9>(x, y); break;
            case 10: x = bulk<10>(x, y); break; case 11: x = bulk<11>(x, y); break; default: x = bulk<12>(x, y); break;
        }
        a.sink[tid] = x;
    }
    // ---------------------------------------------------------------- consumers
    float cx = (float)lane, cy = 1.0f;
    for (int u = 0; u < iters; ++u) {
        bar();
        if (IMG == 2) { cx = bulk<21>(cx, cy); cy = bulk<22>(cy, cx); cx = bulk<23>(cx, cy); cy = bulk<24>(cy, cx); cx = bulk<25>(cx, cy); }
    }
    if (cx == 123.25f) a.sink[tid] = cx;
    if (wave == 0 && lane == 0 && a.stamps) a.stamps[blockIdx.x] = __builtin_amdgcn_s_memtime() - t0;
}

struct Dev {
    float* src; float* sink; long long* stamps;
};

template <int NCONS, int NLOAD, int IMG>
static void run(const Dev& d, const char* name, int leg, int L, int grid, size_t ldsb) {
    static bool once = false;
    if (!once) { once = true; CK(hipFuncSetAttribute((const void*)skel<NCONS, NLOAD, IMG>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); }
    Args a{};
    a.src = d.src; a.sink = d.sink; a.stamps = d.stamps; a.leg = leg; a.L = L; a.cold = 0;
    for (int q = 0; q < 16; ++q) { a.qd[q].base = d.src + q * 64; a.qd[q].bstride = 0; a.qd[q].rs = 640 + q; a.qd[q].cs = 4; a.qd[q].mask = 15; }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int I1 = 64, I2 = 576;
    double best[2] = {1e30, 1e30}; long long tick[2] = {0, 0};
    for (int rep = 0; rep < 5; ++rep)
        for (int w = 0; w < 2; ++w) {
            a.iters = w ? I2 : I1;
            CK(hipEventRecord(e0));
            skel<NCONS, NLOAD, IMG><<<grid, 64 * (NCONS + NLOAD), ldsb>>>(a);
            CK(hipEventRecord(e1));
            CK(hipDeviceSynchronize());
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms * 1e6 < best[w]) {
                best[w] = ms * 1e6;
                std::vector<long long> st(grid);
                CK(hipMemcpy(st.data(), d.stamps, grid * sizeof(long long), hipMemcpyDeviceToHost));
                std::sort(st.begin(), st.end());
                tick[w] = st[grid / 2];
            }
        }
    const double per_ns = (best[1] - best[0]) / (I2 - I1), per_tick = (double)(tick[1] - tick[0]) / (I2 - I1);
    printf("%-34s waves %d+%d img %d leg %d L %4d  grid %4d lds %6zu : %8.1f ns / iteration   (median workgroup %7.1f s_memtime ticks)   launch+preamble %5.1f us\n",
           name, NCONS, NLOAD, IMG, leg, L, grid, ldsb, per_ns, per_tick, (best[0] - I1 * per_ns) * 1e-3);
    fflush(stdout);
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
}

int main() {
    Dev d;
    CK(hipMalloc(&d.src, 8 << 20)); CK(hipMemset(d.src, 0, 8 << 20));
    CK(hipMalloc(&d.sink, 1 << 20)); CK(hipMalloc(&d.stamps, 4096 * sizeof(long long)));
    hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
    printf("# %s, %d CUs, clock %d MHz; one s_barrier per iteration; per-iteration cost = (t(576) - t(64)) / 512, best of 5\n", pr.name, pr.multiProcessorCount, pr.clockRate / 1000);
    const size_t one = 96 * 1024, two = 72 * 1024;   // dynamic LDS that admits one / two workgroups per CU (the ring slots need 65 KB + 1 KB)
    const int grids1[] = {64, 128, 256}, grids2[] = {512};
    printf("\n## A. barrier only: waves per workgroup and workgroups on the chip\n");
    for (int g : grids1) run<4, 1, 0>(d, "barrier only", 0, 0, g, one);
    for (int g : grids2) run<4, 1, 0>(d, "barrier only, 2 WG/CU", 0, 0, g, two);
    for (int g : grids1) run<8, 0, 0>(d, "barrier only, 8 waves", 0, 0, g, one);
    for (int g : grids1) run<3, 1, 0>(d, "barrier only, 4 waves", 0, 0, g, one);
    printf("\n## B. loader instruction path (4 consumers + 1 loader)\n");
    for (int L : {64, 256, 1024}) for (int g : grids1) run<4, 1, 0>(d, "loader SALU chain", 1, L, g, one);
    for (int L : {64, 256}) for (int g : grids1) run<4, 1, 0>(d, "loader 64-bit VALU addresses", 2, L, g, one);
    for (int g : grids2) run<4, 1, 0>(d, "loader 64-bit VALU, 2 WG/CU", 2, 64, g, two);
    for (int g : grids1) run<4, 1, 0>(d, "  + 4 kernarg s_load_x4", 3, 64, g, one);
    for (int g : grids1) run<4, 1, 0>(d, "  + 1 global load + vmcnt(0)", 4, 64, g, one);
    printf("\n## C. real ring fill: 16 x 1 KiB LDS-DMA per iteration (L2-resident source), counted vmcnt\n");
    for (int g : grids1) run<4, 1, 0>(d, "selects + 16 DMA", 5, 64, g, one);
    for (int g : grids2) run<4, 1, 0>(d, "selects + 16 DMA, 2 WG/CU", 5, 64, g, two);
    for (int g : grids1) run<4, 1, 0>(d, "incremental addresses + 16 DMA", 6, 0, g, one);
    for (int g : grids2) run<4, 1, 0>(d, "incremental + 16 DMA, 2 WG/CU", 6, 0, g, two);
    for (int g : grids1) run<4, 2, 0>(d, "two loader waves x 8 DMA", 7, 0, g, one);
    for (int g : grids2) run<4, 2, 0>(d, "two loader waves, 2 WG/CU", 7, 0, g, two);
    printf("\n## D. kernel image: ~38 KB of never-executed code in the kernel (IMG 1), 12 KB executed per iteration by the consumers (IMG 2)\n");
    for (int g : grids1) run<4, 1, 1>(d, "barrier only, cold bulk", 0, 0, g, one);
    for (int g : grids1) run<4, 1, 1>(d, "loader VALU, cold bulk", 2, 64, g, one);
    for (int g : grids1) run<4, 1, 1>(d, "16 DMA incremental, cold bulk", 6, 0, g, one);
    for (int g : grids1) run<4, 1, 2>(d, "executed 16 KB body", 0, 0, g, one);
    for (int g : grids2) run<4, 1, 2>(d, "executed 16 KB body, 2 WG/CU", 0, 0, g, two);
    for (int g : grids1) run<4, 1, 2>(d, "executed body + 16 DMA incr.", 6, 0, g, one);
    return 0;
}
