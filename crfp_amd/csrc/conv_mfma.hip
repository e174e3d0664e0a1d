// 3x3 stride-1 pad-1 convolution as an implicit GEMM on the MFMA, gfx950.  Replaces every "wide" nn.Conv2d(k=3) of
// the reference path (model/CRFP.py:303-317, 449-450, 532, 172-176, 257-261, 747-795; model/LTE.py:40-42).
//
// File map: shared epilogue (conv_epilogue_t) | fp32-MFMA kernel conv3x3_mfma_kernel (v_mfma_f32_32x32x2_f32: the per-op
// C-ABI and the few convs with NCHW / ragged-K sources) | the split-operand kernels on the 16-bit MFMA: the DEFAULT
// conv3x3_split_kernel<1,1,2> (f16x3; bf16x6 as <.,.,3>), and the opt-in experiments (pipelined
// persistent, input-stationary, warp-specialised) | weight packers | launch_conv_mfma.
// The layout notes below are written for the fp32 kernel; the split kernels share orientation, lane map and epilogue.
//
// GEMM orientation:  D[cout][pixel] += A[cout][k] * B[k][pixel]
//   A = weights (rows = 32 output channels of a cout tile), pre-packed so that the 64 lanes of a
//       wave read one contiguous 1 KiB (float4 per lane) per (k-quad-pair, tap): lane (row, half)
//       holds W[row][4 K-channels of quad 2*pair+half][tap].
//   B = activations from an LDS halo tile in Q4 layout: lane (pixel, half) does ONE ds_read_b128
//       = the 4 K-channels of quad 2*pair+half at its (shifted) pixel -> feeds 4 MFMAs.
//   D: lane (pixel = lane&31, half) ends up with 16 rows = four 4-channel groups
//       {8g+4*half .. +3}, g=0..3, i.e. four aligned Q4 elements -> four 16-B stores per tile.
// All layout permutations (virtual concat of several sources, pixel-shuffle on the store side,
// pixel-unshuffle on the load side) are folded into the weight packing (rows / K order) plus
// address arithmetic; no permutation kernel ever runs.
//
// Work decomposition: 256-thread workgroup (4 waves) = 8 rows x 64 px output tile; each wave owns
// 2 rows x 2 half-rows = four 32-pixel MFMA column tiles, times CT cout tiles.  K loop walks the
// input in chunks of 2 quads (8 channels): stage (8+2)x(64+2) halo -> LDS, 9 taps x 4 MFMAs x CT x 4.
#include "crfp_common.h"

#include <cstdlib>
#include <cstring>

namespace CRFP_NS {

constexpr int TW = 64, LW = TW + 2;
// activation loads of the implicit-GEMM kernels.  -DCRFP_CONV_NT (A/B builds): non-temporal, so that a layer's INPUT does not displace its
// OUTPUT from the XCD's 4 MB L2 -- with the XCD-banded tile order the next layer's workgroups run on the XCD that wrote their input.
#ifdef CRFP_CONV_NT
#define CRFP_LDACT(T, p) __builtin_nontemporal_load(reinterpret_cast<const T*>(p))
#else
#define CRFP_LDACT(T, p) (*reinterpret_cast<const T*>(p))
#endif
#ifndef CRFP_TAP_UNROLL
#define CRFP_TAP_UNROLL 3
#endif
#ifndef CRFP_SPLIT_TAP_UNROLL
#define CRFP_SPLIT_TAP_UNROLL 9   // default split kernel: full unroll measured +1.4 % frames/s over 3 (168 VGPRs, still 3 waves/SIMD)
#endif

// tanh / sigmoid on the hardware exp2 + rcp (v_exp_f32, v_rcp_f32: ~1 ulp each) instead of the libm
// versions (~30 VALU instructions each): the 216-channel offset/mask epilogue is transcendental-bound
// otherwise.  |error| < 3e-7 absolute on both, i.e. < 3e-6 px on the +-10 px DCN offsets.
__device__ __forceinline__ float fast_tanh(float v) {
    const float e = __expf(-2.0f * fabsf(v));          // in (0, 1]: no overflow
    const float t = (1.0f - e) * __builtin_amdgcn_rcpf(1.0f + e);
    return copysignf(t, v);
}

__device__ __forceinline__ float apply_act(float v, int act) {
    switch (act) {
        case CRFP_ACT_RELU: return fmaxf(v, 0.0f);
        case CRFP_ACT_LRELU01: return v > 0.0f ? v : 0.1f * v;
        case CRFP_ACT_TANH: return fast_tanh(v);
        case CRFP_ACT_SIGMOID: return fast_sigmoid(v);
        default: return v;
    }
}

// Stage NIN halo elements of one K-quad into registers; the switch on the source kind is hoisted out
// of the (fully unrolled) element loop so every case is straight-line code with static register indices.
template <int NIN>
__device__ __forceinline__ void load_quad_batch(float4 (&r)[NIN], const ConvSrc& s, int n, int kql,
                                                const int (&gy)[NIN], const int (&gx)[NIN], const bool (&ok)[NIN],
                                                int H, int W) {
    const float* base = s.p + (long long)n * s.bstride;
#pragma unroll
    for (int k = 0; k < NIN; ++k) r[k] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    switch (s.kind) {
        case SRC_Q4: {
            const int PW = W + s.pad;
            const act_t* b = as_act(s.p) + (long long)n * s.bstride + (long long)kql * (H + s.pad) * PW * 4;
#pragma unroll
            for (int k = 0; k < NIN; ++k)
                if (ok[k]) { const cf32x4 v = ldq(b + ((long long)gy[k] * PW + gx[k]) * 4); r[k] = make_float4(v.x, v.y, v.z, v.w); }
            break;
        }
        case SRC_NCHW: {
#pragma unroll
            for (int k = 0; k < NIN; ++k)
                if (ok[k]) {
                    const long long o = (long long)gy[k] * W + gx[k], pl = (long long)H * W;
                    const int c0 = 4 * kql;
                    r[k].x = c0 + 0 < s.nch ? base[(c0 + 0) * pl + o] : 0.0f;
                    r[k].y = c0 + 1 < s.nch ? base[(c0 + 1) * pl + o] : 0.0f;
                    r[k].z = c0 + 2 < s.nch ? base[(c0 + 2) * pl + o] : 0.0f;
                    r[k].w = c0 + 3 < s.nch ? base[(c0 + 3) * pl + o] : 0.0f;
                }
            break;
        }
        case SRC_UNSHUF4: {
            const int Qp = kql >> 4, ij = kql & 15, i = ij >> 2, jj = ij & 3;
            const int H4 = 4 * H + s.pad, W4 = 4 * W + s.pad;
            const act_t* b = as_act(s.p) + (long long)n * s.bstride + (long long)Qp * H4 * W4 * 4;
#pragma unroll
            for (int k = 0; k < NIN; ++k)
                if (ok[k]) { const cf32x4 v = ldq(b + ((long long)(4 * gy[k] + i) * W4 + 4 * gx[k] + jj) * 4); r[k] = make_float4(v.x, v.y, v.z, v.w); }
            break;
        }
        case SRC_FLOW2: {
#pragma unroll
            for (int k = 0; k < NIN; ++k)
                if (ok[k]) {
                    const float2 f = *reinterpret_cast<const float2*>(base + ((long long)gy[k] * W + gx[k]) * 2);
                    r[k] = make_float4(f.x, f.y, 0.0f, 0.0f);
                }
            break;
        }
        case SRC_NCHW_SHIFT: {   // the tensor shifted by (sy, sx): its own validity test (ok[] belongs to the unshifted tile)
            const int sy = (s.rsv & 15) - 8, sx = ((s.rsv >> 4) & 15) - 8;
            const float lo = (s.rsv & 256) ? 0.0f : -__builtin_inff();   // ReLU on the way in
            const long long pl = (long long)H * W;
            const int c0 = 4 * kql;
#pragma unroll
            for (int k = 0; k < NIN; ++k) {
                const int y = gy[k] + sy, x = gx[k] + sx;
                if (y >= 0 && y < H && x >= 0 && x < W) {
                    const long long o = (long long)y * W + x;
                    r[k].x = c0 + 0 < s.nch ? fmaxf(base[(c0 + 0) * pl + o], lo) : 0.0f;
                    r[k].y = c0 + 1 < s.nch ? fmaxf(base[(c0 + 1) * pl + o], lo) : 0.0f;
                    r[k].z = c0 + 2 < s.nch ? fmaxf(base[(c0 + 2) * pl + o], lo) : 0.0f;
                    r[k].w = c0 + 3 < s.nch ? fmaxf(base[(c0 + 3) * pl + o], lo) : 0.0f;
                }
            }
            break;
        }
        default: break;
    }
}

// The f16x3 split of 4 values (same arithmetic as split_f16x8_fast further down, which a consuming conv would apply):
// hi[] = fp16(x) pairs, lo[] = fp16((x - fp16(x)) * 2^11) pairs.
__device__ __forceinline__ void split_f16x4(const float (&x)[4], unsigned (&hi)[2], unsigned (&lo)[2]) {
    typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
    typedef float f2_t __attribute__((ext_vector_type(2)));
    float negone = -1.0f;
    asm("" : "+v"(negone));   // opaque: keeps fma(x0, -1, x) a v_fma_mix_f32
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const h2_t a = __builtin_convertvector(f2_t{x[2 * i], x[2 * i + 1]}, h2_t);
        const float r0 = __builtin_fmaf((float)a[0], negone, x[2 * i]);
        const float r1 = __builtin_fmaf((float)a[1], negone, x[2 * i + 1]);
        const h2_t b = __builtin_convertvector(f2_t{r0 * 2048.0f, r1 * 2048.0f}, h2_t);
        hi[i] = __builtin_bit_cast(unsigned, a);
        lo[i] = __builtin_bit_cast(unsigned, b);
    }
}

// Epilogue shared by the fp32-MFMA and the split-bf16-MFMA main loops (identical C/D lane map):
// bias, activation, scale, residual, layout-aware store.
//
// Everything that comes from the kernel arguments (destinations, strides, activation, store mode) is read
// ONCE into registers and the store mode is a template parameter: the first version re-read descriptors through
// scalar loads inside the (pt, g, dst) loops and carried every mode's code -- 7 300 ISA lines, 146 branches,
// 15 k cycles per workgroup (38 % of a 32->32 conv's workgroup lifetime, measured with s_memtime stamps).
// Bias quads and the residual are loaded up front so that the stores leave back to back.
//
// VBIAS = 2: the caller initialised the accumulators with the bias (loads issued at kernel start: no L2 round trip here);
// VBIAS = 1: bias through per-lane vector loads (single epilogue per workgroup).  Inside a loop over cout tiles a vector
// load would force s_waitcnt vmcnt(0), i.e. drain every store of the previous tile's epilogue (vmcnt is in-order and
// counts stores on CDNA4: 12k cycles per tile measured); those callers pass VBIAS=0 (wave-uniform scalar loads,
// selected per lane half) and preload the flow vectors of the lane's PT pixels (flpre, ST_OFFMASK).
struct EpiCtx {
    int H, W, cout, ncq, act, store, n_off_quads, dstH, dstW, lr;
    int xend;       // columns >= xend are not stored (W; the fused pair kernel's 62-column tiles end before their 64 computed columns)
    bool single;    // exactly one destination, starting at quad 0 and taking all of them
    long long* dbg; // diagnostic stamps (null in production)
    float slope, post;
    const float4* bp;
    const act_t* rp;
    const float* flp;
    bool f32dst;         // ST_Q4: the (single) destination is a float tensor also in the bf16 build (flow fields)
    unsigned* ovf;       // fp16-operand overflow word (null: not tracked)
    float* s3p;          // SRC_S3 image of the output (null: none)
    int s3_ngroups;      // cout / 8
    float* dp[CRFP_MAX_DST];
    long long dplane[CRFP_MAX_DST];
    int dpitch[CRFP_MAX_DST], dq0[CRFP_MAX_DST], dq1[CRFP_MAX_DST];
};

__device__ __forceinline__ EpiCtx epi_ctx(const ConvArgs& a, int n) {
    EpiCtx e;
    e.H = a.H; e.W = a.W; e.cout = a.cout; e.act = a.act; e.store = a.store;
    e.xend = a.W;
    e.ncq = (conv_packed_rows(a.cout, a.store, a.ps_r) + 3) >> 2;
    e.n_off_quads = a.n_off_quads; e.dstH = a.dstH; e.dstW = a.dstW;
    e.lr = a.ps_r == 4 ? 2 : 1;                           // ST_PS: r in {2, 4} (checked by the launcher)
    // NONE / RELU / LRELU(0.1) as max(v, slope*v) with slope 1 / 0 / 0.1: exact, branch-free, 2 VALU ops
    e.slope = a.act == CRFP_ACT_RELU ? 0.0f : (a.act == CRFP_ACT_LRELU01 ? 0.1f : 1.0f);
    e.post = a.post_scale;
    e.bp = reinterpret_cast<const float4*>(a.bpk);
    e.rp = a.resid ? as_act(a.resid) + (long long)n * a.resid_bstride : nullptr;
    e.f32dst = a.dst_f32 != 0;
    e.flp = a.flow + (long long)n * a.flow_bstride;
    e.s3p = a.s3_dst ? a.s3_dst + (long long)n * a.s3_bstride : nullptr;
    e.s3_ngroups = a.cout >> 3;
#pragma unroll
    for (int d = 0; d < CRFP_MAX_DST; ++d) {
        const bool on = d < a.ndst;
        // batch stride counts ELEMENTS of the destination's storage type (float for offsets / masks / flow / NCHW planes)
        const bool fdst = a.store == ST_OFFMASK || a.store == ST_NCHW || a.dst_f32;
        e.dp[d] = fdst ? a.dst[d].p + (long long)n * a.dst[d].bstride
                       : reinterpret_cast<float*>(as_act(a.dst[d].p) + (long long)n * a.dst[d].bstride);
        e.dpitch[d] = a.W + a.dst[d].pad;
        e.dplane[d] = (long long)(a.H + a.dst[d].pad) * e.dpitch[d] * 4;
        e.dq0[d] = on ? a.dst[d].q0 : 0;
        e.dq1[d] = on ? a.dst[d].q1 : 0;
    }
    e.single = a.ndst == 1 && a.dst[0].q0 == 0 && a.dst[0].q1 >= e.ncq;
    e.dbg = a.stamps;
    e.ovf = ovf_word(a.ovf, a.ovf_div, a.ovf_add, n, a.ovf_skip0);
    return e;
}

template <int CT, int PT, int RPW, int STORE, int VBIAS, bool SLOWACT>
__device__ __forceinline__ void conv_epilogue_t(const EpiCtx& e, f32x16 (&acc)[CT][PT], int T0, int tx0, int ty0,
                                                int wave, int j, int h, const float2* flpre) {
    const int H = e.H, W = e.W;
    float4 bb[CT][4];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int cq0 = (T0 + ct) * 8 + 2 * g;          // wave-uniform
            if (VBIAS == 2) {
                bb[ct][g] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            } else if (VBIAS == 1) {
                bb[ct][g] = e.bp[cq0 + h];
            } else {
                const float4 b0 = e.bp[cq0], b1 = e.bp[cq0 + 1];
                bb[ct][g] = h ? b1 : b0;
            }
        }

#ifdef CRFP_EPI_DBG
    long long d0 = __builtin_amdgcn_s_memtime(), d1 = 0;
#endif
    float vmax = 0.0f;   // largest |value| this lane stores (fp16-operand range guard, see ConvArgs::ovf)
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
#ifdef CRFP_EPI_DBG
        if (pt == 1) d1 = __builtin_amdgcn_s_memtime();
#endif
        const int y = ty0 + wave * RPW + (pt >> 1), x = tx0 + (pt & 1) * 32 + j;
        if (y >= H || x >= e.xend) continue;
        float2 fl = make_float2(0.0f, 0.0f);
        if (STORE == ST_OFFMASK) fl = flpre ? flpre[pt] : *reinterpret_cast<const float2*>(e.flp + ((long long)y * W + x) * 2);
        float4 rr[CT][4];
        if (e.rp) {   // wave-uniform; all residual quads of this pixel in flight together, before the first store
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int cq = min((T0 + ct) * 8 + 2 * g + h, e.ncq - 1);
                    { const cf32x4 rv = ldq(e.rp + (((long long)cq * H + y) * W + x) * 4); rr[ct][g] = make_float4(rv.x, rv.y, rv.z, rv.w); }
                }
        }
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int cq = (T0 + ct) * 8 + 2 * g + h;
                if (cq >= e.ncq) continue;
                const float4 b = bb[ct][g];
                float v[4] = {acc[ct][pt][4 * g + 0] + b.x, acc[ct][pt][4 * g + 1] + b.y,
                              acc[ct][pt][4 * g + 2] + b.z, acc[ct][pt][4 * g + 3] + b.w};
                if (STORE == ST_OFFMASK) {
                    if (cq < e.n_off_quads) {  // (dy,dx) pairs: 10*tanh(.) + flow flipped to (y,x)
                        v[0] = tanh10_plus(v[0], 10.0f + fl.y);
                        v[1] = tanh10_plus(v[1], 10.0f + fl.x);
                        v[2] = tanh10_plus(v[2], 10.0f + fl.y);
                        v[3] = tanh10_plus(v[3], 10.0f + fl.x);
                    } else {
#pragma unroll
                        for (int c = 0; c < 4; ++c) v[c] = fast_sigmoid(v[c]);
                    }
                } else if (SLOWACT) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) v[c] = apply_act(v[c], e.act) * e.post;
                } else {
#pragma unroll
                    for (int c = 0; c < 4; ++c) v[c] = fmaxf(v[c], e.slope * v[c]) * e.post;   // slope in {1, 0, 0.1}: none / relu / lrelu
                }
                if (STORE != ST_PS && (e.cout & 3)) {  // zero the padding components of a ragged last quad (wave-uniform test)
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        if (4 * cq + c >= e.cout) v[c] = 0.0f;
                }
                if (e.rp) { v[0] += rr[ct][g].x; v[1] += rr[ct][g].y; v[2] += rr[ct][g].z; v[3] += rr[ct][g].w; }
                if (STORE != ST_OFFMASK && STORE != ST_NCHW)   // those outputs never feed an fp16 operand
                    vmax = fmaxf(fmaxf(vmax, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
                // store addresses = per-lane pixel offset (one 32-bit multiply per pixel tile) + wave-uniform 64-bit
                // part per quad: the per-lane 64-bit multiplies of the first version were most of the epilogue's 6 k cycles
#ifndef CRFP_ACT_BF16
                if (STORE == ST_Q4 && e.s3p) {   // SRC_S3 image: this lane's 4 channels are half h of the 8-channel element
                    unsigned hi2[2], lo2[2];
                    split_f16x4(v, hi2, lo2);
                    // v_permlane32_swap: lanes 32-63 of the first operand trade places with lanes 0-31 of the second.  After
                    // it the lower lane (h = 0) holds the x0 words of all 8 channels and the upper lane (same pixel, h = 1)
                    // the x1s words: one 16-byte store per lane instead of two 8-byte ones.
                    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                    const u32x2 r0 = __builtin_amdgcn_permlane32_swap(hi2[0], lo2[0], false, false);
                    const u32x2 r1 = __builtin_amdgcn_permlane32_swap(hi2[1], lo2[1], false, false);
                    const int plane = (T0 + ct) * 4 + g + h * e.s3_ngroups;
                    *reinterpret_cast<uint4*>(e.s3p + (((long long)plane * H + y) * W + x) * 4) = make_uint4(r0.x, r1.x, r0.y, r1.y);
                }
#endif
                if (STORE == ST_OFFMASK || (STORE == ST_Q4 && kActBf16 && e.f32dst)) {   // float destination (offsets / masks / flow), single
                    const int cq0u = (T0 + ct) * 8 + 2 * g;
                    *reinterpret_cast<float4*>(e.dp[0] + (long long)cq0u * e.dplane[0] + h * e.dplane[0] + (y * e.dpitch[0] + x) * 4) =
                        make_float4(v[0], v[1], v[2], v[3]);
                } else if (STORE == ST_Q4) {
                    if (e.single) {   // one destination that takes every quad (wave-uniform)
                        const int cq0u = (T0 + ct) * 8 + 2 * g;
                        stq(as_act(e.dp[0]) + (long long)cq0u * e.dplane[0] + h * e.dplane[0] + (y * e.dpitch[0] + x) * 4,
                            cf32x4{v[0], v[1], v[2], v[3]});
                    } else {
#pragma unroll
                        for (int d = 0; d < CRFP_MAX_DST; ++d)
                            if (cq >= e.dq0[d] && cq < e.dq1[d])
                                stq(as_act(e.dp[d]) + (cq - e.dq0[d]) * e.dplane[d] + (y * e.dpitch[d] + x) * 4,
                                    cf32x4{v[0], v[1], v[2], v[3]});
                    }
                } else if (STORE == ST_PS) {
                    // cq0 is even and r >= 2: both lane halves share (Q, i) and differ by jj = +h
                    const int lr = e.lr, cq0u = (T0 + ct) * 8 + 2 * g;
                    const int Q = cq0u >> (2 * lr), sidx = cq0u & ((1 << (2 * lr)) - 1), i = sidx >> lr, jj = sidx & ((1 << lr) - 1);
                    stq(as_act(e.dp[0]) + ((long long)Q * e.dstH + i) * e.dstW * 4 + jj * 4 + (((y << lr) * e.dstW + (x << lr)) + h) * 4,
                        cf32x4{v[0], v[1], v[2], v[3]});
                } else {  // ST_NCHW
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const int ch = 4 * cq + c;
                        if (ch < e.cout) e.dp[0][((long long)ch * H + y) * W + x] = v[c];
                    }
                }
            }
    }
    // a stored value that the next split-fp16 conv could not represent (>= 65504, inf): raise the sticky word; the output
    // head then poisons the frame with NaN instead of returning plausible garbage
    if (e.ovf && !(vmax < 65504.0f)) atomicOr(e.ovf, 1u);
#ifdef CRFP_EPI_DBG
    if (e.dbg && threadIdx.x == 0) {
        long long* o = e.dbg + (16384 + (long long)blockIdx.x) * 8;
        o[0] = d0; o[1] = d1; o[2] = __builtin_amdgcn_s_memtime(); o[3] = 1;
    }
#endif
}

// e: epi_ctx() made once per workgroup (outside any loop over cout tiles)
template <int CT, int PT, int RPW, int VBIAS = 1>
__device__ __forceinline__ void conv_epilogue(const EpiCtx& e, f32x16 (&acc)[CT][PT], int T0, int tx0, int ty0,
                                              int wave, int j, int h, const float2* flpre = nullptr) {
    const bool slow = e.act == CRFP_ACT_TANH || e.act == CRFP_ACT_SIGMOID;   // API only; the engine never uses them here
    switch (e.store) {   // wave-uniform
        case ST_Q4:
            if (slow) conv_epilogue_t<CT, PT, RPW, ST_Q4, VBIAS, true>(e, acc, T0, tx0, ty0, wave, j, h, flpre);
            else conv_epilogue_t<CT, PT, RPW, ST_Q4, VBIAS, false>(e, acc, T0, tx0, ty0, wave, j, h, flpre);
            break;
        case ST_PS: conv_epilogue_t<CT, PT, RPW, ST_PS, VBIAS, false>(e, acc, T0, tx0, ty0, wave, j, h, flpre); break;
        case ST_OFFMASK: conv_epilogue_t<CT, PT, RPW, ST_OFFMASK, VBIAS, false>(e, acc, T0, tx0, ty0, wave, j, h, flpre); break;
        default:
            if (slow) conv_epilogue_t<CT, PT, RPW, ST_NCHW, VBIAS, true>(e, acc, T0, tx0, ty0, wave, j, h, flpre);
            else conv_epilogue_t<CT, PT, RPW, ST_NCHW, VBIAS, false>(e, acc, T0, tx0, ty0, wave, j, h, flpre);
            break;
    }
}

typedef float f32x4 __attribute__((ext_vector_type(4)));  // native vector: stays in VGPRs (HIP's float4 struct
                                                          // copies lowered to memcpy into a private alloca)
template <int CT, int NW>
__device__ __forceinline__ void load_weight_batch(f32x4 (&rw)[NW], const f32x4* __restrict__ wp, int T0, int npairs,
                                                  int pair, int tid) {
#pragma unroll
    for (int k = 0; k < NW; ++k) {
        const int idx = min(tid + 256 * k, CT * 576 - 1);  // clamp: always a valid load
        const int ct = idx / 576, rem = idx - ct * 576;
        rw[k] = wp[((long long)(T0 + ct) * npairs + pair) * 576 + rem];
    }
}

// CT = cout tiles (of 32) per workgroup, RPW = output rows per wave (tile = 4*RPW rows x 64 px).
// KP = K-chunks (of 8 channels) per staging phase: 1 everywhere except the long-K convolutions of SPyNet (9 x cin virtual channels,
// up to 72 chunks: on its coarse pyramid levels a conv is one or two workgroups walking that many barrier pairs, so two chunks per
// phase halve the fixed part; same chunk order, same accumulation order, same values).
template <int CT, int RPW, int KP = 1>
__global__ __launch_bounds__(256, 2) void conv3x3_mfma_kernel(const ConvArgs a) {
    constexpr int TH = 4 * RPW, LH = TH + 2, PT = 2 * RPW;
    // One K-chunk (2 quads = 8 input channels) lives in LDS at a time: the halo tile of both quads
    // and the packed weights of the chunk for the CT cout tiles.  The NEXT chunk's global loads are
    // issued into registers before the MFMAs of the current chunk start, so HBM/L2 latency hides
    // behind 144*CT MFMAs per wave; LDS is rewritten between two barriers.
    __shared__ float4 tile[2 * KP][LH][LW];
    __shared__ float4 wlds[KP][CT][9 * 64];
    constexpr int NIN = (LH * LW + 255) / 256;       // 3 halo elements per thread per quad
    constexpr int NW = (CT * 9 * 64 + 255) / 256;    // 3 (CT=1) or 5 (CT=2) weight float4 per thread
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int tiles_x = (a.W + TW - 1) / TW;
    // 1-D grid over (tile, cout-tile group) with the groups of a tile adjacent, dealt to the XCDs in contiguous bands:
    // neighbouring tiles share their halo lines and the cout-tile groups of one tile re-read the same input in one L2
    const int ngrp = a.ctiles / CT;
    const int bwork = xcd_band_tile(blockIdx.x, gridDim.x);
    const int btile = bwork / ngrp;
    const int tx0 = (btile % tiles_x) * TW, ty0 = (btile / tiles_x) * TH;
    const int T0 = (bwork - btile * ngrp) * CT;
    const int n = blockIdx.z;                                         // destination "batch item" (ksplit: one per K slice)
    const int n_src = a.ksplit > 0 ? n / a.ksplit : (a.src_bgroup > 0 ? n + n / a.src_bgroup : n);   // source batch item (ConvArgs::ksplit / src_bgroup)
    const int H = a.H, W = a.W;

    // per-thread staging coordinates (constant over the K loop)
    int sgy[NIN], sgx[NIN];
    bool sval[NIN];
#pragma unroll
    for (int k = 0; k < NIN; ++k) {
        const int idx = tid + 256 * k;
        const int r = idx / LW, c = idx - r * LW;
        sgy[k] = ty0 + r - 1;
        sgx[k] = tx0 + c - 1;
        sval[k] = idx < LH * LW && sgy[k] >= 0 && sgy[k] < H && sgx[k] >= 0 && sgx[k] < W;
    }

    f32x16 acc[CT][PT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int pt = 0; pt < PT; ++pt)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[ct][pt][e] = 0.0f;

    const int npairs = a.kq >> 1;
    const int npl = a.ksplit > 0 ? npairs / a.ksplit : npairs;        // K chunks this workgroup walks ...
    const int pair0 = a.ksplit > 0 ? (n % a.ksplit) * npl : 0;        // ... starting here
    const f32x4* __restrict__ wp = reinterpret_cast<const f32x4*>(a.wpk);
    float4 rin0[KP][NIN], rin1[KP][NIN];
    f32x4 rw[KP][NW];

    // loads of the KP chunks of staging phase ST (chunk index clamped past the end: an odd chunk count repeats the last one, unused)
#define CRFP_ISSUE_LOADS(ST)                                                                            \
    _Pragma("unroll") for (int kp = 0; kp < KP; ++kp) {                                                 \
        const int pair_ = pair0 + min(KP * (ST) + kp, npl - 1);                                         \
        {                                                                                               \
            int kql = 2 * pair_, s = 0;                                                                 \
            while (s < a.nsrc - 1 && kql >= a.src[s].nq) { kql -= a.src[s].nq; ++s; }                  \
            load_quad_batch<NIN>(rin0[kp], a.src[s], n_src, kql, sgy, sgx, sval, H, W);                 \
        }                                                                                               \
        {                                                                                               \
            int kql = 2 * pair_ + 1, s = 0;                                                             \
            while (s < a.nsrc - 1 && kql >= a.src[s].nq) { kql -= a.src[s].nq; ++s; }                  \
            load_quad_batch<NIN>(rin1[kp], a.src[s], n_src, kql, sgy, sgx, sval, H, W);                 \
        }                                                                                               \
        load_weight_batch<CT, NW>(rw[kp], wp, T0, npairs, pair_, tid);                                  \
    }

    const int nstages = (npl + KP - 1) / KP;
    CRFP_ISSUE_LOADS(0)
    for (int st = 0; st < nstages; ++st) {
        __syncthreads();  // every wave finished reading the previous chunk
#pragma unroll
        for (int kp = 0; kp < KP; ++kp) {
#pragma unroll
            for (int k = 0; k < NIN; ++k) {
                const int idx = tid + 256 * k;
                if (idx < LH * LW) {
                    (&tile[2 * kp][0][0])[idx] = rin0[kp][k];
                    (&tile[2 * kp + 1][0][0])[idx] = rin1[kp][k];
                }
            }
#pragma unroll
            for (int k = 0; k < NW; ++k) {
                const int idx = tid + 256 * k;
                if (idx < CT * 576) reinterpret_cast<f32x4*>(&wlds[kp][0][0])[idx] = rw[kp][k];
            }
        }
        __syncthreads();
        if (st + 1 < nstages) CRFP_ISSUE_LOADS(st + 1)
#pragma unroll
        for (int kp = 0; kp < KP; ++kp) {
        if (KP * st + kp >= npl) break;   // workgroup-uniform
#pragma unroll CRFP_TAP_UNROLL
        for (int tap = 0; tap < 9; ++tap) {
            const int ky = tap / 3, kx = tap - 3 * ky;
            float4 wa[CT];
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) wa[ct] = wlds[kp][ct][tap * 64 + lane];
            float4 b[PT];
#pragma unroll
            for (int pt = 0; pt < PT; ++pt) b[pt] = tile[2 * kp + h][wave * RPW + (pt >> 1) + ky][(pt & 1) * 32 + j + kx];
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int pt = 0; pt < PT; ++pt) {
                    if ((pt & 1) && tx0 + 32 >= W) continue;   // workgroup-uniform: the tile's right half lies outside a narrow map
                    acc[ct][pt] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[ct].x, b[pt].x, acc[ct][pt], 0, 0, 0);
                    acc[ct][pt] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[ct].y, b[pt].y, acc[ct][pt], 0, 0, 0);
                    acc[ct][pt] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[ct].z, b[pt].z, acc[ct][pt], 0, 0, 0);
                    acc[ct][pt] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[ct].w, b[pt].w, acc[ct][pt], 0, 0, 0);
                }
        }
        }
    }
#undef CRFP_ISSUE_LOADS

    const EpiCtx ec = epi_ctx(a, n);
    conv_epilogue<CT, PT, RPW>(ec, acc, T0, tx0, ty0, wave, j, h);
}

// ================================================================ split-bf16 ("bf16x6") main loop
// fp32-grade convolution on the bf16 MFMA (v_mfma_f32_32x32x16_bf16, 16x the fp32-MFMA rate):
// every fp32 operand x is split exactly into three bf16 terms x = x0 + x1 + x2 (x0 = bf16(x),
// x1 = bf16(x - x0), x2 = bf16(x - x0 - x1); the subtractions are exact in fp32) and the product is
// accumulated in fp32 as  x0w0 + x0w1 + x1w0 + x0w2 + x2w0 + x1w1 ; the dropped terms are <= 2^-24
// relative, i.e. below fp32 rounding (measured: max rel. error 1.9e-6 vs 3.1e-6 for a plain fp32
// matmul at K = 576).  6 bf16 MFMAs replace 8 fp32 MFMAs of 1/16 the rate: 2.67x the throughput of
// the fp32-MFMA path at the same accuracy.  Weights are split once at pack time; activations are
// split while the halo tile is staged (global fp32 -> registers -> 3 bf16 images in LDS).
// K-chunk = 16 channels = 4 quads (lane-half h supplies k = 8h..8h+7 = quads 2h, 2h+1 of the chunk).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#ifndef CRFP_ACT_BF16   // the split-operand schemes exist for fp32 activations only
__device__ __forceinline__ void split_bf16x8(const f32x4& lo, const f32x4& hi, bf16x8& p0, bf16x8& p1, bf16x8& p2) {
    const float x[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const __bf16 a = (__bf16)x[i];
        const float r = x[i] - (float)a;
        const __bf16 b = (__bf16)r;
        const float r2 = r - (float)b;
        p0[i] = a;
        p1[i] = b;
        p2[i] = (__bf16)r2;
    }
}

// ---- split-fp16 ("f16x3"): x = x0 + 2^-11 * x1s with x0 = fp16(x), x1s = fp16((x - x0) * 2^11) (the subtraction and the
// scaling are exact; the scale keeps the residual out of fp16's subnormal range, where the MFMA would flush it).
// Product = x0w0 + 2^-11 * (x0w1s + x1sw0): two fp32 accumulators (hi, lo), THREE MFMAs per 16-channel tap instead of
// six, two LDS images per operand instead of three, and the dropped term x1w1 is <= 2^-22 relative.  CPU emulation
// of the whole 7-frame clip: max |err| 2.2e-6 vs 2.5e-6 for bf16x6 (both at the fp32 summation-order noise floor);
// without the scaling 6.7e-6 if subnormals survive and 1.0e-3 if they are flushed.  Needs |x| < 65504.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
constexpr float F16_RES_SCALE = 2048.0f;

__device__ __forceinline__ void split_f16x8(const f32x4& lo, const f32x4& hi, bf16x8& p0, bf16x8& p1) {
    const float x[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    f16x8 a, b;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const _Float16 h = (_Float16)x[i];
        a[i] = h;
        b[i] = (_Float16)((x[i] - (float)h) * F16_RES_SCALE);
    }
    p0 = __builtin_bit_cast(bf16x8, a);   // LDS / register containers are typed bf16x8 for both schemes
    p1 = __builtin_bit_cast(bf16x8, b);
}

// Same values as split_f16x8 for a halo pixel whose 8 channels all exist (component masks = 15): pixel validity is
// folded into the residual scale (sc = valid ? 2^11 : 0) and an AND on the packed hi words (km = valid ? ~0 : 0), the
// hi parts come out of one v_cvt_pk_f16_f32 per pair and x - x0 out of one v_fma_mix_f32: 3.5 VALU ops per value
// instead of 8.5 (the masked form spends 3 on and/cmp/cndmask per value and converts the hi part twice).
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void split_f16x8_fast(const f32x4& lo, const f32x4& hi, float sc, unsigned km, bf16x8& p0, bf16x8& p1) {
    const float x[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    u32x4 a, b;
    float negone = -1.0f;
    asm("" : "+v"(negone));   // opaque, or the fma is canonicalised to fpext + fsub (two instructions)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const f16x2 h2 = __builtin_convertvector(f32x2{x[2 * i], x[2 * i + 1]}, f16x2);
        const float r0 = __builtin_fmaf((float)h2[0], negone, x[2 * i]);       // = x - x0, exact (v_fma_mix_f32)
        const float r1 = __builtin_fmaf((float)h2[1], negone, x[2 * i + 1]);
        const f16x2 l2 = __builtin_convertvector(f32x2{r0 * sc, r1 * sc}, f16x2);
        a[i] = __builtin_bit_cast(unsigned, h2) & km;
        b[i] = __builtin_bit_cast(unsigned, l2);
    }
    p0 = __builtin_bit_cast(bf16x8, a);
    p1 = __builtin_bit_cast(bf16x8, b);
}

// NP = 3: bf16x6, NP = 2: f16x3.  p[] receives the NP images of 8 channels.
template <int NP>
__device__ __forceinline__ void split_parts(const f32x4& lo, const f32x4& hi, bf16x8 (&p)[NP]);
template <>
__device__ __forceinline__ void split_parts<3>(const f32x4& lo, const f32x4& hi, bf16x8 (&p)[3]) { split_bf16x8(lo, hi, p[0], p[1], p[2]); }
template <>
__device__ __forceinline__ void split_parts<2>(const f32x4& lo, const f32x4& hi, bf16x8 (&p)[2]) { split_f16x8(lo, hi, p[0], p[1]); }

// all products of one (tap, 32-pixel tile): w[] / b[] are the NP images of the A / B fragments
template <int NP>
__device__ __forceinline__ void split_mfma(f32x16& hi, f32x16& lo, const bf16x8 (&w)[NP], const bf16x8 (&b)[NP]) {
    if (NP == 3) {
        f32x16 c = hi;
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[1], b[1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[0], b[NP - 1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[NP - 1], b[0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[0], b[1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[1], b[0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[0], b[0], c, 0, 0, 0);
        hi = c;
    } else {
        const f16x8 w0 = __builtin_bit_cast(f16x8, w[0]), w1 = __builtin_bit_cast(f16x8, w[1]);
        const f16x8 b0 = __builtin_bit_cast(f16x8, b[0]), b1 = __builtin_bit_cast(f16x8, b[1]);
        lo = __builtin_amdgcn_mfma_f32_32x32x16_f16(w0, b1, lo, 0, 0, 0);
        hi = __builtin_amdgcn_mfma_f32_32x32x16_f16(w0, b0, hi, 0, 0, 0);
        lo = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1, b0, lo, 0, 0, 0);
    }
}

__device__ __forceinline__ f32x4 mask_quad(const f32x4& v, int m) {
    return f32x4{(m & 1) ? v.x : 0.0f, (m & 2) ? v.y : 0.0f, (m & 4) ? v.z : 0.0f, (m & 8) ? v.w : 0.0f};
}

// one K-quad of NIN halo elements into registers (native vectors, switch hoisted out of the loop)
template <int NIN>
__device__ __forceinline__ void load_quad_batch_v(f32x4 (&r)[NIN], const ConvSrc& s, int n, int kql,
                                                  const int (&gy)[NIN], const int (&gx)[NIN], const bool (&ok)[NIN],
                                                  int H, int W) {
    const float* base = s.p + (long long)n * s.bstride;
#pragma unroll
    for (int k = 0; k < NIN; ++k) r[k] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    switch (s.kind) {
        case SRC_Q4: {
            const int PW = W + s.pad;
            const float* b = base + (long long)kql * (H + s.pad) * PW * 4;
#pragma unroll
            for (int k = 0; k < NIN; ++k)
                if (ok[k]) r[k] = *reinterpret_cast<const f32x4*>(b + ((long long)gy[k] * PW + gx[k]) * 4);
            break;
        }
        case SRC_NCHW: {
#pragma unroll
            for (int k = 0; k < NIN; ++k)
                if (ok[k]) {
                    const long long o = (long long)gy[k] * W + gx[k], pl = (long long)H * W;
                    const int c0 = 4 * kql;
                    r[k].x = c0 + 0 < s.nch ? base[(c0 + 0) * pl + o] : 0.0f;
                    r[k].y = c0 + 1 < s.nch ? base[(c0 + 1) * pl + o] : 0.0f;
                    r[k].z = c0 + 2 < s.nch ? base[(c0 + 2) * pl + o] : 0.0f;
                    r[k].w = c0 + 3 < s.nch ? base[(c0 + 3) * pl + o] : 0.0f;
                }
            break;
        }
        case SRC_UNSHUF4: {
            const int Qp = kql >> 4, ij = kql & 15, i = ij >> 2, jj = ij & 3;
            const int H4 = 4 * H + s.pad, W4 = 4 * W + s.pad;
            const float* b = base + (long long)Qp * H4 * W4 * 4;
#pragma unroll
            for (int k = 0; k < NIN; ++k)
                if (ok[k]) r[k] = *reinterpret_cast<const f32x4*>(b + ((long long)(4 * gy[k] + i) * W4 + 4 * gx[k] + jj) * 4);
            break;
        }
        case SRC_FLOW2: {
#pragma unroll
            for (int k = 0; k < NIN; ++k)
                if (ok[k]) {
                    const float2 f = *reinterpret_cast<const float2*>(base + ((long long)gy[k] * W + gx[k]) * 2);
                    r[k] = f32x4{f.x, f.y, 0.0f, 0.0f};
                }
            break;
        }
        default: break;
    }
}

// The kernel's body: workgroup wg_x of wg_nx (the launch's work items: (tile, cout-tile group)) of batch item n of plan a.  A function so that
// conv3x3_split_dual_kernel (below) can run TWO plans in one launch.
// KS (round 6, FNet's small maps): blockIdx.z enumerates (batch item, K slice) -- slice z % a.ksplit walks chunks [slice * nchunks / ksplit, ...) of
// source item z / ksplit and stores its raw partial sums as "item" z of a float Q4 destination (the bias rides in slice 0; the launcher passes
// act NONE / no residual / no guard); launch_ksplit_reduce or the pool / resize pass behind the layer adds the slices in order.
template <int CT, int RPW, int NP, bool KS = false>
__device__ __forceinline__ void conv3x3_split_body(const ConvArgs& a, const int wg_x, const int wg_nx, const int n) {
#ifdef CRFP_LAB
    const long long t_entry = __builtin_amdgcn_s_memtime();
#endif
    constexpr int TH = 4 * RPW, LH = TH + 2, PT = 2 * RPW;
    constexpr int NEL = LH * LW;                 // halo pixels
    constexpr int NIN = (NEL + 255) / 256;       // halo pixels per thread; each carries the chunk's 4 quads
    constexpr int WPC = 9 * NP * 64;             // weight vectors per (cout tile, chunk)
    __shared__ bf16x8 tile[NP][2][NEL];          // [split part][quad pair][halo pixel], 16 B each
    __shared__ bf16x8 wlds[CT][WPC];             // [cout tile][(tap, part)][lane]: this chunk's A fragments
    constexpr int NWS = (CT * WPC + 255) / 256;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int tiles_x = (a.W + TW - 1) / TW;
    // 1-D grid over (tile, cout-tile group) with the groups of a tile adjacent, dealt to the XCDs in contiguous bands:
    // neighbouring tiles share their halo lines and the cout-tile groups of one tile re-read the same input in one L2
    const int ngrp = a.ctiles / CT;
    const int bwork = xcd_band_tile(wg_x, wg_nx);
    const int btile = bwork / ngrp;
    const int tx0 = (btile % tiles_x) * TW, ty0 = (btile / tiles_x) * TH;
    const int T0 = (bwork - btile * ngrp) * CT;
    const int kslices = KS ? a.ksplit : 1, kslice = KS ? n % kslices : 0, item = KS ? n / kslices : n;
    const int ns = a.src_bgroup > 0 ? item + item / a.src_bgroup : item;   // source batch item (ConvArgs::src_bgroup)
    const int H = a.H, W = a.W;

    // halo pixels of this thread: clamped coordinates (every load is unconditional and in range; pixels
    // outside the image are zeroed when the registers are converted, not when they are loaded)
    int cgy[NIN], cgx[NIN];
    bool sval[NIN];
#pragma unroll
    for (int t = 0; t < NIN; ++t) {
        const int idx = min(tid + 256 * t, NEL - 1);
        const int r = idx / LW, c = idx - r * LW;
        const int gy = ty0 + r - 1, gx = tx0 + c - 1;
        sval[t] = tid + 256 * t < NEL && gy >= 0 && gy < H && gx >= 0 && gx < W;
        cgy[t] = min(max(gy, 0), H - 1);
        cgx[t] = min(max(gx, 0), W - 1);
    }

    f32x16 acc[CT][PT], acl[CT][PT];   // acl: the 2^-11-scaled partial sums of the f16x3 scheme (dead for bf16x6)
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int pt = 0; pt < PT; ++pt)
#pragma unroll
            for (int e = 0; e < 16; ++e) { acc[ct][pt][e] = 0.0f; acl[ct][pt][e] = 0.0f; }
    if (!KS || kslice == 0) {   // accumulators start at the bias: its loads are the first of the kernel and long landed when the MFMAs begin
        const float4* __restrict__ bp = reinterpret_cast<const float4*>(a.bpk);
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 bq = bp[(T0 + ct) * 8 + 2 * g + h];
#pragma unroll
                for (int pt = 0; pt < PT; ++pt) {
                    acc[ct][pt][4 * g + 0] = bq.x; acc[ct][pt][4 * g + 1] = bq.y;
                    acc[ct][pt][4 * g + 2] = bq.z; acc[ct][pt][4 * g + 3] = bq.w;
                }
            }
    }

    const int nchunks = a.kq >> 2;
    const bf16x8* __restrict__ wp = reinterpret_cast<const bf16x8*>(NP == 3 ? a.wsplit : a.wsplit16);
    f32x4 rq0[NIN], rq1[NIN], rq2[NIN], rq3[NIN];   // the chunk's 4 quads of each of this thread's halo pixels
    bf16x8 rws[NWS];   // weights ride in the same prefetch batch: no global load (vmcnt is in-order!) may sit
                       // inside the MFMA loop, or every tap would wait for the whole input prefetch to land

    // Branch-free issue: each K-quad of the chunk is described by wave-uniform (base, row stride, column
    // stride, component mask); all source kinds share one load form.  Nothing between issue and the LDS
    // write touches a loaded register (a use, a zero-init or a copy forces s_waitcnt and serialises the
    // prefetch behind memory latency -- measured: 5.7k cycles per issue with conditional loads).
    const float* qb0; const float* qb1; const float* qb2; const float* qb3;
    int qrs0, qrs1, qrs2, qrs3, qcs0, qcs1, qcs2, qcs3, qm0 = 0, qm1 = 0, qm2 = 0, qm3 = 0;
#define CRFP_QDESC(QB, QRS, QCS, QM, QI, CH)                                                              \
    {                                                                                                     \
        const QuadDesc d_ = a.qd[4 * (CH) + (QI)];                                                        \
        QB = d_.base + (long long)ns * d_.bstride; QRS = d_.rs; QCS = d_.cs; QM = d_.mask;                \
    }
#define CRFP_SPLIT_ISSUE(CH)                                                                              \
    {                                                                                                     \
        CRFP_QDESC(qb0, qrs0, qcs0, qm0, 0, CH) CRFP_QDESC(qb1, qrs1, qcs1, qm1, 1, CH)                   \
        CRFP_QDESC(qb2, qrs2, qcs2, qm2, 2, CH) CRFP_QDESC(qb3, qrs3, qcs3, qm3, 3, CH)                   \
        _Pragma("unroll") for (int t = 0; t < NIN; ++t) {                                                 \
            rq0[t] = CRFP_LDACT(f32x4, qb0 + cgy[t] * qrs0 + cgx[t] * qcs0);                \
            rq1[t] = CRFP_LDACT(f32x4, qb1 + cgy[t] * qrs1 + cgx[t] * qcs1);                \
            rq2[t] = CRFP_LDACT(f32x4, qb2 + cgy[t] * qrs2 + cgx[t] * qcs2);                \
            rq3[t] = CRFP_LDACT(f32x4, qb3 + cgy[t] * qrs3 + cgx[t] * qcs3);                \
        }                                                                                                 \
        _Pragma("unroll") for (int k = 0; k < NWS; ++k) {                                                 \
            const int idx = min(tid + 256 * k, CT * WPC - 1);                                             \
            const int ct = idx / WPC, rem = idx - ct * WPC;                                               \
            rws[k] = wp[((long long)(T0 + ct) * nchunks + (CH)) * WPC + rem];                             \
        }                                                                                                 \
    }

#ifdef CRFP_LAB
    long long tA = 0, tB = 0, tC = 0, tD = 0, t0 = __builtin_amdgcn_s_memtime();
#endif
    const int ch_begin = KS ? kslice * (nchunks / kslices) : 0, ch_end = KS ? ch_begin + nchunks / kslices : nchunks;
    CRFP_SPLIT_ISSUE(ch_begin)
    for (int ch = ch_begin; ch < ch_end; ++ch) {
        const int m0 = qm0, m1 = qm1, m2 = qm2, m3 = qm3;  // component masks of the chunk now in registers
        __syncthreads();  // every wave finished reading the previous chunk
#ifdef CRFP_LAB
        if (a.stamps) {
            long long t = __builtin_amdgcn_s_memtime(); tA += t - t0; t0 = t;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            t = __builtin_amdgcn_s_memtime(); tC += t - t0; t0 = t;   // diagnostic: load-landing wait booked under C
        }
#endif
        if (NP == 2 && (m0 & 16)) {   // SRC_S3 chunk (wave-uniform): the producer already split it -- copy, zero outside the image
#pragma unroll
            for (int t = 0; t < NIN; ++t) {
                const int idx = tid + 256 * t;
                if (idx < NEL) {
                    const f32x4 z = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
                    tile[0][0][idx] = __builtin_bit_cast(bf16x8, sval[t] ? rq0[t] : z);        // (x0,  channels 0..7 of the chunk)
                    tile[0][1][idx] = __builtin_bit_cast(bf16x8, sval[t] ? rq1[t] : z);        // (x0,  channels 8..15)
                    tile[NP - 1][0][idx] = __builtin_bit_cast(bf16x8, sval[t] ? rq2[t] : z);   // (x1s, channels 0..7)
                    tile[NP - 1][1][idx] = __builtin_bit_cast(bf16x8, sval[t] ? rq3[t] : z);   // (x1s, channels 8..15)
                }
            }
        } else if (NP == 2 && (m0 & m1 & m2 & m3) == 15) {   // wave-uniform: every chunk but a ragged last one
#pragma unroll
            for (int t = 0; t < NIN; ++t) {
                const int idx = tid + 256 * t;
                if (idx < NEL) {
                    const float sc = sval[t] ? F16_RES_SCALE : 0.0f;
                    const unsigned km = sval[t] ? 0xffffffffu : 0u;
                    bf16x8 pa, pb;
                    split_f16x8_fast(rq0[t], rq1[t], sc, km, pa, pb);
                    tile[0][0][idx] = pa; tile[NP - 1][0][idx] = pb;
                    split_f16x8_fast(rq2[t], rq3[t], sc, km, pa, pb);
                    tile[0][1][idx] = pa; tile[NP - 1][1][idx] = pb;
                }
            }
        } else {
#pragma unroll
            for (int t = 0; t < NIN; ++t) {
                const int idx = tid + 256 * t;
                if (idx < NEL) {
                    bf16x8 pp[NP];
                    split_parts<NP>(mask_quad(rq0[t], sval[t] ? m0 : 0), mask_quad(rq1[t], sval[t] ? m1 : 0), pp);
#pragma unroll
                    for (int p = 0; p < NP; ++p) tile[p][0][idx] = pp[p];
                    split_parts<NP>(mask_quad(rq2[t], sval[t] ? m2 : 0), mask_quad(rq3[t], sval[t] ? m3 : 0), pp);
#pragma unroll
                    for (int p = 0; p < NP; ++p) tile[p][1][idx] = pp[p];
                }
            }
        }
#pragma unroll
        for (int k = 0; k < NWS; ++k) {
            const int idx = tid + 256 * k;
            if (idx < CT * WPC) (&wlds[0][0])[idx] = rws[k];
        }
        __syncthreads();
#ifdef CRFP_LAB
        if (a.stamps) { const long long t = __builtin_amdgcn_s_memtime(); tB += t - t0; t0 = t; }
#endif
        if (ch + 1 < ch_end) CRFP_SPLIT_ISSUE(ch + 1)
#ifdef CRFP_LAB
        if (a.stamps) { const long long t = __builtin_amdgcn_s_memtime(); tD += t - t0; t0 = t; }  // diagnostic: issue booked under D
#endif
#pragma unroll CRFP_SPLIT_TAP_UNROLL
        for (int tap = 0; tap < 9; ++tap) {
            const int ky = tap / 3, kx = tap - 3 * ky;
            bf16x8 wa[CT][NP];
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int p = 0; p < NP; ++p) wa[ct][p] = wlds[ct][(tap * NP + p) * 64 + lane];
#pragma unroll
            for (int pt = 0; pt < PT; ++pt) {
                const int pix = (wave * RPW + (pt >> 1) + ky) * LW + (pt & 1) * 32 + j + kx;
                bf16x8 bq[NP];
#pragma unroll
                for (int p = 0; p < NP; ++p) bq[p] = tile[p][h][pix];
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) split_mfma<NP>(acc[ct][pt], acl[ct][pt], wa[ct], bq);
            }
        }
#ifdef CRFP_LAB
        if (a.stamps) { const long long t = __builtin_amdgcn_s_memtime(); tD += t - t0; t0 = t; }
#endif
    }
#undef CRFP_SPLIT_ISSUE
#undef CRFP_QDESC
#ifdef CRFP_LAB
    if (a.stamps) {
        const long long t = __builtin_amdgcn_s_memtime(); tD += t - t0; t0 = t;
        if (tid == 0) {
            long long* o = a.stamps + ((long long)blockIdx.y * gridDim.x + blockIdx.x) * 8;
            o[0] = tA; o[1] = tB; o[2] = tC; o[3] = tD; o[4] = t_entry; o[5] = t;
        }
    }
#endif
    if (NP == 2) {
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int pt = 0; pt < PT; ++pt)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[ct][pt][e] += acl[ct][pt][e] * (1.0f / F16_RES_SCALE);
    }
#ifdef CRFP_LAB
    long long te1 = 0, te2 = 0;
    if (a.stamps) { asm volatile("s_nop 0" ::"v"(acc[0][0][0]), "v"(acc[0][PT - 1][15])); te1 = __builtin_amdgcn_s_memtime(); }
#endif
    const EpiCtx ec = epi_ctx(a, n);
#ifdef CRFP_LAB
    if (a.stamps) { asm volatile("s_waitcnt lgkmcnt(0)" ::"s"(ec.dpitch[0]), "s"(ec.ncq), "s"(ec.slope)); te2 = __builtin_amdgcn_s_memtime(); }
#endif
    conv_epilogue<CT, PT, RPW, 2>(ec, acc, T0, tx0, ty0, wave, j, h);
#ifdef CRFP_LAB
    if (a.stamps) {
        const long long ti = __builtin_amdgcn_s_memtime();      // epilogue issued (stores in flight)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (tid == 0) {
            a.stamps[((long long)blockIdx.y * gridDim.x + blockIdx.x) * 8 + 2] = te1;   // overwrites phase C (unused here)
            a.stamps[((long long)blockIdx.y * gridDim.x + blockIdx.x) * 8 + 3] = te2;   // overwrites phase D
            a.stamps[((long long)blockIdx.y * gridDim.x + blockIdx.x) * 8 + 7] = ti;
            a.stamps[((long long)blockIdx.y * gridDim.x + blockIdx.x) * 8 + 6] = __builtin_amdgcn_s_memtime();
        }
    }
#endif
}

template <int CT, int RPW, int NP>
__global__ __launch_bounds__(256, NP == 2 && CT == 1 && RPW == 1 ? 3 : 2) void conv3x3_split_kernel(const ConvArgs a) {
    conv3x3_split_body<CT, RPW, NP>(a, (int)blockIdx.x, (int)gridDim.x, (int)blockIdx.z);
}
#ifndef CRFP_ACT_BF16
__global__ __launch_bounds__(256, 3) void conv3x3_split_ks_kernel(const ConvArgs a) {
    conv3x3_split_body<1, 1, 2, true>(a, (int)blockIdx.x, (int)gridDim.x, (int)blockIdx.z);
}
#endif

#ifndef CRFP_ACT_BF16
// Round 6: TWO independent convs of the same shape in ONE launch (workgroups [0, w0) run plan a0, the rest plan a1; w0 is a multiple of 8, so a
// workgroup keeps its XCD).  The engine's two 32 -> 64 pixel-shuffle convs behind level 2 (the x4 upsample of the propagated features, dcn_3's
// pre-offset conv) are 1 800 four-wave workgroups each on the chip's 768 slots: 2.34 rounds paid as 3, twice; together 4.69 paid as 5
// (a lock-step batch of 4 clips shows the same arithmetic: 41.0 -> 34.7 us per clip).  Two streams cost more in fork / join than they save
// (profiles/r06_aux_stream_ab.txt).  Each plan's arithmetic is untouched: bit-identical.
template <int CT, int RPW, int NP>
__global__ __launch_bounds__(256, NP == 2 && CT == 1 && RPW == 1 ? 3 : 2) void conv3x3_split_dual_kernel(const ConvArgs a0, const ConvArgs a1, const int w0) {
    const bool second = (int)blockIdx.x >= w0;   // workgroup-uniform
    if (second) conv3x3_split_body<CT, RPW, NP>(a1, (int)blockIdx.x - w0, (int)gridDim.x - w0, (int)blockIdx.z);
    else conv3x3_split_body<CT, RPW, NP>(a0, (int)blockIdx.x, w0, (int)blockIdx.z);
}
#endif

// ---------------------------------------------------------------- f16x3, 8 waves, ONE accumulator ("f16x3s")
// The 4-wave kernel above needs 164 VGPRs (two fp32 accumulators per pixel tile, hi and lo) = 3 waves per SIMD = 768 workgroup
// slots, and a 360 x 640 map is 900 tiles: every 32-cout conv ran a full round plus a 17 % one.  Here the
// whole sum is kept scaled by 2^11 in ONE accumulator:
//     2^11 * x*w  ~=  x0 * A  +  x0 * B  +  x1s * C,    A = fp16(2^11 w),  B = fp16(2^11 (w - A/2^11)),  C = fp16(w)
// (x0 = fp16(x), x1s = fp16(2^11 (x - x0)) as before: same three MFMAs per tap, same exact products, same dropped term;
// the bias enters as 2^11 b and the result leaves through one multiply by 2^-11).  Three weight images instead of two,
// 32 VGPRs less; with 8 waves per workgroup (8 rows x 64 pixels, one weight image per 8 rows instead of per 4) the kernel
// fits 128 VGPRs: 2 workgroups = 16 waves per CU, 450 tiles on 512 slots -- one round.  Needs |w| < 32 (A must stay finite).
constexpr int S8_TH = 8;
constexpr int S8_WPC = 9 * 3 * 64;

// NW = 8: two workgroups per CU (the shipped form).  NW = 16 (A/B builds, -DCRFP_S8_NW=16): ONE 16-row workgroup per CU -- the 27.6 KB
// weight stage is loaded once per CU instead of twice and the halo is 18 / 16 instead of 10 / 8 rows: 104 instead of 139 KB of ingest
// per CU and chunk (profiles/r03_conv_split8_timeline.txt: the prologue is ingest-bound).
template <int NW>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(4, 4))) void conv3x3_split8_kernel(const ConvArgs a) {
    constexpr int S8_NT = 64 * NW, S8_LH = NW + 2, S8_NEL = S8_LH * LW, S8_NIN = (S8_NEL + S8_NT - 1) / S8_NT;
    constexpr int S8_NWS = (S8_WPC + S8_NT - 1) / S8_NT;
    __shared__ bf16x8 tile[2][2][S8_NEL];        // [split part][quad pair][halo pixel]   42.2 KB (NW = 8)
    __shared__ bf16x8 wlds[S8_WPC];              // [(tap, image A/B/C)][lane]            27.6 KB
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#ifdef CRFP_PRIO47   // A/B builds: static priority for the second-dispatched half of the workgroup (MI355X_MICROARCH.md, two waves per SIMD, item 4)
    if (wave >= 4) __builtin_amdgcn_s_setprio(CRFP_PRIO47);
#endif
    const int j = lane & 31, h = lane >> 5;
    const int tiles_x = (a.W + TW - 1) / TW;
    const int ngrp = a.ctiles;
    const int bwork = xcd_band_tile(blockIdx.x, gridDim.x);
    const int btile = bwork / ngrp;
    const int tx0 = (btile % tiles_x) * TW, ty0 = (btile / tiles_x) * NW;
    const int T0 = bwork - btile * ngrp;
    const int n = blockIdx.z;
    const int ns = a.src_bgroup > 0 ? n + n / a.src_bgroup : n;   // source batch item (ConvArgs::src_bgroup)
    const int H = a.H, W = a.W;
#ifdef CRFP_LAB
    // diagnostic timeline (CRFP_STAMP_PTR / CRFP_STAMP_NAME, tools/stamp_conv8.py): absolute s_memtime of wave 0 at kernel entry,
    // first tile in LDS, end of chunk 0's MFMAs, end of the MFMA loop, epilogue stores issued, stores acknowledged
    long long ts[6] = {0, 0, 0, 0, 0, 0};
    if (a.stamps) ts[0] = __builtin_amdgcn_s_memtime();
#define S8_STAMP(I) if (a.stamps) ts[I] = __builtin_amdgcn_s_memtime();
#else
#define S8_STAMP(I)
#endif

    int cgy[S8_NIN], cgx[S8_NIN];
    bool sval[S8_NIN];
#pragma unroll
    for (int t = 0; t < S8_NIN; ++t) {
        const int idx = min(tid + S8_NT * t, S8_NEL - 1);
        const int r = idx / LW, c = idx - r * LW;
        const int gy = ty0 + r - 1, gx = tx0 + c - 1;
        sval[t] = tid + S8_NT * t < S8_NEL && gy >= 0 && gy < H && gx >= 0 && gx < W;
        cgy[t] = min(max(gy, 0), H - 1);
        cgx[t] = min(max(gx, 0), W - 1);
    }

    f32x16 acc[1][2];
    {   // accumulators start at 2^11 * bias
        const float4* __restrict__ bp = reinterpret_cast<const float4*>(a.bpk);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 bq = bp[T0 * 8 + 2 * g + h];
#pragma unroll
            for (int pt = 0; pt < 2; ++pt) {
                acc[0][pt][4 * g + 0] = bq.x * F16_RES_SCALE; acc[0][pt][4 * g + 1] = bq.y * F16_RES_SCALE;
                acc[0][pt][4 * g + 2] = bq.z * F16_RES_SCALE; acc[0][pt][4 * g + 3] = bq.w * F16_RES_SCALE;
            }
        }
    }

    const int nchunks = a.kq >> 2;
    const bf16x8* __restrict__ wp = reinterpret_cast<const bf16x8*>(a.wsplit_sa);
    f32x4 rq0[S8_NIN], rq1[S8_NIN], rq2[S8_NIN], rq3[S8_NIN];
    bf16x8 rws[S8_NWS];
    const float* qb0; const float* qb1; const float* qb2; const float* qb3;
    int qrs0, qrs1, qrs2, qrs3, qcs0, qcs1, qcs2, qcs3, qm0 = 0, qm1 = 0, qm2 = 0, qm3 = 0;
#define CRFP_QDESC(QB_, QRS, QCS, QM, QI, CH)                                                             \
    {                                                                                                     \
        const QuadDesc d_ = a.qd[4 * (CH) + (QI)];                                                        \
        QB_ = d_.base + (long long)ns * d_.bstride; QRS = d_.rs; QCS = d_.cs; QM = d_.mask;                \
    }
#define CRFP_S8_ISSUE(CH)                                                                                 \
    {                                                                                                     \
        CRFP_QDESC(qb0, qrs0, qcs0, qm0, 0, CH) CRFP_QDESC(qb1, qrs1, qcs1, qm1, 1, CH)                   \
        CRFP_QDESC(qb2, qrs2, qcs2, qm2, 2, CH) CRFP_QDESC(qb3, qrs3, qcs3, qm3, 3, CH)                   \
        _Pragma("unroll") for (int t = 0; t < S8_NIN; ++t) {                                              \
            rq0[t] = CRFP_LDACT(f32x4, qb0 + cgy[t] * qrs0 + cgx[t] * qcs0);                \
            rq1[t] = CRFP_LDACT(f32x4, qb1 + cgy[t] * qrs1 + cgx[t] * qcs1);                \
            rq2[t] = CRFP_LDACT(f32x4, qb2 + cgy[t] * qrs2 + cgx[t] * qcs2);                \
            rq3[t] = CRFP_LDACT(f32x4, qb3 + cgy[t] * qrs3 + cgx[t] * qcs3);                \
        }                                                                                                 \
        _Pragma("unroll") for (int k = 0; k < S8_NWS; ++k) {                                              \
            const int idx = min(tid + S8_NT * k, S8_WPC - 1);                                             \
            rws[k] = wp[((long long)T0 * nchunks + (CH)) * S8_WPC + idx];                                 \
        }                                                                                                 \
    }

    CRFP_S8_ISSUE(0)
    for (int ch = 0; ch < nchunks; ++ch) {
        const int m0 = qm0, m1 = qm1, m2 = qm2, m3 = qm3;
        __syncthreads();
        if (m0 & 16) {   // SRC_S3 chunk (wave-uniform): already split by its producer -- copy, zero outside the image
#pragma unroll
            for (int t = 0; t < S8_NIN; ++t) {
                const int idx = tid + S8_NT * t;
                if (idx < S8_NEL) {
                    const f32x4 z = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
                    tile[0][0][idx] = __builtin_bit_cast(bf16x8, sval[t] ? rq0[t] : z);
                    tile[0][1][idx] = __builtin_bit_cast(bf16x8, sval[t] ? rq1[t] : z);
                    tile[1][0][idx] = __builtin_bit_cast(bf16x8, sval[t] ? rq2[t] : z);
                    tile[1][1][idx] = __builtin_bit_cast(bf16x8, sval[t] ? rq3[t] : z);
                }
            }
        } else if ((m0 & m1 & m2 & m3) == 15) {
#pragma unroll
            for (int t = 0; t < S8_NIN; ++t) {
                const int idx = tid + S8_NT * t;
                if (idx < S8_NEL) {
                    const float sc = sval[t] ? F16_RES_SCALE : 0.0f;
                    const unsigned km = sval[t] ? 0xffffffffu : 0u;
                    bf16x8 pa, pb;
                    split_f16x8_fast(rq0[t], rq1[t], sc, km, pa, pb);
                    tile[0][0][idx] = pa; tile[1][0][idx] = pb;
                    split_f16x8_fast(rq2[t], rq3[t], sc, km, pa, pb);
                    tile[0][1][idx] = pa; tile[1][1][idx] = pb;
                }
            }
        } else {
#pragma unroll
            for (int t = 0; t < S8_NIN; ++t) {
                const int idx = tid + S8_NT * t;
                if (idx < S8_NEL) {
                    bf16x8 pp[2];
                    split_parts<2>(mask_quad(rq0[t], sval[t] ? m0 : 0), mask_quad(rq1[t], sval[t] ? m1 : 0), pp);
                    tile[0][0][idx] = pp[0]; tile[1][0][idx] = pp[1];
                    split_parts<2>(mask_quad(rq2[t], sval[t] ? m2 : 0), mask_quad(rq3[t], sval[t] ? m3 : 0), pp);
                    tile[0][1][idx] = pp[0]; tile[1][1][idx] = pp[1];
                }
            }
        }
#pragma unroll
        for (int k = 0; k < S8_NWS; ++k) {
            const int idx = tid + S8_NT * k;
            if (idx < S8_WPC) wlds[idx] = rws[k];
        }
        __syncthreads();
        if (ch == 0) { S8_STAMP(1) }
        if (ch + 1 < nchunks) CRFP_S8_ISSUE(ch + 1)
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int ky = tap / 3, kx = tap - 3 * ky;
            const f16x8 wA = __builtin_bit_cast(f16x8, wlds[(tap * 3 + 0) * 64 + lane]);
            const f16x8 wB = __builtin_bit_cast(f16x8, wlds[(tap * 3 + 1) * 64 + lane]);
            const f16x8 wC = __builtin_bit_cast(f16x8, wlds[(tap * 3 + 2) * 64 + lane]);
#pragma unroll
            for (int pt = 0; pt < 2; ++pt) {
                const int pix = (wave + ky) * LW + pt * 32 + j + kx;
                const f16x8 b0 = __builtin_bit_cast(f16x8, tile[0][h][pix]);
                const f16x8 b1 = __builtin_bit_cast(f16x8, tile[1][h][pix]);
                acc[0][pt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wB, b0, acc[0][pt], 0, 0, 0);
                acc[0][pt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wC, b1, acc[0][pt], 0, 0, 0);
                acc[0][pt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wA, b0, acc[0][pt], 0, 0, 0);
            }
        }
#ifdef CRFP_LAB
        if (a.stamps && ch == 0) { asm volatile("s_nop 0" ::"v"(acc[0][0][0]), "v"(acc[0][1][15])); ts[2] = __builtin_amdgcn_s_memtime(); }
#endif
    }
#ifdef CRFP_LAB
    if (a.stamps) { asm volatile("s_nop 0" ::"v"(acc[0][0][0]), "v"(acc[0][1][15])); ts[3] = __builtin_amdgcn_s_memtime(); }
#endif
#undef CRFP_S8_ISSUE
#undef CRFP_QDESC
#pragma unroll
    for (int pt = 0; pt < 2; ++pt)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[0][pt][e] *= (1.0f / F16_RES_SCALE);
    const EpiCtx ec = epi_ctx(a, n);
    conv_epilogue<1, 2, 1, 2>(ec, acc, T0, tx0, ty0, wave, j, h);
#ifdef CRFP_LAB
    if (a.stamps) {
        ts[4] = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        ts[5] = __builtin_amdgcn_s_memtime();
        if (tid == 0) {
            long long* o = a.stamps + (long long)blockIdx.x * 8;
            for (int q = 0; q < 6; ++q) o[q] = ts[q];
            o[6] = __builtin_amdgcn_s_getreg(6 << 11 | 4 << 6 | 20);   // HW_REG_HW_ID (id 4), offset 0? -> unused placeholder
            o[7] = 1;
        }
    }
#endif
#undef S8_STAMP
}

#ifdef CRFP_LAB
// ---------------------------------------------------------------- f16x3s, persistent over 4-row tiles (lab: lost its A/B)
// Measured, same box, bit-identical output: 32 -> 32 convs 25.0 -> 27.3 us, 64 -> 32 37.2 -> 41.2, block0 43.7 -> 48.8, the pixel-shuffle
// heads 41.2 -> 40.1; clip 10.74 -> 11.06 ms.  What the cross-tile prefetch hides is less than what half-height tiles cost: 5 instead of
// 3.5 ds_read_b128 per MFMA triple (one 32-pixel tile per wave re-reads the A fragments twice as often), the 27.6 KB weight stage per
// 4 rows instead of per 8, a 6 / 4 instead of 10 / 8 halo.  CRFP_S8P=1 selects it in the lab library.
// Same arithmetic, K order and accumulation order as conv3x3_split8_kernel (so the same bits), other work split: a tile is 4 rows x 64
// pixels (wave w: row w / 2, pixel half w & 1 -- ONE 32-pixel MFMA tile per wave, 16 accumulator registers), a 360 x 640 map is 900 tiles,
// and the 512 resident workgroups walk them with the chunk prefetch running ACROSS the tile boundary: the first chunk of tile i + 1 is in
// flight during the last MFMAs and the stores of tile i, so a workgroup's second tile pays neither the load latency of a prologue nor
// the store phase of the first one (profiles/r03_conv_split8_timeline.txt: 28 % + 20 % of a one-tile workgroup's lifetime).  Barriers
// order LDS only (s_waitcnt lgkmcnt(0) + s_barrier): __syncthreads() would wait for the previous tile's stores to be acknowledged.
constexpr int S8P_TH = 4, S8P_LH = S8P_TH + 2, S8P_NEL = S8P_LH * LW, S8P_NT = 512;
constexpr int S8P_NIN = (S8P_NEL + S8P_NT - 1) / S8P_NT, S8P_NWS = (S8_WPC + S8P_NT - 1) / S8P_NT;
__device__ __forceinline__ void s8p_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__global__ __launch_bounds__(S8P_NT) __attribute__((amdgpu_waves_per_eu(4, 4))) void conv3x3_split8p_kernel(const ConvArgs a, int items) {
    __shared__ bf16x8 tile[2][2][S8P_NEL];       // [split part][quad pair][halo pixel]   25.3 KB
    __shared__ bf16x8 wlds[S8_WPC];              // [(tap, image A/B/C)][lane]            27.6 KB
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int row = wave >> 1, half = wave & 1;
    const int tiles_x = (a.W + TW - 1) / TW;
    const int ngrp = a.ctiles;
    const int n = blockIdx.z;
    const int ns = a.src_bgroup > 0 ? n + n / a.src_bgroup : n;   // source batch item (ConvArgs::src_bgroup)
    const int H = a.H, W = a.W;
    const int G = gridDim.x;                     // a multiple of 8: every item of a workgroup lies in its XCD's band
    int item = blockIdx.x;
    if (item >= items) return;
    const int nchunks = a.kq >> 2;
    const bf16x8* __restrict__ wp = reinterpret_cast<const bf16x8*>(a.wsplit_sa);
    const EpiCtx ec = epi_ctx(a, n);

    // load side: the (tile, chunk) whose global loads are in flight
    int ltx0, lty0, lT0;
    int cgy[S8P_NIN], cgx[S8P_NIN];
    bool sval[S8P_NIN];
    f32x4 rq0[S8P_NIN], rq1[S8P_NIN], rq2[S8P_NIN], rq3[S8P_NIN];
    bf16x8 rws[S8P_NWS];
    int qm0 = 0, qm1 = 0, qm2 = 0, qm3 = 0;
#define S8P_SET_TILE(IT)                                                                                  \
    {                                                                                                     \
        const int bw_ = xcd_band_tile((IT), items);                                                       \
        const int bt_ = bw_ / ngrp;                                                                       \
        ltx0 = (bt_ % tiles_x) * TW; lty0 = (bt_ / tiles_x) * S8P_TH; lT0 = bw_ - bt_ * ngrp;             \
        _Pragma("unroll") for (int t = 0; t < S8P_NIN; ++t) {                                             \
            const int idx = min(tid + S8P_NT * t, S8P_NEL - 1);                                           \
            const int r = idx / LW, c = idx - r * LW;                                                     \
            const int gy = lty0 + r - 1, gx = ltx0 + c - 1;                                               \
            sval[t] = tid + S8P_NT * t < S8P_NEL && gy >= 0 && gy < H && gx >= 0 && gx < W;               \
            cgy[t] = min(max(gy, 0), H - 1);                                                              \
            cgx[t] = min(max(gx, 0), W - 1);                                                              \
        }                                                                                                 \
    }
#define S8P_QDESC(QB_, QRS, QCS, QM, QI, CH)                                                              \
    const float* QB_; int QRS, QCS;                                                                       \
    {                                                                                                     \
        const QuadDesc d_ = a.qd[4 * (CH) + (QI)];                                                        \
        QB_ = d_.base + (long long)ns * d_.bstride; QRS = d_.rs; QCS = d_.cs; QM = d_.mask;                \
    }
#define S8P_ISSUE(CH)                                                                                     \
    {                                                                                                     \
        S8P_QDESC(qb0, qrs0, qcs0, qm0, 0, CH) S8P_QDESC(qb1, qrs1, qcs1, qm1, 1, CH)                     \
        S8P_QDESC(qb2, qrs2, qcs2, qm2, 2, CH) S8P_QDESC(qb3, qrs3, qcs3, qm3, 3, CH)                     \
        _Pragma("unroll") for (int t = 0; t < S8P_NIN; ++t) {                                             \
            rq0[t] = CRFP_LDACT(f32x4, qb0 + cgy[t] * qrs0 + cgx[t] * qcs0);                \
            rq1[t] = CRFP_LDACT(f32x4, qb1 + cgy[t] * qrs1 + cgx[t] * qcs1);                \
            rq2[t] = CRFP_LDACT(f32x4, qb2 + cgy[t] * qrs2 + cgx[t] * qcs2);                \
            rq3[t] = CRFP_LDACT(f32x4, qb3 + cgy[t] * qrs3 + cgx[t] * qcs3);                \
        }                                                                                                 \
        _Pragma("unroll") for (int k = 0; k < S8P_NWS; ++k) {                                             \
            const int idx = min(tid + S8P_NT * k, S8_WPC - 1);                                            \
            rws[k] = wp[((long long)lT0 * nchunks + (CH)) * S8_WPC + idx];                                \
        }                                                                                                 \
    }

    S8P_SET_TILE(item)
    S8P_ISSUE(0)
    for (;;) {
        // compute side: this tile
        const int tx0 = ltx0, ty0 = lty0, T0 = lT0;
        f32x16 acc[1][1];
        {   // the accumulator starts at 2^11 * bias
            const float4* __restrict__ bp = reinterpret_cast<const float4*>(a.bpk);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 bq = bp[T0 * 8 + 2 * g + h];
                acc[0][0][4 * g + 0] = bq.x * F16_RES_SCALE; acc[0][0][4 * g + 1] = bq.y * F16_RES_SCALE;
                acc[0][0][4 * g + 2] = bq.z * F16_RES_SCALE; acc[0][0][4 * g + 3] = bq.w * F16_RES_SCALE;
            }
        }
        const int next = item + G;
        for (int ch = 0; ch < nchunks; ++ch) {
            const int m0 = qm0, m1 = qm1, m2 = qm2, m3 = qm3;
            s8p_lds_barrier();                   // every wave is done reading the previous stage
            if (m0 & 16) {   // SRC_S3 chunk (wave-uniform): already split by its producer -- copy, zero outside the image
#pragma unroll
                for (int t = 0; t < S8P_NIN; ++t) {
                    const int idx = tid + S8P_NT * t;
                    if (idx < S8P_NEL) {
                        const f32x4 z = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
                        tile[0][0][idx] = __builtin_bit_cast(bf16x8, sval[t] ? rq0[t] : z);
                        tile[0][1][idx] = __builtin_bit_cast(bf16x8, sval[t] ? rq1[t] : z);
                        tile[1][0][idx] = __builtin_bit_cast(bf16x8, sval[t] ? rq2[t] : z);
                        tile[1][1][idx] = __builtin_bit_cast(bf16x8, sval[t] ? rq3[t] : z);
                    }
                }
            } else if ((m0 & m1 & m2 & m3) == 15) {
#pragma unroll
                for (int t = 0; t < S8P_NIN; ++t) {
                    const int idx = tid + S8P_NT * t;
                    if (idx < S8P_NEL) {
                        const float sc = sval[t] ? F16_RES_SCALE : 0.0f;
                        const unsigned km = sval[t] ? 0xffffffffu : 0u;
                        bf16x8 pa, pb;
                        split_f16x8_fast(rq0[t], rq1[t], sc, km, pa, pb);
                        tile[0][0][idx] = pa; tile[1][0][idx] = pb;
                        split_f16x8_fast(rq2[t], rq3[t], sc, km, pa, pb);
                        tile[0][1][idx] = pa; tile[1][1][idx] = pb;
                    }
                }
            } else {
#pragma unroll
                for (int t = 0; t < S8P_NIN; ++t) {
                    const int idx = tid + S8P_NT * t;
                    if (idx < S8P_NEL) {
                        bf16x8 pp[2];
                        split_parts<2>(mask_quad(rq0[t], sval[t] ? m0 : 0), mask_quad(rq1[t], sval[t] ? m1 : 0), pp);
                        tile[0][0][idx] = pp[0]; tile[1][0][idx] = pp[1];
                        split_parts<2>(mask_quad(rq2[t], sval[t] ? m2 : 0), mask_quad(rq3[t], sval[t] ? m3 : 0), pp);
                        tile[0][1][idx] = pp[0]; tile[1][1][idx] = pp[1];
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < S8P_NWS; ++k) {
                const int idx = tid + S8P_NT * k;
                if (idx < S8_WPC) wlds[idx] = rws[k];
            }
            s8p_lds_barrier();
            if (ch + 1 < nchunks) S8P_ISSUE(ch + 1)
            else if (next < items) { S8P_SET_TILE(next) S8P_ISSUE(0) }   // the next tile's first chunk: in flight during these MFMAs and the stores below
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int ky = tap / 3, kx = tap - 3 * ky;
                const f16x8 wA = __builtin_bit_cast(f16x8, wlds[(tap * 3 + 0) * 64 + lane]);
                const f16x8 wB = __builtin_bit_cast(f16x8, wlds[(tap * 3 + 1) * 64 + lane]);
                const f16x8 wC = __builtin_bit_cast(f16x8, wlds[(tap * 3 + 2) * 64 + lane]);
                const int pix = (row + ky) * LW + half * 32 + j + kx;
                const f16x8 b0 = __builtin_bit_cast(f16x8, tile[0][h][pix]);
                const f16x8 b1 = __builtin_bit_cast(f16x8, tile[1][h][pix]);
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wB, b0, acc[0][0], 0, 0, 0);
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wC, b1, acc[0][0], 0, 0, 0);
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wA, b0, acc[0][0], 0, 0, 0);
            }
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[0][0][e] *= (1.0f / F16_RES_SCALE);
        conv_epilogue<1, 1, 1, 2>(ec, acc, T0, tx0 + half * 32, ty0, row, j, h);
        if (next >= items) break;
        item = next;
    }
#undef S8P_ISSUE
#undef S8P_QDESC
#undef S8P_SET_TILE
}
#endif  // CRFP_LAB

#endif  // !CRFP_ACT_BF16

#ifdef CRFP_ACT_BF16
// ================================================================ bf16-storage main loop
// Activations arrive as bf16 pixel quads (8 bytes): the halo tile is COPIED into one bf16 LDS image (no conversion, no
// split), weights are bf16 (rounded once at pack time), one v_mfma_f32_32x32x16_bf16 per (tap, 16-channel chunk,
// 32-pixel tile) with fp32 accumulators that start at the fp32 bias.  Same tile shape, lane map, K order and epilogue as
// conv3x3_split_kernel<1,1,2> of the fp32 build.  The only fp32 operand left is the flow field of dcn_block.0
// (SRC_FLOW2, QuadDesc::mask bit 5): its (dx, dy) pair has the width of a bf16 quad and is rounded to bf16 on the way
// into LDS.  LDS 22 KB per workgroup.
__device__ __forceinline__ unsigned pack_bf16x2(float a, float b) {
    typedef __bf16 h2_t __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(unsigned, __builtin_convertvector(cf32x2{a, b}, h2_t));
}
// one raw 8-byte K-quad -> its two packed bf16 words; m = component mask (bit 5: fp32 (dx, dy) pair)
__device__ __forceinline__ cu32x2 quad_words(cu32x2 r, int m, bool valid) {
    if (!valid) return cu32x2{0u, 0u};
    if (m & 32) {   // (per-component bit casts of r.x / r.y were folded into pack(x, x) by hipcc 7.2: cast the whole pair)
        typedef __bf16 h2_t __attribute__((ext_vector_type(2)));
        return cu32x2{__builtin_bit_cast(unsigned, __builtin_convertvector(__builtin_bit_cast(cf32x2, r), h2_t)), 0u};
    }
    const unsigned k0 = ((m & 1) ? 0x0000ffffu : 0u) | ((m & 2) ? 0xffff0000u : 0u);
    const unsigned k1 = ((m & 4) ? 0x0000ffffu : 0u) | ((m & 8) ? 0xffff0000u : 0u);
    return cu32x2{r.x & k0, r.y & k1};
}

#ifndef CRFP_BF16_LB
#define CRFP_BF16_LB 5   // workgroups per CU the register allocation aims at.  Round 4, same box, 4-clip lock-step batch: 4 -> 5 takes the kernel
                         // sum per clip 5.67 -> 5.60 ms (32 -> 32 convs 9.4 -> 8.9 us), one-clip calls unchanged; 6 spills (40 B) and loses 6 %
#endif
template <int RPW, bool KS = false>   // KS: K slices over blockIdx.z, see conv3x3_split_body
__global__ __launch_bounds__(256, RPW == 1 ? CRFP_BF16_LB : 3) void conv3x3_bf16_kernel(const ConvArgs a) {
    constexpr int CT = 1, TH = 4 * RPW, LH = TH + 2, PT = 2 * RPW;
    constexpr int NEL = LH * LW;                 // halo pixels
    constexpr int NIN = (NEL + 255) / 256;       // halo pixels per thread; each carries the chunk's 4 quads
    constexpr int WPC = 9 * 64;                  // weight vectors per (cout tile, chunk)
    constexpr int NWS = (WPC + 255) / 256;
    typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
    __shared__ u32x4_t tile[2][NEL];             // [quad pair][halo pixel]: 8 bf16 = 16 B
    __shared__ bf16x8 wlds[WPC];                 // [tap][lane]: this chunk's A fragments
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int tiles_x = (a.W + TW - 1) / TW;
    const int ngrp = a.ctiles;
    const int bwork = xcd_band_tile(blockIdx.x, gridDim.x);
    const int btile = bwork / ngrp;
    const int tx0 = (btile % tiles_x) * TW, ty0 = (btile / tiles_x) * TH;
    const int T0 = bwork - btile * ngrp;
    const int n = blockIdx.z;
    const int kslices = KS ? a.ksplit : 1, kslice = KS ? n % kslices : 0, item = KS ? n / kslices : n;
    const int ns = a.src_bgroup > 0 ? item + item / a.src_bgroup : item;   // source batch item (ConvArgs::src_bgroup)
    const int H = a.H, W = a.W;

    int cgy[NIN], cgx[NIN];
    bool sval[NIN];
#pragma unroll
    for (int t = 0; t < NIN; ++t) {
        const int idx = min(tid + 256 * t, NEL - 1);
        const int r = idx / LW, c = idx - r * LW;
        const int gy = ty0 + r - 1, gx = tx0 + c - 1;
        sval[t] = tid + 256 * t < NEL && gy >= 0 && gy < H && gx >= 0 && gx < W;
        cgy[t] = min(max(gy, 0), H - 1);
        cgx[t] = min(max(gx, 0), W - 1);
    }

    f32x16 acc[CT][PT];
    if (KS && kslice != 0) {
#pragma unroll
        for (int pt = 0; pt < PT; ++pt)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[0][pt][e] = 0.0f;
    } else {   // accumulators start at the bias
        const float4* __restrict__ bp = reinterpret_cast<const float4*>(a.bpk);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 bq = bp[T0 * 8 + 2 * g + h];
#pragma unroll
            for (int pt = 0; pt < PT; ++pt) {
                acc[0][pt][4 * g + 0] = bq.x; acc[0][pt][4 * g + 1] = bq.y;
                acc[0][pt][4 * g + 2] = bq.z; acc[0][pt][4 * g + 3] = bq.w;
            }
        }
    }

    const int nchunks = a.kq >> 2;
    const bf16x8* __restrict__ wp = reinterpret_cast<const bf16x8*>(a.wsplit16);
    cu32x2 rq0[NIN], rq1[NIN], rq2[NIN], rq3[NIN];
    bf16x8 rws[NWS];
    const float* qb0; const float* qb1; const float* qb2; const float* qb3;
    int qrs0, qrs1, qrs2, qrs3, qcs0, qcs1, qcs2, qcs3, qm0 = 0, qm1 = 0, qm2 = 0, qm3 = 0;
#define CRFP_QDESC(QB_, QRS, QCS, QM, QI, CH)                                                             \
    {                                                                                                     \
        const QuadDesc d_ = a.qd[4 * (CH) + (QI)];                                                        \
        QB_ = d_.base + (long long)ns * d_.bstride; QRS = d_.rs; QCS = d_.cs; QM = d_.mask;                \
    }
#define CRFP_BF16_ISSUE(CH)                                                                               \
    {                                                                                                     \
        CRFP_QDESC(qb0, qrs0, qcs0, qm0, 0, CH) CRFP_QDESC(qb1, qrs1, qcs1, qm1, 1, CH)                   \
        CRFP_QDESC(qb2, qrs2, qcs2, qm2, 2, CH) CRFP_QDESC(qb3, qrs3, qcs3, qm3, 3, CH)                   \
        _Pragma("unroll") for (int t = 0; t < NIN; ++t) {                                                 \
            rq0[t] = CRFP_LDACT(cu32x2, qb0 + cgy[t] * qrs0 + cgx[t] * qcs0);               \
            rq1[t] = CRFP_LDACT(cu32x2, qb1 + cgy[t] * qrs1 + cgx[t] * qcs1);               \
            rq2[t] = CRFP_LDACT(cu32x2, qb2 + cgy[t] * qrs2 + cgx[t] * qcs2);               \
            rq3[t] = CRFP_LDACT(cu32x2, qb3 + cgy[t] * qrs3 + cgx[t] * qcs3);               \
        }                                                                                                 \
        _Pragma("unroll") for (int k = 0; k < NWS; ++k) {                                                 \
            const int idx = min(tid + 256 * k, WPC - 1);                                                  \
            rws[k] = wp[((long long)T0 * nchunks + (CH)) * WPC + idx];                                    \
        }                                                                                                 \
    }

    const int ch_begin = KS ? kslice * (nchunks / kslices) : 0, ch_end = KS ? ch_begin + nchunks / kslices : nchunks;
    CRFP_BF16_ISSUE(ch_begin)
    for (int ch = ch_begin; ch < ch_end; ++ch) {
        const int m0 = qm0, m1 = qm1, m2 = qm2, m3 = qm3;
        __syncthreads();  // every wave finished reading the previous chunk
        if ((m0 & m1 & m2 & m3) == 15 && !((m0 | m1 | m2 | m3) & 32)) {   // wave-uniform: 16 real bf16 channels -> plain copy
#pragma unroll
            for (int t = 0; t < NIN; ++t) {
                const int idx = tid + 256 * t;
                if (idx < NEL) {
                    const unsigned km = sval[t] ? 0xffffffffu : 0u;
                    tile[0][idx] = u32x4_t{rq0[t].x & km, rq0[t].y & km, rq1[t].x & km, rq1[t].y & km};
                    tile[1][idx] = u32x4_t{rq2[t].x & km, rq2[t].y & km, rq3[t].x & km, rq3[t].y & km};
                }
            }
        } else {
#pragma unroll
            for (int t = 0; t < NIN; ++t) {
                const int idx = tid + 256 * t;
                if (idx < NEL) {
                    const cu32x2 w0 = quad_words(rq0[t], m0, sval[t]), w1 = quad_words(rq1[t], m1, sval[t]);
                    const cu32x2 w2 = quad_words(rq2[t], m2, sval[t]), w3 = quad_words(rq3[t], m3, sval[t]);
                    tile[0][idx] = u32x4_t{w0.x, w0.y, w1.x, w1.y};
                    tile[1][idx] = u32x4_t{w2.x, w2.y, w3.x, w3.y};
                }
            }
        }
#pragma unroll
        for (int k = 0; k < NWS; ++k) {
            const int idx = tid + 256 * k;
            if (idx < WPC) wlds[idx] = rws[k];
        }
        __syncthreads();
        if (ch + 1 < ch_end) CRFP_BF16_ISSUE(ch + 1)
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int ky = tap / 3, kx = tap - 3 * ky;
            const bf16x8 wa = wlds[tap * 64 + lane];
#pragma unroll
            for (int pt = 0; pt < PT; ++pt) {
                const int pix = (wave * RPW + (pt >> 1) + ky) * LW + (pt & 1) * 32 + j + kx;
                const bf16x8 bq = __builtin_bit_cast(bf16x8, tile[h][pix]);
                acc[0][pt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa, bq, acc[0][pt], 0, 0, 0);
            }
        }
    }
#undef CRFP_BF16_ISSUE
#undef CRFP_QDESC
    const EpiCtx ec = epi_ctx(a, n);
    conv_epilogue<CT, PT, RPW, 2>(ec, acc, T0, tx0, ty0, wave, j, h);
}

// ---------------------------------------------------------------- bf16 storage, 8 waves, two chunks per stage
// The 32-cout layers (one cout tile): workgroup = 8 rows x 64 pixels (wave = one row, two 32-pixel tiles), 450 workgroups for a
// 360 x 640 map = one round at two per CU, and 32 input channels (two 16-channel chunks: tile 42 KB + weights 18 KB) per barrier
// pair -- a 32 -> 32 conv has ONE staging phase instead of two, a 64 -> 32 conv two instead of four.  Same K order, same
// accumulate order per accumulator (chunk-major, taps inside) and the same epilogue as conv3x3_bf16_kernel: identical values.
constexpr int B8_NT = 512, B8_TH = 8, B8_LH = B8_TH + 2, B8_NEL = B8_LH * LW, B8_NIN = (B8_NEL + B8_NT - 1) / B8_NT;
constexpr int B8_WPC = 9 * 64, B8_NWS = (2 * B8_WPC + B8_NT - 1) / B8_NT;

__global__ __launch_bounds__(B8_NT, 2) void conv3x3_bf16x8_kernel(const ConvArgs a) {
    typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
    __shared__ u32x4_t tile[2][2][B8_NEL];       // [chunk of the stage][quad pair][halo pixel]: 8 bf16 = 16 B      42.2 KB
    __shared__ bf16x8 wlds[2 * B8_WPC];          // [chunk][tap][lane]                                             18.4 KB
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int tiles_x = (a.W + TW - 1) / TW;
    const int ngrp = a.ctiles;
    const int bwork = xcd_band_tile(blockIdx.x, gridDim.x);
    const int btile = bwork / ngrp;
    const int tx0 = (btile % tiles_x) * TW, ty0 = (btile / tiles_x) * B8_TH;
    const int T0 = bwork - btile * ngrp;
    const int n = blockIdx.z;
    const int ns = a.src_bgroup > 0 ? n + n / a.src_bgroup : n;   // source batch item (ConvArgs::src_bgroup)
    const int H = a.H, W = a.W;

    int cgy[B8_NIN], cgx[B8_NIN];
    bool sval[B8_NIN];
#pragma unroll
    for (int t = 0; t < B8_NIN; ++t) {
        const int idx = min(tid + B8_NT * t, B8_NEL - 1);
        const int r = idx / LW, c = idx - r * LW;
        const int gy = ty0 + r - 1, gx = tx0 + c - 1;
        sval[t] = tid + B8_NT * t < B8_NEL && gy >= 0 && gy < H && gx >= 0 && gx < W;
        cgy[t] = min(max(gy, 0), H - 1);
        cgx[t] = min(max(gx, 0), W - 1);
    }

    f32x16 acc[1][2];
    {   // accumulators start at the bias
        const float4* __restrict__ bp = reinterpret_cast<const float4*>(a.bpk);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 bq = bp[T0 * 8 + 2 * g + h];
#pragma unroll
            for (int pt = 0; pt < 2; ++pt) {
                acc[0][pt][4 * g + 0] = bq.x; acc[0][pt][4 * g + 1] = bq.y;
                acc[0][pt][4 * g + 2] = bq.z; acc[0][pt][4 * g + 3] = bq.w;
            }
        }
    }

    const int nchunks = a.kq >> 2, nstages = (nchunks + 1) >> 1;
    const bf16x8* __restrict__ wp = reinterpret_cast<const bf16x8*>(a.wsplit16);
    cu32x2 rq[2][4][B8_NIN];   // [chunk of the stage][quad][halo slot]
    int qm[2][4];
    bf16x8 rws[B8_NWS];
    // loads of stage ST: its one or two chunks' quads (per-quad descriptors) and weight fragments (consecutive in the packed image)
#define CRFP_B8_ISSUE(ST)                                                                                 \
    {                                                                                                     \
        _Pragma("unroll") for (int c2 = 0; c2 < 2; ++c2) {                                                \
            const int ch_ = min(2 * (ST) + c2, nchunks - 1);                                              \
            _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                               \
                const QuadDesc d_ = a.qd[4 * ch_ + q];                                                    \
                const float* qb_ = d_.base + (long long)ns * d_.bstride;                                   \
                qm[c2][q] = d_.mask;                                                                      \
                _Pragma("unroll") for (int t = 0; t < B8_NIN; ++t)                                        \
                    rq[c2][q][t] = CRFP_LDACT(cu32x2, qb_ + cgy[t] * d_.rs + cgx[t] * d_.cs); \
            }                                                                                             \
        }                                                                                                 \
        _Pragma("unroll") for (int k = 0; k < B8_NWS; ++k) {                                              \
            const int idx = min(tid + B8_NT * k, 2 * B8_WPC - 1);                                         \
            const int ch_ = min(2 * (ST) + idx / B8_WPC, nchunks - 1);                                    \
            rws[k] = wp[((long long)T0 * nchunks + ch_) * B8_WPC + (idx % B8_WPC)];                       \
        }                                                                                                 \
    }

    CRFP_B8_ISSUE(0)
    for (int st = 0; st < nstages; ++st) {
        const bool two = 2 * st + 1 < nchunks;   // the last stage of an odd chunk count holds one chunk
        __syncthreads();
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2) {
            const int m0 = qm[c2][0], m1 = qm[c2][1], m2 = qm[c2][2], m3 = qm[c2][3];
            if ((m0 & m1 & m2 & m3) == 15 && !((m0 | m1 | m2 | m3) & 32)) {
#pragma unroll
                for (int t = 0; t < B8_NIN; ++t) {
                    const int idx = tid + B8_NT * t;
                    if (idx < B8_NEL) {
                        const unsigned km = sval[t] ? 0xffffffffu : 0u;
                        tile[c2][0][idx] = u32x4_t{rq[c2][0][t].x & km, rq[c2][0][t].y & km, rq[c2][1][t].x & km, rq[c2][1][t].y & km};
                        tile[c2][1][idx] = u32x4_t{rq[c2][2][t].x & km, rq[c2][2][t].y & km, rq[c2][3][t].x & km, rq[c2][3][t].y & km};
                    }
                }
            } else {
#pragma unroll
                for (int t = 0; t < B8_NIN; ++t) {
                    const int idx = tid + B8_NT * t;
                    if (idx < B8_NEL) {
                        const cu32x2 w0 = quad_words(rq[c2][0][t], m0, sval[t]), w1 = quad_words(rq[c2][1][t], m1, sval[t]);
                        const cu32x2 w2 = quad_words(rq[c2][2][t], m2, sval[t]), w3 = quad_words(rq[c2][3][t], m3, sval[t]);
                        tile[c2][0][idx] = u32x4_t{w0.x, w0.y, w1.x, w1.y};
                        tile[c2][1][idx] = u32x4_t{w2.x, w2.y, w3.x, w3.y};
                    }
                }
            }
        }
#pragma unroll
        for (int k = 0; k < B8_NWS; ++k) {
            const int idx = tid + B8_NT * k;
            if (idx < 2 * B8_WPC) wlds[idx] = rws[k];
        }
        __syncthreads();
        if (st + 1 < nstages) CRFP_B8_ISSUE(st + 1)
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2) {
            if (c2 == 1 && !two) break;   // workgroup-uniform
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int ky = tap / 3, kx = tap - 3 * ky;
                const bf16x8 wa = wlds[c2 * B8_WPC + tap * 64 + lane];
#pragma unroll
                for (int pt = 0; pt < 2; ++pt) {
                    const int pix = (wave + ky) * LW + pt * 32 + j + kx;
                    const bf16x8 bq = __builtin_bit_cast(bf16x8, tile[c2][h][pix]);
                    acc[0][pt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa, bq, acc[0][pt], 0, 0, 0);
                }
            }
        }
    }
#undef CRFP_B8_ISSUE
    const EpiCtx ec = epi_ctx(a, n);
    conv_epilogue<1, 2, 1, 2>(ec, acc, T0, tx0, ty0, wave, j, h);
}

// ---------------------------------------------------------------- bf16 storage: TWO 3x3 convs in one launch, the tensor between them in LDS
// conv A (any sources, K in 16-channel chunks, 32 couts, bias + none / relu / lrelu) feeds conv B (32 -> 32) and has no other
// reader: dcn_block.0 -> .2 and res conv1 -> conv2(+x) of every level (model/CRFP.py:331-333, 449-481).  Run separately each
// of them is one round of 450 workgroups whose load / MFMA / store phases add up and the 14.7 MB tensor
// between them is written and read back; here conv A is evaluated on the (8 + 2) x 64 region conv B's 8 x 62 output tile needs
// (halo recompute: 20 pixel tiles of 32 per workgroup instead of 16, 3 + 2 per SIMD), rounded to bf16 exactly where the
// two-kernel path stores it, zeroed outside the image (conv B's zero padding) and kept in LDS as conv B's B-operand image.
// Same K order and per-accumulator accumulation order as conv3x3_bf16_kernel / conv3x3_bf16x8_kernel: identical values.
// LDS 76.7 KB (two workgroups per CU): input chunk tile 25.3 KB (later conv B's 18.4 KB of weights), intermediate 42.2 KB,
// conv A weight stage 9.2 KB.  45 x 11 = 495 workgroups for a 360 x 640 map = one round of the 512 slots.
// (The fp32 build has no such kernel: its intermediate is two fp16 images = 84 KB, with the staging tile and the 27.6 KB
// three-image weight stage 160 KB -- one 8-wave workgroup per CU and 1.9 rounds)
constexpr int P2_NT = 512, P2_OW = 62, P2_IH = 10, P2_IW = 64, P2_LH = 12, P2_LW = 66;
constexpr int P2_NEL = P2_LH * P2_LW, P2_NIN = (P2_NEL + P2_NT - 1) / P2_NT;   // 792 input halo pixels, 2 per thread
constexpr int P2_MP = P2_IW + 2, P2_MEL = P2_IH * P2_MP;                       // intermediate rows of 64 + 2 zero columns
constexpr int P2_WPC = 9 * 64, P2_NWS = (P2_WPC + P2_NT - 1) / P2_NT, P2_NWB = (2 * P2_WPC + P2_NT - 1) / P2_NT;

struct PairB {          // what conv B adds to conv A's plan: its weights, bias, activation, destinations, residual
    const void* wsplit16;
    const float* bpk;
    const float* resid;
    long long resid_bstride;
    ConvDst dst[CRFP_MAX_DST];
    int ndst, cout, act;
    float post_scale;
};

__global__ __launch_bounds__(P2_NT, 2) void conv3x3_bf16_pair_kernel(const ConvArgs a, const PairB b) {
    typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
    __shared__ u32x4_t tileA[2][P2_NEL];         // conv A: [quad pair][input halo pixel] of the current chunk; then conv B's weights
    __shared__ u32x4_t mid[2][2][P2_MEL];        // conv A's output = conv B's input: [chunk][quad pair][pixel], 8 bf16 each
    __shared__ bf16x8 wlds[P2_WPC];              // conv A: [tap][lane] of the current chunk
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int tiles_x = (a.W + P2_OW - 1) / P2_OW;
    const int btile = xcd_band_tile(blockIdx.x, gridDim.x);
    const int tx0 = (btile % tiles_x) * P2_OW, ty0 = (btile / tiles_x) * 8;
    const int n = blockIdx.z;
    const int H = a.H, W = a.W;

    int cgy[P2_NIN], cgx[P2_NIN];
    bool sval[P2_NIN];
#pragma unroll
    for (int t = 0; t < P2_NIN; ++t) {
        const int idx = min(tid + P2_NT * t, P2_NEL - 1);
        const int r = idx / P2_LW, c = idx - r * P2_LW;
        const int gy = ty0 + r - 2, gx = tx0 + c - 2;
        sval[t] = tid + P2_NT * t < P2_NEL && gy >= 0 && gy < H && gx >= 0 && gx < W;
        cgy[t] = min(max(gy, 0), H - 1);
        cgx[t] = min(max(gx, 0), W - 1);
    }
    // the two zero columns behind each intermediate row (read by conv B's taps of the two discarded output columns)
    if (tid < 4 * P2_IH * 2) {
        const int pl = tid / (P2_IH * 2), rc = tid - pl * (P2_IH * 2);
        (&mid[0][0][0])[pl * P2_MEL + (rc >> 1) * P2_MP + P2_IW + (rc & 1)] = u32x4_t{0u, 0u, 0u, 0u};
    }

    // conv A: wave w owns the 32-pixel tiles w, w + 8 (and w + 16 for w < 4) of the 10 x 64 region; tile t = row t >> 1, half t & 1
    const int ntA = wave < 4 ? 3 : 2;
    f32x16 accA[3];
    {
        const float4* __restrict__ bp = reinterpret_cast<const float4*>(a.bpk);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 bq = bp[2 * g + h];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                accA[k][4 * g + 0] = bq.x; accA[k][4 * g + 1] = bq.y; accA[k][4 * g + 2] = bq.z; accA[k][4 * g + 3] = bq.w;
            }
        }
    }
    const int nchunks = a.kq >> 2;
    const bf16x8* __restrict__ wp = reinterpret_cast<const bf16x8*>(a.wsplit16);
    cu32x2 rq0[P2_NIN], rq1[P2_NIN], rq2[P2_NIN], rq3[P2_NIN];
    bf16x8 rws[P2_NWS];
    const float* qb0; const float* qb1; const float* qb2; const float* qb3;
    int qrs0, qrs1, qrs2, qrs3, qcs0, qcs1, qcs2, qcs3, qm0 = 0, qm1 = 0, qm2 = 0, qm3 = 0;
#define CRFP_QDESC(QB_, QRS, QCS, QM, QI, CH)                                                             \
    {                                                                                                     \
        const QuadDesc d_ = a.qd[4 * (CH) + (QI)];                                                        \
        QB_ = d_.base + (long long)n * d_.bstride; QRS = d_.rs; QCS = d_.cs; QM = d_.mask;                \
    }
#define CRFP_P2_ISSUE(CH)                                                                                 \
    {                                                                                                     \
        CRFP_QDESC(qb0, qrs0, qcs0, qm0, 0, CH) CRFP_QDESC(qb1, qrs1, qcs1, qm1, 1, CH)                   \
        CRFP_QDESC(qb2, qrs2, qcs2, qm2, 2, CH) CRFP_QDESC(qb3, qrs3, qcs3, qm3, 3, CH)                   \
        _Pragma("unroll") for (int t = 0; t < P2_NIN; ++t) {                                              \
            rq0[t] = CRFP_LDACT(cu32x2, qb0 + cgy[t] * qrs0 + cgx[t] * qcs0);               \
            rq1[t] = CRFP_LDACT(cu32x2, qb1 + cgy[t] * qrs1 + cgx[t] * qcs1);               \
            rq2[t] = CRFP_LDACT(cu32x2, qb2 + cgy[t] * qrs2 + cgx[t] * qcs2);               \
            rq3[t] = CRFP_LDACT(cu32x2, qb3 + cgy[t] * qrs3 + cgx[t] * qcs3);               \
        }                                                                                                 \
        _Pragma("unroll") for (int k = 0; k < P2_NWS; ++k) {                                              \
            const int idx = min(tid + P2_NT * k, P2_WPC - 1);                                             \
            rws[k] = wp[(long long)(CH) * P2_WPC + idx];                                                  \
        }                                                                                                 \
    }
    CRFP_P2_ISSUE(0)
    bf16x8 rwb[P2_NWB];   // conv B's weights (both chunks), fetched under conv A's last chunk
    for (int ch = 0; ch < nchunks; ++ch) {
        const int m0 = qm0, m1 = qm1, m2 = qm2, m3 = qm3;
        __syncthreads();  // every wave finished reading the previous chunk
        if ((m0 & m1 & m2 & m3) == 15 && !((m0 | m1 | m2 | m3) & 32)) {   // wave-uniform: 16 real bf16 channels -> plain copy
#pragma unroll
            for (int t = 0; t < P2_NIN; ++t) {
                const int idx = tid + P2_NT * t;
                if (idx < P2_NEL) {
                    const unsigned km = sval[t] ? 0xffffffffu : 0u;
                    tileA[0][idx] = u32x4_t{rq0[t].x & km, rq0[t].y & km, rq1[t].x & km, rq1[t].y & km};
                    tileA[1][idx] = u32x4_t{rq2[t].x & km, rq2[t].y & km, rq3[t].x & km, rq3[t].y & km};
                }
            }
        } else {
#pragma unroll
            for (int t = 0; t < P2_NIN; ++t) {
                const int idx = tid + P2_NT * t;
                if (idx < P2_NEL) {
                    const cu32x2 w0 = quad_words(rq0[t], m0, sval[t]), w1 = quad_words(rq1[t], m1, sval[t]);
                    const cu32x2 w2 = quad_words(rq2[t], m2, sval[t]), w3 = quad_words(rq3[t], m3, sval[t]);
                    tileA[0][idx] = u32x4_t{w0.x, w0.y, w1.x, w1.y};
                    tileA[1][idx] = u32x4_t{w2.x, w2.y, w3.x, w3.y};
                }
            }
        }
#pragma unroll
        for (int k = 0; k < P2_NWS; ++k) {
            const int idx = tid + P2_NT * k;
            if (idx < P2_WPC) wlds[idx] = rws[k];
        }
        __syncthreads();
        if (ch + 1 < nchunks) {
            CRFP_P2_ISSUE(ch + 1)
        } else {
            const bf16x8* __restrict__ wb = reinterpret_cast<const bf16x8*>(b.wsplit16);
#pragma unroll
            for (int k = 0; k < P2_NWB; ++k) rwb[k] = wb[min(tid + P2_NT * k, 2 * P2_WPC - 1)];
        }
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int ky = tap / 3, kx = tap - 3 * ky;
            const bf16x8 wa = wlds[tap * 64 + lane];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                if (k == 2 && ntA == 2) break;   // wave-uniform
                const int t = wave + 8 * k;
                const int pix = ((t >> 1) + ky) * P2_LW + (t & 1) * 32 + j + kx;
                const bf16x8 bq = __builtin_bit_cast(bf16x8, tileA[h][pix]);
                accA[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa, bq, accA[k], 0, 0, 0);
            }
        }
    }
#undef CRFP_P2_ISSUE
#undef CRFP_QDESC
    // conv A's epilogue into LDS: activation, zero outside the image, round to bf16 (what the two-kernel path stores), 8 bytes per
    // (cout quad, pixel): channel 8 g + 4 h + r is position 4 h + r of the 8-channel element (chunk g >> 1, quad pair g & 1)
    {
        const float slope = a.act == CRFP_ACT_RELU ? 0.0f : (a.act == CRFP_ACT_LRELU01 ? 0.1f : 1.0f);
        const float post = a.post_scale;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            if (k == 2 && ntA == 2) break;
            const int t = wave + 8 * k, r = t >> 1, c = (t & 1) * 32 + j;
            const int gy = ty0 + r - 1, gx = tx0 + c - 1;
            const bool in = gy >= 0 && gy < H && gx >= 0 && gx < W;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float x = accA[k][4 * g + e];
                    v[e] = in ? fmaxf(x, slope * x) * post : 0.0f;
                }
                cu32x2* dst = reinterpret_cast<cu32x2*>(&mid[g >> 1][g & 1][r * P2_MP + c]) + h;
                *dst = cu32x2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
            }
        }
    }
    __syncthreads();   // conv A's last reads of tileA / wlds are done (and `mid` is on its way)
    bf16x8* const wb_lds = reinterpret_cast<bf16x8*>(&tileA[0][0]);
#pragma unroll
    for (int k = 0; k < P2_NWB; ++k) {
        const int idx = tid + P2_NT * k;
        if (idx < 2 * P2_WPC) wb_lds[idx] = rwb[k];
    }
    f32x16 acc[1][2];
    {
        const float4* __restrict__ bp = reinterpret_cast<const float4*>(b.bpk);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 bq = bp[2 * g + h];
#pragma unroll
            for (int pt = 0; pt < 2; ++pt) {
                acc[0][pt][4 * g + 0] = bq.x; acc[0][pt][4 * g + 1] = bq.y; acc[0][pt][4 * g + 2] = bq.z; acc[0][pt][4 * g + 3] = bq.w;
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int ky = tap / 3, kx = tap - 3 * ky;
            const bf16x8 wa = wb_lds[c2 * P2_WPC + tap * 64 + lane];
#pragma unroll
            for (int pt = 0; pt < 2; ++pt) {
                const int pix = (wave + ky) * P2_MP + pt * 32 + j + kx;
                const bf16x8 bq = __builtin_bit_cast(bf16x8, mid[c2][h][pix]);
                acc[0][pt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa, bq, acc[0][pt], 0, 0, 0);
            }
        }
    // conv B's epilogue: the shared one, on conv B's destinations / residual / activation, 62 columns per tile
    ConvArgs eb = a;
    eb.cout = b.cout; eb.act = b.act; eb.post_scale = b.post_scale; eb.store = ST_Q4; eb.ps_r = 0;
    eb.resid = b.resid; eb.resid_bstride = b.resid_bstride; eb.s3_dst = nullptr; eb.dst_f32 = 0; eb.ndst = b.ndst;
#pragma unroll
    for (int d = 0; d < CRFP_MAX_DST; ++d) eb.dst[d] = b.dst[d];
    EpiCtx ec = epi_ctx(eb, n);
    ec.xend = min(W, tx0 + P2_OW);
    conv_epilogue<1, 2, 1, 2>(ec, acc, 0, tx0, ty0, wave, j, h);
}

#endif  // CRFP_ACT_BF16

#ifdef CRFP_LAB   // experiments that lose to conv3x3_split_kernel<1,1,2>: built only into the lab library (make lab)
// ---------------------------------------------------------------- software-pipelined persistent variant (f16x3)
// One 512-thread workgroup per CU (two waves per SIMD) walks a strided list of 8x64 output tiles; wave = one output
// row (two 32-pixel MFMA column tiles).  The work is a stream of items (tile, K-chunk).  LDS holds TWO items (halo tile
// images 2 x 42 KB + weight fragments 2 x 18 KB); the registers hold two more in raw fp32 form.  While the MFMAs of
// item s run out of LDS buffer s&1 the same waves
//   * issue the global loads of item s+2 (a full item of latency budget; measured wait at use: 16 cycles),
//   * split item s+1 (loaded during item s-1) into its two fp16 images and write them + its weights to buffer (s+1)&1,
// slice by slice between the taps: the MFMA is asynchronous (32 cycles per 32x32x16), so VALU/LDS instructions
// issued between two of them ride in its shadow.  The code between two barriers is branch-free (every thread
// writes both of its halo slots; surplus threads hit a dummy slot), otherwise the scheduler cannot interleave.
// One barrier per item; accumulators start at the bias; the epilogue runs when a tile's last chunk is done.
// History: the first version (bf16x6, 4 waves) needed 6.9 k cycles per item against 3.5 k of MFMA -- LDS operand traffic
// (0.75 ds_read_b128 per MFMA = 93 % of the LDS pipe) and in-order issue behind a full LDS queue with one wave per
// SIMD; 8 waves fixed the issue stalls but bf16x6 stayed LDS-bound (7.9 k per item).  f16x3 moves 2/3 of the bytes.
// Barrier that orders LDS traffic only.  __syncthreads() is a workgroup-scope release fence + s_barrier, and the release
// makes hipcc wait for vmcnt(0): every wave then sits out the write-acknowledge latency of its epilogue stores (2.5-3.3 k
// cycles per item measured) although nobody in the workgroup reads them.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

constexpr int PIPE_NW = 8, PIPE_NT = 64 * PIPE_NW;
constexpr int PIPE_NEL = (PIPE_NW + 2) * LW;                   // 660 halo pixels
constexpr int PIPE_WPC = 9 * 2 * 64;                           // weight vectors per (cout tile, chunk), f16x3
constexpr int PIPE_NWS = (PIPE_WPC + PIPE_NT - 1) / PIPE_NT;   // 3 weight vectors per thread

struct PipeRegs {
    f32x4 q[4][2];        // [K-quad of the chunk][halo element of this thread]
    bf16x8 w[PIPE_NWS];   // this thread's share of the chunk's 18 KB weight fragment image
    int m[4];             // component masks of the 4 quads (wave-uniform)
    bool ok[2];           // halo element inside the image
};

__device__ __forceinline__ void pipe_issue(PipeRegs& R, const ConvArgs& a, const bf16x8* __restrict__ wp, int n, int T0,
                                           int nchunks, int ch, int tx0, int ty0, int tid) {
    const int H = a.H, W = a.W;
    const int ns = a.src_bgroup > 0 ? n + n / a.src_bgroup : n;   // source batch item (ConvArgs::src_bgroup)
    int cgy[2], cgx[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int idx = min(tid + PIPE_NT * t, PIPE_NEL - 1);
        const int r = idx / LW, c = idx - r * LW;
        const int gy = ty0 + r - 1, gx = tx0 + c - 1;
        R.ok[t] = tid + PIPE_NT * t < PIPE_NEL && gy >= 0 && gy < H && gx >= 0 && gx < W;
        cgy[t] = min(max(gy, 0), H - 1);
        cgx[t] = min(max(gx, 0), W - 1);
    }
#pragma unroll
    for (int qi = 0; qi < 4; ++qi) {
        const QuadDesc d = a.qd[4 * ch + qi];
        const float* qb = d.base + (long long)ns * d.bstride;
        R.m[qi] = d.mask;
#pragma unroll
        for (int t = 0; t < 2; ++t) R.q[qi][t] = *reinterpret_cast<const f32x4*>(qb + cgy[t] * d.rs + cgx[t] * d.cs);
    }
#pragma unroll
    for (int k = 0; k < PIPE_NWS; ++k)
        R.w[k] = wp[(long long)(T0 * nchunks + ch) * PIPE_WPC + min(tid + PIPE_NT * k, PIPE_WPC - 1)];
}

// unit U in 0..3: halo element U>>1, quad pair U&1 -> two fp16x8 images.  Slot PIPE_NEL is a dummy.
template <int U>
__device__ __forceinline__ void pipe_split_unit(const PipeRegs& R, bf16x8 (*tl)[2][PIPE_NEL + 1], int tid) {
    constexpr int t = U >> 1, pr = U & 1;
    const int idx = min(tid + PIPE_NT * t, PIPE_NEL);
    bf16x8 p0, p1;
    split_f16x8(mask_quad(R.q[2 * pr][t], R.ok[t] ? R.m[2 * pr] : 0), mask_quad(R.q[2 * pr + 1][t], R.ok[t] ? R.m[2 * pr + 1] : 0),
                p0, p1);
    tl[0][pr][idx] = p0; tl[1][pr][idx] = p1;
}

template <int K>
__device__ __forceinline__ void pipe_put_weight(const PipeRegs& R, bf16x8* wl, int tid) {
    wl[min(tid + PIPE_NT * K, PIPE_WPC)] = R.w[K];   // slot PIPE_WPC is a dummy
}

__device__ __forceinline__ void pipe_tap(f32x16 (&acc)[1][2], f32x16 (&acl)[1][2], const bf16x8* wl,
                                         const bf16x8 (*tl)[2][PIPE_NEL + 1], int tap, int wave, int lane) {
    const int j = lane & 31, h = lane >> 5;
    const int ky = tap / 3, kx = tap - 3 * ky;
    bf16x8 wa[2];
#pragma unroll
    for (int p = 0; p < 2; ++p) wa[p] = wl[(tap * 2 + p) * 64 + lane];
#pragma unroll
    for (int pt = 0; pt < 2; ++pt) {
        const int pix = (wave + ky) * LW + pt * 32 + j + kx;
        bf16x8 bq[2];
#pragma unroll
        for (int p = 0; p < 2; ++p) bq[p] = tl[p][h][pix];
        split_mfma<2>(acc[0][pt], acl[0][pt], wa, bq);
    }
}

__global__ __launch_bounds__(PIPE_NT, 1) void conv3x3_split_pipe_kernel(const ConvArgs a) {
    __shared__ bf16x8 tile[2][2][2][PIPE_NEL + 1];   // [buffer][split part][quad pair][halo pixel (+1 dummy)]  84.6 KB
    __shared__ bf16x8 wlds[2][PIPE_WPC + 1];         // [buffer][(tap, part)][lane] (+1 dummy)                   36.9 KB
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int tiles_x = (a.W + TW - 1) / TW, ntiles = tiles_x * ((a.H + PIPE_NW - 1) / PIPE_NW);
    const int T0 = blockIdx.y, n = blockIdx.z;
    const int ns = a.src_bgroup > 0 ? n + n / a.src_bgroup : n;   // source batch item (ConvArgs::src_bgroup)
    const int nchunks = a.kq >> 2;
    const int my_tiles = (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int nitems = my_tiles * nchunks;
    const bf16x8* __restrict__ wp = reinterpret_cast<const bf16x8*>(a.wsplit16);
    const EpiCtx ec = epi_ctx(a, n);
    float4 bq4[4];   // this lane's bias quads: the accumulators of every tile start there
#pragma unroll
    for (int g = 0; g < 4; ++g) bq4[g] = reinterpret_cast<const float4*>(a.bpk)[T0 * 8 + 2 * g + h];

    // issue cursor (items are issued two ahead of the one being multiplied)
    int ik = 0, ich = 0;
    PipeRegs RA, RB;
#define CRFP_PIPE_ISSUE(R)                                                                                \
    {                                                                                                     \
        if (ik < my_tiles) {                                                                              \
            const int t_ = blockIdx.x + ik * gridDim.x, ty_ = t_ / tiles_x, tx_ = t_ - ty_ * tiles_x;     \
            pipe_issue(R, a, wp, n, T0, nchunks, ich, tx_ * TW, ty_ * PIPE_NW, tid);                      \
        }                                                                                                 \
        if (++ich == nchunks) { ich = 0; ++ik; }                                                          \
    }
#define CRFP_PIPE_ACC_INIT                                                                                \
    _Pragma("unroll") for (int pt = 0; pt < 2; ++pt)                                                      \
        _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                                   \
            acc[0][pt][4 * g + 0] = bq4[g].x; acc[0][pt][4 * g + 1] = bq4[g].y;                           \
            acc[0][pt][4 * g + 2] = bq4[g].z; acc[0][pt][4 * g + 3] = bq4[g].w;                           \
            acl[0][pt][4 * g + 0] = 0.0f; acl[0][pt][4 * g + 1] = 0.0f;                                   \
            acl[0][pt][4 * g + 2] = 0.0f; acl[0][pt][4 * g + 3] = 0.0f;                                   \
        }
    CRFP_PIPE_ISSUE(RA)
    CRFP_PIPE_ISSUE(RB)
    // item 0 -> buffer 0
    pipe_split_unit<0>(RA, tile[0], tid); pipe_split_unit<1>(RA, tile[0], tid);
    pipe_split_unit<2>(RA, tile[0], tid); pipe_split_unit<3>(RA, tile[0], tid);
    pipe_put_weight<0>(RA, wlds[0], tid); pipe_put_weight<1>(RA, wlds[0], tid); pipe_put_weight<2>(RA, wlds[0], tid);
    __syncthreads();

    f32x16 acc[1][2], acl[1][2];
    CRFP_PIPE_ACC_INIT
    int mk = 0, mch = 0;   // item being multiplied

#ifdef CRFP_PIPE_STAMPS
    long long st_wait = 0, st_taps = 0, st_epi = 0, st_bar = 0, st_t = __builtin_amdgcn_s_memtime();
#define CRFP_PST(ACC) { const long long t_ = __builtin_amdgcn_s_memtime(); ACC += t_ - st_t; st_t = t_; }
#else
#define CRFP_PST(ACC)
#endif
    // one item: MFMAs out of LDS buffer BUF; RNEXT (item s+1, landed) is split / copied into buffer BUF^1 between the
    // taps; RFREE (consumed by the previous item) receives the loads of item s+2 first of all
#define CRFP_PIPE_ITEM(BUF, RNEXT, RFREE)                                                                 \
    {                                                                                                     \
        CRFP_PST(st_bar)                                                                                  \
        CRFP_PIPE_ISSUE(RFREE)                                                                            \
        pipe_tap(acc, acl, wlds[BUF], tile[BUF], 0, wave, lane);                                          \
        pipe_split_unit<0>(RNEXT, tile[(BUF) ^ 1], tid);                                                  \
        pipe_tap(acc, acl, wlds[BUF], tile[BUF], 1, wave, lane);                                          \
        pipe_put_weight<0>(RNEXT, wlds[(BUF) ^ 1], tid);                                                  \
        pipe_tap(acc, acl, wlds[BUF], tile[BUF], 2, wave, lane);                                          \
        pipe_split_unit<1>(RNEXT, tile[(BUF) ^ 1], tid);                                                  \
        pipe_tap(acc, acl, wlds[BUF], tile[BUF], 3, wave, lane);                                          \
        pipe_put_weight<1>(RNEXT, wlds[(BUF) ^ 1], tid);                                                  \
        pipe_tap(acc, acl, wlds[BUF], tile[BUF], 4, wave, lane);                                          \
        pipe_split_unit<2>(RNEXT, tile[(BUF) ^ 1], tid);                                                  \
        pipe_tap(acc, acl, wlds[BUF], tile[BUF], 5, wave, lane);                                          \
        pipe_put_weight<2>(RNEXT, wlds[(BUF) ^ 1], tid);                                                  \
        pipe_tap(acc, acl, wlds[BUF], tile[BUF], 6, wave, lane);                                          \
        pipe_split_unit<3>(RNEXT, tile[(BUF) ^ 1], tid);                                                  \
        pipe_tap(acc, acl, wlds[BUF], tile[BUF], 7, wave, lane);                                          \
        pipe_tap(acc, acl, wlds[BUF], tile[BUF], 8, wave, lane);                                          \
        CRFP_PST(st_taps)                                                                                 \
        if (++mch == nchunks) {                                                                           \
            const int t_ = blockIdx.x + mk * gridDim.x, ty_ = t_ / tiles_x, tx_ = t_ - ty_ * tiles_x;     \
            _Pragma("unroll") for (int pt = 0; pt < 2; ++pt)                                              \
                _Pragma("unroll") for (int e = 0; e < 16; ++e) acc[0][pt][e] += acl[0][pt][e] * (1.0f / F16_RES_SCALE); \
            conv_epilogue<1, 2, 1, 2>(ec, acc, T0, tx_ * TW, ty_ * PIPE_NW, wave, j, h);                  \
            CRFP_PIPE_ACC_INIT                                                                            \
            mch = 0; ++mk;                                                                                \
        }                                                                                                 \
        CRFP_PST(st_epi)                                                                                  \
        lds_barrier();     /* buffer BUF^1 complete, every wave done with buffer BUF */                  \
    }

    for (int s = 0; s < nitems; s += 2) {
        CRFP_PIPE_ITEM(0, RB, RA)
        if (s + 1 < nitems) CRFP_PIPE_ITEM(1, RA, RB)
    }
#undef CRFP_PIPE_ITEM
#undef CRFP_PIPE_ISSUE
#undef CRFP_PIPE_ACC_INIT
#ifdef CRFP_PIPE_STAMPS
    if (a.stamps && tid == 0) {
        long long* o = a.stamps + ((long long)blockIdx.y * gridDim.x + blockIdx.x) * 8;
        o[0] = st_wait; o[1] = st_taps; o[2] = st_epi; o[3] = st_bar; o[4] = nitems;
    }
    if (a.stamps && lane == 0 && blockIdx.x == 3) {   // per-wave view of one workgroup
        long long* o = a.stamps + (8192 + wave) * 8;
        o[0] = st_wait; o[1] = st_taps; o[2] = st_epi; o[3] = st_bar; o[4] = nitems;
    }
#endif
}

// ---------------------------------------------------------------- input-stationary variant (short K, many couts)
// For convolutions whose whole K fits in LDS (Cin <= 32: the 32->216 offset/mask conv, the pixel-shuffle
// expanders 32->96 / 24->64 / 32->64) the halo tile is staged and split ONCE per workgroup and the
// workgroup then walks all cout tiles, streaming only the packed weights (27 KB per (cout tile, chunk),
// prefetched into registers during the previous step's MFMAs).  The regular kernel re-stages the same
// input once per cout tile and pays its prologue/epilogue bubble 7x for the 216-channel conv.
template <int NCH, int NWAVES, int NP>
__global__ __launch_bounds__(64 * NWAVES, NWAVES == 4 ? 2 : 1) void conv3x3_split_is_kernel(const ConvArgs a) {
    constexpr int RPW = 1, TH = NWAVES, LH = TH + 2, PT = 2, NT = 64 * NWAVES;
    constexpr int NEL = LH * LW;
    constexpr int NIN = (NEL + NT - 1) / NT;
    constexpr int WPC = 9 * NP * 64;
    constexpr int NWS = (WPC + NT - 1) / NT;
    __shared__ bf16x8 tile[NCH][NP][2][NEL];
    __shared__ bf16x8 wlds[WPC];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int tiles_x = (a.W + TW - 1) / TW;
    const int btile = xcd_band_tile(blockIdx.x, gridDim.x);   // XCD x works on a contiguous band of tiles
    const int tx0 = (btile % tiles_x) * TW, ty0 = (btile / tiles_x) * TH;
    const int n = blockIdx.z;
    const int ns = a.src_bgroup > 0 ? n + n / a.src_bgroup : n;   // source batch item (ConvArgs::src_bgroup)
    const int H = a.H, W = a.W;
    const bf16x8* __restrict__ wp = reinterpret_cast<const bf16x8*>(NP == 3 ? a.wsplit : a.wsplit16);
    const int nsteps = a.ctiles * NCH;

    bf16x8 rws[NWS];
#define CRFP_IS_WLOAD(STEP)                                                                               \
    _Pragma("unroll") for (int k = 0; k < NWS; ++k)                                                       \
        rws[k] = wp[(long long)(STEP) * WPC + min(tid + NT * k, WPC - 1)];
    CRFP_IS_WLOAD(0)   // (cout tile 0, chunk 0): packed index (T*nchunks + ch)*WPC == step*WPC

    // ---- stage + split the whole input tile once
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
        f32x4 rq[4][NIN];
        int msk[4];
        bool ok[NIN];
#pragma unroll
        for (int qi = 0; qi < 4; ++qi) {
            const QuadDesc d = a.qd[4 * ch + qi];
            const float* qb = d.base + (long long)ns * d.bstride;
            msk[qi] = d.mask;
#pragma unroll
            for (int t = 0; t < NIN; ++t) {
                const int idx = min(tid + NT * t, NEL - 1);
                const int r = idx / LW, c = idx - r * LW;
                const int gy = ty0 + r - 1, gx = tx0 + c - 1;
                ok[t] = tid + NT * t < NEL && gy >= 0 && gy < H && gx >= 0 && gx < W;
                rq[qi][t] = *reinterpret_cast<const f32x4*>(qb + min(max(gy, 0), H - 1) * d.rs + min(max(gx, 0), W - 1) * d.cs);
            }
        }
#pragma unroll
        for (int t = 0; t < NIN; ++t) {
            const int idx = tid + NT * t;
            if (idx < NEL) {
                bf16x8 pp[NP];
                split_parts<NP>(mask_quad(rq[0][t], ok[t] ? msk[0] : 0), mask_quad(rq[1][t], ok[t] ? msk[1] : 0), pp);
#pragma unroll
                for (int p = 0; p < NP; ++p) tile[ch][p][0][idx] = pp[p];
                split_parts<NP>(mask_quad(rq[2][t], ok[t] ? msk[2] : 0), mask_quad(rq[3][t], ok[t] ? msk[3] : 0), pp);
#pragma unroll
                for (int p = 0; p < NP; ++p) tile[ch][p][1][idx] = pp[p];
            }
        }
    }

    long long tA = 0, tB = 0, tC = 0, tD = 0, t0 = __builtin_amdgcn_s_memtime();
    if (a.stamps) { const long long t = __builtin_amdgcn_s_memtime(); tA += t - t0; t0 = t; }
    f32x16 acc[1][PT], acl[1][PT];
    const EpiCtx ec = epi_ctx(a, n);
    float2 flpre[PT];   // flow of this lane's pixels (ST_OFFMASK): loaded here, not between two epilogues' stores
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
        const int y = min(ty0 + wave * RPW + (pt >> 1), H - 1), x = min(tx0 + (pt & 1) * 32 + j, W - 1);
        flpre[pt] = a.store == ST_OFFMASK ? *reinterpret_cast<const float2*>(ec.flp + ((long long)y * W + x) * 2)
                                          : make_float2(0.0f, 0.0f);
    }
    for (int step = 0; step < nsteps; ++step) {
        const int ch = step % NCH, ct = step / NCH;
        if (ch == 0) {
#pragma unroll
            for (int pt = 0; pt < PT; ++pt)
#pragma unroll
                for (int e = 0; e < 16; ++e) { acc[0][pt][e] = 0.0f; acl[0][pt][e] = 0.0f; }
        }
        __syncthreads();  // all waves done with the previous step's weights (and, first time, tile staged)
#pragma unroll
        for (int k = 0; k < NWS; ++k) {
            const int idx = tid + NT * k;
            if (idx < WPC) wlds[idx] = rws[k];
        }
        __syncthreads();
        if (a.stamps) { const long long t = __builtin_amdgcn_s_memtime(); tB += t - t0; t0 = t; }
        if (step + 1 < nsteps) { CRFP_IS_WLOAD(step + 1) }
#pragma unroll CRFP_TAP_UNROLL
        for (int tap = 0; tap < 9; ++tap) {
            const int ky = tap / 3, kx = tap - 3 * ky;
            bf16x8 wa[NP];
#pragma unroll
            for (int p = 0; p < NP; ++p) wa[p] = wlds[(tap * NP + p) * 64 + lane];
#pragma unroll
            for (int pt = 0; pt < PT; ++pt) {
                const int pix = (wave * RPW + (pt >> 1) + ky) * LW + (pt & 1) * 32 + j + kx;
                bf16x8 bq[NP];
#pragma unroll
                for (int p = 0; p < NP; ++p) bq[p] = tile[ch][p][h][pix];
                split_mfma<NP>(acc[0][pt], acl[0][pt], wa, bq);
            }
        }
        if (a.stamps) { const long long t = __builtin_amdgcn_s_memtime(); tD += t - t0; t0 = t; }
        if (ch == NCH - 1) {
            if (NP == 2) {
#pragma unroll
                for (int pt = 0; pt < PT; ++pt)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[0][pt][e] += acl[0][pt][e] * (1.0f / F16_RES_SCALE);
            }
            conv_epilogue<1, PT, RPW, 0>(ec, acc, ct, tx0, ty0, wave, j, h, flpre);
        }
        if (a.stamps) { const long long t = __builtin_amdgcn_s_memtime(); tC += t - t0; t0 = t; }
    }
    if (a.stamps && tid == 0) {
        long long* o = a.stamps + (long long)blockIdx.x * 4;
        o[0] = tA; o[1] = tB; o[2] = tC; o[3] = tD;
    }
#undef CRFP_IS_WLOAD
}

// ================================================================ warp-specialised split-bf16 convolution
// Measured on the single-role kernels above (s_memtime stamps): the MFMA phase is only 40-50 % of a
// block's life; the rest is (a) the fp32->3xbf16 split + LDS write of the next chunk, which cannot
// overlap the MFMAs of the same waves, and (b) vmcnt being in-order on CDNA: a wave that has epilogue
// stores in flight must drain them before it can consume a prefetched load (12k cycles per cout tile
// in the 216-channel conv).  Here the roles are split:
//   * 8 compute waves (one output row of 64 px each, 2 per SIMD) only ever read LDS, issue MFMAs and
//     fire their epilogue stores -- they never wait on vmcnt inside the main loop;
//   * 4 loader waves (one per SIMD) own every global load: they fetch the next chunk's halo tile (fp32)
//     and packed weights, split the activations into 3 bf16 images and write them into the OTHER half
//     of a double-buffered LDS tile while the compute waves run the current chunk.
// Two workgroup barriers per chunk: X = compute done with the weight image / loaders done with the next
// tile, Y = weight image of this chunk visible.  Weights (27 KB per chunk) are single-buffered: the
// loaders hold them in registers and copy them in between X and Y (~400 idle compute cycles).
// IS = input-stationary form for Cin <= 32 and many couts: both LDS tile halves hold the (at most two)
// K-chunks for the whole block and the loop runs over (cout tile, chunk) steps streaming only weights.
constexpr int WS_NC = 8, WS_NL = 8, WS_NT = 64 * (WS_NC + WS_NL), WS_TH = 8, WS_LH = WS_TH + 2, WS_NEL = WS_LH * LW;

template <bool IS>
__global__ __launch_bounds__(WS_NT, 1) void conv3x3_split_ws_kernel(const ConvArgs a) {
    constexpr int NLT = 64 * WS_NL;                       // loader threads
    constexpr int NIN = (WS_NEL + NLT - 1) / NLT;         // halo pixels per loader thread (3)
    constexpr int NWS = (27 * 64 + NLT - 1) / NLT;        // weight vectors per loader thread (7)
    __shared__ bf16x8 tile[2][3][2][WS_NEL];
    __shared__ bf16x8 wlds[27 * 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tiles_x = (a.W + TW - 1) / TW;
    const int tx0 = (blockIdx.x % tiles_x) * TW, ty0 = (blockIdx.x / tiles_x) * WS_TH;
    const int n = blockIdx.z;
    const int ns = a.src_bgroup > 0 ? n + n / a.src_bgroup : n;   // source batch item (ConvArgs::src_bgroup)
    const int H = a.H, W = a.W;
    const int nchunks = a.kq >> 2;
    const int T0 = IS ? 0 : blockIdx.y;
    const int nsteps = IS ? a.ctiles * nchunks : nchunks;   // IS: step = ct*nchunks + ch

    if (wave >= WS_NC) {
        // ------------------------------------------------------------ loader role
        const int lt = tid - 64 * WS_NC;
        const bf16x8* __restrict__ wp = reinterpret_cast<const bf16x8*>(a.wsplit);
        int cgy[NIN], cgx[NIN];
        bool sval[NIN];
#pragma unroll
        for (int t = 0; t < NIN; ++t) {
            const int idx = min(lt + NLT * t, WS_NEL - 1);
            const int r = idx / LW, c = idx - r * LW;
            const int gy = ty0 + r - 1, gx = tx0 + c - 1;
            sval[t] = lt + NLT * t < WS_NEL && gy >= 0 && gy < H && gx >= 0 && gx < W;
            cgy[t] = min(max(gy, 0), H - 1);
            cgx[t] = min(max(gx, 0), W - 1);
        }
        // Loader schedule, two register sets (A, B): right after barrier Y(s) the loads of chunk s+2 are
        // issued into the set that was just consumed, THEN the other set (chunk s+1, issued a whole step
        // earlier) is split and written to the idle tile half.  Every load is unconditional (chunk index
        // clamped) so that the in-order vmcnt counts are compile-time constants and the compiler can wait
        // for "all but the loads just issued" instead of vmcnt(0).
        f32x4 qa0[NIN], qa1[NIN], qa2[NIN], qa3[NIN], qb0[NIN], qb1[NIN], qb2[NIN], qb3[NIN];
        bf16x8 wa_[NWS], wb_[NWS];
        int ma0 = 0, ma1 = 0, ma2 = 0, ma3 = 0, mb0 = 0, mb1 = 0, mb2 = 0, mb3 = 0;
#define CRFP_WS_LOAD_IN(R0, R1, R2, R3, M0, M1, M2, M3, CH)                                               \
        {                                                                                                 \
            const QuadDesc d0 = a.qd[4 * (CH)], d1 = a.qd[4 * (CH) + 1], d2 = a.qd[4 * (CH) + 2],         \
                           d3 = a.qd[4 * (CH) + 3];                                                       \
            const float* b0 = d0.base + (long long)ns * d0.bstride;                                        \
            const float* b1 = d1.base + (long long)ns * d1.bstride;                                        \
            const float* b2 = d2.base + (long long)ns * d2.bstride;                                        \
            const float* b3 = d3.base + (long long)ns * d3.bstride;                                        \
            M0 = d0.mask; M1 = d1.mask; M2 = d2.mask; M3 = d3.mask;                                       \
            _Pragma("unroll") for (int t = 0; t < NIN; ++t) {                                             \
                R0[t] = *reinterpret_cast<const f32x4*>(b0 + cgy[t] * d0.rs + cgx[t] * d0.cs);            \
                R1[t] = *reinterpret_cast<const f32x4*>(b1 + cgy[t] * d1.rs + cgx[t] * d1.cs);            \
                R2[t] = *reinterpret_cast<const f32x4*>(b2 + cgy[t] * d2.rs + cgx[t] * d2.cs);            \
                R3[t] = *reinterpret_cast<const f32x4*>(b3 + cgy[t] * d3.rs + cgx[t] * d3.cs);            \
            }                                                                                             \
        }
#define CRFP_WS_LOAD_W(RW, WSTEP)                                                                         \
        _Pragma("unroll") for (int k = 0; k < NWS; ++k)                                                   \
            RW[k] = wp[(long long)(WSTEP) * 1728 + min(lt + NLT * k, 1727)];
#define CRFP_WS_WRITE_IN(R0, R1, R2, R3, M0, M1, M2, M3, BUF)                                             \
        _Pragma("unroll") for (int t = 0; t < NIN; ++t) {                                                 \
            const int idx = lt + NLT * t;                                                                 \
            if (idx < WS_NEL) {                                                                           \
                bf16x8 p0, p1, p2;                                                                        \
                split_bf16x8(mask_quad(R0[t], sval[t] ? M0 : 0), mask_quad(R1[t], sval[t] ? M1 : 0), p0, p1, p2); \
                tile[BUF][0][0][idx] = p0; tile[BUF][1][0][idx] = p1; tile[BUF][2][0][idx] = p2;          \
                split_bf16x8(mask_quad(R2[t], sval[t] ? M2 : 0), mask_quad(R3[t], sval[t] ? M3 : 0), p0, p1, p2); \
                tile[BUF][0][1][idx] = p0; tile[BUF][1][1][idx] = p1; tile[BUF][2][1][idx] = p2;          \
            }                                                                                             \
        }
#define CRFP_WS_WRITE_W(RW)                                                                               \
        _Pragma("unroll") for (int k = 0; k < NWS; ++k) {                                                 \
            const int idx = lt + NLT * k;                                                                 \
            if (idx < 1728) wlds[idx] = RW[k];                                                            \
        }
        // packed weight image of (cout tile T, chunk ch) starts at ((T*nchunks + ch)*27)*64 vectors
        const long long wbase = (long long)T0 * nchunks;
        const int last = nsteps - 1, lastc = nchunks - 1;
        if (IS) {
            CRFP_WS_LOAD_IN(qa0, qa1, qa2, qa3, ma0, ma1, ma2, ma3, 0)
            CRFP_WS_LOAD_W(wa_, wbase)
            CRFP_WS_WRITE_IN(qa0, qa1, qa2, qa3, ma0, ma1, ma2, ma3, 0)
            if (nchunks > 1) {
                CRFP_WS_LOAD_IN(qa0, qa1, qa2, qa3, ma0, ma1, ma2, ma3, 1)
                CRFP_WS_WRITE_IN(qa0, qa1, qa2, qa3, ma0, ma1, ma2, ma3, 1)
            }
            for (int step = 0; step < nsteps; ++step) {
                __syncthreads();  // X
                CRFP_WS_WRITE_W(wa_)
                __syncthreads();  // Y
                CRFP_WS_LOAD_W(wa_, wbase + min(step + 1, last))
            }
        } else {
            CRFP_WS_LOAD_IN(qa0, qa1, qa2, qa3, ma0, ma1, ma2, ma3, 0)
            CRFP_WS_LOAD_W(wa_, wbase)
            CRFP_WS_LOAD_IN(qb0, qb1, qb2, qb3, mb0, mb1, mb2, mb3, min(1, lastc))
            CRFP_WS_LOAD_W(wb_, wbase + min(1, last))
            CRFP_WS_WRITE_IN(qa0, qa1, qa2, qa3, ma0, ma1, ma2, ma3, 0)
            // top of an even step s: tile[s&1] = chunk s, wa_ = weights(s), set B = chunk s+1 (in flight)
            long long sA = 0, sB = 0, sC = 0, sD = 0, t0 = __builtin_amdgcn_s_memtime();
#define CRFP_ST(ACC) if (a.stamps) { const long long t_ = __builtin_amdgcn_s_memtime(); ACC += t_ - t0; t0 = t_; }
            for (int step = 0; step < nsteps; step += 2) {
                __syncthreads();  // X: compute done with wlds and with tile[(step+1)&1]
                CRFP_ST(sD)
                CRFP_WS_WRITE_W(wa_)
                __syncthreads();  // Y
                CRFP_ST(sC)
                CRFP_WS_LOAD_IN(qa0, qa1, qa2, qa3, ma0, ma1, ma2, ma3, min(step + 2, lastc))
                CRFP_WS_LOAD_W(wa_, wbase + min(step + 2, last))
                CRFP_ST(sA)
                if (step + 1 >= nsteps) break;
                CRFP_WS_WRITE_IN(qb0, qb1, qb2, qb3, mb0, mb1, mb2, mb3, (step + 1) & 1)
                CRFP_ST(sB)
                __syncthreads();  // X
                CRFP_ST(sD)
                CRFP_WS_WRITE_W(wb_)
                __syncthreads();  // Y
                CRFP_ST(sC)
                CRFP_WS_LOAD_IN(qb0, qb1, qb2, qb3, mb0, mb1, mb2, mb3, min(step + 3, lastc))
                CRFP_WS_LOAD_W(wb_, wbase + min(step + 3, last))
                CRFP_ST(sA)
                if (step + 2 < nsteps) CRFP_WS_WRITE_IN(qa0, qa1, qa2, qa3, ma0, ma1, ma2, ma3, step & 1)
                CRFP_ST(sB)
            }
            if (a.stamps && lt == 0) {
                long long* o = a.stamps + ((long long)blockIdx.y * gridDim.x + blockIdx.x) * 4 + 4 * 8192;
                o[0] = sA; o[1] = sB; o[2] = sC; o[3] = sD;
            }
#undef CRFP_ST
        }
#undef CRFP_WS_LOAD_IN
#undef CRFP_WS_LOAD_W
#undef CRFP_WS_WRITE_IN
#undef CRFP_WS_WRITE_W
        return;
    }

    // ---------------------------------------------------------------- compute role: wave = output row
    const int j = lane & 31, h = lane >> 5;
    f32x16 acc[1][2];
    const EpiCtx ec = epi_ctx(a, n);
    float2 flpre[2] = {make_float2(0.0f, 0.0f), make_float2(0.0f, 0.0f)};
    if (a.store == ST_OFFMASK) {
        const int y = min(ty0 + wave, H - 1);
#pragma unroll
        for (int pt = 0; pt < 2; ++pt)
            flpre[pt] = *reinterpret_cast<const float2*>(a.flow + (long long)n * a.flow_bstride +
                                                         ((long long)y * W + min(tx0 + pt * 32 + j, W - 1)) * 2);
    }
    long long tA = 0, tB = 0, tC = 0, tD = 0, t0 = __builtin_amdgcn_s_memtime();
    for (int step = 0; step < nsteps; ++step) {
        const int ch = IS ? step % nchunks : step;
        const int buf = IS ? ch : (step & 1);
        if (!IS ? step == 0 : ch == 0) {
#pragma unroll
            for (int pt = 0; pt < 2; ++pt)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[0][pt][e] = 0.0f;
        }
        __syncthreads();  // X
        if (a.stamps) { const long long t = __builtin_amdgcn_s_memtime(); tA += t - t0; t0 = t; }
        __syncthreads();  // Y
        if (a.stamps) { const long long t = __builtin_amdgcn_s_memtime(); tB += t - t0; t0 = t; }
#pragma unroll CRFP_TAP_UNROLL
        for (int tap = 0; tap < 9; ++tap) {
            const int ky = tap / 3, kx = tap - 3 * ky;
            const bf16x8 w0 = wlds[(tap * 3 + 0) * 64 + lane], w1 = wlds[(tap * 3 + 1) * 64 + lane],
                         w2 = wlds[(tap * 3 + 2) * 64 + lane];
#pragma unroll
            for (int pt = 0; pt < 2; ++pt) {
                const int pix = (wave + ky) * LW + pt * 32 + j + kx;
                const bf16x8 b0 = tile[buf][0][h][pix], b1 = tile[buf][1][h][pix], b2 = tile[buf][2][h][pix];
                f32x16 c = acc[0][pt];
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1, b1, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0, b2, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w2, b0, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0, b1, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1, b0, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0, b0, c, 0, 0, 0);
                acc[0][pt] = c;
            }
        }
        if (a.stamps) { const long long t = __builtin_amdgcn_s_memtime(); tD += t - t0; t0 = t; }
        if (IS && ch == nchunks - 1) conv_epilogue<1, 2, 1, 0>(ec, acc, step / nchunks, tx0, ty0, wave, j, h, flpre);
        if (a.stamps) { const long long t = __builtin_amdgcn_s_memtime(); tC += t - t0; t0 = t; }
    }
    if (a.stamps && tid == 0) {
        long long* o = a.stamps + ((long long)blockIdx.y * gridDim.x + blockIdx.x) * 4;
        o[0] = tA; o[1] = tB; o[2] = tC; o[3] = tD;
    }
    if (!IS) conv_epilogue<1, 2, 1, 0>(ec, acc, T0, tx0, ty0, wave, j, h, flpre);
}

#endif  // CRFP_LAB

// split weight pack: wsplit bf16 index =
//   (((((T*nchunks + ch)*9 + tap)*3 + part)*64 + lane)*8 + jj),  lane = half*32 + row,
//   K-quad = 4*ch + 2*half + (jj>>2), component = jj&3
//   np = 3: bf16 triple (bf16x6 scheme); np = 2: fp16 pair, second term scaled by 2^11 (f16x3 scheme), same index with 3 -> 2
__global__ void conv_pack_split_kernel(const ConvArgs a, const float* __restrict__ w, const float* __restrict__ w2,
                                       int cout_split, unsigned short* __restrict__ wsplit, int np) {
    const int nchunks = a.kq >> 2;
    const long long total = (long long)a.ctiles * nchunks * 9 * 64 * 8;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        long long t = idx;
        const int jj = t & 7; t >>= 3;
        const int lane = t & 63; t >>= 6;
        const int tap = t % 9; t /= 9;
        const int ch = t % nchunks;
        const int T = (int)(t / nchunks);
        const int row = lane & 31, half = lane >> 5;
        const int co = conv_row_to_cout(T * 32 + row, a.cout, a.store, a.ps_r);
        int kql = 4 * ch + 2 * half + (jj >> 2), s = 0;
        while (s < a.nsrc - 1 && kql >= a.src[s].nq) { kql -= a.src[s].nq; ++s; }
        int ci = kql < a.src[s].nq ? conv_k_to_cin(a.src[s].kind, a.src[s].nch, kql, jj & 3) : -1;
        if (ci >= 0) ci += a.src[s].cbase;
        float val = 0.0f;
        if (co >= 0 && ci >= 0 && ci < a.cin_total)
            val = co < cout_split ? w[((long long)co * a.cin_total + ci) * 9 + tap]
                                  : w2[((long long)(co - cout_split) * a.cin_total + ci) * 9 + tap];
        const long long base = (((((long long)T * nchunks + ch) * 9 + tap) * np) * 64 + lane) * 8 + jj;
        if (np == 4) {   // f16x3s: A = fp16(2^11 w), B = fp16(2^11 (w - A / 2^11)), C = fp16(w); image stride as for np == 3
            const long long b3 = (((((long long)T * nchunks + ch) * 9 + tap) * 3) * 64 + lane) * 8 + jj;
            const _Float16 pa = (_Float16)(val * 2048.0f);
            const _Float16 pb = (_Float16)((val - (float)pa * (1.0f / 2048.0f)) * 2048.0f);
            wsplit[b3] = __builtin_bit_cast(unsigned short, pa);
            wsplit[b3 + 64 * 8] = __builtin_bit_cast(unsigned short, pb);
            wsplit[b3 + 2 * 64 * 8] = __builtin_bit_cast(unsigned short, (_Float16)val);
        } else if (np == 1) {   // bf16 build: the weight IS its bf16 rounding
            wsplit[base] = __builtin_bit_cast(unsigned short, (__bf16)val);
        } else if (np == 3) {
            const __bf16 p0 = (__bf16)val;
            const float r = val - (float)p0;
            const __bf16 p1 = (__bf16)r;
            const __bf16 p2 = (__bf16)(r - (float)p1);
            wsplit[base] = __builtin_bit_cast(unsigned short, p0);
            wsplit[base + 64 * 8] = __builtin_bit_cast(unsigned short, p1);
            wsplit[base + 2 * 64 * 8] = __builtin_bit_cast(unsigned short, p2);
        } else {
            const _Float16 p0 = (_Float16)val;
            const _Float16 p1 = (_Float16)((val - (float)p0) * 2048.0f);   // F16_RES_SCALE
            wsplit[base] = __builtin_bit_cast(unsigned short, p0);
            wsplit[base + 64 * 8] = __builtin_bit_cast(unsigned short, p1);
        }
    }
}

// product build: the fp16 pair image only; lab build: bf16 triple image followed by the fp16 pair image
#ifdef CRFP_LAB
size_t conv_split_weight_bytes(const ConvArgs& a) { return (size_t)a.ctiles * (a.kq >> 2) * 9 * (3 + 2 + 3) * 64 * 16; }
size_t conv_split16_offset_bytes(const ConvArgs& a) { return (size_t)a.ctiles * (a.kq >> 2) * 9 * 3 * 64 * 16; }
#elif defined(CRFP_ACT_BF16)
size_t conv_split_weight_bytes(const ConvArgs& a) { return (size_t)a.ctiles * (a.kq >> 2) * 9 * 1 * 64 * 16; }   // one bf16 image
size_t conv_split16_offset_bytes(const ConvArgs&) { return 0; }
#else
// the fp16 pair image (4-wave kernel), then the three images of the single-accumulator scheme (8-wave kernel)
size_t conv_split_weight_bytes(const ConvArgs& a) { return (size_t)a.ctiles * (a.kq >> 2) * 9 * (2 + 3) * 64 * 16; }
size_t conv_split16_offset_bytes(const ConvArgs&) { return 0; }
#endif
#ifndef CRFP_ACT_BF16
static size_t conv_split_sa_offset_bytes(const ConvArgs& a) {
#ifdef CRFP_LAB
    return (size_t)a.ctiles * (a.kq >> 2) * 9 * 5 * 64 * 16;
#else
    return (size_t)a.ctiles * (a.kq >> 2) * 9 * 2 * 64 * 16;
#endif
}
#endif

int launch_conv_pack_split(const ConvArgs& a, const float* w, const float* w2, int cout_split, void* wsplit,
                           hipStream_t s) {
    if (a.kq & 3) { set_error("conv_pack_split: kq %d not a multiple of 4", a.kq); return CRFP_E_BADARG; }
    const long long total = (long long)a.ctiles * (a.kq >> 2) * 9 * 64 * 8;
    const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
#ifdef CRFP_LAB
    conv_pack_split_kernel<<<blocks, 256, 0, s>>>(a, w, w2, w2 ? cout_split : a.cout, (unsigned short*)wsplit, 3);
    CRFP_CHECK_LAUNCH();
#endif
    conv_pack_split_kernel<<<blocks, 256, 0, s>>>(a, w, w2, w2 ? cout_split : a.cout,
                                                  (unsigned short*)((char*)wsplit + conv_split16_offset_bytes(a)), kActBf16 ? 1 : 2);
    CRFP_CHECK_LAUNCH();
#ifndef CRFP_ACT_BF16
    conv_pack_split_kernel<<<blocks, 256, 0, s>>>(a, w, w2, w2 ? cout_split : a.cout,
                                                  (unsigned short*)((char*)wsplit + conv_split_sa_offset_bytes(a)), 4);
    CRFP_CHECK_LAUNCH();
#endif
    return 0;
}

// ---------------------------------------------------------------- weight packing (device side)
// wpk float index = ((((T*npairs + pair)*9 + tap)*2 + half)*32 + row)*4 + comp
// Rows whose reference channel is >= cout_split come from a second weight tensor (w2/bias2): used to
// run dcn_offset and dcn_mask (same input, model/CRFP.py:337,339) as ONE convolution.
__global__ void conv_pack_kernel(const ConvArgs a, const float* __restrict__ w, const float* __restrict__ bias,
                                 const float* __restrict__ w2, const float* __restrict__ bias2, int cout_split,
                                 float* __restrict__ wpk, float* __restrict__ bpk) {
    const int npairs = a.kq >> 1;
    const long long total = (long long)a.ctiles * npairs * 9 * 64 * 4;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        long long t = idx;
        const int comp = t & 3; t >>= 2;
        const int row = t & 31; t >>= 5;
        const int half = t & 1; t >>= 1;
        const int tap = t % 9; t /= 9;
        const int pair = t % npairs;
        const int T = (int)(t / npairs);
        const int co = conv_row_to_cout(T * 32 + row, a.cout, a.store, a.ps_r);
        int kql = 2 * pair + half, s = 0;
        while (s < a.nsrc - 1 && kql >= a.src[s].nq) { kql -= a.src[s].nq; ++s; }
        int ci = kql < a.src[s].nq ? conv_k_to_cin(a.src[s].kind, a.src[s].nch, kql, comp) : -1;
        if (ci >= 0) ci += a.src[s].cbase;
        float val = 0.0f;
        if (co >= 0 && ci >= 0 && ci < a.cin_total)
            val = co < cout_split ? w[((long long)co * a.cin_total + ci) * 9 + tap]
                                  : w2[((long long)(co - cout_split) * a.cin_total + ci) * 9 + tap];
        if (kActBf16) val = (float)(__bf16)val;   // bf16 build: every conv weight of the engine is a bf16 value
        wpk[idx] = val;
    }
    const int nb = a.ctiles * 32;
    for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < nb; r += gridDim.x * blockDim.x) {
        const int co = conv_row_to_cout(r, a.cout, a.store, a.ps_r);
        bpk[r] = co < 0 ? 0.0f : (co < cout_split ? (bias ? bias[co] : 0.0f) : bias2[co - cout_split]);
    }
}

size_t conv_packed_weight_floats(const ConvArgs& a) { return (size_t)a.ctiles * (a.kq >> 1) * 9 * 64 * 4; }

int launch_conv_pack(const ConvArgs& a, const float* w, const float* bias, const float* w2, const float* bias2,
                     int cout_split, float* wpk, float* bpk, hipStream_t s) {
    const long long total = (long long)conv_packed_weight_floats(a);
    const int blocks = (int)((total + 255) / 256 < 1024 ? (total + 255) / 256 : 1024);
    conv_pack_kernel<<<blocks, 256, 0, s>>>(a, w, bias, w2, bias2, w2 ? cout_split : a.cout, wpk, bpk);
    CRFP_CHECK_LAUNCH();
    return 0;
}

// Precision of the engine's wide convolutions and of dcn_g8's GEMM.  Default: the split-fp16 scheme (fp32-grade, needs
// |operand| < 65504, guarded by the overflow word, see ConvArgs::ovf).  Strict: plain fp32 MFMA everywhere -- selected per
// call through ConvArgs::strict (CRFP_DSV_STRICT_F32 of the C-ABI) or for the whole process with CRFP_PRECISION=f32
// (read once; the per-family knobs of round 1, CRFP_CONV_MODE=f32 / CRFP_DCN_MODE=f32, are read by the lab library only).
#ifndef CRFP_ACT_BF16

}  // namespace CRFP_NS
namespace crfp {
bool precision_env_strict(int family) {
    const char* p = getenv("CRFP_PRECISION");
    if (p && !strcmp(p, "f32")) return true;
#ifdef CRFP_LAB   // the round-1 per-family knobs (CRFP_CONV_MODE / CRFP_DCN_MODE) live on in the lab library only
    const char* l = getenv(family == 0 ? "CRFP_CONV_MODE" : "CRFP_DCN_MODE");
    return l && !strcmp(l, "f32");
#else
    (void)family;
    return false;
#endif
}
}  // namespace crfp
namespace CRFP_NS {
#endif

#ifdef CRFP_LAB
static const char* lab_conv_mode() { static const char* m = getenv("CRFP_CONV_MODE"); return m ? m : "f16x3"; }
static int lab_knob(const char* k, int dflt) { const char* v = getenv(k); return v ? atoi(v) : dflt; }
#endif

// SRC_S3 sources / s3_dst are understood by conv3x3_split_kernel<1,1,2> (the default) only
bool conv_s3_supported() {
    static const bool ok = [] {
        if (kActBf16 || precision_env_strict(0)) return false;
#ifdef CRFP_LAB
        if (strcmp(lab_conv_mode(), "f16x3")) return false;
        if (lab_knob("CRFP_SPLIT_WS", 0) || lab_knob("CRFP_SPLIT_IS", 0) || lab_knob("CRFP_SPLIT_PIPE", 0)) return false;
        if (lab_knob("CRFP_SPLIT_RPW", 1) != 1 || lab_knob("CRFP_SPLIT_CT", 1) != 1 || !lab_knob("CRFP_CONV_S3", 1)) return false;
#endif
        return true;
    }();
    return ok;
}

static bool uses_s3_dst_only_4wave(const ConvArgs&) { return false; }   // the shared epilogue writes S3 images from either kernel

// per-quad load descriptors (wave-uniform in the kernels: one s_load per quad); filled into the caller's private plan copy
static int build_quad_descs(ConvArgs& am, const char* name) {
    const ConvArgs& a = am;
    {
        int q = 0;
        for (int i = 0; i < a.nsrc; ++i)
            for (int k = 0; k < a.src[i].nq; ++k, ++q) {
                const ConvSrc& sr = a.src[i];
                QuadDesc& d = am.qd[q];
                d.bstride = sr.bstride; d.rsv = 0;
                // strides and offsets are in FLOATS (4 bytes) for both builds: an activation quad spans kQuadBytes / 4 of them
                constexpr int QF = kQuadBytes / 4;
                if (sr.kind == SRC_Q4) {
                    d.bstride = sr.bstride * (long long)sizeof(act_t) / 4;
                    d.rs = (a.W + sr.pad) * QF; d.cs = QF; d.mask = 15;
                    d.base = sr.p + (long long)k * (a.H + sr.pad) * d.rs;
                } else if (sr.kind == SRC_UNSHUF4) {
                    const int W4 = 4 * a.W + sr.pad, ij = k & 15;
                    d.bstride = sr.bstride * (long long)sizeof(act_t) / 4;
                    d.rs = 4 * W4 * QF; d.cs = 4 * QF; d.mask = 15;
                    d.base = sr.p + ((long long)(k >> 4) * (4 * a.H + sr.pad) * W4 + (ij >> 2) * W4 + (ij & 3)) * QF;
                } else if (sr.kind == SRC_FLOW2) {
                    d.rs = 2 * a.W; d.cs = 2; d.mask = kActBf16 ? (32 | 3) : 3; d.base = sr.p;   // bit 5: fp32 (dx, dy) pair
                } else if (sr.kind == SRC_S3) {
                    // chunk-local quad qi -> 16-byte plane: qi 0/1 = x0 image, channels 0..7 / 8..15 of the chunk; qi 2/3 = x1s image
                    if ((q & 3) != (k & 3) || (sr.nq & 3) || sr.pad) {
                        set_error("conv_mfma %s: an SRC_S3 source must start on a 16-channel chunk, hold whole chunks and be unpadded", name);
                        return CRFP_E_BADARG;
                    }
                    const int qi = k & 3, plane = (qi >> 1) * (sr.nq >> 1) + 2 * (k >> 2) + (qi & 1);
                    d.rs = a.W * 4; d.cs = 4; d.mask = 16 | 15;
                    d.base = sr.p + (long long)plane * a.H * d.rs;
                } else {
                    d.rs = 0; d.cs = 0; d.mask = 0; d.base = a.wpk; d.bstride = 0;
                }
            }
    }
    return 0;
}

int launch_conv_mfma(const ConvArgs& a, const char* name, hipStream_t s) {
    if (a.kq & 1 || a.kq < 2 || a.ctiles < 1 || a.nsrc < 1 || a.nsrc > CRFP_MAX_SRC) {
        set_error("conv_mfma %s: bad plan (kq=%d ctiles=%d nsrc=%d)", name, a.kq, a.ctiles, a.nsrc);
        return CRFP_E_BADARG;
    }
    if (a.store == ST_PS && ((a.ps_r != 2 && a.ps_r != 4) || a.act == CRFP_ACT_TANH || a.act == CRFP_ACT_SIGMOID)) {
        set_error("conv_mfma %s: pixel-shuffle store supports r in {2, 4} with none/relu/lrelu (r=%d act=%d)", name, a.ps_r, a.act);
        return CRFP_E_UNSUPPORTED;
    }
    if (a.ksplit > 0) {
        // K split over blockIdx.z exists in ONE launch branch (the fp32-MFMA kernel over SRC_NCHW_SHIFT views, two K-quads per step) and
        // writes raw partial sums: anything else would drop trailing chunks, apply bias / activation once per slice, or keep grid.z = N
        bool shift_only = true;
        for (int i = 0; i < a.nsrc; ++i) shift_only = shift_only && a.src[i].kind == SRC_NCHW_SHIFT;
        if (!shift_only || ((a.kq >> 1) % a.ksplit) != 0 || a.act != CRFP_ACT_NONE || a.post_scale != 1.0f || a.resid || a.store != ST_NCHW) {
            set_error("conv_mfma %s: ksplit = %d needs SRC_NCHW_SHIFT sources only, (kq / 2) %% ksplit == 0 (kq = %d), no activation / scale / residual "
                      "and an NCHW store of the partial sums", name, a.ksplit, a.kq);
            return CRFP_E_UNSUPPORTED;
        }
    }
    static const bool env_strict = precision_env_strict(0);
    const bool use_split = !(env_strict || a.strict);
    bool ct2 = a.ctiles % 2 == 0;
    bool nchw_src = false, s3_src = false;
    for (int i = 0; i < a.nsrc; ++i) { nchw_src |= a.src[i].kind == SRC_NCHW || a.src[i].kind == SRC_NCHW_SHIFT; s3_src |= a.src[i].kind == SRC_S3; }
    const bool split = a.wsplit && use_split && (a.kq & 3) == 0 && !nchw_src && a.kq <= CRFP_MAX_KQ;
    const bool uses_s3 = a.s3_dst != nullptr || s3_src;
    if (uses_s3 && (!split || !conv_s3_supported() || (a.s3_dst && (a.store != ST_Q4 || (a.cout & 7))))) {
        set_error("conv_mfma %s: SRC_S3 / s3_dst need the default f16x3 kernel, ST_Q4 and cout %% 8 == 0", name);
        return CRFP_E_UNSUPPORTED;
    }
    int split_rpw = 1;
#ifdef CRFP_LAB
    split_rpw = lab_knob("CRFP_SPLIT_RPW", 1);
    const int split_ct = lab_knob("CRFP_SPLIT_CT", 1);
    if (lab_knob("CRFP_CONV_CT", 2) < 2) ct2 = false;
    if (split && split_ct < 2) ct2 = false;
#else
    if (split) ct2 = false;  // <2,1> needs 93 KB of LDS (1 workgroup per CU): slower than 2 x <1,1>
#endif
    const int TH = (ct2 || (split && split_rpw == 1)) ? 4 : 8;
    const int tiles = ((a.W + TW - 1) / TW) * ((a.H + TH - 1) / TH);
    const double px = (double)a.N * a.H * a.W;
    double in_ch = 0;
    for (int i = 0; i < a.nsrc; ++i) in_ch += a.src[i].kind == SRC_ZERO ? 0 : a.src[i].nch;
    ProfScope prof(name, s, px * (in_ch + a.cout) * (double)sizeof(act_t) + (double)a.cout * in_ch * 9 * 4.0,
                   2.0 * px * a.cout * in_ch * 9.0);
    ConvArgs& am = const_cast<ConvArgs&>(a);  // callers pass a private, mutable plan copy
    am.stamps = nullptr;
#ifdef CRFP_LAB
    // diagnostic: CRFP_STAMP_PTR=<device address> CRFP_STAMP_NAME=<launch site> records phase cycles per block
    static const char* stamp_name = getenv("CRFP_STAMP_NAME");
    static long long* stamp_ptr = getenv("CRFP_STAMP_PTR") ? (long long*)strtoull(getenv("CRFP_STAMP_PTR"), nullptr, 0) : nullptr;
    am.stamps = (stamp_ptr && stamp_name && !strcmp(stamp_name, name)) ? stamp_ptr : nullptr;
#endif
    am.wsplit16 = a.wsplit ? (const char*)a.wsplit + conv_split16_offset_bytes(a) : nullptr;
#ifndef CRFP_ACT_BF16
    am.wsplit_sa = a.wsplit ? (const char*)a.wsplit + conv_split_sa_offset_bytes(a) : nullptr;
#endif
    if (env_strict || a.strict) am.ovf = nullptr;   // nothing downstream turns this output into an fp16 operand
    if (a.kq <= CRFP_MAX_KQ) {
        const int rc = build_quad_descs(am, name);
        if (rc) return rc;
    }
#ifdef CRFP_LAB
    const bool use_f16 = strcmp(lab_conv_mode(), "bf16x6") != 0;
    // warp-specialised variant: measured 346 vs 351.5 frames/s for the single-role kernels (loader issue is
    // throttled by the ~12 B/clk/CU the memory system delivers) -> kept as an opt-in experiment
    const bool use_ws = lab_knob("CRFP_SPLIT_WS", 0) == 1;
    // input-stationary variant: wins for bf16x6 (65 KB workgroups, 2 per CU); with f16x3 the plain kernel runs 3 workgroups
    // per CU and is faster even for the 216-channel conv (139.8 vs 147.6 us), so it is opt-in there
    const bool use_is = lab_knob("CRFP_SPLIT_IS", use_f16 ? 0 : 1) != 0;
    const bool use_pipe = lab_knob("CRFP_SPLIT_PIPE", 0) == 1;
    const int pipe_wgs = lab_knob("CRFP_PIPE_WGS", 256);
    if (split && use_ws) {
        const int wtiles = ((a.W + TW - 1) / TW) * ((a.H + WS_TH - 1) / WS_TH);
        if (use_is && a.kq <= 8 && a.ctiles >= 2) {
            conv3x3_split_ws_kernel<true><<<dim3(wtiles, 1, a.N), WS_NT, 0, s>>>(a);
        } else {
            conv3x3_split_ws_kernel<false><<<dim3(wtiles, a.ctiles, a.N), WS_NT, 0, s>>>(a);
        }
        CRFP_CHECK_LAUNCH();
        return 0;
    }
    if (split && use_is && a.kq <= 8 && a.ctiles >= 2) {
        // input-stationary: whole K in LDS, one workgroup per 4x64 tile walks every cout tile
        dim3 grid(((a.W + TW - 1) / TW) * ((a.H + 7) / 8), 1, a.N);
        const int is_waves = lab_knob("CRFP_IS_WAVES", 8);
        if (use_f16 && is_waves == 4) {
            dim3 grid4(((a.W + TW - 1) / TW) * ((a.H + 3) / 4), 1, a.N);
            if (a.kq == 4) conv3x3_split_is_kernel<1, 4, 2><<<grid4, 256, 0, s>>>(am);
            else conv3x3_split_is_kernel<2, 4, 2><<<grid4, 256, 0, s>>>(am);
        } else if (use_f16) {
            if (a.kq == 4) conv3x3_split_is_kernel<1, 8, 2><<<grid, 512, 0, s>>>(am);
            else conv3x3_split_is_kernel<2, 8, 2><<<grid, 512, 0, s>>>(am);
        } else {
            if (a.kq == 4) conv3x3_split_is_kernel<1, 8, 3><<<grid, 512, 0, s>>>(am);
            else conv3x3_split_is_kernel<2, 8, 3><<<grid, 512, 0, s>>>(am);
        }
        CRFP_CHECK_LAUNCH();
        return 0;
    }
    if (split && use_pipe) {
        // persistent: one workgroup per CU walks tiles blockIdx.x, blockIdx.x + gridDim.x, ...
        const int ntl = ((a.W + TW - 1) / TW) * ((a.H + PIPE_NW - 1) / PIPE_NW);
        const int per = (ntl + pipe_wgs - 1) / pipe_wgs;           // tiles per workgroup
        dim3 grid((ntl + per - 1) / per, a.ctiles, a.N);           // balanced shares
        conv3x3_split_pipe_kernel<<<grid, PIPE_NT, 0, s>>>(a);
        CRFP_CHECK_LAUNCH();
        return 0;
    }
    if (split && (ct2 || split_rpw != 1 || !use_f16)) {
        if (ct2) {
            dim3 grid(tiles * (a.ctiles / 2), 1, a.N);
            if (use_f16) conv3x3_split_kernel<2, 1, 2><<<grid, 256, 0, s>>>(am);
            else conv3x3_split_kernel<2, 1, 3><<<grid, 256, 0, s>>>(am);
        } else if (split_rpw == 1) {
            conv3x3_split_kernel<1, 1, 3><<<dim3(tiles * a.ctiles, 1, a.N), 256, 0, s>>>(am);
        } else {
            dim3 grid(tiles * a.ctiles, 1, a.N);
            if (use_f16) conv3x3_split_kernel<1, 2, 2><<<grid, 256, 0, s>>>(am);
            else conv3x3_split_kernel<1, 2, 3><<<grid, 256, 0, s>>>(am);
        }
        CRFP_CHECK_LAUNCH();
        return 0;
    }
#endif
    if (split) {
#ifdef CRFP_ACT_BF16
        // (8-row tiles, conv3x3_bf16_kernel<2>: 1.25 instead of 1.5 ds_read_b128 per MFMA and half the weight staging, but 450
        // workgroups on 256 CUs -- measured neutral to -8 % per conv, so 4-row tiles stay.  A register prefetch two chunks deep
        // (126 VGPRs, still 4 workgroups per CU) was also slower: 32->32 conv 15.2 -> 16.3 us, offset / mask head 70.8 -> 74.0.)
        // 8-wave kernel, two chunks per staging phase, for the 32-cout layers with an even chunk count: same-box 32 -> 32 convs 16.0 ->
        // 15.1 us, conv2(+x) 17.7 -> 17.0, conv_fuse 22.2 -> 21.6, identical values; block0 (5 chunks: a half-empty last stage) 25.4 -> 26.0,
        // so odd chunk counts keep the 4-wave kernel
        // (for layers with several cout tiles it loses: FNet dec2a 36 -> 43 us, enc3b 22 -> 26, the pixel-shuffle heads +0..1 us)
        // ... and only while its 8-row tiles fill no more than one round of the chip's 512 slots (two 60 KB workgroups per CU): a lock-step
        // batch of clips is several rounds, where tile granularity no longer matters and the 4-wave kernel's four workgroups per CU overlap
        // more of each other's load / MFMA / store phases (round 4, same box, 4 clips: res.main0 15.1 -> 14.1 us per clip, conv_fuse 14.7 -> 14.0,
        // conv1 9.9 -> 9.2; CRFP_BF16_X8_MAX_WGS overrides the threshold for A/B runs)
#ifdef CRFP_LAB
        static const int x8_max_wgs = getenv("CRFP_BF16_X8_MAX_WGS") ? atoi(getenv("CRFP_BF16_X8_MAX_WGS")) : 512;   // A/B knob, lab library only
#else
        constexpr int x8_max_wgs = 512;   // fixed in the product: the x8 / 4-wave choice is part of the per-clip bit-identity contract
#endif
        if (a.ctiles == 1 && ((a.kq >> 2) & 1) == 0 && (long long)a.N * ((a.W + TW - 1) / TW) * ((a.H + B8_TH - 1) / B8_TH) <= x8_max_wgs) {
            const int tiles8 = ((a.W + TW - 1) / TW) * ((a.H + B8_TH - 1) / B8_TH);
            conv3x3_bf16x8_kernel<<<dim3(tiles8 * a.ctiles, 1, a.N), B8_NT, 0, s>>>(am);
        } else
            conv3x3_bf16_kernel<1><<<dim3(tiles * a.ctiles, 1, a.N), 256, 0, s>>>(am);
#else
        // 8-wave single-accumulator kernel for the convs with one cout tile (the 32-cout layers: one round of 450 workgroups
        // instead of 1.17 rounds of 900; same-box: conv1 26.4 -> 24.6 us, conv2 28.7 -> 26.1, block0 46.0 -> 43.2, main0 40.0 ->
        // 38.0, clip -1.5 %); with several cout tiles the 4-wave kernel stays (offset/mask head 114.0 vs 117.7 us)
#ifdef CRFP_LAB
        static const int s8_max_wgs = getenv("CRFP_F32_S8_MAX_WGS") ? atoi(getenv("CRFP_F32_S8_MAX_WGS")) : (1 << 30);   // A/B knob (round 4), lab library only
#else
        constexpr int s8_max_wgs = 1 << 30;
#endif
        if (a.ctiles == 1 && !uses_s3_dst_only_4wave(a) && (long long)a.N * ((a.W + TW - 1) / TW) * ((a.H + S8_TH - 1) / S8_TH) <= s8_max_wgs) {
#ifdef CRFP_S8_NW
            constexpr int s8nw = CRFP_S8_NW;
#else
            constexpr int s8nw = S8_TH;
#endif
            const int tiles8 = ((a.W + TW - 1) / TW) * ((a.H + s8nw - 1) / s8nw);
#ifdef CRFP_LAB
            static const int s8p = getenv("CRFP_S8P") ? atoi(getenv("CRFP_S8P")) : 0;   // the persistent 4-row form (lost: see the kernel)
            const int items4 = tiles * a.ctiles;
            if (s8p && items4 > 512) conv3x3_split8p_kernel<<<dim3(512, 1, a.N), S8P_NT, 0, s>>>(am, items4);
            else
#endif
            conv3x3_split8_kernel<s8nw><<<dim3(tiles8 * a.ctiles, 1, a.N), 64 * s8nw, 0, s>>>(am);
        } else
            conv3x3_split_kernel<1, 1, 2><<<dim3(tiles * a.ctiles, 1, a.N), 256, 0, s>>>(am);
#endif
    } else if (a.src[0].kind == SRC_NCHW_SHIFT) {   // SPyNet's 7x7-as-3x3 convolutions: 18 ... 144 K-quads, two chunks per staging phase
        const int tiles4 = ((a.W + TW - 1) / TW) * ((a.H + 3) / 4);
        const int nz = a.N * (a.ksplit > 0 ? a.ksplit : 1);
        if (ct2) conv3x3_mfma_kernel<2, 1, 2><<<dim3(tiles4 * (a.ctiles / 2), 1, nz), 256, 0, s>>>(a);
        else conv3x3_mfma_kernel<1, 1, 2><<<dim3(tiles4 * a.ctiles, 1, nz), 256, 0, s>>>(a);
    } else if (ct2) {
        conv3x3_mfma_kernel<2, 1><<<dim3(tiles * (a.ctiles / 2), 1, a.N), 256, 0, s>>>(a);
    } else if ((long long)tiles * a.ctiles * a.N < 512) {
        // few 8-row tiles (the LR-resolution convs of a single streamed frame: 115 workgroups for 180 x 320): 4-row tiles spread the
        // same latency-bound work over twice as many CUs
        const int tiles4 = ((a.W + TW - 1) / TW) * ((a.H + 3) / 4);
        conv3x3_mfma_kernel<1, 1><<<dim3(tiles4 * a.ctiles, 1, a.N), 256, 0, s>>>(a);
    } else {
        conv3x3_mfma_kernel<1, 2><<<dim3(tiles * a.ctiles, 1, a.N), 256, 0, s>>>(a);
    }
    CRFP_CHECK_LAUNCH();
    return 0;
}


// Two convs of the same shape in one launch (fp32 build: conv3x3_split_dual_kernel); anything it does not cover runs as two launches.
int launch_conv_mfma_dual(const ConvArgs& a0, const char* name0, const ConvArgs& a1, const char* name1, const char* name_both, hipStream_t s) {
#ifndef CRFP_ACT_BF16
#ifndef CRFP_CONV_DUAL
#define CRFP_CONV_DUAL 1   // A/B builds: 0 = always two launches
#endif
    auto plain_split = [](const ConvArgs& a) {
        bool q4 = true;
        for (int i = 0; i < a.nsrc; ++i) q4 = q4 && (a.src[i].kind == SRC_Q4 || a.src[i].kind == SRC_S3 || a.src[i].kind == SRC_ZERO);
        return a.wsplit && !a.strict && !precision_env_strict(0) && (a.kq & 3) == 0 && a.kq <= CRFP_MAX_KQ && a.ctiles > 1 && q4 && a.ksplit == 0 && !a.s3_dst &&
               (a.store != ST_PS || ((a.ps_r == 2 || a.ps_r == 4) && a.act != CRFP_ACT_TANH && a.act != CRFP_ACT_SIGMOID));
    };
    bool lab = false;
#ifdef CRFP_LAB
    lab = true;   // the lab library keeps its kernel-selection knobs: two launches
#endif
    if (CRFP_CONV_DUAL && !lab && plain_split(a0) && plain_split(a1) && a0.N == a1.N && a0.H == a1.H && a0.W == a1.W && a0.ctiles == a1.ctiles &&
        conv_s3_supported()) {
        const int tiles = ((a0.W + TW - 1) / TW) * ((a0.H + 3) / 4), w0 = tiles * a0.ctiles;
        if ((w0 & 7) == 0) {
            double bytes = 0, flops = 0;
            for (const ConvArgs* a : {&a0, &a1}) {
                double in_ch = 0;
                for (int i = 0; i < a->nsrc; ++i) in_ch += a->src[i].kind == SRC_ZERO ? 0 : a->src[i].nch;
                const double px = (double)a->N * a->H * a->W;
                bytes += px * (in_ch + a->cout) * (double)sizeof(act_t) + (double)a->cout * in_ch * 9 * 4.0;
                flops += 2.0 * px * a->cout * in_ch * 9.0;
            }
            ProfScope prof(name_both, s, bytes, flops);
            ConvArgs* am[2] = {&const_cast<ConvArgs&>(a0), &const_cast<ConvArgs&>(a1)};   // callers pass private, mutable plan copies
            const char* nm[2] = {name0, name1};
            for (int k = 0; k < 2; ++k) {
                am[k]->stamps = nullptr;
                am[k]->wsplit16 = (const char*)am[k]->wsplit + conv_split16_offset_bytes(*am[k]);
                am[k]->wsplit_sa = (const char*)am[k]->wsplit + conv_split_sa_offset_bytes(*am[k]);
                const int rc = build_quad_descs(*am[k], nm[k]);
                if (rc) return rc;
            }
            conv3x3_split_dual_kernel<1, 1, 2><<<dim3(2 * w0, 1, a0.N), 256, 0, s>>>(*am[0], *am[1], w0);
            CRFP_CHECK_LAUNCH();
            return 0;
        }
    }
#endif
    const int rc = launch_conv_mfma(a0, name0, s);
    return rc ? rc : launch_conv_mfma(a1, name1, s);
}

// ---------------------------------------------------------------- K slices for small maps (round 6: FNet's deep layers) -- measured, NOT shipped
// A conv over a small map is a handful of workgroups each walking ALL its 16-channel chunks in turn: FNet's 256 -> 256 layer on a 22 x 40 map is 48
// workgroups x 16 chunks = 53 us on a chip of 256 CUs.  conv_auto_ksplit cuts K into `ks` slices that run as workgroups of their own (the rule
// reads the layer's geometry only, never the batch size: one pair per call and a whole clip's pairs must take the same slices, or the
// one-frame-per-call results would stop being bit-identical to the clip call's); launch_conv_ksplit runs them -- slice k of item n stores its raw
// partial sums (bias in slice 0) as float Q4 item n * ks + k of `part` -- and launch_ksplit_reduce, or the pool / resize pass that reads the layer
// anyway, adds the slices in order, applies the activation, rounds to the storage type and guards the fp16 operand range, as the conv's
// epilogue would have.  Same box, product builds (profiles/r06_fnet_ksplit_ab.txt): fp32 FNet of ONE pair 318 -> 270 us and the one-frame-per-call
// rate without the resident promise 600 -> 619 frames/s -- but the six pairs of a clip 598 -> 690 us (their layers already fill the chip; the
// partial tensors are 50 MB per pair of extra traffic) and the fp32 HEADLINE clip 9.177 -> 9.266 ms; bf16 build: one pair 219 -> 215 us (its
// chunks are three times cheaper: nothing to shorten), six pairs 350 -> 495 us.  A rule that may not look at the batch size cannot have the one
// without the other, so the product keeps every layer in one piece; the lab library takes the slices with CRFP_CONV_KSPLIT=1 (fp32 build) / 2 (both).
int conv_auto_ksplit(int H, int W, int ctiles, int kq) {
#ifdef CRFP_LAB
    static const int on = getenv("CRFP_CONV_KSPLIT") ? atoi(getenv("CRFP_CONV_KSPLIT")) : 0;
#else
    constexpr int on = 0;
#endif
    if (on < (kActBf16 ? 2 : 1) || (kq & 3)) return 1;
    const int nch = kq >> 2;
    const long long wgs = (long long)((W + TW - 1) / TW) * ((H + 3) / 4) * ctiles;   // 4-row tiles of the four-wave kernels
    int ks = 1;
    while (ks < 8 && nch % (2 * ks) == 0 && nch / (2 * ks) >= 2 && wgs * 2 * ks <= 384) ks *= 2;
    return ks;
}

// a: the layer's plan as launch_conv_mfma takes it, with a.ksplit = ks, a.N = items (not items * ks), one ST_Q4 destination = `part`
// (bstride = floats of one (item, slice) = cout-quads * H * W * 4); activation, residual and guard belong to the reduce pass
int launch_conv_ksplit(const ConvArgs& a, const char* name, hipStream_t s) {
    bool q4 = true;
    for (int i = 0; i < a.nsrc; ++i) q4 = q4 && (a.src[i].kind == SRC_Q4 || a.src[i].kind == SRC_ZERO);
    const int nch = a.kq >> 2;
    if (a.ksplit < 2 || (a.kq & 3) || a.kq > CRFP_MAX_KQ || nch % a.ksplit || !q4 || !a.wsplit || a.strict || precision_env_strict(0) || a.store != ST_Q4 ||
        a.ndst != 1 || a.resid || a.s3_dst || a.post_scale != 1.0f || (a.cout & 3)) {
        set_error("conv_ksplit %s: unsupported plan (ksplit=%d kq=%d store=%d ndst=%d strict=%d)", name, a.ksplit, a.kq, a.store, a.ndst, a.strict);
        return CRFP_E_UNSUPPORTED;
    }
    const int tiles = ((a.W + TW - 1) / TW) * ((a.H + 3) / 4);
    double in_ch = 0;
    for (int i = 0; i < a.nsrc; ++i) in_ch += a.src[i].kind == SRC_ZERO ? 0 : a.src[i].nch;
    const double px = (double)a.N * a.H * a.W;
    ProfScope prof(name, s, px * (in_ch * (double)sizeof(act_t) + a.cout * 4.0 * a.ksplit) + (double)a.cout * in_ch * 9 * 4.0, 2.0 * px * a.cout * in_ch * 9.0);
    ConvArgs& am = const_cast<ConvArgs&>(a);   // callers pass a private, mutable plan copy
    am.stamps = nullptr;
    am.act = CRFP_ACT_NONE;
    am.dst_f32 = 1;
    am.ovf = nullptr;
    am.wsplit16 = (const char*)a.wsplit + conv_split16_offset_bytes(a);
#ifndef CRFP_ACT_BF16
    am.wsplit_sa = (const char*)a.wsplit + conv_split_sa_offset_bytes(a);
#endif
    const int rc = build_quad_descs(am, name);
    if (rc) return rc;
    const dim3 grid(tiles * a.ctiles, 1, a.N * a.ksplit);
#ifdef CRFP_ACT_BF16
    conv3x3_bf16_kernel<1, true><<<grid, 256, 0, s>>>(am);
#else
    conv3x3_split_ks_kernel<<<grid, 256, 0, s>>>(am);
#endif
    CRFP_CHECK_LAUNCH();
    return 0;
}

// out[n] (Q4, storage type) = act(sum over k of part[n * ks + k]) in slice order; values an fp16 operand cannot hold raise item n's status word
__global__ void ksplit_reduce_kernel(const float* __restrict__ part, long long pb, int ks, act_t* __restrict__ out, long long ob, long long quads, int act,
                                     unsigned* ovf, int ovf_div, int ovf_add) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= quads) return;
    const int n = blockIdx.y;
    const cf32x4 v = ksplit_load(part, pb, ks, n, i * 4, act, ovf_word(ovf, ovf_div, ovf_add, n));
    stq(out + (long long)n * ob + i * 4, v);
}

int launch_ksplit_reduce(const float* part, long long pb, int ks, float* out, long long ob, int N, int nq, int H, int W, int act, unsigned* ovf, int ovf_div,
                         int ovf_add, hipStream_t s) {
    const long long quads = (long long)nq * H * W;
    ProfScope prof("ksplit_reduce_q4", s, (double)N * quads * (16.0 * ks + 4.0 * sizeof(act_t)), 0);
    ksplit_reduce_kernel<<<dim3((unsigned)((quads + 255) / 256), N), 256, 0, s>>>(part, pb, ks, as_act(out), ob, quads, act, ovf, ovf_div, ovf_add);
    CRFP_CHECK_LAUNCH();
    return 0;
}

#ifdef CRFP_ACT_BF16
// conv A -> conv B in one launch (conv3x3_bf16_pair_kernel); both plans as launch_conv_mfma takes them.  Conv A's output tensor is
// never written: it must have no other reader.
int launch_conv_pair(const ConvArgs& a, const ConvArgs& b, const char* name, hipStream_t s) {
    if ((a.kq & 3) || a.kq > CRFP_MAX_KQ || a.ctiles != 1 || a.cout != 32 || a.store != ST_Q4 || !a.wsplit || a.act == CRFP_ACT_TANH ||
        a.act == CRFP_ACT_SIGMOID || b.nsrc != 1 || b.src[0].kind != SRC_Q4 || b.kq != 8 || b.ctiles != 1 || b.cout != 32 || b.store != ST_Q4 ||
        !b.wsplit || b.act == CRFP_ACT_TANH || b.act == CRFP_ACT_SIGMOID || b.s3_dst || b.dst_f32 || a.N != b.N || a.H != b.H || a.W != b.W) {
        set_error("conv_pair %s: unsupported pair (A: kq=%d cout=%d store=%d; B: nsrc=%d kq=%d cout=%d store=%d)", name, a.kq, a.cout, a.store,
                  b.nsrc, b.kq, b.cout, b.store);
        return CRFP_E_UNSUPPORTED;
    }
    for (int i = 0; i < a.nsrc; ++i)
        if (a.src[i].kind == SRC_NCHW || a.src[i].kind == SRC_S3) { set_error("conv_pair %s: NCHW / S3 sources are not supported", name); return CRFP_E_UNSUPPORTED; }
    const double px = (double)a.N * a.H * a.W;
    double in_ch = 0;
    for (int i = 0; i < a.nsrc; ++i) in_ch += a.src[i].kind == SRC_ZERO ? 0 : a.src[i].nch;
    // algorithmic work of the PAIR: A's inputs + B's outputs (+ residual) cross HBM, the 32-channel tensor between them does not
    ProfScope prof(name, s, px * (in_ch + 32 + (b.resid ? 32 : 0)) * (double)sizeof(act_t) + (32.0 * in_ch + 32.0 * 32.0) * 9 * 4.0,
                   2.0 * px * 32 * (in_ch + 32) * 9.0);
    ConvArgs am = a;
    am.stamps = nullptr;
    am.wsplit16 = (const char*)a.wsplit + conv_split16_offset_bytes(a);
    const int rc = build_quad_descs(am, name);
    if (rc) return rc;
    PairB pb;
    memset(&pb, 0, sizeof(pb));
    pb.wsplit16 = (const char*)b.wsplit + conv_split16_offset_bytes(b);
    pb.bpk = b.bpk; pb.resid = b.resid; pb.resid_bstride = b.resid_bstride;
    for (int d = 0; d < CRFP_MAX_DST; ++d) pb.dst[d] = b.dst[d];
    pb.ndst = b.ndst; pb.cout = b.cout; pb.act = b.act; pb.post_scale = b.post_scale;
    am.ovf = b.ovf; am.ovf_div = b.ovf_div; am.ovf_add = b.ovf_add;
    const int tiles = ((a.W + P2_OW - 1) / P2_OW) * ((a.H + 7) / 8);
    conv3x3_bf16_pair_kernel<<<dim3(tiles, 1, a.N), P2_NT, 0, s>>>(am, pb);
    CRFP_CHECK_LAUNCH();
    return 0;
}
#endif

}  // namespace CRFP_NS
