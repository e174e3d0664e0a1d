// Narrow 3x3 convolutions at 8x resolution (Cin <= 12, Cout <= 4): HBM-bound stencils on the VALU.
// Replaces the C<=10 nn.Conv2d layers that run on 1440x2560 maps in the reference:
// encoder_hr (model/LTE.py:105-110), dcn_3's dcn_block / conv_fuse / dcn_offset / dcn_mask
// (model/CRFP.py:297-314 with mid_channels=4), forward_resblocks_3 (:1431-1432), conv_tttf (:1421)
// and conv_last (:1466-1468).  MFMA would waste >= 75 % of its rows on 4 output channels, so these
// are per-pixel FMAs with the weights in scalar registers; inputs are Q4 quads (one 16-B load per
// tap per 4 input channels).  Fused epilogues: activation, residual add, the fovea blend that
// produces the new recurrent state (model/CRFP.py:1672-1675), the RGB head + bilinear base
// (:1678-1683) and dcn_3's shared (dy,dx,mask) triple (:337-347).
#include "crfp_common.h"
#include <cstdlib>
#include <cstring>

namespace CRFP_NS {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float n_act(float v, int act) {
    switch (act) {
        case CRFP_ACT_RELU: return fmaxf(v, 0.0f);
        case CRFP_ACT_LRELU01: return v > 0.0f ? v : 0.1f * v;
        case CRFP_ACT_TANH: return tanhf(v);
        case CRFP_ACT_SIGMOID: return 1.0f / (1.0f + expf(-v));
        default: return v;
    }
}

// Workgroup = 256 threads walks a strided list of 64 x 16 output tiles (persistent).  Per tile the (16+2) x (64+2)
// halo of every input quad goes global -> registers -> LDS (coalesced 16-B loads); thread (tx, ty) then produces the 4
// vertically adjacent pixels (tx, 4*ty .. 4*ty+3): 18 halo + 9 weight ds_read_b128 per input quad feed 4 x 9 x 4 fp32
// MFMAs (v_mfma_f32_4x4x1: the four couts of a pixel per instruction; CRFP_NARROW_MFMA=0 builds the v_fma_f32 form with
// broadcast weight reads).  The loads of tile t+1 are issued right after tile t reached LDS, so they fly during
// the FMAs and stores of tile t: the one-tile-per-workgroup version spent 7.7-12 k cycles per tile waiting for its
// loads and 4 k in an epilogue that re-read its arguments (s_memtime stamps), i.e. HBM idled while it computed.
#ifndef CRFP_NARROW_KY_UNROLL
#define CRFP_NARROW_KY_UNROLL 3
#endif
#ifndef CRFP_NARROW_OCC1
#ifdef CRFP_ACT_BF16
#define CRFP_NARROW_OCC1 5   // bf16 build (bf16-MFMA form): 9.5 KB of LDS per input quad, 100 / 140 / 160 VGPRs
#else
#define CRFP_NARROW_OCC1 4   // workgroups per CU (launch bound and persistent grid) for KQ = 1
#endif
#endif
#ifndef CRFP_NARROW_OCC2
#define CRFP_NARROW_OCC2 3   // ... for KQ = 2 (KQ = 3: 2, LDS-bound)
#endif
#ifndef CRFP_NARROW_OCC3
#define CRFP_NARROW_OCC3 2   // (bf16 build: 3 fits at 160 VGPRs but measured slower, dcn3.block0 43.0 -> 45.6 us)
#endif
#ifndef CRFP_NARROW_MFMA
#define CRFP_NARROW_MFMA 1
#endif
constexpr int NTW = 64, NTH = 16, NLW = NTW + 2, NLH = NTH + 2;
constexpr int NST = (NLH * NLW + 255) / 256;  // 5 halo elements per thread per quad

// raw halo element as it sits in the prefetch registers -> fp32 quad for the LDS tile
#ifdef CRFP_ACT_BF16
__device__ __forceinline__ cu32x2 raw_flow(const char* p) { return *reinterpret_cast<const cu32x2*>(p); }
// fp32 (dx, dy) of a flow element -> bf16 quad (dx_hi, dy_hi, dx_lo, dy_lo): hi + lo carries 16 mantissa bits
__device__ __forceinline__ cu32x2 flow_words(cu32x2 r) {
    typedef __bf16 b2_t __attribute__((ext_vector_type(2)));
    const cf32x2 fl = __builtin_bit_cast(cf32x2, r);
    const b2_t hi = __builtin_convertvector(fl, b2_t);
    const cf32x2 back = __builtin_convertvector(hi, cf32x2);
    const b2_t lo = __builtin_convertvector(cf32x2{fl.x - back.x, fl.y - back.y}, b2_t);
    return cu32x2{__builtin_bit_cast(unsigned, hi), __builtin_bit_cast(unsigned, lo)};
}
__device__ __forceinline__ f32x4 raw_to_quad(cu32x2 r, bool flow) {
    const cf32x2 fl = __builtin_bit_cast(cf32x2, r);   // whole-pair cast (see quad_words in conv_mfma.hip)
    return flow ? f32x4{fl.x, fl.y, 0.0f, 0.0f} : quad_from_bits(r);
}
#else
__device__ __forceinline__ f32x4 raw_flow(const char* p) {
    const float2 f = *reinterpret_cast<const float2*>(p);
    return f32x4{f.x, f.y, 0.0f, 0.0f};
}
__device__ __forceinline__ f32x4 raw_to_quad(f32x4 r, bool) { return r; }
#endif

// PyTorch's source index for align_corners=False (same as resample.hip's src_index): the x8 bilinear base of the output
// head recomputed from the LR frame (NarrowArgs::base_lr) instead of read back from a staged quad
__device__ __forceinline__ void narrow_src_index(int dst, float scale, int in_size, int& i0, int& i1, float& l0, float& l1) {
    float src = scale * ((float)dst + 0.5f) - 0.5f;
    if (src < 0.0f) src = 0.0f;
    i0 = min((int)floorf(src), in_size - 1);
    i1 = min(i0 + 1, in_size - 1);
    l1 = fminf(fmaxf(src - (float)i0, 0.0f), 1.0f);
    l0 = 1.0f - l1;
}

// the second destination of an NE_PLAIN epilogue (NarrowArgs::dst2): the arithmetic of lrelu_q4_to_p4_kernel (resample.hip) on the value
// as dst holds it (bf16 storage: after its rounding)
__device__ __forceinline__ void narrow_store_state(act_t* d2, long long ppix, cf32x4 v, float& vmax) {
#ifdef CRFP_ACT_BF16
    v = quad_from_bits(quad_to_bits(v));
#endif
    const cf32x4 o = cf32x4{v.x > 0.0f ? v.x : 0.1f * v.x, v.y > 0.0f ? v.y : 0.1f * v.y, v.z > 0.0f ? v.z : 0.1f * v.z, v.w > 0.0f ? v.w : 0.1f * v.w};
    stq(d2 + ppix * 4, o);
    vmax = fmaxf(fmaxf(vmax, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
}

// GATE: NarrowArgs::gate holds, per 64 x 16 tile of this launch's map, whether the fovea mask has a set pixel within `gate_h` tiles of the
// tile (mask_gate_kernel, resample.hip).  Tiles without one are not computed at all.  NE_PLAIN (encoder_hr, model/CRFP.py:1545-1547): their
// outputs only ever feed pixels the fovea blend deselects.  NE_BLEND (conv_tttf + blend, :1672-1675): there the new state is lrelu(state),
// which a plain streaming pass has written before this launch (launch_lrelu_q4_to_p4, resample.hip).
// ST2 (NE_PLAIN, round 6): the launch has a second destination (NarrowArgs::dst2); a template parameter so that every other instantiation keeps
// the epilogue it had (as a run-time branch it cost the bf16 build's stencils ~1 us each: profiles/r06_headline_ab.txt)
template <int KQ, int EPI, bool GATE = false, bool ST2 = false>
__global__ __launch_bounds__(256, KQ == 1 ? CRFP_NARROW_OCC1 : (KQ == 2 ? CRFP_NARROW_OCC2 : CRFP_NARROW_OCC3)) void conv3x3_narrow_kernel(const NarrowArgs a) {
#ifdef CRFP_ACT_BF16
    __shared__ cu32x2 tile[KQ][NLH][NLW];   // bf16 quads as they sit in HBM (8 bytes)
#else
    __shared__ float4 tile[KQ][NLH][NLW];
    __shared__ float4 wl[9 * KQ * 4];  // FMA form: [tap][kq][cin comp] -> float4 over cout (broadcast reads)
                                       // MFMA form: [tap][kq][cout] -> float4 over cin comp (lane reads row cout = lane & 3)
#endif
    const int tid = threadIdx.x, tx = tid & 63, ty = tid >> 6;
#ifndef CRFP_ACT_BF16
    if (tid < 9 * KQ * 4) {
        const float4 wv = reinterpret_cast<const float4*>(a.wpk)[tid];   // packed: (tap, kq, cin comp) -> 4 couts
        if (CRFP_NARROW_MFMA) {
            float* wf = reinterpret_cast<float*>(wl) + (tid >> 2) * 16 + (tid & 3);
            wf[0] = wv.x; wf[4] = wv.y; wf[8] = wv.z; wf[12] = wv.w;
        } else {
            wl[tid] = wv;
        }
    }
#endif
    const int n = blockIdx.z;
    const int H = a.H, W = a.W;
    const int tiles_x = (W + NTW - 1) / NTW, ntiles = tiles_x * ((H + NTH - 1) / NTH);

    // ---- everything that comes from the kernel arguments, once
    const char* qbase[KQ];    // plane of K-quad k of its source (byte pointer: act_t quads, or a float [H][W][2] flow field)
    int qpitch[KQ];           // row pitch in pixels
    bool qflow[KQ];           // [H][W][2] flow field instead of a Q4 plane
#pragma unroll
    for (int k = 0; k < KQ; ++k) {
        int kql = k, s = 0;
        while (s < a.nsrc - 1 && kql >= a.src[s].nq) { kql -= a.src[s].nq; ++s; }
        const ConvSrc src = a.src[s];
        qflow[k] = src.kind == SRC_FLOW2;
        qpitch[k] = W + src.pad;
        qbase[k] = qflow[k] ? reinterpret_cast<const char*>(src.p + (long long)n * src.bstride)
                            : reinterpret_cast<const char*>(as_act(src.p) + (long long)n * src.bstride + (long long)kql * (H + src.pad) * qpitch[k] * 4);
    }
#ifdef CRFP_ACT_BF16
    // bf16 storage: the channel mixing runs on v_mfma_f32_16x16x32_bf16 in a block-diagonal arrangement.  Lane l = 16 g + n
    // supplies, as B column n / K group g, 8 bf16 of ITS OWN pixel (two taps x 4 channels, straight from the bf16 LDS tile);
    // A row 4 g' + c carries the weights of cout c in K group g' only.  D row 4 g + c, column n -- the registers of lane
    // 16 g + n -- is then the tap pair's contribution to the 4 couts of that lane's pixel: 64 pixels x 8 K x 4 couts per
    // 16-cycle MFMA, 5 MFMAs per input quad and output row where the fp32 4x4x1 form issued 36 of 8 cycles (same products:
    // bf16 activations x bf16 weights are exact in fp32; fp32 accumulate).  The A fragments (20 VGPRs per input quad) are
    // built once per workgroup.  A flow source keeps fp32 accuracy as (dx_hi, dy_hi, dx_lo, dy_lo) against (w_dx, w_dy, w_dx, w_dy).
    typedef __bf16 nb16x8 __attribute__((ext_vector_type(8)));
    typedef __bf16 nb16x2 __attribute__((ext_vector_type(2)));
    typedef unsigned nu32x4 __attribute__((ext_vector_type(4)));
    nb16x8 aw[KQ][5];
    {
        const int lane = tid & 63, c = lane & 3;
        const bool diag = (lane >> 4) == ((lane & 15) >> 2);
#pragma unroll
        for (int k = 0; k < KQ; ++k)
#pragma unroll
            for (int st = 0; st < 5; ++st) {
                nu32x4 wds;
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    const int tap = 2 * st + hf;
                    float v[4] = {0.0f, 0.0f, 0.0f, 0.0f};
                    if (tap < 9) {
#pragma unroll
                        for (int comp = 0; comp < 4; ++comp) {
                            const int cc = qflow[k] ? (comp & 1) : comp;
                            const float wv = a.wpk[((tap * KQ + k) * 4 + cc) * 4 + c];   // already a bf16 value (narrow_pack_kernel)
                            v[comp] = diag ? wv : 0.0f;
                        }
                    }
                    wds[2 * hf] = __builtin_bit_cast(unsigned, __builtin_convertvector(cf32x2{v[0], v[1]}, nb16x2));
                    wds[2 * hf + 1] = __builtin_bit_cast(unsigned, __builtin_convertvector(cf32x2{v[2], v[3]}, nb16x2));
                }
                aw[k][st] = __builtin_bit_cast(nb16x8, wds);
            }
    }
#endif
    const float4 bias = *reinterpret_cast<const float4*>(a.bpk);
    const int cout = a.cout, act = a.act;
    // NONE / RELU / LRELU(0.1) as max(v, slope*v) with slope 1 / 0 / 0.1 (exact); tanh / sigmoid take the slow branch
    const float slope = act == CRFP_ACT_RELU ? 0.0f : (act == CRFP_ACT_LRELU01 ? 0.1f : 1.0f);
    const bool slow_act = act == CRFP_ACT_TANH || act == CRFP_ACT_SIGMOID;
    const float post = a.post_scale;
    float* const dst = a.dst + (long long)n * a.dst_bstride;          // NE_LAST / NE_OFFMASK3: float tensors
    act_t* const dsta = as_act(a.dst) + (long long)n * a.dst_bstride;   // NE_PLAIN / NE_BLEND: activation (act_t) tensors
    const int dpitch = W + a.dst_pad;                       // Q4 destination may be padded (P4)
    const act_t* const resid = a.resid ? as_act(a.resid) + (long long)n * a.resid_bstride : nullptr;
    act_t* const dst2 = ST2 ? as_act(a.dst2) + (long long)n * a.dst2_bstride : nullptr;
    const uint8_t* const mask = EPI == NE_BLEND ? a.mask + (long long)n * a.mask_bstride : nullptr;
    const act_t* const basep = EPI == NE_LAST && a.base ? as_act(a.base) + (long long)n * a.base_bstride : nullptr;
    const float* const baselr = EPI == NE_LAST && !a.base ? a.base_lr + (long long)n * a.base_bstride : nullptr;
    // fp16-operand range guard (ConvArgs::ovf): the output head turns the frame into NaN once the sticky word is set
    unsigned* const ovfw = ovf_word(a.ovf, a.ovf_div, a.ovf_add, n);
    const bool poison = EPI == NE_LAST && ovfw && *ovfw != 0;
    float vmax = 0.0f;
    const float* const flowp = EPI == NE_OFFMASK3 ? a.flow + (long long)n * a.flow_bstride : nullptr;
    const bool y_only = a.y_only != 0;

#ifdef CRFP_ACT_BF16
    typedef cu32x2 rawq_t;    // 8 bytes: 4 bf16 of a pixel quad, or the (dx, dy) floats of a flow element
#else
    typedef f32x4 rawq_t;
#endif
#ifdef CRFP_LAB
    const int NPROBE = a.rsv;   // lab timing experiments (results wrong), bits: 1 = no MFMA loop, 2 = no stores, 4 = no global loads
#else
    constexpr int NPROBE = 0;
#endif
    rawq_t r[KQ][NST];
    bool okr[NST];   // halo element of the tile held in r[] lies inside the image
    // Every load of a tile is issued before anything consumes it, UNCONDITIONALLY at clamped coordinates: a predicated
    // load (`if (inside) r = load`) makes hipcc wait for each load right behind its issue (the select that merges the
    // zero needs the data) -- the first version's ISA read `G V(0) G V(0) ...`, five full memory latencies per quad.
    // Outside-image elements are zeroed when the registers are written to LDS.
#define CRFP_NARROW_LOAD(T)                                                                               \
    {                                                                                                     \
        const int ty_ = (T) / tiles_x, x0_ = ((T) - ty_ * tiles_x) * NTW, y0_ = ty_ * NTH;                \
        int cgy[NST], cgx[NST];                                                                           \
        _Pragma("unroll") for (int t = 0; t < NST; ++t) {                                                 \
            const int idx = min(tid + 256 * t, NLH * NLW - 1);                                            \
            const int rr = idx / NLW, c = idx - rr * NLW;                                                 \
            const int gy = y0_ + rr - 1, gx = x0_ + c - 1;                                                \
            okr[t] = tid + 256 * t < NLH * NLW && gy >= 0 && gy < H && gx >= 0 && gx < W;                 \
            cgy[t] = min(max(gy, 0), H - 1);                                                              \
            cgx[t] = min(max(gx, 0), W - 1);                                                              \
        }                                                                                                 \
        if (!(NPROBE & 4))                                                                                \
        _Pragma("unroll") for (int k = 0; k < KQ; ++k) {                                                  \
            if (qflow[k]) {   /* wave-uniform */                                                          \
                _Pragma("unroll") for (int t = 0; t < NST; ++t)                                           \
                    r[k][t] = raw_flow(qbase[k] + ((long long)cgy[t] * W + cgx[t]) * 8);                  \
            } else {                                                                                      \
                _Pragma("unroll") for (int t = 0; t < NST; ++t)                                           \
                    r[k][t] = *reinterpret_cast<const rawq_t*>(qbase[k] + ((long long)cgy[t] * qpitch[k] + cgx[t]) * kQuadBytes); \
            }                                                                                             \
        }                                                                                                 \
    }

    // (same-box A/B: -1..-5 % for the Q4 -> Q4 stencils, +6 % for the NCHW-plane head, which keeps the strided walk)
    constexpr bool BAND = EPI != NE_LAST;
    const int xq = ntiles >> 3, xr = ntiles & 7, xcd = blockIdx.x & 7;
    const int band0 = BAND ? xcd * xq + min(xcd, xr) : 0;
    const int band1 = BAND ? band0 + xq + (xcd < xr ? 1 : 0) : ntiles;
    const int t_step = BAND ? ((int)gridDim.x - xcd + 7) >> 3 : (int)gridDim.x;   // workgroups on this XCD
    int t_cur = BAND ? band0 + (blockIdx.x >> 3) : (int)blockIdx.x;
    // gate flags of this batch item: 4 bytes per tile (halo 0 .. 3), this launch reads byte gate_h
    const uint8_t* const gate = GATE ? a.gate + (long long)n * a.gate_bstride + a.gate_h : nullptr;
    if (GATE)
        while (t_cur < band1 && !gate[4 * t_cur]) t_cur += t_step;
    if (t_cur >= band1) return;
    CRFP_NARROW_LOAD(t_cur)
    for (;;) {
#pragma unroll
        for (int k = 0; k < KQ; ++k)
#pragma unroll
            for (int t = 0; t < NST; ++t) {
                const int idx = tid + 256 * t;
#ifdef CRFP_ACT_BF16
                if (idx < NLH * NLW) (&tile[k][0][0])[idx] = okr[t] ? (qflow[k] ? flow_words(r[k][t]) : r[k][t]) : cu32x2{0u, 0u};
#else
                if (idx < NLH * NLW) reinterpret_cast<f32x4*>(&tile[k][0][0])[idx] = okr[t] ? raw_to_quad(r[k][t], qflow[k]) : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#endif
            }
        __syncthreads();
        int t_next = t_cur + t_step;
        if (GATE)
            while (t_next < band1 && !gate[4 * t_next]) t_next += t_step;
        if (t_next < band1) CRFP_NARROW_LOAD(t_next)     // flies during the FMAs and stores below

        f32x4 acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = f32x4{bias.x, bias.y, bias.z, bias.w};
#ifdef CRFP_ACT_BF16
#pragma unroll
        for (int k = 0; k < KQ; ++k)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int st = 0; st < 5; ++st) {
                    const int t0 = 2 * st, t1 = 2 * st + 1;
                    const cu32x2 q0 = tile[k][4 * ty + i + t0 / 3][tx + t0 % 3];
                    const cu32x2 q1 = t1 < 9 ? tile[k][4 * ty + i + t1 / 3][tx + t1 % 3] : cu32x2{0u, 0u};
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aw[k][st], __builtin_bit_cast(nb16x8, nu32x4{q0.x, q0.y, q1.x, q1.y}),
                                                                     acc[i], 0, 0, 0);
                }
#else
        // The k loop is deliberately NOT unrolled (hipcc otherwise hoists every quad's reads).  ky is: with the MFMA form
        // a quad needs 18 halo + 9 weight ds_read_b128 (27 instead of 45 when the rows shared by the ky are re-read) and
        // stays at 108 / 128 / 149 VGPRs for KQ = 1 / 2 / 3 (+0.7 % clip); the FMA form spilled when ky was unrolled.
#pragma unroll 1
        for (int k = 0; k < ((NPROBE & 1) ? 0 : KQ); ++k) {
#pragma unroll (CRFP_NARROW_KY_UNROLL)
            for (int ky = 0; ky < 3; ++ky) {
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const float4* wq = &wl[((ky * 3 + kx) * KQ + k) * 4];
                    if (CRFP_NARROW_MFMA) {
                        // v_mfma_f32_4x4x1_16b_f32: 16 blocks of (4 couts x 1) x (1 x 4 pixels); lane l supplies A row l & 3
                        // (its cout's weight) and B column l & 3 (its own pixel's value) of block l >> 2 and receives the 4
                        // couts of its own pixel in the 4 result registers -- one 8-cycle MFMA where the FMA form issues four
                        // 4-cycle v_fma_f32 (packed FP32 would do the same but is off, see Makefile).  fp32 in, fp32 out.
                        const float4 wv = wq[tx & 3];
                        f32x4 u[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i) u[i] = reinterpret_cast<const f32x4&>(tile[k][4 * ty + ky + i][tx + kx]);
#pragma unroll
                        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(wv.x, u[i].x, acc[i], 0, 0, 0);
#pragma unroll
                        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(wv.y, u[i].y, acc[i], 0, 0, 0);
#pragma unroll
                        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(wv.z, u[i].z, acc[i], 0, 0, 0);
#pragma unroll
                        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(wv.w, u[i].w, acc[i], 0, 0, 0);
                        continue;
                    }
                    const float4 w0 = wq[0], w1 = wq[1], w2 = wq[2], w3 = wq[3];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float4 u = tile[k][4 * ty + ky + i][tx + kx];
                        acc[i][0] = fmaf(w3.x, u.w, fmaf(w2.x, u.z, fmaf(w1.x, u.y, fmaf(w0.x, u.x, acc[i][0]))));
                        acc[i][1] = fmaf(w3.y, u.w, fmaf(w2.y, u.z, fmaf(w1.y, u.y, fmaf(w0.y, u.x, acc[i][1]))));
                        acc[i][2] = fmaf(w3.z, u.w, fmaf(w2.z, u.z, fmaf(w1.z, u.y, fmaf(w0.z, u.x, acc[i][2]))));
                        acc[i][3] = fmaf(w3.w, u.w, fmaf(w2.w, u.z, fmaf(w1.w, u.y, fmaf(w0.w, u.x, acc[i][3]))));
                    }
                }
            }
        }
#endif

        const int tyi = t_cur / tiles_x, x = (t_cur - tyi * tiles_x) * NTW + tx, y0 = tyi * NTH;
        if (x < W && !(NPROBE & 2)) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int y = y0 + 4 * ty + i;
                if (y >= H) break;
                const long long pix = (long long)y * W + x;
                const long long dpix = (long long)y * dpitch + x;
                if (EPI == NE_PLAIN) {
                    float v[4];
                    if (slow_act) {
#pragma unroll
                        for (int o = 0; o < 4; ++o) v[o] = n_act(acc[i][o], act) * post;
                    } else {
#pragma unroll
                        for (int o = 0; o < 4; ++o) v[o] = fmaxf(acc[i][o], slope * acc[i][o]) * post;
                    }
#pragma unroll
                    for (int o = 0; o < 4; ++o)
                        if (o >= cout) v[o] = 0.0f;
                    if (resid) {
                        const cf32x4 rv = ldq(resid + pix * 4);
                        v[0] += rv.x; v[1] += rv.y; v[2] += rv.z; v[3] += rv.w;
                    }
                    stq(dsta + dpix * 4, cf32x4{v[0], v[1], v[2], v[3]});
                    if (ST2) narrow_store_state(dst2, (long long)y * (W + 1) + x, cf32x4{v[0], v[1], v[2], v[3]}, vmax);
                } else if (EPI == NE_BLEND) {
#ifdef CRFP_ACT_BF16
                    const cf32x4 centre = quad_from_bits(tile[0][4 * ty + i + 1][tx + 1]);
#else
                    const float4 centre = tile[0][4 * ty + i + 1][tx + 1];
#endif
                    const bool m = mask[pix] != 0;
                    float v[4] = {m ? acc[i][0] : centre.x, m ? acc[i][1] : centre.y, m ? acc[i][2] : centre.z,
                                  m ? acc[i][3] : centre.w};
#pragma unroll
                    for (int o = 0; o < 4; ++o) v[o] = v[o] > 0.0f ? v[o] : 0.1f * v[o];
                    vmax = fmaxf(fmaxf(vmax, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));   // the state feeds a split-fp16 conv
                    stq(dsta + dpix * 4, cf32x4{v[0], v[1], v[2], v[3]});
                } else if (EPI == NE_LAST) {
                    cf32x4 b;
                    if (basep) {
                        b = ldq(basep + pix * 4);
                    } else {   // bilinear x8 of the LR frame [3][H/8][W/8], the arithmetic of hr_prep_kernel
                        const int lh = H >> 3, lw = W >> 3;
                        int y0, y1, x0, x1;
                        float ly0, ly1, lx0, lx1;
                        narrow_src_index(y, 0.125f, lh, y0, y1, ly0, ly1);
                        narrow_src_index(x, 0.125f, lw, x0, x1, lx0, lx1);
                        float u[3];
#pragma unroll
                        for (int c = 0; c < 3; ++c) {
                            const float* p = baselr + (long long)c * lh * lw;
                            u[c] = ly0 * (lx0 * p[y0 * lw + x0] + lx1 * p[y0 * lw + x1]) + ly1 * (lx0 * p[y1 * lw + x0] + lx1 * p[y1 * lw + x1]);
                        }
                        b = cf32x4{u[0], u[1], u[2], 0.0f};
                    }
                    const float bad = poison ? __builtin_nanf("") : 0.0f;
                    if (y_only) {
                        dst[pix] = acc[i][0] + (0.299f * b.x + 0.587f * b.y + 0.114f * b.z) + bad;
                    } else {
                        const long long plane = (long long)H * W;
                        dst[pix] = acc[i][0] + b.x + bad;
                        dst[plane + pix] = acc[i][1] + b.y + bad;
                        dst[2 * plane + pix] = acc[i][2] + b.z + bad;
                    }
                } else {  // NE_OFFMASK3
                    const float2 f = *reinterpret_cast<const float2*>(flowp + pix * 2);
                    *reinterpret_cast<float4*>(dst + dpix * 4) =
                        make_float4(tanh10_plus(acc[i][0], 10.0f + f.y), tanh10_plus(acc[i][1], 10.0f + f.x),
                                    fast_sigmoid(acc[i][2]), 0.0f);   // libm tanhf/expf were the whole 42 us of this conv
                }
            }
        }
        if ((EPI == NE_BLEND || ST2) && ovfw && !(vmax < 65504.0f)) { atomicOr(ovfw, 1u); vmax = 0.0f; }
        if (t_next >= band1) break;
        t_cur = t_next;
        // every wave is done reading the tile before it is overwritten.  LDS-only barrier: __syncthreads() would also
        // wait (vmcnt(0)) for the write acknowledgement of the stores above, which nobody in the workgroup reads
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
#undef CRFP_NARROW_LOAD
}

#if !defined(CRFP_ACT_BF16) && defined(CRFP_LAB)
// ---------------------------------------------------------------- round-3 experiment: the same stencil WITHOUT the LDS tile
// The persistent LDS-tile kernel above spends 9.3 of the 26.6 us of a 4 -> 4 conv with loads, MFMAs and stores all removed
// (per-tile index arithmetic, LDS staging, two barriers per tile, ramp).  Here a thread owns R vertically adjacent
// pixels and reads their (R + 2) x 3 neighbourhood straight from global memory (each element is read by ~9 lanes: L1 / L2
// hits; 4.5-6 x 16 B per pixel and input quad through the texture path = 7-9 us per quad at the L1 rate), zero padding through
// the buffer range check (an invalid corner gets an out-of-range offset: one v_cndmask on the offset instead of four on the
// data).  No barrier after the weight table, no persistence, 6 (R = 2) or 4 (R = 4) waves per SIMD.  Same MFMA chains in the
// same order: bit-identical to conv3x3_narrow_kernel.  NE_PLAIN, Q4 sources only.  Lab: CRFP_NARROW_DIRECT=2|4.
template <int KQ, int R>
__global__ __launch_bounds__(256) void conv3x3_narrow_direct_kernel(const NarrowArgs a) {
    __shared__ float4 wl[9 * KQ * 4];
    const int tid = threadIdx.x, tx = tid & 63, ty = tid >> 6;
    if (tid < 9 * KQ * 4) {
        const float4 wv = reinterpret_cast<const float4*>(a.wpk)[tid];
        float* wf = reinterpret_cast<float*>(wl) + (tid >> 2) * 16 + (tid & 3);
        wf[0] = wv.x; wf[4] = wv.y; wf[8] = wv.z; wf[12] = wv.w;
    }
    const int n = blockIdx.z, H = a.H, W = a.W;
    const int x = blockIdx.x * 64 + tx, y0 = (blockIdx.y * 4 + ty) * R;
    const float4 bias = *reinterpret_cast<const float4*>(a.bpk);
    f32x4 acc[R];
#pragma unroll
    for (int i = 0; i < R; ++i) acc[i] = f32x4{bias.x, bias.y, bias.z, bias.w};
    bool rowok[R + 2], colok[3];
#pragma unroll
    for (int r = 0; r < R + 2; ++r) rowok[r] = y0 + r - 1 >= 0 && y0 + r - 1 < H;
#pragma unroll
    for (int c = 0; c < 3; ++c) colok[c] = x + c - 1 >= 0 && x + c - 1 < W;
    __syncthreads();
#pragma unroll 1
    for (int k = 0; k < KQ; ++k) {
        int kql = k, sidx = 0;
        while (sidx < a.nsrc - 1 && kql >= a.src[sidx].nq) { kql -= a.src[sidx].nq; ++sidx; }
        const ConvSrc src = a.src[sidx];
        const int pitch = (W + src.pad) * 16, plane = (H + src.pad) * pitch;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            (char*)const_cast<float*>(src.p + (long long)n * src.bstride) + (long long)kql * plane, 0, plane, 0x00020000);
        const int o00 = (y0 - 1) * pitch + (x - 1) * 16;
        f32x4 nb[R + 2][3];
#pragma unroll
        for (int r = 0; r < R + 2; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const int off = rowok[r] && colok[c] ? o00 + r * pitch + c * 16 : 0x7ffffff0;
                nb[r][c] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0));
            }
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const float4 wv = wl[((ky * 3 + kx) * KQ + k) * 4 + (tx & 3)];
#pragma unroll
                for (int i = 0; i < R; ++i) acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(wv.x, nb[ky + i][kx].x, acc[i], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < R; ++i) acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(wv.y, nb[ky + i][kx].y, acc[i], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < R; ++i) acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(wv.z, nb[ky + i][kx].z, acc[i], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < R; ++i) acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(wv.w, nb[ky + i][kx].w, acc[i], 0, 0, 0);
            }
    }
    if (x >= W) return;
    const float slope = a.act == CRFP_ACT_RELU ? 0.0f : (a.act == CRFP_ACT_LRELU01 ? 0.1f : 1.0f);
    const act_t* const resid = a.resid ? as_act(a.resid) + (long long)n * a.resid_bstride : nullptr;
    act_t* const dsta = as_act(a.dst) + (long long)n * a.dst_bstride;
    const int dpitch = W + a.dst_pad;
#pragma unroll
    for (int i = 0; i < R; ++i) {
        const int y = y0 + i;
        if (y >= H) break;
        float v[4];
#pragma unroll
        for (int o = 0; o < 4; ++o) v[o] = o < a.cout ? fmaxf(acc[i][o], slope * acc[i][o]) * a.post_scale : 0.0f;
        if (resid) {
            const cf32x4 rv = ldq(resid + ((long long)y * W + x) * 4);
            v[0] += rv.x; v[1] += rv.y; v[2] += rv.z; v[3] += rv.w;
        }
        stq(dsta + ((long long)y * dpitch + x) * 4, cf32x4{v[0], v[1], v[2], v[3]});
    }
}
#endif

#ifndef CRFP_ACT_BF16
#ifndef CRFP_SEQ_FAST
#define CRFP_SEQ_FAST 1   // A/B builds: 0 = the general loader / epilogue for every tile (quad-sequential and chain kernels)
#endif
// ---------------------------------------------------------------- quad-sequential form of the multi-quad stencils (round 6, fp32 build)
// conv3x3_narrow_kernel stages ALL KQ input quads of a tile at once: 57 KB of LDS and 149 VGPRs for KQ = 3 = two workgroups per CU, whose
// load / MFMA / store phases then add up instead of overlapping (destructive probes, profiles/r06_narrow_probes.txt: dcn3.block0 67.0 us,
// without its MFMAs 51.0, without its loads 50.8, without both 23.7).  Here the unit of work is (tile, input quad): ONE quad's halo in LDS
// (19 KB) and in the prefetch registers (20 VGPRs), the 4 x 4 accumulators carried over the KQ units of a tile, the next unit's loads in
// flight during this unit's MFMAs -- the resources of the KQ = 1 kernel, four workgroups per CU.  Same products in the same order (k outer,
// then ky, kx, channel): bit-identical to conv3x3_narrow_kernel<KQ, NE_PLAIN>.
template <int KQ>
__global__ __launch_bounds__(256, CRFP_NARROW_OCC1) void conv3x3_narrow_seq_kernel(const NarrowArgs a) {
    __shared__ float4 tile[NLH][NLW];
    __shared__ float4 wl[9 * KQ * 4];   // [tap][kq][cout] -> float4 over cin comp (lane reads row cout = lane & 3)
    const int tid = threadIdx.x, tx = tid & 63, ty = tid >> 6;
    if (tid < 9 * KQ * 4) {
        const float4 wv = reinterpret_cast<const float4*>(a.wpk)[tid];
        float* wf = reinterpret_cast<float*>(wl) + (tid >> 2) * 16 + (tid & 3);
        wf[0] = wv.x; wf[4] = wv.y; wf[8] = wv.z; wf[12] = wv.w;
    }
    const int n = blockIdx.z;
    const int H = a.H, W = a.W;
    const int tiles_x = (W + NTW - 1) / NTW, ntiles = tiles_x * ((H + NTH - 1) / NTH);
    const char* qbase[KQ];
    int qpitch[KQ];
    bool qflow[KQ];
#pragma unroll
    for (int k = 0; k < KQ; ++k) {
        int kql = k, s = 0;
        while (s < a.nsrc - 1 && kql >= a.src[s].nq) { kql -= a.src[s].nq; ++s; }
        const ConvSrc src = a.src[s];
        qflow[k] = src.kind == SRC_FLOW2;
        qpitch[k] = W + src.pad;
        qbase[k] = qflow[k] ? reinterpret_cast<const char*>(src.p + (long long)n * src.bstride)
                            : reinterpret_cast<const char*>(src.p + (long long)n * src.bstride + (long long)kql * (H + src.pad) * qpitch[k] * 4);
    }
    const float4 bias = *reinterpret_cast<const float4*>(a.bpk);
    const int cout = a.cout, act = a.act;
    const float slope = act == CRFP_ACT_RELU ? 0.0f : (act == CRFP_ACT_LRELU01 ? 0.1f : 1.0f);
    const float post = a.post_scale;
    float* const dsta = a.dst + (long long)n * a.dst_bstride;
    const int dpitch = W + a.dst_pad;
    const float* const resid = a.resid ? a.resid + (long long)n * a.resid_bstride : nullptr;
    const bool plain_out = CRFP_SEQ_FAST && cout == 4 && post == 1.0f;   // (uniform) nothing to mask or scale in the epilogue

    f32x4 r[NST];
    bool okr[NST];
    bool r_in = false;   // (uniform) the halo held in r[] lies inside the image as a whole: it goes to LDS without the zero-padding selects
    // this thread's halo elements as byte offsets from the halo's first pixel in a Q4 plane of pitch W -- the same for every tile
    unsigned hoff[NST];
    bool pitch_w = CRFP_SEQ_FAST != 0;   // (uniform) every source plane has pitch W (no padded source): the fast path's precondition
#pragma unroll
    for (int k = 0; k < KQ; ++k) pitch_w = pitch_w && qpitch[k] == W;
#pragma unroll
    for (int t = 0; t < NST; ++t) {
        const int idx = min(tid + 256 * t, NLH * NLW - 1), rr = idx / NLW;
        hoff[t] = (unsigned)(rr * W + (idx - rr * NLW)) * 16u;
    }
    // The halo of quad K_ of tile T.  The stencils are bound by instruction issue as much as by HBM (the 4x4x1 MFMAs of a quad are 1 152
    // issue cycles per wave; the index arithmetic of the general loader was about as many), so tiles whose halo lies inside the image (93 % of a
    // 1440 x 2560 map) take a fast path: a scalar tile base + one 32-bit multiply-add per element, no clamps, no validity flags.  General path:
    // every load issued unconditionally at clamped coordinates (see conv3x3_narrow_kernel).
#define CRFP_SEQ_LOAD(T, K_)                                                                              \
    {                                                                                                     \
        const int ty_ = (T) / tiles_x, x0_ = ((T) - ty_ * tiles_x) * NTW, y0_ = ty_ * NTH;                \
        const char* qb_ = qbase[0]; int qp_ = qpitch[0]; bool qf_ = qflow[0];                             \
        _Pragma("unroll") for (int kk = 1; kk < KQ; ++kk)                                                 \
            if ((K_) == kk) { qb_ = qbase[kk]; qp_ = qpitch[kk]; qf_ = qflow[kk]; }                       \
        r_in = pitch_w && x0_ >= 1 && y0_ >= 1 && x0_ + NLW - 1 <= W && y0_ + NLH - 1 <= H;               \
        if (r_in) {   /* raw buffer loads: scalar tile offset + the thread's 32-bit element offset, no 64-bit address arithmetic */ \
            const int so_ = ((y0_ - 1) * W + (x0_ - 1)) * 16;                                             \
            if (qf_) {                                                                                    \
                const __amdgpu_buffer_rsrc_t rs_ = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(qb_), 0, H * W * 8, 0x00020000); \
                _Pragma("unroll") for (int t = 0; t < NST; ++t) {                                         \
                    const cf32x2 f_ = __builtin_bit_cast(cf32x2, __builtin_amdgcn_raw_buffer_load_b64(rs_, (int)(hoff[t] >> 1), so_ >> 1, 0)); \
                    r[t] = f32x4{f_.x, f_.y, 0.0f, 0.0f};                                                 \
                }                                                                                         \
            } else {                                                                                      \
                const __amdgpu_buffer_rsrc_t rs_ = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(qb_), 0, H * W * 16, 0x00020000); \
                _Pragma("unroll") for (int t = 0; t < NST; ++t)                                           \
                    r[t] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_, (int)hoff[t], so_, 0)); \
            }                                                                                             \
        } else {                                                                                          \
            _Pragma("unroll") for (int t = 0; t < NST; ++t) {                                             \
                const int idx_ = min(tid + 256 * t, NLH * NLW - 1), hr_ = idx_ / NLW, hc_ = idx_ - hr_ * NLW; \
                const int gy = y0_ + hr_ - 1, gx = x0_ + hc_ - 1;                                         \
                okr[t] = tid + 256 * t < NLH * NLW && gy >= 0 && gy < H && gx >= 0 && gx < W;             \
                const int cgy = min(max(gy, 0), H - 1), cgx = min(max(gx, 0), W - 1);                     \
                if (qf_) r[t] = raw_flow(qb_ + ((long long)cgy * W + cgx) * 8);                           \
                else r[t] = *reinterpret_cast<const f32x4*>(qb_ + ((long long)cgy * qp_ + cgx) * 16);     \
            }                                                                                             \
        }                                                                                                 \
    }
    const int xq = ntiles >> 3, xr = ntiles & 7, xcd = blockIdx.x & 7;
    const int band0 = xcd * xq + min(xcd, xr), band1 = band0 + xq + (xcd < xr ? 1 : 0);
    const int t_step = ((int)gridDim.x - xcd + 7) >> 3;   // workgroups on this XCD
    int t_cur = band0 + (blockIdx.x >> 3), k_cur = 0;
    if (t_cur >= band1) return;
    CRFP_SEQ_LOAD(t_cur, 0)
    f32x4 acc[4];
    for (;;) {
        if (r_in) {
#pragma unroll
            for (int t = 0; t < NST; ++t)
                if (256 * t + 255 < NLH * NLW || tid + 256 * t < NLH * NLW) reinterpret_cast<f32x4*>(&tile[0][0])[tid + 256 * t] = r[t];
        } else {
#pragma unroll
            for (int t = 0; t < NST; ++t) {
                const int idx = tid + 256 * t;
                if (idx < NLH * NLW) reinterpret_cast<f32x4*>(&tile[0][0])[idx] = okr[t] ? r[t] : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            }
        }
        __syncthreads();
        // the next unit: the next quad of this tile, or quad 0 of the workgroup's next tile
        const bool last_k = k_cur == KQ - 1;
        const int t_next = last_k ? t_cur + t_step : t_cur, k_next = last_k ? 0 : k_cur + 1;
        if (t_next < band1) CRFP_SEQ_LOAD(t_next, k_next)     // flies during the MFMAs (and stores) below
        if (k_cur == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = f32x4{bias.x, bias.y, bias.z, bias.w};
        }
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const float4 wv = wl[((ky * 3 + kx) * KQ + k_cur) * 4 + (tx & 3)];
                f32x4 u[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) u[i] = reinterpret_cast<const f32x4&>(tile[4 * ty + ky + i][tx + kx]);
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(wv.x, u[i].x, acc[i], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(wv.y, u[i].y, acc[i], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(wv.z, u[i].z, acc[i], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(wv.w, u[i].w, acc[i], 0, 0, 0);
            }
        }
        if (last_k) {
            const int tyi = t_cur / tiles_x, x0 = (t_cur - tyi * tiles_x) * NTW, x = x0 + tx, y0 = tyi * NTH;
            if (plain_out && x0 + NTW <= W && y0 + NTH <= H) {   // (uniform) a whole tile, four couts, no scale: 32-bit offsets from the tile's base
                float* const dt = dsta + ((long long)y0 * dpitch + x0) * 4;
                const float* const rt_ = resid ? resid + ((long long)y0 * W + x0) * 4 : nullptr;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    cf32x4 v;
#pragma unroll
                    for (int o = 0; o < 4; ++o) v[o] = fmaxf(acc[i][o], slope * acc[i][o]);
                    if (rt_) {
                        const cf32x4 rv = ldq(rt_ + (unsigned)((4 * ty + i) * W + tx) * 4u);
                        v[0] += rv.x; v[1] += rv.y; v[2] += rv.z; v[3] += rv.w;
                    }
                    stq(dt + (unsigned)((4 * ty + i) * dpitch + tx) * 4u, v);
                }
            } else if (x < W) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int y = y0 + 4 * ty + i;
                    if (y >= H) break;
                    float v[4];
#pragma unroll
                    for (int o = 0; o < 4; ++o) v[o] = o < cout ? fmaxf(acc[i][o], slope * acc[i][o]) * post : 0.0f;
                    if (resid) {
                        const cf32x4 rv = ldq(resid + ((long long)y * W + x) * 4);
                        v[0] += rv.x; v[1] += rv.y; v[2] += rv.z; v[3] += rv.w;
                    }
                    stq(dsta + ((long long)y * dpitch + x) * 4, cf32x4{v[0], v[1], v[2], v[3]});
                }
            }
        }
        if (t_next >= band1) break;
        t_cur = t_next; k_cur = k_next;
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // LDS-only barrier (see conv3x3_narrow_kernel)
    }
#undef CRFP_SEQ_LOAD
}
#endif

// ---------------------------------------------------------------- two stencils in one pass (A -> B)
// conv A (KQA input quads -> 4 channels, activation) feeding conv B (that quad -> 4 channels, NE_PLAIN with optional residual
// or the dcn_3 offset/mask epilogue) without the round trip of A's output through HBM: at 8x resolution every tensor is
// 59 MB (fp32 @A), a write + a read of it is ~19 us of HBM time, plus one launch and one persistent-loop ramp.  Used where
// A's output has no other reader: encoder_hr.0 -> .2, dcn_3.conv_fuse -> offset/mask head, res3.conv1 -> conv2(+x).
// Per 16 x 64 output tile: the (16+4) x (64+4) halo of A's inputs is staged in LDS as before; A is evaluated on the
// (16+2) x (64+2) region B needs -- 1188 pixels, thread t takes pixels t, t+256, ... (five 4x4x1-MFMA chains per lane; the
// MFMA form pairs every lane with its own pixel, so any pixel-to-lane map works) -- its activated result goes to a second
// LDS tile, ZERO where the pixel lies outside the image (B's zero padding pads A's OUTPUT); B then runs exactly like the
// single-conv kernel out of that tile.  A costs 1.16x its pixels and 9 instead of 4.5 LDS reads per pixel and tap row.
constexpr int PLW = NTW + 4, PLH = NTH + 4;             // A's input halo
constexpr int PMW = NTW + 2, PMH = NTH + 2;             // intermediate (A's output) tile
constexpr int PSTA = (PLH * PLW + 255) / 256;           // 6 halo elements per thread per quad
constexpr int PNA = (PMH * PMW + 255) / 256;            // 5 intermediate pixels per thread

template <int KQA, int EPIB>
__global__ __launch_bounds__(256, KQA == 1 ? 3 : 2) void conv3x3_narrow_pair_kernel(const NarrowArgs a, const NarrowArgs b) {
    __shared__ float4 tileA[KQA][PLH][PLW];
    __shared__ float4 tileB[PMH][PMW];
    __shared__ float4 wlA[9 * KQA * 4], wlB[9 * 4];      // MFMA form: [tap][kq][cout] -> float4 over cin comp
    const int tid = threadIdx.x, tx = tid & 63, ty = tid >> 6;
    if (tid < 9 * KQA * 4) {
        const float4 wv = reinterpret_cast<const float4*>(a.wpk)[tid];
        float* wf = reinterpret_cast<float*>(wlA) + (tid >> 2) * 16 + (tid & 3);
        wf[0] = wv.x; wf[4] = wv.y; wf[8] = wv.z; wf[12] = wv.w;
    }
    if (tid < 9 * 4) {
        const float4 wv = reinterpret_cast<const float4*>(b.wpk)[tid];
        float* wf = reinterpret_cast<float*>(wlB) + (tid >> 2) * 16 + (tid & 3);
        wf[0] = wv.x; wf[4] = wv.y; wf[8] = wv.z; wf[12] = wv.w;
    }
    const int n = blockIdx.z;
    const int H = a.H, W = a.W;
    const int tiles_x = (W + NTW - 1) / NTW, ntiles = tiles_x * ((H + NTH - 1) / NTH);
    const char* qbase[KQA];
    int qpitch[KQA];
    bool qflow[KQA];
#pragma unroll
    for (int k = 0; k < KQA; ++k) {
        int kql = k, s = 0;
        while (s < a.nsrc - 1 && kql >= a.src[s].nq) { kql -= a.src[s].nq; ++s; }
        const ConvSrc src = a.src[s];
        qflow[k] = src.kind == SRC_FLOW2;
        qpitch[k] = W + src.pad;
        qbase[k] = qflow[k] ? reinterpret_cast<const char*>(src.p + (long long)n * src.bstride)
                            : reinterpret_cast<const char*>(as_act(src.p) + (long long)n * src.bstride + (long long)kql * (H + src.pad) * qpitch[k] * 4);
    }
    const float4 biasA = *reinterpret_cast<const float4*>(a.bpk), biasB = *reinterpret_cast<const float4*>(b.bpk);
    const float slopeA = a.act == CRFP_ACT_RELU ? 0.0f : (a.act == CRFP_ACT_LRELU01 ? 0.1f : 1.0f);
    const float slopeB = b.act == CRFP_ACT_RELU ? 0.0f : (b.act == CRFP_ACT_LRELU01 ? 0.1f : 1.0f);
    const int coutB = b.cout;
    float* const dst = b.dst + (long long)n * b.dst_bstride;
    act_t* const dsta = as_act(b.dst) + (long long)n * b.dst_bstride;
    const int dpitch = W + b.dst_pad;
    const act_t* const resid = b.resid ? as_act(b.resid) + (long long)n * b.resid_bstride : nullptr;
    const float* const flowp = EPIB == NE_OFFMASK3 ? b.flow + (long long)n * b.flow_bstride : nullptr;
    act_t* const dst2 = EPIB == NE_PLAIN && b.dst2 ? as_act(b.dst2) + (long long)n * b.dst2_bstride : nullptr;
    unsigned* const ovfw = dst2 ? ovf_word(b.ovf, b.ovf_div, b.ovf_add, n) : nullptr;
    float vmax = 0.0f;

#ifdef CRFP_ACT_BF16
    typedef cu32x2 rawq_t;
#else
    typedef f32x4 rawq_t;
#endif
    rawq_t r[KQA][PSTA];
    bool okr[PSTA];
#define CRFP_PAIR_LOAD(T)                                                                                 \
    {                                                                                                     \
        const int ty_ = (T) / tiles_x, x0_ = ((T) - ty_ * tiles_x) * NTW, y0_ = ty_ * NTH;                \
        int cgy[PSTA], cgx[PSTA];                                                                         \
        _Pragma("unroll") for (int t = 0; t < PSTA; ++t) {                                                \
            const int idx = min(tid + 256 * t, PLH * PLW - 1);                                            \
            const int rr = idx / PLW, c = idx - rr * PLW;                                                 \
            const int gy = y0_ + rr - 2, gx = x0_ + c - 2;                                                \
            okr[t] = tid + 256 * t < PLH * PLW && gy >= 0 && gy < H && gx >= 0 && gx < W;                 \
            cgy[t] = min(max(gy, 0), H - 1);                                                              \
            cgx[t] = min(max(gx, 0), W - 1);                                                              \
        }                                                                                                 \
        _Pragma("unroll") for (int k = 0; k < KQA; ++k) {                                                 \
            if (qflow[k]) {                                                                               \
                _Pragma("unroll") for (int t = 0; t < PSTA; ++t)                                          \
                    r[k][t] = raw_flow(qbase[k] + ((long long)cgy[t] * W + cgx[t]) * 8);                  \
            } else {                                                                                      \
                _Pragma("unroll") for (int t = 0; t < PSTA; ++t)                                          \
                    r[k][t] = *reinterpret_cast<const rawq_t*>(qbase[k] + ((long long)cgy[t] * qpitch[k] + cgx[t]) * kQuadBytes); \
            }                                                                                             \
        }                                                                                                 \
    }

    const int xq = ntiles >> 3, xr = ntiles & 7, xcd = blockIdx.x & 7;
    const int band0 = xcd * xq + min(xcd, xr), band1 = band0 + xq + (xcd < xr ? 1 : 0);
    const int t_step = ((int)gridDim.x - xcd + 7) >> 3;
    int t_cur = band0 + (blockIdx.x >> 3);
    if (t_cur >= band1) return;
    CRFP_PAIR_LOAD(t_cur)
    for (;;) {
#pragma unroll
        for (int k = 0; k < KQA; ++k)
#pragma unroll
            for (int t = 0; t < PSTA; ++t) {
                const int idx = tid + 256 * t;
                if (idx < PLH * PLW) reinterpret_cast<f32x4*>(&tileA[k][0][0])[idx] = okr[t] ? raw_to_quad(r[k][t], qflow[k]) : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            }
        __syncthreads();
        const int t_next = t_cur + t_step;
        if (t_next < band1) CRFP_PAIR_LOAD(t_next)
        const int tyi = t_cur / tiles_x, x0 = (t_cur - tyi * tiles_x) * NTW, y0 = tyi * NTH;

        // ---- conv A on the (NTH+2) x (NTW+2) region, result -> tileB
        {
            f32x4 acc[PNA];
            int pr[PNA], pc[PNA];
#pragma unroll
            for (int i = 0; i < PNA; ++i) {
                const int idx = min(tid + 256 * i, PMH * PMW - 1);   // surplus lanes recompute the last pixel (their store is skipped)
                pr[i] = idx / PMW; pc[i] = idx - pr[i] * PMW;
                acc[i] = f32x4{biasA.x, biasA.y, biasA.z, biasA.w};
            }
#pragma unroll 1
            for (int k = 0; k < KQA; ++k)
#pragma unroll 1
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        const float4 wv = wlA[((ky * 3 + kx) * KQA + k) * 4 + (tx & 3)];
                        f32x4 u[PNA];
#pragma unroll
                        for (int i = 0; i < PNA; ++i) u[i] = reinterpret_cast<const f32x4&>(tileA[k][pr[i] + ky][pc[i] + kx]);
#pragma unroll
                        for (int i = 0; i < PNA; ++i) acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(wv.x, u[i].x, acc[i], 0, 0, 0);
#pragma unroll
                        for (int i = 0; i < PNA; ++i) acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(wv.y, u[i].y, acc[i], 0, 0, 0);
#pragma unroll
                        for (int i = 0; i < PNA; ++i) acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(wv.z, u[i].z, acc[i], 0, 0, 0);
#pragma unroll
                        for (int i = 0; i < PNA; ++i) acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(wv.w, u[i].w, acc[i], 0, 0, 0);
                    }
#pragma unroll
            for (int i = 0; i < PNA; ++i) {
                if (tid + 256 * i >= PMH * PMW) continue;
                const int gy = y0 + pr[i] - 1, gx = x0 + pc[i] - 1;
                const bool in = gy >= 0 && gy < H && gx >= 0 && gx < W;
                f32x4 v;
#pragma unroll
                for (int o = 0; o < 4; ++o) v[o] = (in && o < a.cout) ? fmaxf(acc[i][o], slopeA * acc[i][o]) * a.post_scale : 0.0f;
#ifdef CRFP_ACT_BF16
                v = quad_from_bits(quad_to_bits(v));   // the intermediate is an activation tensor: rounded as if it had been stored
#endif
                reinterpret_cast<f32x4&>(tileB[pr[i]][pc[i]]) = v;
            }
        }
        __syncthreads();

        // ---- conv B out of tileB (one input quad), 4 vertically adjacent pixels per thread
        f32x4 accB[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) accB[i] = f32x4{biasB.x, biasB.y, biasB.z, biasB.w};
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const float4 wv = wlB[(ky * 3 + kx) * 4 + (tx & 3)];
                f32x4 u[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) u[i] = reinterpret_cast<const f32x4&>(tileB[4 * ty + ky + i][tx + kx]);
#pragma unroll
                for (int i = 0; i < 4; ++i) accB[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(wv.x, u[i].x, accB[i], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 4; ++i) accB[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(wv.y, u[i].y, accB[i], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 4; ++i) accB[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(wv.z, u[i].z, accB[i], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 4; ++i) accB[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(wv.w, u[i].w, accB[i], 0, 0, 0);
            }
        const int x = x0 + tx;
        if (x < W) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int y = y0 + 4 * ty + i;
                if (y >= H) break;
                const long long pix = (long long)y * W + x, dpix = (long long)y * dpitch + x;
                if (EPIB == NE_PLAIN) {
                    float v[4];
#pragma unroll
                    for (int o = 0; o < 4; ++o) v[o] = o < coutB ? fmaxf(accB[i][o], slopeB * accB[i][o]) * b.post_scale : 0.0f;
                    if (resid) {
                        const cf32x4 rv = ldq(resid + pix * 4);
                        v[0] += rv.x; v[1] += rv.y; v[2] += rv.z; v[3] += rv.w;
                    }
                    stq(dsta + dpix * 4, cf32x4{v[0], v[1], v[2], v[3]});
                    if (dst2) narrow_store_state(dst2, (long long)y * (W + 1) + x, cf32x4{v[0], v[1], v[2], v[3]}, vmax);
                } else {   // NE_OFFMASK3
                    const float2 f = *reinterpret_cast<const float2*>(flowp + pix * 2);
                    *reinterpret_cast<float4*>(dst + dpix * 4) =
                        make_float4(tanh10_plus(accB[i][0], 10.0f + f.y), tanh10_plus(accB[i][1], 10.0f + f.x), fast_sigmoid(accB[i][2]), 0.0f);
                }
            }
        }
        if (ovfw && !(vmax < 65504.0f)) { atomicOr(ovfw, 1u); vmax = 0.0f; }
        if (t_next >= band1) break;
        t_cur = t_next;
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
#undef CRFP_PAIR_LOAD
}

#ifndef CRFP_ACT_BF16
// ---------------------------------------------------------------- three stencils in one pass (A -> B -> C), round 6, fp32 build
// The 8x-resolution chains of the reference run conv after conv on 59 MB tensors (fp32 @A) that have exactly one reader:
//   dcn_3:  dcn_block.0 (up, warped state, flow -> 4) -> dcn_block.2 (4 -> 4) -> conv_fuse (cat(., pre-offset) -> 4)      model/CRFP.py:303-308,333-336
//   forward_resblocks_3:  main.0 (cat(up, aligned) -> 4) -> conv1 (4 -> 4, ReLU) -> conv2 (4 -> 4) + x                     model/CRFP.py:1626-1630
// As three launches each chain writes and re-reads two such tensors (236 MB of the 501 MB / 354 MB the chain moves).  Here a workgroup
// produces a 16 x 60 tile of C's output from the (16+6) x (60+6) halo of A's inputs: A is evaluated on the 20 x 64 region B needs -- lane =
// column, wave w = rows 5 w .. 5 w + 4, so a lane's five vertically adjacent pixels share their halo rows as in the single-conv kernel
// (conv3x3_narrow_pair_kernel's flat pixel map cost twice the LDS reads) -- its activated output goes to an LDS tile, ZERO where the pixel
// lies outside the image (B's zero padding pads A's OUTPUT); B likewise on 18 x 62 into a second tile; C on 16 x 60 from that tile (and, for
// conv_fuse, from one more global quad).  Global quads pass through LDS one at a time (the quad-sequential form above): 47 KB of LDS and
// ~110 VGPRs = three workgroups per CU.  RES: C adds A's output of the same pixel (the residual block's x, still in A's tile) and may write
// the second destination (NarrowArgs::dst2).  Each conv accumulates bias, then k-outer / ky / kx / channel as its single kernel does:
// bit-identical to the three launches.  Cost: A runs on 1.33x and B on 1.16x the pixels.
constexpr int KCW = 60, KCH = 16;                   // C's output tile
constexpr int KAW = KCW + 4, KAH = KCH + 4;         // A's region (64 x 20)
constexpr int KBW = KCW + 2, KBH = KCH + 2;         // B's region (62 x 18)
constexpr int KIW = KCW + 6, KIH = KCH + 6;         // halo of A's inputs (66 x 22)
constexpr int KST = (KIH * KIW + 255) / 256;        // 6 halo elements per thread and quad

template <int KQA, int KQG, bool RES>   // KQA: A's input quads (global); KQG: global quads of C behind its LDS quad (0 / 1)
__global__ __launch_bounds__(256, 3) void conv3x3_narrow_chain_kernel(const NarrowArgs a, const NarrowArgs b, const NarrowArgs c) {
    __shared__ float4 buf0[KIH][KIW];   // one global quad of A (22 x 66); later B's output (rows < 18, columns < 62)
    __shared__ float4 t1[KAH][KAW];     // A's output (20 x 64); KQG: later C's global quad (rows < 18, columns < 62)
    __shared__ float4 wlA[9 * KQA * 4], wlB[9 * 4], wlC[9 * (1 + KQG) * 4];   // [tap][kq][cout] -> float4 over cin comp
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#define CRFP_CHAIN_W(WL, ARGS, NQ)                                                                        \
    if (tid < 9 * (NQ) * 4) {                                                                             \
        const float4 wv = reinterpret_cast<const float4*>((ARGS).wpk)[tid];                               \
        float* wf = reinterpret_cast<float*>(WL) + (tid >> 2) * 16 + (tid & 3);                           \
        wf[0] = wv.x; wf[4] = wv.y; wf[8] = wv.z; wf[12] = wv.w;                                          \
    }
    CRFP_CHAIN_W(wlA, a, KQA) CRFP_CHAIN_W(wlB, b, 1) CRFP_CHAIN_W(wlC, c, 1 + KQG)
#undef CRFP_CHAIN_W
    const int n = blockIdx.z;
    const int H = a.H, W = a.W;
    const int tiles_x = (W + KCW - 1) / KCW, ntiles = tiles_x * ((H + KCH - 1) / KCH);
    // global quads: A's KQA, then C's (its quad 1)
    const char* qbase[KQA + KQG];
    int qpitch[KQA + KQG];
    bool qflow[KQA + KQG];
#pragma unroll
    for (int k = 0; k < KQA + KQG; ++k) {
        const NarrowArgs& g = k < KQA ? a : c;
        int kql = k < KQA ? k : 1, s = 0;
        while (s < g.nsrc - 1 && kql >= g.src[s].nq) { kql -= g.src[s].nq; ++s; }
        const ConvSrc src = g.src[s];
        qflow[k] = src.kind == SRC_FLOW2;
        qpitch[k] = W + src.pad;
        qbase[k] = qflow[k] ? reinterpret_cast<const char*>(src.p + (long long)n * src.bstride)
                            : reinterpret_cast<const char*>(src.p + (long long)n * src.bstride + (long long)kql * (H + src.pad) * qpitch[k] * 4);
    }
    const float4 biasA = *reinterpret_cast<const float4*>(a.bpk), biasB = *reinterpret_cast<const float4*>(b.bpk),
                 biasC = *reinterpret_cast<const float4*>(c.bpk);
    const float slopeA = a.act == CRFP_ACT_RELU ? 0.0f : (a.act == CRFP_ACT_LRELU01 ? 0.1f : 1.0f);
    const float slopeB = b.act == CRFP_ACT_RELU ? 0.0f : (b.act == CRFP_ACT_LRELU01 ? 0.1f : 1.0f);
    const float slopeC = c.act == CRFP_ACT_RELU ? 0.0f : (c.act == CRFP_ACT_LRELU01 ? 0.1f : 1.0f);
    float* const dst = c.dst + (long long)n * c.dst_bstride;
    const int dpitch = W + c.dst_pad;
    float* const dst2 = c.dst2 ? c.dst2 + (long long)n * c.dst2_bstride : nullptr;
    unsigned* const ovfw = dst2 ? ovf_word(c.ovf, c.ovf_div, c.ovf_add, n) : nullptr;
    float vmax = 0.0f;

    f32x4 r[KST];
    bool okr[KST];
    bool r_in = false;   // (uniform) the region held in r[] lies inside the image as a whole (see conv3x3_narrow_seq_kernel's fast path)
    // this thread's elements of A's input halo as byte offsets from the halo's first pixel in a Q4 plane of pitch W
    unsigned hoff[KST];
    bool pitch_w = CRFP_SEQ_FAST != 0;
#pragma unroll
    for (int k = 0; k < KQA; ++k) pitch_w = pitch_w && qpitch[k] == W;
#pragma unroll
    for (int t = 0; t < KST; ++t) {
        const int idx = min(tid + 256 * t, KIH * KIW - 1), rr = idx / KIW;
        hoff[t] = (unsigned)(rr * W + (idx - rr * KIW)) * 16u;
    }
    const bool plain_out = CRFP_SEQ_FAST && c.cout == 4 && c.post_scale == 1.0f;
    // global quad Q_ over the RW x RH region whose top-left pixel is (x0 - OFF, y0 - OFF) of tile T.  A's quads (OFF == 3) of a tile whose halo
    // lies inside the image: raw buffer loads at scalar tile offset + hoff[]; otherwise unconditional loads at clamped coordinates
#define CRFP_CHAIN_LOAD(T, Q_, RW, RH, OFF)                                                               \
    {                                                                                                     \
        const int ty_ = (T) / tiles_x, x0_ = ((T) - ty_ * tiles_x) * KCW, y0_ = ty_ * KCH;                \
        const char* qb_ = qbase[Q_]; const int qp_ = qpitch[Q_];                                          \
        r_in = (OFF) == 3 && pitch_w && x0_ >= 3 && y0_ >= 3 && x0_ - 3 + KIW <= W && y0_ - 3 + KIH <= H; \
        if (r_in) {                                                                                       \
            const int so_ = ((y0_ - 3) * W + (x0_ - 3)) * 16;                                             \
            if (qflow[Q_]) {                                                                              \
                const __amdgpu_buffer_rsrc_t rs_ = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(qb_), 0, H * W * 8, 0x00020000); \
                _Pragma("unroll") for (int t = 0; t < KST; ++t) {                                         \
                    const cf32x2 f_ = __builtin_bit_cast(cf32x2, __builtin_amdgcn_raw_buffer_load_b64(rs_, (int)(hoff[t] >> 1), so_ >> 1, 0)); \
                    r[t] = f32x4{f_.x, f_.y, 0.0f, 0.0f};                                                 \
                }                                                                                         \
            } else {                                                                                      \
                const __amdgpu_buffer_rsrc_t rs_ = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(qb_), 0, H * W * 16, 0x00020000); \
                _Pragma("unroll") for (int t = 0; t < KST; ++t)                                           \
                    r[t] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_, (int)hoff[t], so_, 0)); \
            }                                                                                             \
        } else                                                                                            \
        _Pragma("unroll") for (int t = 0; t < KST; ++t) {                                                 \
            const int idx = min(tid + 256 * t, (RW) * (RH) - 1);                                          \
            const int rr = idx / (RW), cc = idx - rr * (RW);                                              \
            const int gy = y0_ + rr - (OFF), gx = x0_ + cc - (OFF);                                       \
            okr[t] = tid + 256 * t < (RW) * (RH) && gy >= 0 && gy < H && gx >= 0 && gx < W;               \
            const int cgy = min(max(gy, 0), H - 1), cgx = min(max(gx, 0), W - 1);                         \
            if (qflow[Q_]) r[t] = raw_flow(qb_ + ((long long)cgy * W + cgx) * 8);                         \
            else r[t] = *reinterpret_cast<const f32x4*>(qb_ + ((long long)cgy * qp_ + cgx) * 16);         \
        }                                                                                                 \
    }
#define CRFP_CHAIN_SYNC asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // LDS-only barrier (see conv3x3_narrow_kernel)
    // NR rows x 9 taps of one input quad out of LDS tile SRC (any row pitch) into acc[0 .. NR - 1]: rows R0 + i, column C0 of the stage's region
#define CRFP_CHAIN_MFMA(SRC, WL, KQT, K_, NR, R0, C0)                                                     \
    _Pragma("unroll") for (int ky = 0; ky < 3; ++ky) {                                                    \
        _Pragma("unroll") for (int kx = 0; kx < 3; ++kx) {                                                \
            const float4 wv = (WL)[((ky * 3 + kx) * (KQT) + (K_)) * 4 + (lane & 3)];                      \
            f32x4 u[NR];                                                                                  \
            _Pragma("unroll") for (int i = 0; i < (NR); ++i) u[i] = reinterpret_cast<const f32x4&>((SRC)[(R0) + i + ky][(C0) + kx]); \
            _Pragma("unroll") for (int i = 0; i < (NR); ++i) acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(wv.x, u[i].x, acc[i], 0, 0, 0); \
            _Pragma("unroll") for (int i = 0; i < (NR); ++i) acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(wv.y, u[i].y, acc[i], 0, 0, 0); \
            _Pragma("unroll") for (int i = 0; i < (NR); ++i) acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(wv.z, u[i].z, acc[i], 0, 0, 0); \
            _Pragma("unroll") for (int i = 0; i < (NR); ++i) acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(wv.w, u[i].w, acc[i], 0, 0, 0); \
        }                                                                                                 \
    }
    const int xq = ntiles >> 3, xr = ntiles & 7, xcd = blockIdx.x & 7;
    const int band0 = xcd * xq + min(xcd, xr), band1 = band0 + xq + (xcd < xr ? 1 : 0);
    const int t_step = ((int)gridDim.x - xcd + 7) >> 3;
    int t_cur = band0 + (blockIdx.x >> 3);
    if (t_cur >= band1) return;
    CRFP_CHAIN_LOAD(t_cur, 0, KIW, KIH, 3)
    // B's rows / column of this thread (18 x 62 region on the 20 x 64 thread grid: surplus threads recompute the last row / column, unwritten)
    const int rB0 = min(5 * wave, KBH - 5), cB = min(lane, KBW - 1), cC = min(lane, KCW - 1);
    for (;;) {
        const int tyi = t_cur / tiles_x, x0 = (t_cur - tyi * tiles_x) * KCW, y0 = tyi * KCH;
        const int t_next = t_cur + t_step;
        // (uniform) A's 20 x 64 region lies inside the image: no zero padding to apply to A's and B's outputs
        const bool reg_in = CRFP_SEQ_FAST && x0 >= 2 && y0 >= 2 && x0 - 2 + KAW <= W && y0 - 2 + KAH <= H;
        f32x4 acc[5];
        // ---- conv A: its global quads one after the other through buf0
#pragma unroll
        for (int k = 0; k < KQA; ++k) {
            if (r_in) {
#pragma unroll
                for (int t = 0; t < KST; ++t)
                    if (256 * t + 255 < KIH * KIW || tid + 256 * t < KIH * KIW) reinterpret_cast<f32x4*>(&buf0[0][0])[tid + 256 * t] = r[t];
            } else {
#pragma unroll
                for (int t = 0; t < KST; ++t) {
                    const int idx = tid + 256 * t;
                    if (idx < KIH * KIW) reinterpret_cast<f32x4*>(&buf0[0][0])[idx] = okr[t] ? r[t] : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
                }
            }
            CRFP_CHAIN_SYNC
            // the next global quad flies during these MFMAs: A's next one, C's, or quad 0 of the workgroup's next tile
            if (k + 1 < KQA) CRFP_CHAIN_LOAD(t_cur, k + 1, KIW, KIH, 3)
            else if (KQG) CRFP_CHAIN_LOAD(t_cur, KQA, KBW, KBH, 1)
            else if (t_next < band1) CRFP_CHAIN_LOAD(t_next, 0, KIW, KIH, 3)
            if (k == 0) {
#pragma unroll
                for (int i = 0; i < 5; ++i) acc[i] = f32x4{biasA.x, biasA.y, biasA.z, biasA.w};
            }
            CRFP_CHAIN_MFMA(buf0, wlA, KQA, k, 5, 5 * wave, lane)
            CRFP_CHAIN_SYNC   // every wave is done reading buf0
        }
        // A's activated output -> t1, zero outside the image
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int gy = y0 - 2 + 5 * wave + i, gx = x0 - 2 + lane;
            const bool in = reg_in || (gy >= 0 && gy < H && gx >= 0 && gx < W);
            f32x4 v;
#pragma unroll
            for (int o = 0; o < 4; ++o) v[o] = (in && o < a.cout) ? fmaxf(acc[i][o], slopeA * acc[i][o]) * a.post_scale : 0.0f;
            reinterpret_cast<f32x4&>(t1[5 * wave + i][lane]) = v;
        }
        CRFP_CHAIN_SYNC
        // ---- conv B out of t1 on the 18 x 62 region
#pragma unroll
        for (int i = 0; i < 5; ++i) acc[i] = f32x4{biasB.x, biasB.y, biasB.z, biasB.w};
        CRFP_CHAIN_MFMA(t1, wlB, 1, 0, 5, rB0, cB)
        if (KQG) CRFP_CHAIN_SYNC   // t1 is about to receive C's global quad: every wave is done reading A's output
        if (5 * wave < KBH && lane < KBW) {
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                const int row = rB0 + i;
                if (row < 5 * wave) continue;   // (wave 3: rows 13, 14 belong to wave 2)
                const int gy = y0 - 1 + row, gx = x0 - 1 + lane;
                const bool in = reg_in || (gy >= 0 && gy < H && gx >= 0 && gx < W);
                f32x4 v;
#pragma unroll
                for (int o = 0; o < 4; ++o) v[o] = (in && o < b.cout) ? fmaxf(acc[i][o], slopeB * acc[i][o]) * b.post_scale : 0.0f;
                reinterpret_cast<f32x4&>(buf0[row][lane]) = v;
            }
        }
        if (KQG) {
#pragma unroll
            for (int t = 0; t < KST; ++t) {
                const int idx = tid + 256 * t, rr = idx / KBW, cc = idx - rr * KBW;
                if (idx < KBH * KBW) reinterpret_cast<f32x4&>(t1[rr][cc]) = okr[t] ? r[t] : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            }
            if (t_next < band1) CRFP_CHAIN_LOAD(t_next, 0, KIW, KIH, 3)
        }
        CRFP_CHAIN_SYNC
        // ---- conv C on the 16 x 60 tile: quad 0 = B's output (buf0), quad 1 = its global quad (t1)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = f32x4{biasC.x, biasC.y, biasC.z, biasC.w};
        CRFP_CHAIN_MFMA(buf0, wlC, 1 + KQG, 0, 4, 4 * wave, cC)
        if (KQG) CRFP_CHAIN_MFMA(t1, wlC, 1 + KQG, 1, 4, 4 * wave, cC)
        const int x = x0 + lane;
        if (plain_out && x0 + KCW <= W && y0 + KCH <= H) {   // (uniform) a whole tile, four couts, no scale: 32-bit offsets from the tile's base
            if (lane < KCW) {
                float* const dt = dst + ((long long)y0 * dpitch + x0) * 4;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    cf32x4 v;
#pragma unroll
                    for (int o = 0; o < 4; ++o) v[o] = fmaxf(acc[i][o], slopeC * acc[i][o]);
                    if (RES) {
                        const f32x4 rv = reinterpret_cast<const f32x4&>(t1[4 * wave + i + 2][lane + 2]);
                        v[0] += rv.x; v[1] += rv.y; v[2] += rv.z; v[3] += rv.w;
                    }
                    stq(dt + (unsigned)((4 * wave + i) * dpitch + lane) * 4u, v);
                    if (dst2) narrow_store_state(dst2 + ((long long)y0 * (W + 1) + x0) * 4, (long long)(unsigned)((4 * wave + i) * (W + 1) + lane), v, vmax);
                }
            }
        } else if (lane < KCW && x < W) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int y = y0 + 4 * wave + i;
                if (y >= H) break;
                float v[4];
#pragma unroll
                for (int o = 0; o < 4; ++o) v[o] = o < c.cout ? fmaxf(acc[i][o], slopeC * acc[i][o]) * c.post_scale : 0.0f;
                if (RES) {   // + A's output of this pixel (what the three-launch path re-reads from HBM as conv2's residual)
                    const f32x4 rv = reinterpret_cast<const f32x4&>(t1[4 * wave + i + 2][lane + 2]);
                    v[0] += rv.x; v[1] += rv.y; v[2] += rv.z; v[3] += rv.w;
                }
                stq(dst + ((long long)y * dpitch + x) * 4, cf32x4{v[0], v[1], v[2], v[3]});
                if (dst2) narrow_store_state(dst2, (long long)y * (W + 1) + x, cf32x4{v[0], v[1], v[2], v[3]}, vmax);
            }
        }
        if (ovfw && !(vmax < 65504.0f)) { atomicOr(ovfw, 1u); vmax = 0.0f; }
        if (t_next >= band1) break;
        t_cur = t_next;
        CRFP_CHAIN_SYNC   // buf0 and t1 are free for the next tile
    }
#undef CRFP_CHAIN_MFMA
#undef CRFP_CHAIN_SYNC
#undef CRFP_CHAIN_LOAD
}

// A -> B -> C in one launch.  C's sources: quad 0 = B's output, then (optionally) one global quad; RES: C's residual is A's output.
int launch_narrow_chain(const NarrowArgs& a, const NarrowArgs& b, const NarrowArgs& c, bool res, const char* name, hipStream_t s) {
    auto fast = [](const NarrowArgs& g) { return g.act != CRFP_ACT_TANH && g.act != CRFP_ACT_SIGMOID; };
    const int kqg = c.kq - 1;
    if (a.kq < 1 || a.kq > 3 || b.kq != 1 || kqg < 0 || kqg > 1 || (res && kqg) || a.epi != NE_PLAIN || b.epi != NE_PLAIN || c.epi != NE_PLAIN ||
        a.resid || b.resid || c.resid || a.dst2 || b.dst2 || !fast(a) || !fast(b) || !fast(c) || a.H != b.H || a.W != b.W || a.H != c.H || a.W != c.W ||
        (kqg && c.src[1].kind != SRC_Q4)) {
        set_error("conv_narrow_chain %s: unsupported chain (kqA=%d kqB=%d kqC=%d res=%d)", name, a.kq, b.kq, c.kq, (int)res);
        return CRFP_E_UNSUPPORTED;
    }
    double in_ch = 0;
    for (int i = 0; i < a.nsrc; ++i) in_ch += a.src[i].nch;
    if (kqg) in_ch += c.src[1].nch;
    const double px = (double)a.N * a.H * a.W;
    ProfScope prof(name, s, px * (in_ch + c.cout + (c.dst2 ? 4 : 0)) * 4.0, 2.0 * px * 9.0 * (in_ch * a.cout + 4.0 * b.cout + 4.0 * c.cout));
    const int ntl = ((a.W + KCW - 1) / KCW) * ((a.H + KCH - 1) / KCH);
    const int share = (ntl + 256 * 3 - 1) / (256 * 3);
    dim3 grid((ntl + share - 1) / share, 1, a.N);
    if (res) {
        switch (a.kq) {
            case 1: conv3x3_narrow_chain_kernel<1, 0, true><<<grid, 256, 0, s>>>(a, b, c); break;
            case 2: conv3x3_narrow_chain_kernel<2, 0, true><<<grid, 256, 0, s>>>(a, b, c); break;
            default: conv3x3_narrow_chain_kernel<3, 0, true><<<grid, 256, 0, s>>>(a, b, c); break;
        }
    } else if (kqg) {
        switch (a.kq) {
            case 1: conv3x3_narrow_chain_kernel<1, 1, false><<<grid, 256, 0, s>>>(a, b, c); break;
            case 2: conv3x3_narrow_chain_kernel<2, 1, false><<<grid, 256, 0, s>>>(a, b, c); break;
            default: conv3x3_narrow_chain_kernel<3, 1, false><<<grid, 256, 0, s>>>(a, b, c); break;
        }
    } else {
        switch (a.kq) {
            case 1: conv3x3_narrow_chain_kernel<1, 0, false><<<grid, 256, 0, s>>>(a, b, c); break;
            case 2: conv3x3_narrow_chain_kernel<2, 0, false><<<grid, 256, 0, s>>>(a, b, c); break;
            default: conv3x3_narrow_chain_kernel<3, 0, false><<<grid, 256, 0, s>>>(a, b, c); break;
        }
    }
    CRFP_CHECK_LAUNCH();
    return 0;
}
#endif

#ifdef CRFP_ACT_BF16
// ---------------------------------------------------------------- three stencils in one pass (A -> B -> C), bf16 storage (round 6)
// forward_resblocks_3's main.0 -> conv1 -> conv2 + x (+ the new state) as ONE launch, the twin of the fp32 build's conv3x3_narrow_chain_kernel: the
// same 16 x 60 output tile, A on 20 x 64 and B on 18 x 62 (lane = column, five / four rows per wave), global quads through LDS one at a time; the
// tensors between the convs are the bf16 quads the three-launch path would have stored (rounded at the same points), the channel mixing runs on
// the block-diagonal v_mfma_f32_16x16x32_bf16 of conv3x3_narrow_kernel in its order (quad outer, then the five tap pairs): bit-identical.
// LDS 21.8 KB + the A fragments of the (KQA + 2) quads (5 KB each: 64 lanes x 16 B x 5 tap pairs), read once per tap pair and stage.
constexpr int KCW = 60, KCH = 16;
constexpr int KAW = KCW + 4, KAH = KCH + 4;
constexpr int KBW = KCW + 2, KBH = KCH + 2;
constexpr int KIW = KCW + 6, KIH = KCH + 6;
constexpr int KST = (KIH * KIW + 255) / 256;

template <int KQA, int KQG, bool RES>   // KQG: global quads of C behind its LDS quad (0 / 1); RES: C adds A's output (+ the second destination)
__global__ __launch_bounds__(256, 3) void conv3x3_narrow_chain_kernel(const NarrowArgs a, const NarrowArgs b, const NarrowArgs c) {
    static_assert(!(RES && KQG), "the residual chain has no global quad in C");
    __shared__ cu32x2 buf0[KIH][KIW];   // one global quad of A (22 x 66); later B's output (rows < 18, columns < 62)
    __shared__ cu32x2 t1[KAH][KAW];     // A's output (20 x 64): B's input; RES: C's residual; KQG: later C's global quad (rows < 18, columns < 62)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = blockIdx.z;
    const int H = a.H, W = a.W;
    const int tiles_x = (W + KCW - 1) / KCW, ntiles = tiles_x * ((H + KCH - 1) / KCH);
    const char* qbase[KQA + KQG];
    int qpitch[KQA + KQG];
    bool qflow[KQA + KQG];
#pragma unroll
    for (int k = 0; k < KQA + KQG; ++k) {
        const NarrowArgs& g = k < KQA ? a : c;
        int kql = k < KQA ? k : 1, s = 0;
        while (s < g.nsrc - 1 && kql >= g.src[s].nq) { kql -= g.src[s].nq; ++s; }
        const ConvSrc src = g.src[s];
        qflow[k] = src.kind == SRC_FLOW2;
        qpitch[k] = W + src.pad;
        qbase[k] = qflow[k] ? reinterpret_cast<const char*>(src.p + (long long)n * src.bstride)
                            : reinterpret_cast<const char*>(as_act(src.p) + (long long)n * src.bstride + (long long)kql * (H + src.pad) * qpitch[k] * 4);
    }
    typedef __bf16 nb16x8 __attribute__((ext_vector_type(8)));
    typedef __bf16 nb16x2 __attribute__((ext_vector_type(2)));
    typedef unsigned nu32x4 __attribute__((ext_vector_type(4)));
    // block-diagonal A fragments (conv3x3_narrow_kernel) of A's quads, B and C: built once per workgroup into LDS ([quad][tap pair][lane])
    __shared__ nb16x8 awl[(KQA + 2 + KQG) * 5][64];
    if (tid < 64) {
        const int cc_ = lane & 3;
        const bool diag = (lane >> 4) == ((lane & 15) >> 2);
#define CRFP_CHAIN_FRAG(QI, WPK, KQT, K_, FLOW_)                                                          \
    _Pragma("unroll") for (int st = 0; st < 5; ++st) {                                                    \
        nu32x4 wds;                                                                                       \
        _Pragma("unroll") for (int hf = 0; hf < 2; ++hf) {                                                \
            const int tap = 2 * st + hf;                                                                  \
            float v[4] = {0.0f, 0.0f, 0.0f, 0.0f};                                                        \
            if (tap < 9) {                                                                                \
                _Pragma("unroll") for (int comp = 0; comp < 4; ++comp) {                                  \
                    const int cc = (FLOW_) ? (comp & 1) : comp;                                           \
                    const float wv = (WPK)[((tap * (KQT) + (K_)) * 4 + cc) * 4 + cc_];                    \
                    v[comp] = diag ? wv : 0.0f;                                                           \
                }                                                                                         \
            }                                                                                             \
            wds[2 * hf] = __builtin_bit_cast(unsigned, __builtin_convertvector(cf32x2{v[0], v[1]}, nb16x2)); \
            wds[2 * hf + 1] = __builtin_bit_cast(unsigned, __builtin_convertvector(cf32x2{v[2], v[3]}, nb16x2)); \
        }                                                                                                 \
        awl[(QI) * 5 + st][lane] = __builtin_bit_cast(nb16x8, wds);                                       \
    }
#pragma unroll
        for (int k = 0; k < KQA; ++k) CRFP_CHAIN_FRAG(k, a.wpk, KQA, k, qflow[k])
        CRFP_CHAIN_FRAG(KQA, b.wpk, 1, 0, false)
        CRFP_CHAIN_FRAG(KQA + 1, c.wpk, 1 + KQG, 0, false)
        if (KQG) CRFP_CHAIN_FRAG(KQA + 2, c.wpk, 1 + KQG, 1, qflow[KQA + KQG - 1])
#undef CRFP_CHAIN_FRAG
    }
    const float4 biasA = *reinterpret_cast<const float4*>(a.bpk), biasB = *reinterpret_cast<const float4*>(b.bpk),
                 biasC = *reinterpret_cast<const float4*>(c.bpk);
    const float slopeA = a.act == CRFP_ACT_RELU ? 0.0f : (a.act == CRFP_ACT_LRELU01 ? 0.1f : 1.0f);
    const float slopeB = b.act == CRFP_ACT_RELU ? 0.0f : (b.act == CRFP_ACT_LRELU01 ? 0.1f : 1.0f);
    const float slopeC = c.act == CRFP_ACT_RELU ? 0.0f : (c.act == CRFP_ACT_LRELU01 ? 0.1f : 1.0f);
    act_t* const dst = as_act(c.dst) + (long long)n * c.dst_bstride;
    const int dpitch = W + c.dst_pad;
    act_t* const dst2 = c.dst2 ? as_act(c.dst2) + (long long)n * c.dst2_bstride : nullptr;
    unsigned* const ovfw = dst2 ? ovf_word(c.ovf, c.ovf_div, c.ovf_add, n) : nullptr;
    float vmax = 0.0f;

    cu32x2 r[KST];
    bool okr[KST];
#define CRFP_CHAIN_LOAD(T, Q_, RW, RH, OFF)                                                               \
    {                                                                                                     \
        const int ty_ = (T) / tiles_x, x0_ = ((T) - ty_ * tiles_x) * KCW, y0_ = ty_ * KCH;                \
        _Pragma("unroll") for (int t = 0; t < KST; ++t) {                                                 \
            const int idx = min(tid + 256 * t, (RW) * (RH) - 1);                                          \
            const int rr = idx / (RW), cc = idx - rr * (RW);                                              \
            const int gy = y0_ + rr - (OFF), gx = x0_ + cc - (OFF);                                       \
            okr[t] = tid + 256 * t < (RW) * (RH) && gy >= 0 && gy < H && gx >= 0 && gx < W;               \
            const int cgy = min(max(gy, 0), H - 1), cgx = min(max(gx, 0), W - 1);                         \
            if (qflow[Q_]) r[t] = raw_flow(qbase[Q_] + ((long long)cgy * W + cgx) * 8);                   \
            else r[t] = *reinterpret_cast<const cu32x2*>(qbase[Q_] + ((long long)cgy * qpitch[Q_] + cgx) * 8); \
        }                                                                                                 \
    }
#define CRFP_CHAIN_SYNC asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    // NR rows of one input quad out of LDS tile SRC into acc[0 .. NR - 1]: rows R0 + i, column C0 of the stage's region
    // (tap pair outer, rows inner: every accumulator still sees its tap pairs in order)
#define CRFP_CHAIN_MFMA(SRC, QI, NR, R0, C0)                                                              \
    _Pragma("unroll") for (int st = 0; st < 5; ++st) {                                                    \
        const nb16x8 af_ = awl[(QI) * 5 + st][lane];                                                      \
        _Pragma("unroll") for (int i = 0; i < (NR); ++i) {                                                \
            const int t0 = 2 * st, t1_ = 2 * st + 1;                                                      \
            const cu32x2 q0 = (SRC)[(R0) + i + t0 / 3][(C0) + t0 % 3];                                    \
            const cu32x2 q1 = t1_ < 9 ? (SRC)[(R0) + i + t1_ / 3][(C0) + t1_ % 3] : cu32x2{0u, 0u};       \
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af_, __builtin_bit_cast(nb16x8, nu32x4{q0.x, q0.y, q1.x, q1.y}), acc[i], 0, 0, 0); \
        }                                                                                                 \
    }
    const int xq = ntiles >> 3, xr = ntiles & 7, xcd = blockIdx.x & 7;
    const int band0 = xcd * xq + min(xcd, xr), band1 = band0 + xq + (xcd < xr ? 1 : 0);
    const int t_step = ((int)gridDim.x - xcd + 7) >> 3;
    int t_cur = band0 + (blockIdx.x >> 3);
    if (t_cur >= band1) return;
    CRFP_CHAIN_LOAD(t_cur, 0, KIW, KIH, 3)
    const int rB0 = min(5 * wave, KBH - 5), cB = min(lane, KBW - 1), cC = min(lane, KCW - 1);
    for (;;) {
        const int tyi = t_cur / tiles_x, x0 = (t_cur - tyi * tiles_x) * KCW, y0 = tyi * KCH;
        const int t_next = t_cur + t_step;
        f32x4 acc[5];
        // ---- conv A: its global quads one after the other through buf0
#pragma unroll
        for (int k = 0; k < KQA; ++k) {
#pragma unroll
            for (int t = 0; t < KST; ++t) {
                const int idx = tid + 256 * t;
                if (idx < KIH * KIW) (&buf0[0][0])[idx] = okr[t] ? (qflow[k] ? flow_words(r[t]) : r[t]) : cu32x2{0u, 0u};
            }
            CRFP_CHAIN_SYNC
            if (k + 1 < KQA) CRFP_CHAIN_LOAD(t_cur, k + 1, KIW, KIH, 3)
            else if (KQG) CRFP_CHAIN_LOAD(t_cur, KQA, KBW, KBH, 1)
            else if (t_next < band1) CRFP_CHAIN_LOAD(t_next, 0, KIW, KIH, 3)
            if (k == 0) {
#pragma unroll
                for (int i = 0; i < 5; ++i) acc[i] = f32x4{biasA.x, biasA.y, biasA.z, biasA.w};
            }
            CRFP_CHAIN_MFMA(buf0, k, 5, 5 * wave, lane)
            CRFP_CHAIN_SYNC   // every wave is done reading buf0
        }
        // A's activated output -> t1 as the bf16 quads the three-launch path stores, zero outside the image
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int gy = y0 - 2 + 5 * wave + i, gx = x0 - 2 + lane;
            const bool in = gy >= 0 && gy < H && gx >= 0 && gx < W;
            cf32x4 v;
#pragma unroll
            for (int o = 0; o < 4; ++o) v[o] = (in && o < a.cout) ? fmaxf(acc[i][o], slopeA * acc[i][o]) * a.post_scale : 0.0f;
            t1[5 * wave + i][lane] = quad_to_bits(v);
        }
        CRFP_CHAIN_SYNC
        // ---- conv B out of t1 on the 18 x 62 region -> buf0
#pragma unroll
        for (int i = 0; i < 5; ++i) acc[i] = f32x4{biasB.x, biasB.y, biasB.z, biasB.w};
        CRFP_CHAIN_MFMA(t1, KQA, 5, rB0, cB)
        if (KQG) CRFP_CHAIN_SYNC   // t1 is about to receive C's global quad: every wave is done reading A's output
        if (5 * wave < KBH && lane < KBW) {
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                const int row = rB0 + i;
                if (row < 5 * wave) continue;
                const int gy = y0 - 1 + row, gx = x0 - 1 + lane;
                const bool in = gy >= 0 && gy < H && gx >= 0 && gx < W;
                cf32x4 v;
#pragma unroll
                for (int o = 0; o < 4; ++o) v[o] = (in && o < b.cout) ? fmaxf(acc[i][o], slopeB * acc[i][o]) * b.post_scale : 0.0f;
                buf0[row][lane] = quad_to_bits(v);
            }
        }
        if (KQG) {
#pragma unroll
            for (int t = 0; t < KST; ++t) {
                const int idx = tid + 256 * t, rr = idx / KBW, cc = idx - rr * KBW;
                if (idx < KBH * KBW) t1[rr][cc] = okr[t] ? (qflow[KQA] ? flow_words(r[t]) : r[t]) : cu32x2{0u, 0u};
            }
            if (t_next < band1) CRFP_CHAIN_LOAD(t_next, 0, KIW, KIH, 3)
        }
        CRFP_CHAIN_SYNC
        // ---- conv C on the 16 x 60 tile out of buf0, + A's output of the pixel (the residual block's x), + the new state
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = f32x4{biasC.x, biasC.y, biasC.z, biasC.w};
        CRFP_CHAIN_MFMA(buf0, KQA + 1, 4, 4 * wave, cC)
        if (KQG) CRFP_CHAIN_MFMA(t1, KQA + 2, 4, 4 * wave, cC)
        const int x = x0 + lane;
        if (lane < KCW && x < W) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int y = y0 + 4 * wave + i;
                if (y >= H) break;
                float v[4];
#pragma unroll
                for (int o = 0; o < 4; ++o) v[o] = o < c.cout ? fmaxf(acc[i][o], slopeC * acc[i][o]) * c.post_scale : 0.0f;
                if (RES) {
                    const cf32x4 rv = quad_from_bits(t1[4 * wave + i + 2][lane + 2]);
                    v[0] += rv.x; v[1] += rv.y; v[2] += rv.z; v[3] += rv.w;
                }
                stq(dst + ((long long)y * dpitch + x) * 4, cf32x4{v[0], v[1], v[2], v[3]});
                if (dst2) narrow_store_state(dst2, (long long)y * (W + 1) + x, cf32x4{v[0], v[1], v[2], v[3]}, vmax);
            }
        }
        if (ovfw && !(vmax < 65504.0f)) { atomicOr(ovfw, 1u); vmax = 0.0f; }
        if (t_next >= band1) break;
        t_cur = t_next;
        CRFP_CHAIN_SYNC   // buf0 and t1 are free for the next tile
    }
#undef CRFP_CHAIN_MFMA
#undef CRFP_CHAIN_SYNC
#undef CRFP_CHAIN_LOAD
}

// A -> B -> C in one launch.  C's sources: quad 0 = B's output, then (optionally) one global quad; res: C's residual is A's output.
int launch_narrow_chain(const NarrowArgs& a, const NarrowArgs& b, const NarrowArgs& c, bool res, const char* name, hipStream_t s) {
    auto fast = [](const NarrowArgs& g) { return g.act != CRFP_ACT_TANH && g.act != CRFP_ACT_SIGMOID; };
    const int kqg = c.kq - 1;
    if (a.kq < 1 || a.kq > 3 || b.kq != 1 || kqg < 0 || kqg > 1 || (res && kqg) || (res && a.kq > 2) || (!res && !kqg) || a.epi != NE_PLAIN || b.epi != NE_PLAIN ||
        c.epi != NE_PLAIN || a.resid || b.resid || c.resid || a.dst2 || b.dst2 || !fast(a) || !fast(b) || !fast(c) || a.H != b.H || a.W != b.W || a.H != c.H ||
        a.W != c.W || (kqg && c.src[1].kind != SRC_Q4)) {
        set_error("conv_narrow_chain %s: unsupported chain (kqA=%d kqB=%d kqC=%d res=%d)", name, a.kq, b.kq, c.kq, (int)res);
        return CRFP_E_UNSUPPORTED;
    }
    double in_ch = 0;
    for (int i = 0; i < a.nsrc; ++i) in_ch += a.src[i].nch;
    if (kqg) in_ch += c.src[1].nch;
    const double px = (double)a.N * a.H * a.W;
    ProfScope prof(name, s, px * (in_ch + c.cout + (c.dst2 ? 4 : 0)) * (double)sizeof(act_t), 2.0 * px * 9.0 * (in_ch * a.cout + 4.0 * b.cout + 4.0 * c.cout));
    const int ntl = ((a.W + KCW - 1) / KCW) * ((a.H + KCH - 1) / KCH);
    const int share = (ntl + 256 * 3 - 1) / (256 * 3);
    dim3 grid((ntl + share - 1) / share, 1, a.N);
    if (res) {
        if (a.kq == 1) conv3x3_narrow_chain_kernel<1, 0, true><<<grid, 256, 0, s>>>(a, b, c);
        else conv3x3_narrow_chain_kernel<2, 0, true><<<grid, 256, 0, s>>>(a, b, c);
    } else {
        switch (a.kq) {
            case 1: conv3x3_narrow_chain_kernel<1, 1, false><<<grid, 256, 0, s>>>(a, b, c); break;
            case 2: conv3x3_narrow_chain_kernel<2, 1, false><<<grid, 256, 0, s>>>(a, b, c); break;
            default: conv3x3_narrow_chain_kernel<3, 1, false><<<grid, 256, 0, s>>>(a, b, c); break;
        }
    }
    CRFP_CHECK_LAUNCH();
    return 0;
}
#endif

int launch_narrow_pair(const NarrowArgs& a, const NarrowArgs& b, const char* name, hipStream_t s) {
    if (a.kq < 1 || a.kq > 3 || a.cout < 1 || a.cout > 4 || b.kq != 1 || b.cout < 1 || b.cout > 4 ||
        (b.epi != NE_PLAIN && b.epi != NE_OFFMASK3) || a.epi != NE_PLAIN || a.resid || a.act == CRFP_ACT_TANH ||
        a.act == CRFP_ACT_SIGMOID || b.act == CRFP_ACT_TANH || b.act == CRFP_ACT_SIGMOID || a.H != b.H || a.W != b.W || a.N != b.N) {
        set_error("conv_narrow_pair %s: unsupported pair (kqA=%d kqB=%d epiA=%d epiB=%d)", name, a.kq, b.kq, a.epi, b.epi);
        return CRFP_E_UNSUPPORTED;
    }
    double in_ch = 0;
    for (int i = 0; i < a.nsrc; ++i) in_ch += a.src[i].nch;
    const double px = (double)a.N * a.H * a.W;
    const double outb = b.epi == NE_OFFMASK3 ? (3 + 2) * 4.0 : (b.cout + (b.resid ? 4 : 0) + (b.dst2 ? 4 : 0)) * (double)sizeof(act_t);
    ProfScope prof(name, s, px * (in_ch * (double)sizeof(act_t) + outb), 2.0 * px * 9.0 * (in_ch * a.cout + 4.0 * b.cout));
    const int ntl = ((a.W + NTW - 1) / NTW) * ((a.H + NTH - 1) / NTH);
    const int per_cu = a.kq == 1 ? 3 : 2;
    const int share = (ntl + 256 * per_cu - 1) / (256 * per_cu);
    dim3 grid((ntl + share - 1) / share, 1, a.N);
#define CRFP_PAIR_LAUNCH(KQ_)                                                                          \
    if (b.epi == NE_PLAIN) conv3x3_narrow_pair_kernel<KQ_, NE_PLAIN><<<grid, 256, 0, s>>>(a, b);       \
    else conv3x3_narrow_pair_kernel<KQ_, NE_OFFMASK3><<<grid, 256, 0, s>>>(a, b);
    switch (a.kq) {
        case 1: CRFP_PAIR_LAUNCH(1) break;
        case 2: CRFP_PAIR_LAUNCH(2) break;
        default: CRFP_PAIR_LAUNCH(3) break;
    }
#undef CRFP_PAIR_LAUNCH
    CRFP_CHECK_LAUNCH();
    return 0;
}

// wpk[((tap*KQ + kq)*4 + comp)*4 + o] = W[o][cin(kq,comp)][tap]
__global__ void narrow_pack_kernel(const NarrowArgs a, const float* __restrict__ w, const float* __restrict__ bias,
                                   const float* __restrict__ w2, const float* __restrict__ bias2, int cout_split,
                                   float* __restrict__ wpk, float* __restrict__ bpk) {
    const int total = 9 * a.kq * 16;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const int o = idx & 3, comp = (idx >> 2) & 3;
        const int k = (idx >> 4) % a.kq, tap = (idx >> 4) / a.kq;
        int kql = k, s = 0;
        while (s < a.nsrc - 1 && kql >= a.src[s].nq) { kql -= a.src[s].nq; ++s; }
        int ci = kql < a.src[s].nq ? conv_k_to_cin(a.src[s].kind, a.src[s].nch, kql, comp) : -1;
        if (ci >= 0) ci += a.src[s].cbase;
        float val = 0.0f;
        if (o < a.cout && ci >= 0 && ci < a.cin_total)
            val = o < cout_split ? w[((long long)o * a.cin_total + ci) * 9 + tap]
                                 : w2[((long long)(o - cout_split) * a.cin_total + ci) * 9 + tap];
        if (kActBf16) val = (float)(__bf16)val;   // bf16 build: every conv weight of the engine is a bf16 value
        wpk[idx] = val;
    }
    if (blockIdx.x == 0 && threadIdx.x < 4) {
        const int o = threadIdx.x;
        bpk[o] = o >= a.cout ? 0.0f : (o < cout_split ? bias[o] : bias2[o - cout_split]);
    }
}

size_t narrow_packed_weight_floats(const NarrowArgs& a) { return (size_t)9 * a.kq * 16; }

int launch_narrow_pack(const NarrowArgs& a, const float* w, const float* bias, const float* w2, const float* bias2,
                       int cout_split, float* wpk, float* bpk, hipStream_t s) {
    narrow_pack_kernel<<<2, 256, 0, s>>>(a, w, bias, w2, bias2, w2 ? cout_split : a.cout, wpk, bpk);
    CRFP_CHECK_LAUNCH();
    return 0;
}

int launch_narrow(const NarrowArgs& a_in, const char* name, hipStream_t s) {
    NarrowArgs a = a_in;
    a.stamps = nullptr;
#ifdef CRFP_LAB
    static const char* stamp_name = getenv("CRFP_STAMP_NAME");
    static long long* stamp_ptr = getenv("CRFP_STAMP_PTR") ? (long long*)strtoull(getenv("CRFP_STAMP_PTR"), nullptr, 0) : nullptr;
    a.stamps = (stamp_ptr && stamp_name && !strcmp(stamp_name, name)) ? stamp_ptr : nullptr;
    static const int nprobe = getenv("CRFP_NARROW_PROBE") ? atoi(getenv("CRFP_NARROW_PROBE")) : 0;
    a.rsv = nprobe;
#endif
    if (a.kq < 1 || a.kq > 3 || a.cout < 1 || a.cout > 4) {
        set_error("conv_narrow %s: unsupported kq=%d cout=%d", name, a.kq, a.cout);
        return CRFP_E_UNSUPPORTED;
    }
    double in_ch = 0;
    for (int i = 0; i < a.nsrc; ++i) in_ch += a.src[i].nch;
    const double px = (double)a.N * a.H * a.W;
    double extra = 0;
    if (a.epi == NE_BLEND) extra = 1.0 / 4;          // u8 mask
    if (a.epi == NE_LAST) extra = 3;                 // base quad (3 used)
    if (a.epi == NE_OFFMASK3) extra = 2;             // flow
    if (a.resid) extra += 4;
    if (a.dst2) extra += 4;
    // mask-gated launches touch a data-dependent share of the map: they are booked with no algorithmic bytes / flops (their time still counts)
    const double bytes = a.gate ? 0.0 : px * (in_ch + (a.epi == NE_OFFMASK3 ? 3 : a.cout) + extra) * (double)sizeof(act_t);
    ProfScope prof(name, s, bytes, a.gate ? 0.0 : 2.0 * px * in_ch * a.cout * 9.0);
#if !defined(CRFP_ACT_BF16) && defined(CRFP_LAB)
    static const int direct = getenv("CRFP_NARROW_DIRECT") ? atoi(getenv("CRFP_NARROW_DIRECT")) : 0;
    bool q4only = true;
    for (int i = 0; i < a.nsrc; ++i) q4only &= a.src[i].kind == SRC_Q4;
    if (direct && a.epi == NE_PLAIN && q4only && a.act != CRFP_ACT_TANH && a.act != CRFP_ACT_SIGMOID) {
        const int R = direct == 2 ? 2 : 4;
        dim3 grid((a.W + 63) / 64, (a.H + 4 * R - 1) / (4 * R), a.N);
#define CRFP_DIRECT_LAUNCH(KQ_) if (R == 2) conv3x3_narrow_direct_kernel<KQ_, 2><<<grid, 256, 0, s>>>(a); else conv3x3_narrow_direct_kernel<KQ_, 4><<<grid, 256, 0, s>>>(a);
        switch (a.kq) {
            case 1: CRFP_DIRECT_LAUNCH(1) break;
            case 2: CRFP_DIRECT_LAUNCH(2) break;
            default: CRFP_DIRECT_LAUNCH(3) break;
        }
#undef CRFP_DIRECT_LAUNCH
        CRFP_CHECK_LAUNCH();
        return 0;
    }
#endif
    // persistent: a few workgroups per CU walk the tiles (ceil-balanced shares)
    const int ntl = ((a.W + NTW - 1) / NTW) * ((a.H + NTH - 1) / NTH);
#ifdef CRFP_LAB
    static const int per_cu_env = getenv("CRFP_NARROW_WGS_PER_CU") ? atoi(getenv("CRFP_NARROW_WGS_PER_CU")) : 0;   // tuning knob
#else
    constexpr int per_cu_env = 0;
#endif
    const int per_cu = per_cu_env > 0 ? per_cu_env : (a.kq == 1 ? CRFP_NARROW_OCC1 : (a.kq == 2 ? CRFP_NARROW_OCC2 : CRFP_NARROW_OCC3));
    const int share = (ntl + 256 * per_cu - 1) / (256 * per_cu);
    dim3 grid((ntl + share - 1) / share, 1, a.N);
#ifndef CRFP_ACT_BF16
    {
    // Round 6: the plain stencils (dcn_3.dcn_block.0 / .2, dcn_3.conv_fuse; forward_resblocks_3 when its chain is off) in the quad-sequential form.
    // (The bf16 build runs both 8x conv chains as one launch each -- conv3x3_narrow_chain_kernel -- and keeps conv3x3_narrow_kernel for the rest;
    // a bf16 twin of this form measured dcn3.block0 42.1 -> 37.0 us before the chains superseded it: profiles/r06_narrow_seq_ab.txt.)
#ifndef CRFP_NARROW_SEQ
#define CRFP_NARROW_SEQ 1
#endif
#ifndef CRFP_NARROW_SEQ_MINKQ
#define CRFP_NARROW_SEQ_MINKQ 1   // the one-quad plain stencils too: the form carries the interior-tile fast path (A/B builds: 2)
#endif
#ifdef CRFP_LAB   // lab library: CRFP_NARROW_SEQ=0 at run time keeps conv3x3_narrow_kernel for the plain stencils
    static const bool seq_on = getenv("CRFP_NARROW_SEQ") ? atoi(getenv("CRFP_NARROW_SEQ")) != 0 : CRFP_NARROW_SEQ != 0;
#else
    constexpr bool seq_on = CRFP_NARROW_SEQ != 0;
#endif
    if (seq_on && !a.gate && !a.dst2 && a.epi == NE_PLAIN && a.kq >= CRFP_NARROW_SEQ_MINKQ && a.act != CRFP_ACT_TANH && a.act != CRFP_ACT_SIGMOID) {
        const int share4 = (ntl + 256 * CRFP_NARROW_OCC1 - 1) / (256 * CRFP_NARROW_OCC1);
        dim3 grid4((ntl + share4 - 1) / share4, 1, a.N);
        if (a.kq == 1) conv3x3_narrow_seq_kernel<1><<<grid4, 256, 0, s>>>(a);
        else if (a.kq == 2) conv3x3_narrow_seq_kernel<2><<<grid4, 256, 0, s>>>(a);
        else conv3x3_narrow_seq_kernel<3><<<grid4, 256, 0, s>>>(a);
        CRFP_CHECK_LAUNCH();
        return 0;
    }
    }
#endif
    if (a.gate) {   // mask-gated forms (engine.hip: encoder_hr and the fovea blend)
        if (a.epi == NE_PLAIN && a.kq == 1) conv3x3_narrow_kernel<1, NE_PLAIN, true><<<grid, 256, 0, s>>>(a);
        else if (a.epi == NE_PLAIN && a.kq == 2) conv3x3_narrow_kernel<2, NE_PLAIN, true><<<grid, 256, 0, s>>>(a);
        else if (a.epi == NE_BLEND && a.kq == 2) conv3x3_narrow_kernel<2, NE_BLEND, true><<<grid, 256, 0, s>>>(a);
        else { set_error("conv_narrow %s: no mask-gated form for kq=%d epi=%d", name, a.kq, a.epi); return CRFP_E_UNSUPPORTED; }
        CRFP_CHECK_LAUNCH();
        return 0;
    }
    if (a.dst2) {   // the second destination (the new state): forward_resblocks_3's last conv, one input quad
        if (a.epi != NE_PLAIN || a.kq != 1) { set_error("conv_narrow %s: a second destination needs a plain one-quad stencil (kq=%d epi=%d)", name, a.kq, a.epi); return CRFP_E_UNSUPPORTED; }
        conv3x3_narrow_kernel<1, NE_PLAIN, false, true><<<grid, 256, 0, s>>>(a);
        CRFP_CHECK_LAUNCH();
        return 0;
    }
#define CRFP_NARROW_LAUNCH(KQ_)                                                                    \
    switch (a.epi) {                                                                               \
        case NE_PLAIN: conv3x3_narrow_kernel<KQ_, NE_PLAIN><<<grid, 256, 0, s>>>(a); break;        \
        case NE_BLEND: conv3x3_narrow_kernel<KQ_, NE_BLEND><<<grid, 256, 0, s>>>(a); break;        \
        case NE_LAST: conv3x3_narrow_kernel<KQ_, NE_LAST><<<grid, 256, 0, s>>>(a); break;          \
        default: conv3x3_narrow_kernel<KQ_, NE_OFFMASK3><<<grid, 256, 0, s>>>(a); break;           \
    }
    switch (a.kq) {
        case 1: CRFP_NARROW_LAUNCH(1) break;
        case 2: CRFP_NARROW_LAUNCH(2) break;
        default: CRFP_NARROW_LAUNCH(3) break;
    }
#undef CRFP_NARROW_LAUNCH
    CRFP_CHECK_LAUNCH();
    return 0;
}

}  // namespace CRFP_NS
