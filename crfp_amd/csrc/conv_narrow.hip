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

namespace crfp {

__device__ __forceinline__ float4 narrow_load(const ConvSrc& s, int n, int kql, int gy, int gx, int H, int W) {
    const float* base = s.p + (long long)n * s.bstride;
    if (s.kind == SRC_FLOW2) {
        const float2 f = *reinterpret_cast<const float2*>(base + ((long long)gy * W + gx) * 2);
        return make_float4(f.x, f.y, 0.0f, 0.0f);
    }
    return *reinterpret_cast<const float4*>(base + (((long long)kql * H + gy) * W + gx) * 4);
}

__device__ __forceinline__ float n_act(float v, int act) {
    switch (act) {
        case CRFP_ACT_RELU: return fmaxf(v, 0.0f);
        case CRFP_ACT_LRELU01: return v > 0.0f ? v : 0.1f * v;
        case CRFP_ACT_TANH: return tanhf(v);
        case CRFP_ACT_SIGMOID: return 1.0f / (1.0f + expf(-v));
        default: return v;
    }
}

template <int KQ>
__global__ __launch_bounds__(256) void conv3x3_narrow_kernel(const NarrowArgs a) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int n = blockIdx.z;
    const int H = a.H, W = a.W;
    if (x >= W || y >= H) return;

    // resolve K-quad -> (source, local quad) once (uniform)
    int ksrc[KQ], klocal[KQ];
#pragma unroll
    for (int k = 0; k < KQ; ++k) {
        int kql = k, s = 0;
        while (s < a.nsrc - 1 && kql >= a.src[s].nq) { kql -= a.src[s].nq; ++s; }
        ksrc[k] = s;
        klocal[k] = kql;
    }

    float acc[4] = {a.bpk[0], a.bpk[1], a.bpk[2], a.bpk[3]};
    float4 centre = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    const float* __restrict__ w = a.wpk;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        const int gy = y + tap / 3 - 1, gx = x + tap % 3 - 1;
        const bool in = gy >= 0 && gy < H && gx >= 0 && gx < W;
#pragma unroll
        for (int k = 0; k < KQ; ++k) {
            float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            if (in) v = narrow_load(a.src[ksrc[k]], n, klocal[k], gy, gx, H, W);
            if (tap == 4 && k == 0) centre = v;
            const float* wk = w + (tap * KQ + k) * 16;
#pragma unroll
            for (int o = 0; o < 4; ++o)
                acc[o] = fmaf(wk[12 + o], v.w, fmaf(wk[8 + o], v.z, fmaf(wk[4 + o], v.y, fmaf(wk[o], v.x, acc[o]))));
        }
    }

    const long long pix = (long long)y * W + x;
    if (a.epi == NE_PLAIN) {
        float v[4];
#pragma unroll
        for (int o = 0; o < 4; ++o) v[o] = o < a.cout ? n_act(acc[o], a.act) * a.post_scale : 0.0f;
        if (a.resid) {
            const float4 r = *reinterpret_cast<const float4*>(a.resid + (long long)n * a.resid_bstride + pix * 4);
            v[0] += r.x; v[1] += r.y; v[2] += r.z; v[3] += r.w;
        }
        *reinterpret_cast<float4*>(a.dst + (long long)n * a.dst_bstride + pix * 4) = make_float4(v[0], v[1], v[2], v[3]);
    } else if (a.epi == NE_BLEND) {
        const bool m = a.mask[(long long)n * a.mask_bstride + pix] != 0;
        float v[4] = {m ? acc[0] : centre.x, m ? acc[1] : centre.y, m ? acc[2] : centre.z, m ? acc[3] : centre.w};
#pragma unroll
        for (int o = 0; o < 4; ++o) v[o] = v[o] > 0.0f ? v[o] : 0.1f * v[o];
        *reinterpret_cast<float4*>(a.dst + (long long)n * a.dst_bstride + pix * 4) = make_float4(v[0], v[1], v[2], v[3]);
    } else if (a.epi == NE_LAST) {
        const float4 b = *reinterpret_cast<const float4*>(a.base + (long long)n * a.base_bstride + pix * 4);
        float* o = a.dst + (long long)n * a.dst_bstride;
        if (a.y_only) {
            o[pix] = acc[0] + (0.299f * b.x + 0.587f * b.y + 0.114f * b.z);
        } else {
            const long long plane = (long long)H * W;
            o[pix] = acc[0] + b.x;
            o[plane + pix] = acc[1] + b.y;
            o[2 * plane + pix] = acc[2] + b.z;
        }
    } else {  // NE_OFFMASK3
        const float2 f = *reinterpret_cast<const float2*>(a.flow + (long long)n * a.flow_bstride + pix * 2);
        *reinterpret_cast<float4*>(a.dst + (long long)n * a.dst_bstride + pix * 4) =
            make_float4(10.0f * tanhf(acc[0]) + f.y, 10.0f * tanhf(acc[1]) + f.x, 1.0f / (1.0f + expf(-acc[2])), 0.0f);
    }
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

int launch_narrow(const NarrowArgs& a, const char* name, hipStream_t s) {
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
    ProfScope prof(name, s, px * (in_ch + (a.epi == NE_OFFMASK3 ? 3 : a.cout) + extra) * 4.0,
                   2.0 * px * in_ch * a.cout * 9.0);
    dim3 grid((a.W + 63) / 64, (a.H + 3) / 4, a.N);
    switch (a.kq) {
        case 1: conv3x3_narrow_kernel<1><<<grid, 256, 0, s>>>(a); break;
        case 2: conv3x3_narrow_kernel<2><<<grid, 256, 0, s>>>(a); break;
        default: conv3x3_narrow_kernel<3><<<grid, 256, 0, s>>>(a); break;
    }
    CRFP_CHECK_LAUNCH();
    return 0;
}

}  // namespace crfp
