// 3x3 stride-1 pad-1 convolution as an implicit GEMM on the fp32 MFMA (v_mfma_f32_32x32x2_f32),
// gfx950.  Replaces every "wide" nn.Conv2d(k=3) of the reference path (model/CRFP.py:303-317,
// 449-450, 532, 172-176, 257-261, 747-795; model/LTE.py:40-42).
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

namespace crfp {

constexpr int TW = 64, LW = TW + 2;
#ifndef CRFP_TAP_UNROLL
#define CRFP_TAP_UNROLL 3
#endif

__device__ __forceinline__ float4 load_src_quad(const ConvSrc& s, int n, int kql, int gy, int gx, int H, int W) {
    const float* base = s.p + (long long)n * s.bstride;
    switch (s.kind) {
        case SRC_Q4:
            return *reinterpret_cast<const float4*>(base + (((long long)kql * H + gy) * W + gx) * 4);
        case SRC_NCHW: {
            float v[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int ch = 4 * kql + c;
                v[c] = ch < s.nch ? base[((long long)ch * H + gy) * W + gx] : 0.0f;
            }
            return make_float4(v[0], v[1], v[2], v[3]);
        }
        case SRC_UNSHUF4: {
            const int Qp = kql >> 4, ij = kql & 15, i = ij >> 2, jj = ij & 3;
            const int H4 = 4 * H, W4 = 4 * W;
            return *reinterpret_cast<const float4*>(base + (((long long)Qp * H4 + 4 * gy + i) * W4 + 4 * gx + jj) * 4);
        }
        case SRC_FLOW2: {
            const float2 f = *reinterpret_cast<const float2*>(base + ((long long)gy * W + gx) * 2);
            return make_float4(f.x, f.y, 0.0f, 0.0f);
        }
        default:
            return make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
}

__device__ __forceinline__ float apply_act(float v, int act) {
    switch (act) {
        case CRFP_ACT_RELU: return fmaxf(v, 0.0f);
        case CRFP_ACT_LRELU01: return v > 0.0f ? v : 0.1f * v;
        case CRFP_ACT_TANH: return tanhf(v);
        case CRFP_ACT_SIGMOID: return 1.0f / (1.0f + expf(-v));
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
            const float* b = base + (long long)kql * H * W * 4;
#pragma unroll
            for (int k = 0; k < NIN; ++k)
                if (ok[k]) r[k] = *reinterpret_cast<const float4*>(b + ((long long)gy[k] * W + gx[k]) * 4);
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
            const int H4 = 4 * H, W4 = 4 * W;
            const float* b = base + (long long)Qp * H4 * W4 * 4;
#pragma unroll
            for (int k = 0; k < NIN; ++k)
                if (ok[k]) r[k] = *reinterpret_cast<const float4*>(b + ((long long)(4 * gy[k] + i) * W4 + 4 * gx[k] + jj) * 4);
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
        default: break;
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
template <int CT, int RPW>
__global__ __launch_bounds__(256, 2) void conv3x3_mfma_kernel(const ConvArgs a) {
    constexpr int TH = 4 * RPW, LH = TH + 2, PT = 2 * RPW;
    // One K-chunk (2 quads = 8 input channels) lives in LDS at a time: the halo tile of both quads
    // and the packed weights of the chunk for the CT cout tiles.  The NEXT chunk's global loads are
    // issued into registers before the MFMAs of the current chunk start, so HBM/L2 latency hides
    // behind 144*CT MFMAs per wave; LDS is rewritten between two barriers.
    __shared__ float4 tile[2][LH][LW];
    __shared__ float4 wlds[CT][9 * 64];
    constexpr int NIN = (LH * LW + 255) / 256;       // 3 halo elements per thread per quad
    constexpr int NW = (CT * 9 * 64 + 255) / 256;    // 3 (CT=1) or 5 (CT=2) weight float4 per thread
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int tiles_x = (a.W + TW - 1) / TW;
    const int tx0 = (blockIdx.x % tiles_x) * TW, ty0 = (blockIdx.x / tiles_x) * TH;
    const int T0 = blockIdx.y * CT;
    const int n = blockIdx.z;
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
    const f32x4* __restrict__ wp = reinterpret_cast<const f32x4*>(a.wpk);
    float4 rin0[NIN], rin1[NIN];
    f32x4 rw[NW];

#define CRFP_ISSUE_LOADS(PAIR)                                                                          \
    {                                                                                                   \
        {                                                                                               \
            int kql = 2 * (PAIR), s = 0;                                                                \
            while (s < a.nsrc - 1 && kql >= a.src[s].nq) { kql -= a.src[s].nq; ++s; }                  \
            load_quad_batch<NIN>(rin0, a.src[s], n, kql, sgy, sgx, sval, H, W);                         \
        }                                                                                               \
        {                                                                                               \
            int kql = 2 * (PAIR) + 1, s = 0;                                                            \
            while (s < a.nsrc - 1 && kql >= a.src[s].nq) { kql -= a.src[s].nq; ++s; }                  \
            load_quad_batch<NIN>(rin1, a.src[s], n, kql, sgy, sgx, sval, H, W);                         \
        }                                                                                               \
        load_weight_batch<CT, NW>(rw, wp, T0, npairs, (PAIR), tid);                                     \
    }

    CRFP_ISSUE_LOADS(0)
    for (int pair = 0; pair < npairs; ++pair) {
        __syncthreads();  // every wave finished reading the previous chunk
#pragma unroll
        for (int k = 0; k < NIN; ++k) {
            const int idx = tid + 256 * k;
            if (idx < LH * LW) {
                (&tile[0][0][0])[idx] = rin0[k];
                (&tile[1][0][0])[idx] = rin1[k];
            }
        }
#pragma unroll
        for (int k = 0; k < NW; ++k) {
            const int idx = tid + 256 * k;
            if (idx < CT * 576) reinterpret_cast<f32x4*>(&wlds[0][0])[idx] = rw[k];
        }
        __syncthreads();
        if (pair + 1 < npairs) CRFP_ISSUE_LOADS(pair + 1)
#pragma unroll CRFP_TAP_UNROLL
        for (int tap = 0; tap < 9; ++tap) {
            const int ky = tap / 3, kx = tap - 3 * ky;
            float4 wa[CT];
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) wa[ct] = wlds[ct][tap * 64 + lane];
            float4 b[PT];
#pragma unroll
            for (int pt = 0; pt < PT; ++pt) b[pt] = tile[h][wave * RPW + (pt >> 1) + ky][(pt & 1) * 32 + j + kx];
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int pt = 0; pt < PT; ++pt) {
                    acc[ct][pt] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[ct].x, b[pt].x, acc[ct][pt], 0, 0, 0);
                    acc[ct][pt] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[ct].y, b[pt].y, acc[ct][pt], 0, 0, 0);
                    acc[ct][pt] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[ct].z, b[pt].z, acc[ct][pt], 0, 0, 0);
                    acc[ct][pt] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[ct].w, b[pt].w, acc[ct][pt], 0, 0, 0);
                }
        }
    }
#undef CRFP_ISSUE_LOADS

    // ---------------- epilogue: bias, activation, scale, residual, layout-aware store
    const int nrows = conv_packed_rows(a.cout, a.store, a.ps_r);
    const int ncq = (nrows + 3) >> 2;
    const float4* __restrict__ bp = reinterpret_cast<const float4*>(a.bpk);
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
        const int y = ty0 + wave * RPW + (pt >> 1), x = tx0 + (pt & 1) * 32 + j;
        if (y >= H || x >= W) continue;
        float2 fl = make_float2(0.0f, 0.0f);
        if (a.store == ST_OFFMASK)
            fl = *reinterpret_cast<const float2*>(a.flow + (long long)n * a.flow_bstride + ((long long)y * W + x) * 2);
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int cq = (T0 + ct) * 8 + 2 * g + h;
                if (cq >= ncq) continue;
                const float4 bb = bp[cq];
                float v[4] = {acc[ct][pt][4 * g + 0] + bb.x, acc[ct][pt][4 * g + 1] + bb.y,
                              acc[ct][pt][4 * g + 2] + bb.z, acc[ct][pt][4 * g + 3] + bb.w};
                if (a.store == ST_OFFMASK) {
                    if (cq < a.n_off_quads) {  // (dy,dx) pairs: 10*tanh(.) + flow flipped to (y,x)
                        v[0] = 10.0f * tanhf(v[0]) + fl.y;
                        v[1] = 10.0f * tanhf(v[1]) + fl.x;
                        v[2] = 10.0f * tanhf(v[2]) + fl.y;
                        v[3] = 10.0f * tanhf(v[3]) + fl.x;
                    } else {
#pragma unroll
                        for (int c = 0; c < 4; ++c) v[c] = 1.0f / (1.0f + expf(-v[c]));
                    }
                } else {
#pragma unroll
                    for (int c = 0; c < 4; ++c) v[c] = apply_act(v[c], a.act) * a.post_scale;
                }
                if (a.store != ST_PS) {  // zero the padding components of a ragged last quad
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        if (4 * cq + c >= a.cout) v[c] = 0.0f;
                }
                if (a.resid) {
                    const float4 r = *reinterpret_cast<const float4*>(
                        a.resid + (long long)n * a.resid_bstride + (((long long)cq * H + y) * W + x) * 4);
                    v[0] += r.x; v[1] += r.y; v[2] += r.z; v[3] += r.w;
                }
                if (a.store == ST_Q4 || a.store == ST_OFFMASK) {
#pragma unroll
                    for (int d = 0; d < CRFP_MAX_DST; ++d)
                        if (d < a.ndst && cq >= a.dst[d].q0 && cq < a.dst[d].q1)
                            *reinterpret_cast<float4*>(a.dst[d].p + (long long)n * a.dst[d].bstride +
                                                       (((long long)(cq - a.dst[d].q0) * H + y) * W + x) * 4) =
                                make_float4(v[0], v[1], v[2], v[3]);
                } else if (a.store == ST_PS) {
                    const int r = a.ps_r, r2 = r * r;
                    const int Q = cq / r2, s = cq - Q * r2, i = s / r, jj = s - i * r;
                    *reinterpret_cast<float4*>(a.dst[0].p + (long long)n * a.dst[0].bstride +
                                               (((long long)Q * a.dstH + y * r + i) * a.dstW + x * r + jj) * 4) =
                        make_float4(v[0], v[1], v[2], v[3]);
                } else {  // ST_NCHW
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const int ch = 4 * cq + c;
                        if (ch < a.cout)
                            a.dst[0].p[(long long)n * a.dst[0].bstride + ((long long)ch * H + y) * W + x] = v[c];
                    }
                }
            }
    }
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

int launch_conv_mfma(const ConvArgs& a, const char* name, hipStream_t s) {
    if (a.kq & 1 || a.kq < 2 || a.ctiles < 1 || a.nsrc < 1 || a.nsrc > CRFP_MAX_SRC) {
        set_error("conv_mfma %s: bad plan (kq=%d ctiles=%d nsrc=%d)", name, a.kq, a.ctiles, a.nsrc);
        return CRFP_E_BADARG;
    }
    static const int max_ct = getenv("CRFP_CONV_CT") ? atoi(getenv("CRFP_CONV_CT")) : 2;  // tuning knob
    const bool ct2 = a.ctiles % 2 == 0 && max_ct >= 2;
    const int TH = ct2 ? 4 : 8;
    const int tiles = ((a.W + TW - 1) / TW) * ((a.H + TH - 1) / TH);
    const double px = (double)a.N * a.H * a.W;
    double in_ch = 0;
    for (int i = 0; i < a.nsrc; ++i) in_ch += a.src[i].kind == SRC_ZERO ? 0 : a.src[i].nch;
    ProfScope prof(name, s, px * (in_ch + a.cout) * 4.0 + (double)a.cout * in_ch * 9 * 4.0,
                   2.0 * px * a.cout * in_ch * 9.0);
    if (ct2) {
        dim3 grid(tiles, a.ctiles / 2, a.N);
        conv3x3_mfma_kernel<2, 1><<<grid, 256, 0, s>>>(a);
    } else {
        dim3 grid(tiles, a.ctiles, a.N);
        conv3x3_mfma_kernel<1, 2><<<grid, 256, 0, s>>>(a);
    }
    CRFP_CHECK_LAUNCH();
    return 0;
}

}  // namespace crfp
