// SPyNet (reference model/CRFP.py:554-741; `conv` with ReLU BEFORE the convolution :145-152): the optional flow network of
// the path.  CRFP_DSV itself runs FNet (model/CRFP.py:1405-1406 -- the SPyNet line is commented out), so this is not on
// the per-frame hot path; it is built as ONE native call (crfp_spynet_forward) over plain NCHW fp32 tensors:
//   resize both frames to a multiple of 32 (bilinear, align_corners=False) fused with (x - mean) / std   (:586-591,620-621,716-726)
//   5 x avg_pool2d(2, 2)                                                                                  (:624-636)
//   per level, coarse to fine: flow_up = 2 * bilinear_x2(flow, align_corners=True) (:647-652); warped = flow_warp(supp,
//   flow_up, border) (:655-657); out = 5 x (ReLU -> conv7x7) on the virtual concat [ref | warped | flow_up] (:658-659,
//   :693-734); flow = flow_up + out (:660)
//   resize the flow back (align_corners=False) and rescale its components by w / w_up, h / h_up           (:728-739)
// Round 3: the 30 7x7 convolutions of the pyramid run on the fp32 MFMA as 3x3 convolutions over nine shifted views of their input
// (spy_conv7_mfma below): 28.6 -> ~5 ms per 192 x 320 pair.  The direct fp32 convolution (thread = one pixel x 16 output
// channels, 16 x 16 pixel tile, input halo staged in LDS four channels at a time, weights through the scalar cache) stays
// behind the stand-alone crfp_convkxk_f32 operator, which has no workspace to pack weights into.
#include "crfp_common.h"

#include <cstring>

namespace crfp {

constexpr int SPY_T = 16, SPY_CI = 4, SPY_CO = 16;

struct SpyConvArgs {
    const float* src[3];   // virtual concat of up to 3 NCHW tensors
    int nch[3];
    int nsrc;
    const float* w;        // [cout][cin][K][K]
    const float* b;        // [cout]
    const float* resid;    // optional [n][cout][H][W], added to the result
    float* out;            // [n][cout][H][W]
    int N, cin, cout, H, W, pre_relu;
};

template <int K>
__global__ __launch_bounds__(256) void spy_conv_kernel(const SpyConvArgs a) {
    constexpr int P = K / 2, LT = SPY_T + K - 1;
    __shared__ float tile[SPY_CI][LT][LT + 1];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int x0 = blockIdx.x * SPY_T, y0 = blockIdx.y * SPY_T;
    const int cob = a.cout > SPY_CO ? (a.cout + SPY_CO - 1) / SPY_CO : 1;
    const int n = blockIdx.z / cob, co0 = (blockIdx.z - n * cob) * SPY_CO;
    const int H = a.H, W = a.W;
    const long long HW = (long long)H * W;
    float acc[SPY_CO];
#pragma unroll
    for (int o = 0; o < SPY_CO; ++o) acc[o] = co0 + o < a.cout ? a.b[co0 + o] : 0.0f;
    for (int c0 = 0; c0 < a.cin; c0 += SPY_CI) {
        __syncthreads();
        for (int idx = threadIdx.x; idx < SPY_CI * LT * LT; idx += 256) {
            const int ci = idx / (LT * LT), rem = idx - ci * LT * LT, r = rem / LT, c = rem - r * LT;
            const int ch = c0 + ci, gy = y0 + r - P, gx = x0 + c - P;
            float v = 0.0f;
            if (ch < a.cin && gy >= 0 && gy < H && gx >= 0 && gx < W) {
                int s = 0, cl = ch;
                while (s < a.nsrc - 1 && cl >= a.nch[s]) { cl -= a.nch[s]; ++s; }
                v = a.src[s][((long long)n * a.nch[s] + cl) * HW + (long long)gy * W + gx];
                if (a.pre_relu) v = fmaxf(v, 0.0f);
            }
            tile[ci][r][c] = v;
        }
        __syncthreads();
#pragma unroll
        for (int ci = 0; ci < SPY_CI; ++ci) {
            if (c0 + ci >= a.cin) break;
            for (int ky = 0; ky < K; ++ky)
#pragma unroll
                for (int kx = 0; kx < K; ++kx) {
                    const float v = tile[ci][ty + ky][tx + kx];
                    const float* wp = a.w + ((long long)co0 * a.cin + c0 + ci) * K * K + ky * K + kx;   // workgroup-uniform
#pragma unroll
                    for (int o = 0; o < SPY_CO; ++o)
                        if (co0 + o < a.cout) acc[o] = fmaf(wp[(long long)o * a.cin * K * K], v, acc[o]);
                }
        }
    }
    const int x = x0 + tx, y = y0 + ty;
    if (x >= W || y >= H) return;
#pragma unroll
    for (int o = 0; o < SPY_CO; ++o)
        if (co0 + o < a.cout) {
            const long long oi = ((long long)n * a.cout + co0 + o) * HW + (long long)y * W + x;
            a.out[oi] = acc[o] + (a.resid ? a.resid[oi] : 0.0f);
        }
}

static int launch_spy_conv(const SpyConvArgs& a, int K, hipStream_t s) {
    const int cob = (a.cout + SPY_CO - 1) / SPY_CO;
    dim3 grid((a.W + SPY_T - 1) / SPY_T, (a.H + SPY_T - 1) / SPY_T, a.N * cob);
    ProfScope prof("spynet_conv", s, (double)a.N * a.H * a.W * (a.cin + a.cout) * 4.0, 2.0 * a.N * a.H * a.W * a.cin * a.cout * K * K);
    switch (K) {
        case 3: spy_conv_kernel<3><<<grid, 256, 0, s>>>(a); break;
        case 5: spy_conv_kernel<5><<<grid, 256, 0, s>>>(a); break;
        case 7: spy_conv_kernel<7><<<grid, 256, 0, s>>>(a); break;
        default: set_error("convkxk: kernel size %d unsupported (3, 5, 7)", K); return CRFP_E_UNSUPPORTED;
    }
    CRFP_CHECK_LAUNCH();
    return 0;
}

// ---- 7x7 convolution on the MFMA (round 3).  A 7x7 kernel padded to 9x9 is a 3x3 grid of 3x3 blocks: with S_ij = the input
// shifted by (3 i - 3, 3 j - 3) (zero outside the image),
//     conv7(x)[y, x] = sum_{i, j} conv3(S_ij; W9_ij)[y, x],   W9_ij[ky][kx] = W7[3 i + ky - 1][3 j + kx - 1] (0 outside 0..6),
// i.e. ONE 3x3 convolution over 9 cin "virtual" channels.  The nine shifted views are nine SRC_NCHW_SHIFT sources of the same
// tensor (conv_mfma.hip: shifted read, own zero-padding test, ReLU on the way in), so the library's implicit-GEMM kernel
// (v_mfma_f32_32x32x2_f32: fp32 in, fp32 accumulate, exact fp32 products) runs it unchanged; 81 / 49 of the taps are structural
// zeros.  Weights are rearranged (spy_w9_kernel) and packed per call into the workspace.  Needs cin % 8 == 0 (two K-quads per step of the fp32-MFMA kernel).
__global__ void spy_w9_kernel(const float* __restrict__ w7, float* __restrict__ w9, int cout, int cin) {
    const int total = cout * 9 * cin * 9;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const int tap = idx % 9, vc = (idx / 9) % (9 * cin), o = idx / (81 * cin);
        const int sblk = vc / cin, c = vc - sblk * cin, i = sblk / 3, j = sblk - 3 * i;
        const int u = 3 * i + tap / 3 - 1, v = 3 * j + tap % 3 - 1;
        w9[idx] = (u >= 0 && u < 7 && v >= 0 && v < 7) ? w7[((long long)o * cin + c) * 49 + u * 7 + v] : 0.0f;
    }
}

// [ref(3) | warped(3) | flow(2)] -> one 8-channel NCHW tensor (the first conv of a basic module reads one source nine times)
__global__ void spy_cat8_kernel(const float* __restrict__ a3, const float* __restrict__ b3, const float* __restrict__ c2,
                                float* __restrict__ out, long long HW) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int n = blockIdx.y;
    if (i >= HW) return;
#pragma unroll
    for (int c = 0; c < 3; ++c) out[((long long)n * 8 + c) * HW + i] = a3[((long long)n * 3 + c) * HW + i];
#pragma unroll
    for (int c = 0; c < 3; ++c) out[((long long)n * 8 + 3 + c) * HW + i] = b3[((long long)n * 3 + c) * HW + i];
#pragma unroll
    for (int c = 0; c < 2; ++c) out[((long long)n * 8 + 6 + c) * HW + i] = c2[((long long)n * 2 + c) * HW + i];
}

__global__ void spy_add_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = b[i] + a[i];   // flow_up + out, the operand order of the reference (:660)
}

// partial sums of the K-split form: 9 slices x up to 64 channels of a map of at most a quarter of the padded frame (px pixels)
static size_t spy_partial_floats(int n, size_t px) { return (size_t)9 * 64 * (px / 4) * (size_t)(n > 0 ? n : 1); }
static size_t spy_mfma_scratch_floats(int cin, int cout) {
    const size_t ctiles = (cout + 31) / 32, kq = 9 * (size_t)cin / 4;
    return align_up((size_t)cout * 81 * cin, 64) + align_up(ctiles * (kq / 2) * 9 * 64 * 4, 64) + align_up(ctiles * 32, 64);
}

// out[n][c][px] = bias[c] + sum over the 9 K slices, in slice order (deterministic)
__global__ void spy_sum9_kernel(const float* __restrict__ part, const float* __restrict__ bias, float* __restrict__ out, long long per,
                                long long HW, int cout, int n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= per * n) return;
    const long long b = i / per, r = i - b * per;
    float acc = bias[(int)(r / HW)];
#pragma unroll
    for (int sb = 0; sb < 9; ++sb) acc += part[(b * 9 + sb) * per + r];
    out[i] = acc;
}

// out[n,cout,H,W] = conv7x7(relu?(x[n,cin,H,W])) + bias on the fp32 MFMA; scratch: spy_mfma_scratch_floats(cin, cout) floats
static int spy_conv7_mfma(const float* x, int cin, const float* w7, const float* bias, float* out, int cout, int n, int H, int W,
                          int pre_relu, float* scratch, size_t partial_floats, hipStream_t s) {
    ConvArgs a;
    memset(&a, 0, sizeof(a));
    a.nsrc = 9;
    for (int sb = 0; sb < 9; ++sb) {
        ConvSrc& d = a.src[sb];
        d.p = x; d.bstride = (long long)cin * H * W; d.kind = SRC_NCHW_SHIFT; d.nch = cin; d.nq = cin / 4; d.cbase = sb * cin; d.pad = 0;
        d.rsv = (3 * (sb / 3) - 3 + 8) | ((3 * (sb % 3) - 3 + 8) << 4) | (pre_relu ? 256 : 0);
    }
    a.kq = 9 * cin / 4; a.cin_total = 9 * cin; a.cout = cout; a.ctiles = (cout + 31) / 32;
    a.N = n; a.H = H; a.W = W; a.act = CRFP_ACT_NONE; a.post_scale = 1.0f; a.store = ST_NCHW;
    a.ndst = 1; a.dst[0].p = out; a.dst[0].bstride = (long long)cout * H * W; a.dst[0].q0 = 0; a.dst[0].q1 = (cout + 3) / 4;
    float* w9 = scratch;
    float* wpk = w9 + align_up((size_t)cout * 81 * cin, 64);
    float* bpk = wpk + align_up(conv_packed_weight_floats(a), 64);
    float* part = bpk + align_up((size_t)a.ctiles * 32, 64);
    const int total = cout * 81 * cin;
    spy_w9_kernel<<<(total + 255) / 256 < 512 ? (total + 255) / 256 : 512, 256, 0, s>>>(w7, w9, cout, cin);
    CRFP_CHECK_LAUNCH();
    // Coarse pyramid levels: the whole conv is a handful of workgroups, each walking all 9 cin / 8 K chunks on the (slow) fp32 MFMA --
    // ~200 us whatever the map size.  There every shifted view gets its own workgroup (ConvArgs::ksplit = 9) that writes a partial sum,
    // and a fixed-order pass adds bias + the nine partials: deterministic, 9x the parallelism.
    const long long wgs = (long long)((W + 63) / 64) * ((H + 3) / 4) * a.ctiles * n;
    const bool split = wgs * 9 <= 2048 && (size_t)9 * n * cout * H * W <= partial_floats;
    int rc = launch_conv_pack(a, w9, split ? nullptr : bias, nullptr, nullptr, cout, wpk, bpk, s);
    if (rc) return rc;
    a.wpk = wpk; a.bpk = bpk;
    if (!split) return launch_conv_mfma(a, "spynet_conv7_mfma", s);
    a.ksplit = 9;
    a.dst[0].p = part;
    rc = launch_conv_mfma(a, "spynet_conv7_mfma_ksplit", s);
    if (rc) return rc;
    const long long per = (long long)cout * H * W;
    spy_sum9_kernel<<<(unsigned)((per * n + 255) / 256), 256, 0, s>>>(part, bias, out, per, (long long)H * W, cout, n);
    CRFP_CHECK_LAUNCH();
    return 0;
}

// F.interpolate(size=(OH,OW), bilinear, align_corners=False) of [n,3,H,W] fused with (x - mean[c]) / std[c]
__global__ void spy_resize_norm_kernel(const float* __restrict__ x, float* __restrict__ out, int H, int W, int OH, int OW,
                                       float sh, float sw, float m0, float m1, float m2, float s0, float s1, float s2) {
    const int ox = blockIdx.x * 64 + (threadIdx.x & 63), oy = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (ox >= OW || oy >= OH) return;
    const long long nc = blockIdx.z;
    const int c = (int)(nc % 3);
    const float mean = c == 0 ? m0 : (c == 1 ? m1 : m2), sd = c == 0 ? s0 : (c == 1 ? s1 : s2);
    float sy = sh * ((float)oy + 0.5f) - 0.5f, sx = sw * ((float)ox + 0.5f) - 0.5f;
    sy = sy < 0.0f ? 0.0f : sy; sx = sx < 0.0f ? 0.0f : sx;
    const int y0 = min((int)floorf(sy), H - 1), x0 = min((int)floorf(sx), W - 1), y1 = min(y0 + 1, H - 1), x1 = min(x0 + 1, W - 1);
    const float ly1 = fminf(fmaxf(sy - (float)y0, 0.0f), 1.0f), lx1 = fminf(fmaxf(sx - (float)x0, 0.0f), 1.0f);
    const float ly0 = 1.0f - ly1, lx0 = 1.0f - lx1;
    const float* p = x + nc * H * W;
    const float v = ly0 * (lx0 * p[(long long)y0 * W + x0] + lx1 * p[(long long)y0 * W + x1]) +
                    ly1 * (lx0 * p[(long long)y1 * W + x0] + lx1 * p[(long long)y1 * W + x1]);
    out[(nc * OH + oy) * OW + ox] = (v - mean) / sd;
}

// F.interpolate(scale_factor=2, bilinear, align_corners=True) * mul on NCHW planes: src = dst * (in - 1) / (out - 1)
__global__ void upsample_ac_nchw_kernel(const float* __restrict__ x, float* __restrict__ out, int H, int W, int OH, int OW,
                                        float mul) {
    const int ox = blockIdx.x * 64 + (threadIdx.x & 63), oy = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (ox >= OW || oy >= OH) return;
    const long long nc = blockIdx.z;
    const float rh = OH > 1 ? (float)(H - 1) / (float)(OH - 1) : 0.0f, rw = OW > 1 ? (float)(W - 1) / (float)(OW - 1) : 0.0f;
    const float sy = rh * (float)oy, sx = rw * (float)ox;
    const int y0 = min((int)sy, H - 1), x0 = min((int)sx, W - 1), y1 = min(y0 + 1, H - 1), x1 = min(x0 + 1, W - 1);
    const float ly1 = sy - (float)y0, lx1 = sx - (float)x0, ly0 = 1.0f - ly1, lx0 = 1.0f - lx1;
    const float* p = x + nc * H * W;
    out[(nc * OH + oy) * OW + ox] = mul * (ly0 * (lx0 * p[(long long)y0 * W + x0] + lx1 * p[(long long)y0 * W + x1]) +
                                           ly1 * (lx0 * p[(long long)y1 * W + x0] + lx1 * p[(long long)y1 * W + x1]));
}

// flow_warp(x, flow, padding_mode='border') on NCHW planes with an NCHW flow (channel 0 = dx, 1 = dy): the reference's
// coordinate round trip (model/CRFP.py:118-128) and ATen's border clamp
__global__ void spy_warp_border_kernel(const float* __restrict__ x, const float* __restrict__ flow, float* __restrict__ out,
                                       int C, int H, int W) {
    const int px = blockIdx.x * 64 + (threadIdx.x & 63), py = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (px >= W || py >= H) return;
    const int n = blockIdx.z;
    const long long HW = (long long)H * W, pix = (long long)py * W + px;
    const float fx = flow[(long long)n * 2 * HW + pix], fy = flow[((long long)n * 2 + 1) * HW + pix];
    const float dw = (float)(W - 1 > 1 ? W - 1 : 1), dh = (float)(H - 1 > 1 ? H - 1 : 1);
    const float gx = 2.0f * ((float)px + fx) / dw - 1.0f, gy = 2.0f * ((float)py + fy) / dh - 1.0f;
    float ix = (gx + 1.0f) * ((float)(W - 1) / 2.0f), iy = (gy + 1.0f) * ((float)(H - 1) / 2.0f);
    ix = fminf(fmaxf(ix, 0.0f), (float)(W - 1));
    iy = fminf(fmaxf(iy, 0.0f), (float)(H - 1));
    const float flx = floorf(ix), fly = floorf(iy);
    const int x0 = (int)flx, y0 = (int)fly, x1 = min(x0 + 1, W - 1), y1 = min(y0 + 1, H - 1);
    const float lx = ix - flx, ly = iy - fly, hx = 1.0f - lx, hy = 1.0f - ly;
    // corners beyond the last row / column carry weight 0 under border padding (ix <= W - 1)
    for (int c = 0; c < C; ++c) {
        const float* p = x + ((long long)n * C + c) * HW;
        out[((long long)n * C + c) * HW + pix] = p[(long long)y0 * W + x0] * (hy * hx) + p[(long long)y0 * W + x1] * (hy * lx) +
                                                 p[(long long)y1 * W + x0] * (ly * hx) + p[(long long)y1 * W + x1] * (ly * lx);
    }
}

// out[n,c] = mul[c] * resize(x[n,c]) (align_corners=False): the final flow resize + per-component rescale (:728-739)
__global__ void spy_resize_scale2_kernel(const float* __restrict__ x, float* __restrict__ out, int H, int W, int OH, int OW,
                                         float sh, float sw, float mulx, float muly) {
    const int ox = blockIdx.x * 64 + (threadIdx.x & 63), oy = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (ox >= OW || oy >= OH) return;
    const long long nc = blockIdx.z;
    float sy = sh * ((float)oy + 0.5f) - 0.5f, sx = sw * ((float)ox + 0.5f) - 0.5f;
    sy = sy < 0.0f ? 0.0f : sy; sx = sx < 0.0f ? 0.0f : sx;
    const int y0 = min((int)floorf(sy), H - 1), x0 = min((int)floorf(sx), W - 1), y1 = min(y0 + 1, H - 1), x1 = min(x0 + 1, W - 1);
    const float ly1 = fminf(fmaxf(sy - (float)y0, 0.0f), 1.0f), lx1 = fminf(fmaxf(sx - (float)x0, 0.0f), 1.0f);
    const float ly0 = 1.0f - ly1, lx0 = 1.0f - lx1;
    const float* p = x + nc * H * W;
    out[(nc * OH + oy) * OW + ox] = ((nc & 1) ? muly : mulx) *
        (ly0 * (lx0 * p[(long long)y0 * W + x0] + lx1 * p[(long long)y0 * W + x1]) +
         ly1 * (lx0 * p[(long long)y1 * W + x0] + lx1 * p[(long long)y1 * W + x1]));
}

static inline dim3 px_grid(int W, int H, int Z) { return dim3((W + 63) / 64, (H + 3) / 4, Z); }
static inline int up32(int v) { return v % 32 == 0 ? v : 32 * (v / 32 + 1); }

}  // namespace crfp

using namespace crfp;

extern "C" {

int crfp_convkxk_f32(const float* x, const float* weight, const float* bias, float* out, int n, int cin, int cout, int h, int w,
                     int k, int pre_relu, void* stream) {
    if (!x || !weight || !bias || !out || n < 1 || cin < 1 || cout < 1 || h < 1 || w < 1) { set_error("convkxk: bad argument"); return CRFP_E_BADARG; }
    SpyConvArgs a;
    memset(&a, 0, sizeof(a));
    a.src[0] = x; a.nch[0] = cin; a.nsrc = 1; a.w = weight; a.b = bias; a.out = out;
    a.N = n; a.cin = cin; a.cout = cout; a.H = h; a.W = w; a.pre_relu = pre_relu;
    return launch_spy_conv(a, k, (hipStream_t)stream);
}

int crfp_upsample_bilinear_ac_f32(const float* x, float* out, int n, int c, int h, int w, int oh, int ow, float mul, void* stream) {
    if (!x || !out || n < 1 || c < 1 || h < 1 || w < 1 || oh < 1 || ow < 1) { set_error("upsample_ac: bad argument"); return CRFP_E_BADARG; }
    upsample_ac_nchw_kernel<<<px_grid(ow, oh, n * c), 256, 0, (hipStream_t)stream>>>(x, out, h, w, oh, ow, mul);
    CRFP_CHECK_LAUNCH();
    return 0;
}

size_t crfp_spynet_workspace_bytes(int n, int h, int w) {
    if (n < 1 || h < 1 || w < 1) return 0;
    const size_t px = (size_t)up32(h) * up32(w);
    // pyramids of both frames (3 ch, sum over levels < 4/3), warped (3), flow x2 (2 + 2), conv ping-pong (64 + 64), the first conv's
    // 8-channel input, and the rearranged + packed weights of one 7x7 conv at a time (largest: 32 -> 64)
    return align_up((size_t)n * px * sizeof(float) * (2 * 4 + 3 + 4 + 128 + 8), 256) + align_up((spy_mfma_scratch_floats(32, 64) + spy_partial_floats(n, px)) * sizeof(float), 256) + 4096;
}

int crfp_spynet_forward(const float* const* params, const float* ref, const float* supp, float* flow, int n, int h, int w,
                        void* workspace, size_t workspace_bytes, void* stream) {
    if (!params || !ref || !supp || !flow || n < 1 || h < 1 || w < 1) { set_error("spynet: bad argument"); return CRFP_E_BADARG; }
    for (int i = 0; i < CRFP_SPYNET_NUM_PARAMS; ++i)
        if (!params[i]) { set_error("spynet: parameter %d is null", i); return CRFP_E_BADARG; }
    if (!workspace || workspace_bytes < crfp_spynet_workspace_bytes(n, h, w)) { set_error("spynet: workspace too small"); return CRFP_E_WORKSPACE; }
    hipStream_t s = (hipStream_t)stream;
    const int hu = up32(h), wu = up32(w);
    float* p = (float*)workspace;
    float* pyr[2][6];
    for (int f = 0; f < 2; ++f)
        for (int l = 0; l < 6; ++l) { pyr[f][l] = p; p += (size_t)n * 3 * (hu >> l) * (wu >> l); }   // l = 0 finest
    float* warped = p; p += (size_t)n * 3 * hu * wu;
    float* fl[2] = {p, p + (size_t)n * 2 * hu * wu}; p += (size_t)n * 4 * hu * wu;
    float* t0 = p; float* t1 = p + (size_t)n * 64 * hu * wu; p += (size_t)n * 128 * hu * wu;
    float* x8 = p; p += (size_t)n * 8 * hu * wu;
    float* wscratch = p;
    // torch computes in / out in float for size= resizes
    const float sh = (float)h / (float)hu, sw = (float)w / (float)wu;
    for (int f = 0; f < 2; ++f) {
        spy_resize_norm_kernel<<<px_grid(wu, hu, n * 3), 256, 0, s>>>(f ? supp : ref, pyr[f][0], h, w, hu, wu, sh, sw, 0.485f, 0.456f,
                                                                     0.406f, 0.229f, 0.224f, 0.225f);
        CRFP_CHECK_LAUNCH();
        for (int l = 1; l < 6; ++l) {
            int rc = launch_avgpool2_nchw(pyr[f][l - 1], pyr[f][l], n, 3, hu >> (l - 1), wu >> (l - 1), s);
            if (rc) return rc;
        }
    }
    // coarse (level index 5 = 1/32) to fine
    int cur = 0;
    if (hipMemsetAsync(fl[0], 0, (size_t)n * 2 * (hu >> 5) * (wu >> 5) * sizeof(float), s) != hipSuccess) { set_error("spynet: memset failed"); return 1; }
    static const int chans[6] = {8, 32, 64, 32, 16, 2};
    for (int level = 0; level < 6; ++level) {
        const int l = 5 - level, H = hu >> l, W = wu >> l;
        float* flow_up = fl[cur];
        if (level > 0) {   // flow_up = 2 * bilinear_x2(flow, align_corners=True)
            flow_up = fl[cur ^ 1];
            upsample_ac_nchw_kernel<<<px_grid(W, H, n * 2), 256, 0, s>>>(fl[cur], flow_up, H / 2, W / 2, H, W, 2.0f);
            CRFP_CHECK_LAUNCH();
            cur ^= 1;
        }
        spy_warp_border_kernel<<<px_grid(W, H, n), 256, 0, s>>>(pyr[1][l], flow_up, warped, 3, H, W);
        CRFP_CHECK_LAUNCH();
        const float* in = nullptr;
        spy_cat8_kernel<<<dim3((unsigned)(((long long)H * W + 255) / 256), n), 256, 0, s>>>(pyr[0][l], warped, flow_up, x8, (long long)H * W);
        CRFP_CHECK_LAUNCH();
        for (int j = 0; j < 5; ++j) {
            // 8 -> 32 -> 64 -> 32 -> 16 -> 2 on the MFMA; flow = flow_up + out of the last one (:660) in a tiny pass of its own
            float* o = (j & 1) ? t1 : t0;
            int rc = spy_conv7_mfma(j == 0 ? x8 : in, chans[j], params[(level * 5 + j) * 2], params[(level * 5 + j) * 2 + 1], o, chans[j + 1], n, H, W,
                                    1, wscratch, spy_partial_floats(n, (size_t)hu * wu), s);
            if (rc) return rc;
            in = o;
            if (j == 4) {
                const long long tot = (long long)n * 2 * H * W;
                spy_add_kernel<<<(unsigned)((tot + 255) / 256), 256, 0, s>>>(o, flow_up, fl[cur ^ 1], tot);
                CRFP_CHECK_LAUNCH();
            }
        }
        cur ^= 1;
    }
    spy_resize_scale2_kernel<<<px_grid(w, h, n * 2), 256, 0, s>>>(fl[cur], flow, hu, wu, h, w, (float)hu / (float)h, (float)wu / (float)w,
                                                                   (float)w / (float)wu, (float)h / (float)hu);
    CRFP_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
