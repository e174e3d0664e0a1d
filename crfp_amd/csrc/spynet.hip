// SPyNet (reference model/CRFP.py:554-741; `conv` with ReLU BEFORE the convolution :145-152): the optional flow network of
// the path.  CRFP_DSV itself runs FNet (model/CRFP.py:1405-1406 -- the SPyNet line is commented out), so this is not on
// the per-frame hot path; it is built as ONE native call (crfp_spynet_forward) over plain NCHW fp32 tensors:
//   resize both frames to a multiple of 32 (bilinear, align_corners=False) fused with (x - mean) / std   (:586-591,620-621,716-726)
//   5 x avg_pool2d(2, 2)                                                                                  (:624-636)
//   per level, coarse to fine: flow_up = 2 * bilinear_x2(flow, align_corners=True) (:647-652); warped = flow_warp(supp,
//   flow_up, border) (:655-657); out = 5 x (ReLU -> conv7x7) on the virtual concat [ref | warped | flow_up] (:658-659,
//   :693-734); flow = flow_up + out (:660)
//   resize the flow back (align_corners=False) and rescale its components by w / w_up, h / h_up           (:728-739)
// The 7x7 convolutions are direct fp32 convolutions: thread = one pixel x 16 output channels, 16 x 16 pixel tile, input
// halo staged in LDS four channels at a time (ReLU applied on the way in), weights through the scalar cache (they are
// uniform over the workgroup).  fp32 FMA throughout.
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
    // pyramids of both frames (3 ch, sum over levels < 4/3), warped (3), flow x2 (2 + 2), conv ping-pong (64 + 64)
    return align_up((size_t)n * px * sizeof(float) * (2 * 4 + 3 + 4 + 128), 256) + 4096;
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
    float* t0 = p; float* t1 = p + (size_t)n * 64 * hu * wu;
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
        for (int j = 0; j < 5; ++j) {
            SpyConvArgs a;
            memset(&a, 0, sizeof(a));
            if (j == 0) { a.src[0] = pyr[0][l]; a.nch[0] = 3; a.src[1] = warped; a.nch[1] = 3; a.src[2] = flow_up; a.nch[2] = 2; a.nsrc = 3; }
            else { a.src[0] = in; a.nch[0] = chans[j]; a.nsrc = 1; }
            a.w = params[(level * 5 + j) * 2]; a.b = params[(level * 5 + j) * 2 + 1];
            a.N = n; a.cin = chans[j]; a.cout = chans[j + 1]; a.H = H; a.W = W; a.pre_relu = 1;
            float* o = j == 4 ? fl[cur ^ 1] : ((j & 1) ? t1 : t0);
            a.out = o;
            a.resid = j == 4 ? flow_up : nullptr;   // flow = flow_up + out
            int rc = launch_spy_conv(a, 7, s);
            if (rc) return rc;
            in = o;
        }
        cur ^= 1;
    }
    spy_resize_scale2_kernel<<<px_grid(w, h, n * 2), 256, 0, s>>>(fl[cur], flow, hu, wu, h, w, (float)hu / (float)h, (float)wu / (float)w,
                                                                   (float)w / (float)wu, (float)h / (float)hu);
    CRFP_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
