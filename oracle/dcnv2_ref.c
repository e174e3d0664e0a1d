/* ORACLE -- TEST INFRASTRUCTURE ONLY (never linked or loaded by the product path).
 *
 * Plain-C restatement of the third-party op the reference imports as `dcn_v2.DCNv2`
 * (reference model/CRFP.py:6, ctor call :318-320, forward call :350; package named in
 * README.md:26 = github.com/jinfagang/DCNv2_latest, no version pinned -> parity unpinned by
 * the reference).  Follows the published DCNv2 algorithm: modulated deformable im2col
 * (bilinear sample per tap per deformable group, each out-of-range corner contributes 0,
 * whole sample 0 unless -1 < p < size) followed by a [Cout] x [Cin*K] contraction + bias.
 * Independent of oracle/crfp_oracle.py::dcnv2 (scalar loops, double accumulation) so the two
 * cross-check each other in tests/test_oracle_dcn.py.
 *
 * Also holds scalar restatements of flow_warp (reference model/CRFP.py:90-130, through
 * grid_sample's align_corners=True un-normalisation) used as a second opinion on the
 * coordinate arithmetic.
 */
#include <math.h>
#include <stddef.h>

static float corner(const float* plane, int H, int W, int y, int x) {
    if (y < 0 || y > H - 1 || x < 0 || x > W - 1) return 0.0f;
    return plane[(size_t)y * W + x];
}

static float bilinear_zero(const float* plane, int H, int W, float py, float px) {
    if (!(py > -1.0f && px > -1.0f && py < (float)H && px < (float)W)) return 0.0f;
    float fy = floorf(py), fx = floorf(px);
    int y0 = (int)fy, x0 = (int)fx;
    float ly = py - fy, lx = px - fx, hy = 1.0f - ly, hx = 1.0f - lx;
    return hy * hx * corner(plane, H, W, y0, x0) + hy * lx * corner(plane, H, W, y0, x0 + 1) +
           ly * hx * corner(plane, H, W, y0 + 1, x0) + ly * lx * corner(plane, H, W, y0 + 1, x0 + 1);
}

/* x[B,C,H,W] offset[B,2*dg*K,H,W] mask[B,dg*K,H,W] weight[O,C,k,k] bias[O] -> out[B,O,H,W] */
int dcnv2_ref_forward(const float* x, const float* offset, const float* mask, const float* weight,
                      const float* bias, float* out, int B, int C, int O, int H, int W, int k,
                      int pad, int dil, int dg) {
    if (C % dg != 0 || k < 1) return -1;
    const int K = k * k, cpg = C / dg;
    const size_t HW = (size_t)H * W;
    for (int b = 0; b < B; ++b)
        for (int y = 0; y < H; ++y)
            for (int xw = 0; xw < W; ++xw) {
                double acc[512];
                if (O > 512) return -2;
                for (int o = 0; o < O; ++o) acc[o] = (double)bias[o];
                for (int g = 0; g < dg; ++g)
                    for (int t = 0; t < K; ++t) {
                        int ky = t / k, kx = t % k;
                        const float* offp = offset + ((size_t)b * 2 * dg * K + 2 * (g * K + t)) * HW;
                        float oy = offp[(size_t)y * W + xw];
                        float ox = offp[HW + (size_t)y * W + xw];
                        float m = mask[((size_t)b * dg * K + g * K + t) * HW + (size_t)y * W + xw];
                        float py = (float)(y - pad + ky * dil) + oy;
                        float px = (float)(xw - pad + kx * dil) + ox;
                        for (int cc = 0; cc < cpg; ++cc) {
                            int c = g * cpg + cc;
                            float v = bilinear_zero(x + ((size_t)b * C + c) * HW, H, W, py, px) * m;
                            if (v != 0.0f)
                                for (int o = 0; o < O; ++o)
                                    acc[o] += (double)weight[((size_t)o * C + c) * K + t] * (double)v;
                        }
                    }
                for (int o = 0; o < O; ++o) out[((size_t)b * O + o) * HW + (size_t)y * W + xw] = (float)acc[o];
            }
    return 0;
}

/* x[N,C,H,W], flow[N,H,W,2] (dx,dy) -> out[N,C,H,W]; padding_mode 0 = zeros, 1 = border */
int flow_warp_ref(const float* x, const float* flow, float* out, int N, int C, int H, int W, int border) {
    const size_t HW = (size_t)H * W;
    const float dw = (float)(W - 1 > 1 ? W - 1 : 1), dh = (float)(H - 1 > 1 ? H - 1 : 1);
    for (int n = 0; n < N; ++n)
        for (int y = 0; y < H; ++y)
            for (int xw = 0; xw < W; ++xw) {
                const float* f = flow + (((size_t)n * H + y) * W + xw) * 2;
                float gx = 2.0f * ((float)xw + f[0]) / dw - 1.0f;
                float gy = 2.0f * ((float)y + f[1]) / dh - 1.0f;
                float ix = (gx + 1.0f) * ((float)(W - 1) / 2.0f);
                float iy = (gy + 1.0f) * ((float)(H - 1) / 2.0f);
                if (border) {
                    ix = fminf(fmaxf(ix, 0.0f), (float)(W - 1));
                    iy = fminf(fmaxf(iy, 0.0f), (float)(H - 1));
                }
                float fx = floorf(ix), fy = floorf(iy);
                int x0 = (int)fx, y0 = (int)fy;
                float lx = ix - fx, ly = iy - fy, hx = 1.0f - lx, hy = 1.0f - ly;
                for (int c = 0; c < C; ++c) {
                    const float* p = x + ((size_t)n * C + c) * HW;
                    out[((size_t)n * C + c) * HW + (size_t)y * W + xw] =
                        hy * hx * corner(p, H, W, y0, x0) + hy * lx * corner(p, H, W, y0, x0 + 1) +
                        ly * hx * corner(p, H, W, y0 + 1, x0) + ly * lx * corner(p, H, W, y0 + 1, x0 + 1);
                }
            }
    return 0;
}
