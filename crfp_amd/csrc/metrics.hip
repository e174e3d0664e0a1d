// Masked PSNR + SSIM partial sums in one pass over an image pair.
// Replaces utils.calc_psnr_and_ssim_cuda -> psnr_cuda / ssim_cuda / _ssim (reference utils.py:166-185,187-240,242-254):
//   x' = x*mul + add                              (the data-dependent range conversion of utils.py:244-250, chosen by the host)
//   mse  = sum_c,p m(p) (a'-b')^2 / (sum_p m(p) * C)
//   ssim = sum_c,p m(p) S_c(p)    / (sum_p m(p) * C),   S from the 11x11 gaussian (sigma 1.5) windowed means, zero padding,
//          C1 = 0.01^2, C2 = 0.03^2 (image range [0,1]).
// The reference filters with the 121-tap outer-product window; here the window is applied separably (rows, then columns) in
// fp32 -- same weights g[i]*g[j] up to fp32 rounding order.  acc[0] += sum m (a'-b')^2, acc[1] += sum m S, acc[2] += sum m.
#include "crfp_common.h"

namespace crfp {

constexpr int SW = 64, SH = 16, SR = 5, SLW = SW + 2 * SR, SLH = SH + 2 * SR;   // output tile, window radius, halo tile

struct SsimWin { float g[11]; };

__global__ __launch_bounds__(256) void psnr_ssim_partial_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                                const uint8_t* __restrict__ mask, double* __restrict__ acc,
                                                                int C, int H, int W, float mul, float add, SsimWin win) {
    __shared__ float ta[SLH][SLW], tb[SLH][SLW];
    __shared__ float hb[5][SLH][SW];        // row-filtered a, b, a^2, b^2, ab
    __shared__ double red[3][4];
    const int tid = threadIdx.x;
    const int nc = blockIdx.z, n = nc / C;
    const int x0 = blockIdx.x * SW, y0 = blockIdx.y * SH;
    const float* pa = a + (long long)nc * H * W;
    const float* pb = b + (long long)nc * H * W;
    for (int i = tid; i < SLH * SLW; i += 256) {
        const int r = i / SLW, c = i - r * SLW;
        const int gy = y0 + r - SR, gx = x0 + c - SR;
        const bool in = gy >= 0 && gy < H && gx >= 0 && gx < W;   // F.conv2d zero padding applies to the converted image
        ta[r][c] = in ? pa[(long long)gy * W + gx] * mul + add : 0.0f;
        tb[r][c] = in ? pb[(long long)gy * W + gx] * mul + add : 0.0f;
    }
    __syncthreads();
    for (int i = tid; i < SLH * SW; i += 256) {
        const int r = i / SW, c = i - r * SW;
        float s0 = 0, s1 = 0, s2 = 0, s3 = 0, s4 = 0;
#pragma unroll
        for (int k = 0; k < 11; ++k) {
            const float va = ta[r][c + k], vb = tb[r][c + k], g = win.g[k];
            s0 += g * va; s1 += g * vb; s2 += g * (va * va); s3 += g * (vb * vb); s4 += g * (va * vb);
        }
        hb[0][r][c] = s0; hb[1][r][c] = s1; hb[2][r][c] = s2; hb[3][r][c] = s3; hb[4][r][c] = s4;
    }
    __syncthreads();
    double se = 0.0, ss = 0.0, sm = 0.0;
    for (int i = tid; i < SH * SW; i += 256) {
        const int r = i / SW, c = i - r * SW;
        const int gy = y0 + r, gx = x0 + c;
        if (gy >= H || gx >= W) continue;
        float m[5] = {0, 0, 0, 0, 0};
#pragma unroll
        for (int k = 0; k < 11; ++k) {
            const float g = win.g[k];
#pragma unroll
            for (int q = 0; q < 5; ++q) m[q] += g * hb[q][r + k][c];
        }
        const float mu1 = m[0], mu2 = m[1];
        const float mu1_sq = mu1 * mu1, mu2_sq = mu2 * mu2, mu12 = mu1 * mu2;
        const float s1 = m[2] - mu1_sq, s2 = m[3] - mu2_sq, s12 = m[4] - mu12;
        const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
        const float S = ((2.0f * mu12 + C1) * (2.0f * s12 + C2)) / ((mu1_sq + mu2_sq + C1) * (s1 + s2 + C2));
        const float mk = mask ? (mask[(long long)n * H * W + (long long)gy * W + gx] ? 1.0f : 0.0f) : 1.0f;
        const float d = ta[r + SR][c + SR] - tb[r + SR][c + SR];
        se += (double)(mk * d * d);
        ss += (double)(mk * S);
        if (nc - n * C == 0) sm += (double)mk;     // the mask is shared by the C channels: count it once
    }
    for (int o = 32; o > 0; o >>= 1) { se += __shfl_down(se, o); ss += __shfl_down(ss, o); sm += __shfl_down(sm, o); }
    const int lane = tid & 63, wv = tid >> 6;
    if (lane == 0) { red[0][wv] = se; red[1][wv] = ss; red[2][wv] = sm; }
    __syncthreads();
    if (tid == 0) {
        atomicAdd(&acc[0], red[0][0] + red[0][1] + red[0][2] + red[0][3]);
        atomicAdd(&acc[1], red[1][0] + red[1][1] + red[1][2] + red[1][3]);
        atomicAdd(&acc[2], red[2][0] + red[2][1] + red[2][2] + red[2][3]);
    }
}

int launch_psnr_ssim_partial(const float* a, const float* b, const uint8_t* mask, double* acc, int N, int C, int H, int W,
                             float mul, float add, hipStream_t s) {
    SsimWin win;
    {   // utils.gaussian(11, 1.5): float32 tensor of exp(...) normalised by its float32 sum
        float g[11], sum = 0.0f;
        for (int x = 0; x < 11; ++x) { g[x] = (float)exp(-(double)((x - 5) * (x - 5)) / (2.0 * 1.5 * 1.5)); }
        for (int x = 0; x < 11; ++x) sum += g[x];
        for (int x = 0; x < 11; ++x) win.g[x] = g[x] / sum;
    }
    ProfScope prof("psnr_ssim_partial", s, (double)N * C * H * W * 8.0, (double)N * C * H * W * 2.0 * 5 * 22);
    dim3 grid((W + SW - 1) / SW, (H + SH - 1) / SH, N * C);
    psnr_ssim_partial_kernel<<<grid, 256, 0, s>>>(a, b, mask, acc, C, H, W, mul, add, win);
    CRFP_CHECK_LAUNCH();
    return 0;
}

}  // namespace crfp
