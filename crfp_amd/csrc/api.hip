// Per-operator C-ABI entry points (NCHW tensors as the reference's Python operators pass them).
#include "crfp_common.h"

#include <cstring>

using namespace crfp;

static size_t q4_bytes(int n, int c, int h, int w) { return align_up((size_t)n * ((c + 3) / 4) * h * w * 16, 256); }
static size_t p4_guard(int w) { return align_up((size_t)(w + 2) * 16, 256); }  // zeroed guard in front of plane 0
static size_t p4_bytes(int n, int c, int h, int w) {
    return p4_guard(w) + align_up((size_t)n * ((c + 3) / 4) * (h + 1) * (w + 1) * 16, 256);
}

// offset [n,2,h,w] + mask [n,1,h,w] -> the compact quad (dy, dx, mask, 0) dcn3_kernel reads
__global__ void offmask3_pack_kernel(const float* __restrict__ offset, const float* __restrict__ mask, float* __restrict__ out, long long hw, long long total) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const long long n = i / hw, p = i - n * hw;
    reinterpret_cast<float4*>(out)[i] = make_float4(offset[(2 * n) * hw + p], offset[(2 * n + 1) * hw + p], mask[n * hw + p], 0.0f);
}

extern "C" {

size_t crfp_flow_warp_workspace_bytes(int n, int c, int h, int w) { return 2 * q4_bytes(n, c, h, w); }

int crfp_flow_warp_f32(const float* x, const float* flow, float* out, int n, int c, int h, int w, int padding_mode,
                       void* workspace, size_t workspace_bytes, void* stream) {
    if (!x || !flow || !out || n < 1 || c < 1 || h < 1 || w < 1) { set_error("flow_warp: bad argument"); return CRFP_E_BADARG; }
    if (padding_mode != CRFP_PAD_ZEROS && padding_mode != CRFP_PAD_BORDER) { set_error("flow_warp: padding_mode %d unsupported", padding_mode); return CRFP_E_UNSUPPORTED; }
    if (!workspace || workspace_bytes < crfp_flow_warp_workspace_bytes(n, c, h, w)) { set_error("flow_warp: workspace too small"); return CRFP_E_WORKSPACE; }
    hipStream_t s = (hipStream_t)stream;
    float* xin = (float*)workspace;
    float* xo = (float*)((char*)workspace + q4_bytes(n, c, h, w));
    const int nq = (c + 3) / 4;
    const long long bs = (long long)nq * h * w * 4;
    int rc = launch_nchw_to_q4(x, xin, n, c, h, w, 0, s);
    if (!rc) rc = launch_flow_warp_q4(xin, bs, flow, (long long)h * w * 2, xo, bs, n, nq, h, w, padding_mode == CRFP_PAD_BORDER, 0, s);
    if (!rc) rc = launch_q4_to_nchw(xo, out, n, c, h, w, 0, s);
    return rc;
}

size_t crfp_dcnv2_workspace_bytes(int n, int cin, int cout, int h, int w, int k, int dg) {
    if (cin == 32 && cout == 32 && dg == 8 && k == 3)
        return p4_bytes(n, 32, h, w) + q4_bytes(n, 32, h, w) + q4_bytes(n, 216, h, w) + align_up(36 * 2 * 32 * 4 * sizeof(float), 256);
    return 256;
}

int crfp_dcnv2_forward_f32(const float* x, const float* offset, const float* mask, const float* weight,
                           const float* bias, float* out, int n, int cin, int cout, int h, int w, int k, int pad,
                           int dil, int dg, void* workspace, size_t workspace_bytes, void* stream) {
    if (!x || !offset || !mask || !weight || !bias || !out || n < 1 || h < 1 || w < 1) { set_error("dcnv2: bad argument"); return CRFP_E_BADARG; }
    if (k != 3 || pad != 1 || dil != 1) { set_error("dcnv2: only kernel 3, padding 1, dilation 1 (got %d,%d,%d)", k, pad, dil); return CRFP_E_UNSUPPORTED; }
    if (dg < 1 || cin % dg != 0 || cout < 1) { set_error("dcnv2: cin %d not divisible by deformable_groups %d", cin, dg); return CRFP_E_BADARG; }
    hipStream_t s = (hipStream_t)stream;
    if (cin == 32 && cout == 32 && dg == 8) {
        if (!workspace || workspace_bytes < crfp_dcnv2_workspace_bytes(n, cin, cout, h, w, k, dg)) { set_error("dcnv2: workspace too small"); return CRFP_E_WORKSPACE; }
        char* p = (char*)workspace;
        float* xq = (float*)(p + p4_guard(w)); p += p4_bytes(n, 32, h, w);   // P4: padded planes, guard + pads zeroed below
        float* oq = (float*)p; p += q4_bytes(n, 32, h, w);
        float* om = (float*)p; p += q4_bytes(n, 216, h, w);
        float* wpk = (float*)p;
        if (hipMemsetAsync((char*)xq - p4_guard(w), 0, p4_bytes(n, 32, h, w), s) != hipSuccess) { set_error("dcnv2: memset failed"); return 1; }
        int rc = launch_nchw_to_q4(x, xq, n, 32, h, w, 1, s);
        if (!rc) rc = launch_offmask_nchw_to_q4(offset, mask, om, n, 144, 72, h, w, s);
        if (!rc) rc = launch_dcn_g8_pack(weight, wpk, s);
        if (!rc) rc = launch_dcn_g8(xq, 8LL * (h + 1) * (w + 1) * 4, om, 54LL * h * w * 4, wpk, bias, oq, 8LL * h * w * 4, n, h, w, s);
        if (!rc) rc = launch_q4_to_nchw(oq, out, n, 32, h, w, 0, s);
        return rc;
    }
    return launch_dcn_generic(x, offset, mask, weight, bias, out, n, cin, cout, h, w, dg, s);
}

// ---- DCNv2 4 -> 4, one deformable group, ONE (dy, dx) and ONE mask per pixel shared by the 9 taps (the
// `offset_mask_shared_across_taps` form).  The reference builds this by tiling the 2 + 1 channels 9x before the DCNv2 call
// (model/CRFP.py:341-347); here they stay compact: offset [n,2,h,w], mask [n,1,h,w].
size_t crfp_dcnv2_shared_workspace_bytes(int n, int c, int h, int w) {
    if (n < 1 || c != 4 || h < 1 || w < 1) return 0;
    return p4_bytes(n, 4, h, w) + 2 * q4_bytes(n, 4, h, w);
}

int crfp_dcnv2_shared_f32(const float* x, const float* offset, const float* mask, const float* weight, const float* bias, float* out,
                          int n, int cin, int cout, int h, int w, void* workspace, size_t workspace_bytes, void* stream) {
    if (!x || !offset || !mask || !weight || !bias || !out || n < 1 || h < 1 || w < 1) { set_error("dcnv2_shared: bad argument"); return CRFP_E_BADARG; }
    if (cin != 4 || cout != 4) { set_error("dcnv2_shared: the shared-offset form is built for 4 -> 4 channels (got %d -> %d)", cin, cout); return CRFP_E_UNSUPPORTED; }
    if (!workspace || workspace_bytes < crfp_dcnv2_shared_workspace_bytes(n, 4, h, w)) { set_error("dcnv2_shared: workspace too small"); return CRFP_E_WORKSPACE; }
    hipStream_t s = (hipStream_t)stream;
    char* p = (char*)workspace;
    float* xq = (float*)(p + p4_guard(w)); p += p4_bytes(n, 4, h, w);   // P4: padded plane, guard + pads zero
    float* om = (float*)p; p += q4_bytes(n, 4, h, w);
    float* oq = (float*)p;
    if (hipMemsetAsync((char*)xq - p4_guard(w), 0, p4_bytes(n, 4, h, w), s) != hipSuccess) { set_error("dcnv2_shared: memset failed"); return 1; }
    int rc = launch_nchw_to_q4(x, xq, n, 4, h, w, 1, s);
    if (!rc) {   // quad = (dy, dx, mask, 0)
        const long long hw = (long long)h * w, total = hw * n;
        offmask3_pack_kernel<<<(unsigned)((total + 255) / 256), 256, 0, s>>>(offset, mask, om, hw, total);
        if (hipGetLastError() != hipSuccess) { set_error("dcnv2_shared: pack launch failed"); rc = 1; }
    }
    if (!rc) rc = launch_dcn3(xq, (long long)(h + 1) * (w + 1) * 4, om, (long long)h * w * 4, weight, bias, oq, (long long)h * w * 4, n, h, w, s);
    if (!rc) rc = launch_q4_to_nchw(oq, out, n, 4, h, w, 0, s);
    return rc;
}

// (cin = cout = 32, dg = 8) with the weights packed once: crfp_dcnv2_g8_pack_f32 -> crfp_dcnv2_g8_packed_f32
size_t crfp_dcnv2_g8_packed_bytes(void) { return align_up(36 * 2 * 32 * 4 * sizeof(float), 256); }

int crfp_dcnv2_g8_pack_f32(const float* weight, void* packed, size_t packed_bytes, void* stream) {
    if (!weight || !packed) { set_error("dcnv2_g8_pack: bad argument"); return CRFP_E_BADARG; }
    if (packed_bytes < crfp_dcnv2_g8_packed_bytes()) { set_error("dcnv2_g8_pack: packed buffer too small"); return CRFP_E_WORKSPACE; }
    return launch_dcn_g8_pack(weight, (float*)packed, (hipStream_t)stream);
}

int crfp_dcnv2_g8_packed_f32(const float* x, const float* offset, const float* mask, const void* packed, const float* bias, float* out,
                             int n, int h, int w, void* workspace, size_t workspace_bytes, void* stream) {
    if (!x || !offset || !mask || !packed || !bias || !out || n < 1 || h < 1 || w < 1) { set_error("dcnv2_g8_packed: bad argument"); return CRFP_E_BADARG; }
    if (!workspace || workspace_bytes < crfp_dcnv2_workspace_bytes(n, 32, 32, h, w, 3, 8)) { set_error("dcnv2_g8_packed: workspace too small"); return CRFP_E_WORKSPACE; }
    hipStream_t s = (hipStream_t)stream;
    char* p = (char*)workspace;
    float* xq = (float*)(p + p4_guard(w)); p += p4_bytes(n, 32, h, w);
    float* oq = (float*)p; p += q4_bytes(n, 32, h, w);
    float* om = (float*)p;
    if (hipMemsetAsync((char*)xq - p4_guard(w), 0, p4_bytes(n, 32, h, w), s) != hipSuccess) { set_error("dcnv2_g8_packed: memset failed"); return 1; }
    int rc = launch_nchw_to_q4(x, xq, n, 32, h, w, 1, s);
    if (!rc) rc = launch_offmask_nchw_to_q4(offset, mask, om, n, 144, 72, h, w, s);
    if (!rc) rc = launch_dcn_g8(xq, 8LL * (h + 1) * (w + 1) * 4, om, 54LL * h * w * 4, (const float*)packed, bias, oq, 8LL * h * w * 4, n, h, w, s);
    if (!rc) rc = launch_q4_to_nchw(oq, out, n, 32, h, w, 0, s);
    return rc;
}

static ConvArgs api_conv_plan(int cin, int cout, int act, float post_scale) {
    ConvArgs a;
    memset(&a, 0, sizeof(a));
    a.nsrc = 1;
    a.src[0].kind = SRC_NCHW;
    a.src[0].nch = cin;
    a.src[0].nq = (cin + 3) / 4;
    a.kq = a.src[0].nq;
    if (a.kq & 1) {
        a.src[1].kind = SRC_ZERO;
        a.src[1].nch = 1;
        a.src[1].nq = 1;
        a.src[1].cbase = cin;
        a.nsrc = 2;
        a.kq += 1;
    }
    a.cin_total = cin;
    a.cout = cout;
    a.store = ST_NCHW;
    a.act = act;
    a.post_scale = post_scale;
    a.ctiles = (cout + 31) / 32;
    return a;
}

size_t crfp_conv3x3_workspace_bytes(int n, int cin, int cout, int h, int w) {
    (void)n; (void)h; (void)w;
    ConvArgs a = api_conv_plan(cin, cout, 0, 1.0f);
    return align_up(conv_packed_weight_floats(a) * sizeof(float), 256) + align_up((size_t)a.ctiles * 32 * sizeof(float), 256);
}

// Weights packed once (crfp_conv3x3_pack_f32), reused by every crfp_conv3x3_packed_f32 call: the operator without the per-call repack
size_t crfp_conv3x3_packed_bytes(int cin, int cout) { return crfp_conv3x3_workspace_bytes(1, cin, cout, 1, 1); }

int crfp_conv3x3_pack_f32(const float* weight, const float* bias, int cin, int cout, void* packed, size_t packed_bytes, void* stream) {
    if (!weight || !packed || cin < 1 || cout < 1) { set_error("conv3x3_pack: bad argument"); return CRFP_E_BADARG; }
    if (packed_bytes < crfp_conv3x3_packed_bytes(cin, cout)) { set_error("conv3x3_pack: packed buffer too small"); return CRFP_E_WORKSPACE; }
    ConvArgs a = api_conv_plan(cin, cout, 0, 1.0f);
    float* wpk = (float*)packed;
    float* bpk = (float*)((char*)packed + align_up(conv_packed_weight_floats(a) * sizeof(float), 256));
    return launch_conv_pack(a, weight, bias, nullptr, nullptr, cout, wpk, bpk, (hipStream_t)stream);
}

int crfp_conv3x3_packed_f32(const float* x, const void* packed, float* out, int n, int cin, int cout, int h, int w, int act,
                            float post_scale, void* stream) {
    if (!x || !packed || !out || n < 1 || cin < 1 || cout < 1 || h < 1 || w < 1) { set_error("conv3x3_packed: bad argument"); return CRFP_E_BADARG; }
    if (act < CRFP_ACT_NONE || act > CRFP_ACT_SIGMOID) { set_error("conv3x3_packed: unknown activation %d", act); return CRFP_E_BADARG; }
    ConvArgs a = api_conv_plan(cin, cout, act, post_scale);
    a.src[0].p = x;
    a.src[0].bstride = (long long)cin * h * w;
    a.ndst = 1;
    a.dst[0].p = out;
    a.dst[0].bstride = (long long)cout * h * w;
    a.dst[0].q0 = 0;
    a.dst[0].q1 = (cout + 3) / 4;
    a.N = n; a.H = h; a.W = w;
    a.wpk = (const float*)packed;
    a.bpk = (const float*)((const char*)packed + align_up(conv_packed_weight_floats(a) * sizeof(float), 256));
    return launch_conv_mfma(a, "conv_mfma:api_nchw", (hipStream_t)stream);
}

int crfp_conv3x3_f32(const float* x, const float* weight, const float* bias, float* out, int n, int cin, int cout, int h,
                     int w, int act, float post_scale, void* workspace, size_t workspace_bytes, void* stream) {
    if (!x || !weight || !out || n < 1 || cin < 1 || cout < 1 || h < 1 || w < 1) { set_error("conv3x3: bad argument"); return CRFP_E_BADARG; }
    if (act < CRFP_ACT_NONE || act > CRFP_ACT_SIGMOID) { set_error("conv3x3: unknown activation %d", act); return CRFP_E_BADARG; }
    if (!workspace || workspace_bytes < crfp_conv3x3_workspace_bytes(n, cin, cout, h, w)) { set_error("conv3x3: workspace too small"); return CRFP_E_WORKSPACE; }
    hipStream_t s = (hipStream_t)stream;
    ConvArgs a = api_conv_plan(cin, cout, act, post_scale);
    float* wpk = (float*)workspace;
    float* bpk = (float*)((char*)workspace + align_up(conv_packed_weight_floats(a) * sizeof(float), 256));
    int rc = launch_conv_pack(a, weight, bias, nullptr, nullptr, cout, wpk, bpk, s);
    if (rc) return rc;
    a.src[0].p = x;
    a.src[0].bstride = (long long)cin * h * w;
    a.ndst = 1;
    a.dst[0].p = out;
    a.dst[0].bstride = (long long)cout * h * w;
    a.dst[0].q0 = 0;
    a.dst[0].q1 = (cout + 3) / 4;
    a.N = n; a.H = h; a.W = w;
    a.wpk = wpk;
    a.bpk = bpk;
    return launch_conv_mfma(a, "conv_mfma:api_nchw", s);
}

// ---- the conv operator with everything SURVEY 8(b) lists for it: up to two inputs (a fused torch.cat), optional residual, activation,
// load through pixel_unshuffle(4), store plain into a channel slice of a wider tensor or through pixel_shuffle(r).
static ConvArgs api_conv_ex_plan(int cin, int cin2, int cout, int act, float post_scale, int unshuffle_r, int shuffle_r) {
    ConvArgs a;
    memset(&a, 0, sizeof(a));
    int kq = 0;
    a.nsrc = 0;
    ConvSrc& s0 = a.src[a.nsrc++];
    s0.kind = unshuffle_r == 4 ? SRC_UNSHUF4 : SRC_NCHW;
    s0.nch = cin;
    s0.nq = unshuffle_r == 4 ? 16 * ((cin / 16 + 3) / 4) : (cin + 3) / 4;   // SRC_UNSHUF4: 16 sub-positions per hi-res quad
    s0.cbase = 0;
    kq += s0.nq;
    if (cin2 > 0) {
        ConvSrc& s1 = a.src[a.nsrc++];
        s1.kind = SRC_NCHW; s1.nch = cin2; s1.nq = (cin2 + 3) / 4; s1.cbase = cin;
        kq += s1.nq;
    }
    if (kq & 1) {
        ConvSrc& z = a.src[a.nsrc++];
        z.kind = SRC_ZERO; z.nch = 1; z.nq = 1; z.cbase = cin + cin2;
        kq += 1;
    }
    a.kq = kq;
    a.cin_total = cin + cin2;
    a.cout = cout;
    a.store = shuffle_r > 1 ? ST_PS : ST_NCHW;
    a.ps_r = shuffle_r > 1 ? shuffle_r : 0;
    a.act = act;
    a.post_scale = post_scale;
    a.ctiles = (conv_packed_rows(cout, a.store, a.ps_r) + 31) / 32;
    return a;
}

static bool conv_ex_args_ok(int cin, int cin2, int cout, int act, int unshuffle_r, int shuffle_r, bool has_resid) {
    if (cin < 1 || cin2 < 0 || cout < 1) { set_error("conv3x3_ex: bad channel counts (%d, %d -> %d)", cin, cin2, cout); return false; }
    if (act < CRFP_ACT_NONE || act > CRFP_ACT_SIGMOID) { set_error("conv3x3_ex: unknown activation %d", act); return false; }
    if (unshuffle_r != 0 && unshuffle_r != 1 && unshuffle_r != 4) { set_error("conv3x3_ex: pixel_unshuffle load supports r = 4 (got %d)", unshuffle_r); return false; }
    if (unshuffle_r == 4 && (cin2 > 0 || cin % 16)) { set_error("conv3x3_ex: pixel_unshuffle(4) load takes one input with cin %% 16 == 0"); return false; }
    if (shuffle_r != 0 && shuffle_r != 1 && shuffle_r != 2 && shuffle_r != 4) { set_error("conv3x3_ex: pixel_shuffle store supports r in {2, 4} (got %d)", shuffle_r); return false; }
    if (shuffle_r > 1 && (cout % (shuffle_r * shuffle_r) || has_resid || act == CRFP_ACT_TANH || act == CRFP_ACT_SIGMOID)) {
        set_error("conv3x3_ex: pixel_shuffle store needs cout %% r^2 == 0, no residual, activation none / relu / lrelu");
        return false;
    }
    return true;
}

size_t crfp_conv3x3_ex_workspace_bytes(int n, int cin, int cin2, int cout, int h, int w, int unshuffle_r, int shuffle_r, int has_residual) {
    if (n < 1 || h < 1 || w < 1 || !conv_ex_args_ok(cin, cin2, cout, CRFP_ACT_NONE, unshuffle_r, shuffle_r, false)) return 0;
    const ConvArgs a = api_conv_ex_plan(cin, cin2, cout, 0, 1.0f, unshuffle_r, shuffle_r);
    size_t b = align_up(conv_packed_weight_floats(a) * sizeof(float), 256) + align_up((size_t)a.ctiles * 32 * sizeof(float), 256);
    if (unshuffle_r == 4) b += q4_bytes(n, cin / 16, 4 * h, 4 * w);
    if (has_residual) b += q4_bytes(n, cout, h, w);
    if (shuffle_r > 1) b += q4_bytes(n, cout / (shuffle_r * shuffle_r), h * shuffle_r, w * shuffle_r);
    return b;
}

/* x: [n, cin, h, w]  (unshuffle_r = 4: [n, cin / 16, 4h, 4w], read as pixel_unshuffle(x, 4));  x2: optional [n, cin2, h, w];
 * weight [cout, cin + cin2, 3, 3];  residual: optional [n, cout, h, w], added after activation and post_scale;
 * out: [n, out_ctotal, h, w], channels [out_c0, out_c0 + cout) written (shuffle_r > 1: [n, cout / r^2, h r, w r], whole). */
int crfp_conv3x3_ex_f32(const float* x, int cin, const float* x2, int cin2, const float* weight, const float* bias, const float* residual,
                        float* out, int n, int cout, int h, int w, int act, float post_scale, int unshuffle_r, int shuffle_r,
                        int out_c0, int out_ctotal, void* workspace, size_t workspace_bytes, void* stream) {
    if (!x || !weight || !out || n < 1 || h < 1 || w < 1 || (cin2 > 0 && !x2)) { set_error("conv3x3_ex: bad argument"); return CRFP_E_BADARG; }
    if (!conv_ex_args_ok(cin, cin2, cout, act, unshuffle_r, shuffle_r, residual != nullptr)) return CRFP_E_UNSUPPORTED;
    if (shuffle_r <= 1 && (out_c0 < 0 || out_ctotal < out_c0 + cout)) { set_error("conv3x3_ex: channel slice [%d, %d) outside %d channels", out_c0, out_c0 + cout, out_ctotal); return CRFP_E_BADARG; }
    if (!workspace || workspace_bytes < crfp_conv3x3_ex_workspace_bytes(n, cin, cin2, cout, h, w, unshuffle_r, shuffle_r, residual != nullptr)) {
        set_error("conv3x3_ex: workspace too small");
        return CRFP_E_WORKSPACE;
    }
    hipStream_t s = (hipStream_t)stream;
    ConvArgs a = api_conv_ex_plan(cin, cin2, cout, act, post_scale, unshuffle_r, shuffle_r);
    char* p = (char*)workspace;
    float* wpk = (float*)p; p += align_up(conv_packed_weight_floats(a) * sizeof(float), 256);
    float* bpk = (float*)p; p += align_up((size_t)a.ctiles * 32 * sizeof(float), 256);
    int rc = launch_conv_pack(a, weight, bias, nullptr, nullptr, cout, wpk, bpk, s);
    if (rc) return rc;
    if (unshuffle_r == 4) {   // the hi-res tensor as Q4, read through SRC_UNSHUF4
        float* xq = (float*)p; p += q4_bytes(n, cin / 16, 4 * h, 4 * w);
        rc = launch_nchw_to_q4(x, xq, n, cin / 16, 4 * h, 4 * w, 0, s);
        if (rc) return rc;
        a.src[0].p = xq;
        a.src[0].bstride = (long long)((cin / 16 + 3) / 4) * (4 * h) * (4 * w) * 4;
    } else {
        a.src[0].p = x;
        a.src[0].bstride = (long long)cin * h * w;
    }
    if (cin2 > 0) { a.src[1].p = x2; a.src[1].bstride = (long long)cin2 * h * w; }
    if (residual) {
        float* rq = (float*)p; p += q4_bytes(n, cout, h, w);
        rc = launch_nchw_to_q4(residual, rq, n, cout, h, w, 0, s);
        if (rc) return rc;
        a.resid = rq;
        a.resid_bstride = (long long)((cout + 3) / 4) * h * w * 4;
    }
    a.N = n; a.H = h; a.W = w;
    a.wpk = wpk; a.bpk = bpk;
    a.ndst = 1;
    if (shuffle_r > 1) {
        const int co = cout / (shuffle_r * shuffle_r), H2 = h * shuffle_r, W2 = w * shuffle_r;
        float* oq = (float*)p;
        a.dst[0].p = oq; a.dst[0].bstride = (long long)((co + 3) / 4) * H2 * W2 * 4; a.dst[0].q0 = 0; a.dst[0].q1 = (co + 3) / 4;
        a.dstH = H2; a.dstW = W2;
        rc = launch_conv_mfma(a, "conv_mfma:api_ex_ps", s);
        if (!rc) rc = launch_q4_to_nchw(oq, out, n, co, H2, W2, 0, s);
        return rc;
    }
    a.dst[0].p = out + (long long)out_c0 * h * w;
    a.dst[0].bstride = (long long)out_ctotal * h * w;
    a.dst[0].q0 = 0;
    a.dst[0].q1 = (cout + 3) / 4;
    return launch_conv_mfma(a, "conv_mfma:api_ex", s);
}

int crfp_upsample_bilinear_f32(const float* x, float* out, int n, int c, int h, int w, int oh, int ow, float scale_h,
                               float scale_w, float mul, void* stream) {
    if (!x || !out || n < 1 || c < 1 || h < 1 || w < 1 || oh < 1 || ow < 1) { set_error("upsample: bad argument"); return CRFP_E_BADARG; }
    return launch_upsample_nchw(x, out, n, c, h, w, oh, ow, scale_h, scale_w, mul, (hipStream_t)stream);
}

int crfp_avgpool2_f32(const float* x, float* out, int n, int c, int h, int w, void* stream) {
    if (!x || !out || n < 1 || c < 1 || h < 2 || w < 2) { set_error("avgpool2: bad argument"); return CRFP_E_BADARG; }
    return launch_avgpool2_nchw(x, out, n, c, h, w, (hipStream_t)stream);
}

// fused a-15: conv_tttf on [state | x_hr], fovea blend, LeakyReLU -> new state; conv_last + bilinear x8 base -> frame
size_t crfp_fovea_head_workspace_bytes(int n, int h, int w) {
    const int H = 8 * h, W = 8 * w;
    return 4 * q4_bytes(n, 4, H, W) + align_up((size_t)n * 3 * H * W * sizeof(float), 256) + 2 * align_up((9 * 2 * 16 + 4) * sizeof(float), 256);
}

int crfp_fovea_head_f32(const float* state, const float* x_hr, const unsigned char* mask, const float* lr, const float* w_tttf,
                        const float* b_tttf, const float* w_last, const float* b_last, float* new_state, float* out, int n,
                        int h, int w, int y_only, void* workspace, size_t workspace_bytes, void* stream) {
    if (!state || !x_hr || !mask || !lr || !w_tttf || !b_tttf || !w_last || !b_last || !new_state || !out || n < 1 || h < 1 || w < 1) {
        set_error("fovea_head: bad argument");
        return CRFP_E_BADARG;
    }
    if (!workspace || workspace_bytes < crfp_fovea_head_workspace_bytes(n, h, w)) { set_error("fovea_head: workspace too small"); return CRFP_E_WORKSPACE; }
    hipStream_t s = (hipStream_t)stream;
    const int H = 8 * h, W = 8 * w;
    char* p = (char*)workspace;
    float* sq = (float*)p; p += q4_bytes(n, 4, H, W);
    float* xq = (float*)p; p += q4_bytes(n, 4, H, W);
    float* nq = (float*)p; p += q4_bytes(n, 4, H, W);
    float* bq = (float*)p; p += q4_bytes(n, 4, H, W);
    float* up = (float*)p; p += align_up((size_t)n * 3 * H * W * sizeof(float), 256);
    float* pk_t = (float*)p; p += align_up((9 * 2 * 16 + 4) * sizeof(float), 256);
    float* pk_l = (float*)p;
    const long long qs = (long long)H * W * 4;
    auto plan = [&](int nsrc, int cin, int cout, int epi) {
        NarrowArgs a;
        memset(&a, 0, sizeof(a));
        for (int i = 0; i < nsrc; ++i) { a.src[i].kind = SRC_Q4; a.src[i].nch = 4; a.src[i].nq = 1; a.src[i].cbase = 4 * i; a.src[i].bstride = qs; }
        a.nsrc = nsrc; a.kq = nsrc; a.cin_total = cin; a.cout = cout; a.act = CRFP_ACT_NONE; a.epi = epi; a.y_only = y_only;
        a.post_scale = 1.0f; a.N = n; a.H = H; a.W = W;
        return a;
    };
    NarrowArgs t = plan(2, 8, 4, NE_BLEND), l = plan(1, 4, y_only ? 1 : 3, NE_LAST);
    int rc = launch_nchw_to_q4(state, sq, n, 4, H, W, 0, s);
    if (!rc) rc = launch_nchw_to_q4(x_hr, xq, n, 4, H, W, 0, s);
    if (!rc) rc = launch_upsample_nchw(lr, up, n, 3, h, w, H, W, 0.125f, 0.125f, 1.0f, s);   // nn.Upsample(x8, bilinear, align_corners=False)
    if (!rc) rc = launch_nchw_to_q4(up, bq, n, 3, H, W, 0, s);
    if (!rc) rc = launch_narrow_pack(t, w_tttf, b_tttf, nullptr, nullptr, 0, pk_t, pk_t + 9 * 2 * 16, s);
    if (!rc) rc = launch_narrow_pack(l, w_last, b_last, nullptr, nullptr, 0, pk_l, pk_l + 9 * 1 * 16, s);
    if (rc) return rc;
    t.src[0].p = sq; t.src[1].p = xq; t.wpk = pk_t; t.bpk = pk_t + 9 * 2 * 16; t.dst = nq; t.dst_bstride = qs;
    t.mask = mask; t.mask_bstride = (long long)H * W;
    rc = launch_narrow(t, "conv_narrow:tttf_blend", s);
    l.src[0].p = nq; l.wpk = pk_l; l.bpk = pk_l + 9 * 1 * 16; l.dst = out; l.dst_bstride = (long long)(y_only ? 1 : 3) * H * W;
    l.base = bq; l.base_bstride = qs;
    if (!rc) rc = launch_narrow(l, "conv_narrow:last_plus_base", s);
    if (!rc) rc = launch_q4_to_nchw(nq, new_state, n, 4, H, W, 0, s);
    return rc;
}

int crfp_psnr_ssim_partial_f32(const float* a, const float* b, const unsigned char* mask, double* acc, int n, int c, int h, int w,
                               float mul, float add, void* stream) {
    if (!a || !b || !acc || n < 1 || c < 1 || h < 1 || w < 1) { set_error("psnr_ssim_partial: bad argument"); return CRFP_E_BADARG; }
    return launch_psnr_ssim_partial(a, b, mask, acc, n, c, h, w, mul, add, (hipStream_t)stream);
}

int crfp_psnr_partial_f32(const float* a, const float* b, double* acc, int n, int c, int h, int w, void* stream) {
    if (!a || !b || !acc || n < 1 || c < 1 || h < 1 || w < 1) { set_error("psnr_partial: bad argument"); return CRFP_E_BADARG; }
    return launch_psnr_partial(a, b, acc, n, c, h, w, (hipStream_t)stream);
}

}  // extern "C"
