// Engine-level schedule of the reference's benchmark-only wiring model/CRFP_runtime.py::MRCF_simple_v18.forward(lrs, fvs, warp_size)
// (:8469-8664; the model test_runtime.py builds, :41, :142): ONE C-ABI call per clip over the same HIP kernels as the CRFP_DSV engine
// (engine.hip), every intermediate in the library's Q4 / P4 layouts.  Round 2 composed this wiring in Python from the per-operator entry
// points (NCHW API tensors, fp32 MFMA, a layout conversion on both sides of every operator): 3.0 ms per 1080p frame.
//
// How the wiring differs from CRFP_DSV (restated from the cited lines; oracle/runtime_oracle.py is the CPU restatement, the golden of
// the reference class itself is tests/golden/runtime_small.npz):
//   * FNet, the warps and all four DCNs only see the top-left warp_size window (:8487, :8533-8620); the 8x state carried to the next
//     frame is that window (:8650).  Here every windowed tensor is a COMPACT tensor of window size (crop / paste kernels move the
//     24-channel 2x features, the 4-channel 8x features and the state between frame and window), so the conv / warp / DCN kernels run
//     unchanged on window-sized maps.
//   * prev2 = downsample(state) and prev2_w = downsample(warp(state)) (:8536-8537) instead of warp(downsample(state)).
//   * levels 0-2 all start from the SAME upsample(x_lr) features and only produce the carried 8 channels + the offset feature (:8549-8599).
//   * ResidualBlocksWithInputConv_v2 (:511-556): conv1 on the windowed first input pasted over conv2 of the full-frame second input,
//     LeakyReLU, one bottleneck residual block (C -> C/2 -> C, :406-462).  On levels 0-2 both inputs have window size, so conv2's result
//     is overwritten entirely and is not computed; on level 3 conv2 runs on the full 8x frame and conv1's window result is pasted in.
//     Frame 0 uses the separate forward_resblocks_k_ modules on the full frame (:8617-8637).
//   * the fovea arrives as an (Hf, Wf) crop fed twice to encoder_hr (:8507); conv_tttf runs on the top-left (Hf, Wf) crop of the 8x
//     features (zero padding at the crop's own border) and its result is pasted back before the LeakyReLU (:8645-8648).
// fp32 storage, default (split-fp16) precision only.
#include "crfp_common.h"

#ifndef CRFP_ACT_BF16

#include <cstdlib>
#include <cstring>
#include <string>
#include <mutex>
#include <vector>

namespace crfp {
namespace rt {

struct ConvDef { const char* stem; int cout, cin; };
// state_dict order of crfp_amd.model.CRFP_runtime.MRCF_simple_v18 (mid_channels = 32) = the reference's (:8364-8467)
static const ConvDef kRt[79] = {
    {"spynet.encoder1.0", 32, 6}, {"spynet.encoder1.2", 32, 32}, {"spynet.encoder2.0", 64, 32},
    {"spynet.encoder2.2", 64, 64}, {"spynet.encoder3.0", 128, 64}, {"spynet.encoder3.2", 128, 128},
    {"spynet.decoder1.0", 256, 128}, {"spynet.decoder1.2", 256, 256}, {"spynet.decoder2.0", 128, 256},
    {"spynet.decoder2.2", 128, 128}, {"spynet.decoder3.0", 64, 128}, {"spynet.decoder3.2", 64, 64},
    {"spynet.flow.0", 32, 64}, {"spynet.flow.2", 2, 32},
    {"dcn_0.dcn_block.0", 32, 66}, {"dcn_0.dcn_block.2", 32, 32}, {"dcn_0.dcn_offset", 144, 32},
    {"dcn_0.dcn_mask", 72, 32}, {"dcn_0.dcn", 32, 32},
    {"dcn_1.conv_fuse", 32, 64}, {"dcn_1.dcn_block.0", 32, 66}, {"dcn_1.dcn_block.2", 32, 32},
    {"dcn_1.dcn_offset", 144, 32}, {"dcn_1.dcn_mask", 72, 32}, {"dcn_1.dcn", 32, 32},
    {"dcn_2.conv_fuse", 32, 64}, {"dcn_2.dcn_block.0", 32, 66}, {"dcn_2.dcn_block.2", 32, 32},
    {"dcn_2.dcn_offset", 144, 32}, {"dcn_2.dcn_mask", 72, 32}, {"dcn_2.dcn", 32, 32},
    {"dcn_3.upsample.upsample_conv", 64, 32}, {"dcn_3.conv_fuse", 4, 8}, {"dcn_3.dcn_block.0", 4, 10},
    {"dcn_3.dcn_block.2", 4, 4}, {"dcn_3.dcn_offset", 2, 4}, {"dcn_3.dcn_mask", 1, 4}, {"dcn_3.dcn", 4, 4},
    {"encoder_lr.slice1.0", 32, 3}, {"encoder_lr.slice1.2", 32, 32}, {"encoder_hr.slice1.0", 4, 6},
    {"encoder_hr.slice1.2", 4, 4}, {"conv_tttf", 4, 8},
    {"forward_resblocks_0_.conv1", 32, 24}, {"forward_resblocks_0_.conv2", 32, 8}, {"forward_resblocks_0_.main.1.0.conv1", 16, 32},
    {"forward_resblocks_0_.main.1.0.conv2", 32, 16},
    {"forward_resblocks_1_.conv1", 32, 24}, {"forward_resblocks_1_.conv2", 32, 8}, {"forward_resblocks_1_.main.1.0.conv1", 16, 32},
    {"forward_resblocks_1_.main.1.0.conv2", 32, 16},
    {"forward_resblocks_2_.conv1", 32, 24}, {"forward_resblocks_2_.conv2", 32, 8}, {"forward_resblocks_2_.main.1.0.conv1", 16, 32},
    {"forward_resblocks_2_.main.1.0.conv2", 32, 16},
    {"forward_resblocks_3_.conv1", 4, 4}, {"forward_resblocks_3_.conv2", 4, 1}, {"forward_resblocks_3_.main.1.0.conv1", 2, 4},
    {"forward_resblocks_3_.main.1.0.conv2", 4, 2},
    {"forward_resblocks_0.conv1", 32, 64}, {"forward_resblocks_0.conv2", 32, 32}, {"forward_resblocks_0.main.1.0.conv1", 16, 32},
    {"forward_resblocks_0.main.1.0.conv2", 32, 16},
    {"forward_resblocks_1.conv1", 32, 64}, {"forward_resblocks_1.conv2", 32, 32}, {"forward_resblocks_1.main.1.0.conv1", 16, 32},
    {"forward_resblocks_1.main.1.0.conv2", 32, 16},
    {"forward_resblocks_2.conv1", 32, 64}, {"forward_resblocks_2.conv2", 32, 32}, {"forward_resblocks_2.main.1.0.conv1", 16, 32},
    {"forward_resblocks_2.main.1.0.conv2", 32, 16},
    {"forward_resblocks_3.conv1", 4, 8}, {"forward_resblocks_3.conv2", 4, 4}, {"forward_resblocks_3.main.1.0.conv1", 2, 4},
    {"forward_resblocks_3.main.1.0.conv2", 4, 2},
    {"downsample.downsample_conv", 32, 64}, {"upsample.upsample_conv", 96, 32}, {"upsample_post.upsample_conv", 64, 24},
    {"conv_last", 3, 4}};
constexpr int RT_CI_LAST = 78;
static int rt_cout(int ci, int y_only) { return (ci == RT_CI_LAST && y_only) ? 1 : kRt[ci].cout; }
static int ci_dcn(int lvl, int which) { static const int base[3] = {13, 19, 25}; return base[lvl] + which; }   // 0 fuse .. 5 dcn (engine.hip)

enum ItemType { T_MFMA = 0, T_NARROW = 1, T_DCN8 = 2, T_RAW = 3 };
struct Item {
    int type = T_MFMA;
    ConvArgs c;
    NarrowArgs nw;
    int w1 = -1, w2 = -1;
    size_t off_w = 0, off_b = 0, n_w = 0, n_b = 0, off_s = 0, n_s = 0;
    const char* name = "";
};
enum {
    RI_F0 = 0, RI_ENC_LR0 = 14, RI_ENC_LR1, RI_UPS, RI_DOWN,
    RI_LVL0,                                   // per level 8: FUSE, DB0, DB1, OMF, DCNW, C1 (64 -> 32), B1 (32 -> 16), B2 (16 -> 32 + x)
    RI_FIRST0 = RI_LVL0 + 24,                  // per level 3: conv1 (24 -> 32), B1, B2 of forward_resblocks_k_
    RI_UPP = RI_FIRST0 + 9, RI_POFF,
    RI_EH0, RI_EH1, RI_D3B0, RI_D3B1, RI_D3FUSE, RI_D3OM, RI_D3W,
    RI_R3_C2, RI_R3_C1, RI_R3_B1, RI_R3_B2, RI_R3F_C1, RI_R3F_B1, RI_R3F_B2, RI_TTTF, RI_LAST, RI_COUNT
};
enum { L_FUSE = 0, L_DB0, L_DB1, L_OMF, L_DCNW, L_C1, L_B1, L_B2 };
static int it_lvl(int l, int which) { return RI_LVL0 + 8 * l + which; }
static int it_first(int l, int which) { return RI_FIRST0 + 3 * l + which; }

struct SrcSpec { int kind, nch; };

struct Model {
    Item items[RI_COUNT];
    size_t total_floats = 0;
    int y_only;

    void add_mfma(int id, const char* name, int ci, int ci2, std::vector<SrcSpec> srcs, int store, int ps_r, int act, float post_scale = 1.0f) {
        Item& it = items[id];
        it.type = T_MFMA; it.name = name; it.w1 = ci; it.w2 = ci2;
        ConvArgs& a = it.c;
        memset(&a, 0, sizeof(a));
        int kq = 0, cbase = 0;
        for (auto& s : srcs) {
            ConvSrc& d = a.src[a.nsrc++];
            d.kind = s.kind; d.nch = s.nch; d.nq = src_quads(s.kind, s.nch); d.cbase = cbase;
            cbase += s.nch; kq += d.nq;
        }
        if (kq & 3) {   // K is consumed in 16-channel chunks
            ConvSrc& d = a.src[a.nsrc++];
            d.kind = SRC_ZERO; d.nq = 4 - (kq & 3); d.nch = d.nq; d.cbase = cbase; kq += d.nq;
        }
        a.kq = kq; a.cin_total = kRt[ci].cin;
        a.cout = rt_cout(ci, y_only) + (ci2 >= 0 ? rt_cout(ci2, y_only) : 0);
        a.store = store; a.ps_r = ps_r; a.act = act; a.post_scale = post_scale;
        a.ctiles = (conv_packed_rows(a.cout, store, ps_r) + 31) / 32;
        it.n_w = conv_packed_weight_floats(a);
        it.n_b = (size_t)a.ctiles * 32;
        it.n_s = conv_split_weight_bytes(a) / sizeof(float);
    }
    void add_narrow(int id, const char* name, int ci, int ci2, std::vector<SrcSpec> srcs, int act, int epi) {
        Item& it = items[id];
        it.type = T_NARROW; it.name = name; it.w1 = ci; it.w2 = ci2;
        NarrowArgs& a = it.nw;
        memset(&a, 0, sizeof(a));
        int kq = 0, cbase = 0;
        for (auto& s : srcs) {
            ConvSrc& d = a.src[a.nsrc++];
            d.kind = s.kind; d.nch = s.nch; d.nq = src_quads(s.kind, s.nch); d.cbase = cbase;
            cbase += s.nch; kq += d.nq;
        }
        a.kq = kq; a.cin_total = kRt[ci].cin;
        a.cout = rt_cout(ci, y_only) + (ci2 >= 0 ? rt_cout(ci2, y_only) : 0);
        a.act = act; a.epi = epi; a.y_only = y_only; a.post_scale = 1.0f;
        it.n_w = narrow_packed_weight_floats(a);
        it.n_b = 4;
    }

    explicit Model(int y_only_) : y_only(y_only_) {
        const int Q = SRC_Q4, FS = SRC_S3;
        static const char* fn[14] = {"conv_mfma:fnet.enc1a", "conv_mfma:fnet.enc1b", "conv_mfma:fnet.enc2a", "conv_mfma:fnet.enc2b",
                                     "conv_mfma:fnet.enc3a", "conv_mfma:fnet.enc3b", "conv_mfma:fnet.dec1a", "conv_mfma:fnet.dec1b",
                                     "conv_mfma:fnet.dec2a", "conv_mfma:fnet.dec2b", "conv_mfma:fnet.dec3a", "conv_mfma:fnet.dec3b",
                                     "conv_mfma:fnet.flow0", "conv_mfma:fnet.flow2"};
        add_mfma(RI_F0, fn[0], 0, -1, {{Q, 3}, {Q, 3}}, ST_Q4, 0, CRFP_ACT_RELU);
        for (int i = 1; i < 13; ++i) add_mfma(RI_F0 + i, fn[i], i, -1, {{Q, kRt[i].cin}}, ST_Q4, 0, CRFP_ACT_RELU);
        add_mfma(RI_F0 + 13, fn[13], 13, -1, {{Q, 32}}, ST_Q4, 0, CRFP_ACT_TANH, 256.0f);
        add_mfma(RI_ENC_LR0, "conv_mfma:enc_lr0", 38, -1, {{Q, 3}}, ST_Q4, 0, CRFP_ACT_LRELU01);
        add_mfma(RI_ENC_LR1, "conv_mfma:enc_lr1", 39, -1, {{Q, 32}}, ST_Q4, 0, CRFP_ACT_LRELU01);
        add_mfma(RI_UPS, "conv_mfma:upsample_ps2", 76, -1, {{Q, 32}}, ST_PS, 2, CRFP_ACT_NONE);
        add_mfma(RI_DOWN, "conv_mfma:downsample_unshuf4", 75, -1, {{SRC_UNSHUF4, 64}}, ST_Q4, 0, CRFP_ACT_NONE);
        for (int l = 0; l < 3; ++l) {
            if (l > 0) add_mfma(it_lvl(l, L_FUSE), "conv_mfma:dcn.conv_fuse", ci_dcn(l, 0), -1, {{Q, 32}, {FS, 32}}, ST_Q4, 0, CRFP_ACT_LRELU01);
            add_mfma(it_lvl(l, L_DB0), "conv_mfma:dcn.block0", ci_dcn(l, 1), -1, {{Q, 24}, {Q, 8}, {Q, 32}, {SRC_FLOW2, 2}}, ST_Q4, 0, CRFP_ACT_LRELU01);
            add_mfma(it_lvl(l, L_DB1), "conv_mfma:dcn.block2", ci_dcn(l, 2), -1, {{Q, 32}}, ST_Q4, 0, CRFP_ACT_LRELU01);
            add_mfma(it_lvl(l, L_OMF), "conv_mfma:dcn.offset_mask_fused", ci_dcn(l, 3), ci_dcn(l, 4), {{FS, 32}}, ST_DCNFUSE, 0, CRFP_ACT_NONE);
            Item& dw = items[it_lvl(l, L_DCNW)];
            dw.type = T_DCN8; dw.name = "dcn_g8_weights"; dw.w1 = ci_dcn(l, 5); dw.n_w = 2 * 36 * 2 * 32 * 4; dw.n_b = 32;
            // forward_resblocks_l (v2): conv1 on [cur(24) | carry(8) | aligned(32)] (:8571); conv2 is overwritten entirely (see the header)
            add_mfma(it_lvl(l, L_C1), "conv_mfma:rt.res.conv1", 59 + 4 * l, -1, {{Q, 24}, {Q, 8}, {Q, 32}}, ST_Q4, 0, CRFP_ACT_LRELU01);
            add_mfma(it_lvl(l, L_B1), "conv_mfma:rt.res.bneck1", 61 + 4 * l, -1, {{Q, 32}}, ST_Q4, 0, CRFP_ACT_RELU);
            add_mfma(it_lvl(l, L_B2), "conv_mfma:rt.res.bneck2_add", 62 + 4 * l, -1, {{Q, 16}}, ST_Q4, 0, CRFP_ACT_NONE);
            add_mfma(it_first(l, 0), "conv_mfma:rt.res_first.conv1", 43 + 4 * l, -1, {{Q, 24}}, ST_Q4, 0, CRFP_ACT_LRELU01);
            add_mfma(it_first(l, 1), "conv_mfma:rt.res_first.bneck1", 45 + 4 * l, -1, {{Q, 32}}, ST_Q4, 0, CRFP_ACT_RELU);
            add_mfma(it_first(l, 2), "conv_mfma:rt.res_first.bneck2_add", 46 + 4 * l, -1, {{Q, 16}}, ST_Q4, 0, CRFP_ACT_NONE);
        }
        add_mfma(RI_UPP, "conv_mfma:upsample_post_ps4", 77, -1, {{Q, 24}}, ST_PS, 4, CRFP_ACT_LRELU01);
        add_mfma(RI_POFF, "conv_mfma:dcn3.preoffset_ps4", 31, -1, {{FS, 32}}, ST_PS, 4, CRFP_ACT_NONE, 2.0f);
        add_narrow(RI_EH0, "conv_narrow:enc_hr0", 40, -1, {{Q, 3}, {Q, 3}}, CRFP_ACT_LRELU01, NE_PLAIN);
        add_narrow(RI_EH1, "conv_narrow:enc_hr1", 41, -1, {{Q, 4}}, CRFP_ACT_LRELU01, NE_PLAIN);
        add_narrow(RI_D3B0, "conv_narrow:dcn3.block0", 33, -1, {{Q, 4}, {Q, 4}, {SRC_FLOW2, 2}}, CRFP_ACT_LRELU01, NE_PLAIN);
        add_narrow(RI_D3B1, "conv_narrow:dcn3.block2", 34, -1, {{Q, 4}}, CRFP_ACT_LRELU01, NE_PLAIN);
        add_narrow(RI_D3FUSE, "conv_narrow:dcn3.conv_fuse", 32, -1, {{Q, 4}, {Q, 4}}, CRFP_ACT_LRELU01, NE_PLAIN);
        add_narrow(RI_D3OM, "conv_narrow:dcn3.offset_mask", 35, 36, {{Q, 4}}, CRFP_ACT_NONE, NE_OFFMASK3);
        Item& d3 = items[RI_D3W];
        d3.type = T_RAW; d3.name = "dcn3_weights"; d3.w1 = 37; d3.n_w = 4 * 4 * 9; d3.n_b = 4;
        add_narrow(RI_R3_C2, "conv_narrow:rt.res3.conv2_full", 72, -1, {{Q, 4}}, CRFP_ACT_LRELU01, NE_PLAIN);
        add_narrow(RI_R3_C1, "conv_narrow:rt.res3.conv1_window", 71, -1, {{Q, 4}, {Q, 4}}, CRFP_ACT_LRELU01, NE_PLAIN);
        add_narrow(RI_R3_B1, "conv_narrow:rt.res3.bneck1", 73, -1, {{Q, 4}}, CRFP_ACT_RELU, NE_PLAIN);
        add_narrow(RI_R3_B2, "conv_narrow:rt.res3.bneck2_add", 74, -1, {{Q, 2}}, CRFP_ACT_NONE, NE_PLAIN);
        add_narrow(RI_R3F_C1, "conv_narrow:rt.res3_first.conv1", 55, -1, {{Q, 4}}, CRFP_ACT_LRELU01, NE_PLAIN);
        add_narrow(RI_R3F_B1, "conv_narrow:rt.res3_first.bneck1", 57, -1, {{Q, 4}}, CRFP_ACT_RELU, NE_PLAIN);
        add_narrow(RI_R3F_B2, "conv_narrow:rt.res3_first.bneck2_add", 58, -1, {{Q, 2}}, CRFP_ACT_NONE, NE_PLAIN);
        add_narrow(RI_TTTF, "conv_narrow:rt.tttf_crop", 42, -1, {{Q, 4}, {Q, 4}}, CRFP_ACT_LRELU01, NE_PLAIN);
        add_narrow(RI_LAST, "conv_narrow:last_plus_base", RT_CI_LAST, -1, {{Q, 4}}, CRFP_ACT_NONE, NE_LAST);
        size_t cur = 0;
        for (int i = 0; i < RI_COUNT; ++i) {
            Item& it = items[i];
            if (it.w1 < 0) continue;
            it.off_w = cur; cur += (it.n_w + 63) / 64 * 64;
            it.off_b = cur; cur += (it.n_b + 63) / 64 * 64;
            it.off_s = cur; cur += (it.n_s + 63) / 64 * 64;
        }
        total_floats = cur;
    }
};
static const Model& model_for(int y_only) {
    static const Model m0(0), m1(1);
    return y_only ? m1 : m0;
}

// ------------------------------------------------------------------ workspace
struct Layout {
    size_t cur = 0;
    int t, h, w, fh, fw, wh8, ww8, wh2, ww2, whl, wwl;   // window at 8x / 2x / LR resolution
    int h1, w1, h2, w2, h3, w3;                           // FNet levels on the LR window
    size_t status, state, carry, prev2;
    size_t lr_q4, lrw_q4, e_lr0, x_lr, fvq, eh0, x_hr, flow_lr;
    size_t fa0, fa1, fp1, fb0, fb1, fp2, fc0, fc1, fp3, fd0, fd1, fu1, fe0, fe1, fu2, ff0, ff1, fu3, fg0, fg1;
    // [2]: written by the state-independent pre-work of frame i into set i & 1 while frame i - 1 still reads the other set; [3]: one per
    // level (the three levels of a frame run on three streams)
    size_t prop0[2], prop_a, prop_b, y0f, y1f, tmp8f, flow2[2], flow8[2], state_w, prev2w, carryw, win[2], fa[3], fb[3], offfeat[3], aligned[3], y0[3], y1[3];
    size_t up_full[2], upw[2], poff, g0, g1, g2, al3, featf[2], z1, feat2, tcrop;

    // Q4 tensor of nq quads (pad = 1: P4 planes with a zeroed guard in front, see engine.hip); kind 1 = [H][W][2] floats
    size_t take(int N, int nq, int H, int W, int kind = 0, int pad = 0) {
        const size_t elems = kind == 0 ? (size_t)N * nq * (H + pad) * (W + pad) * 4 : (size_t)N * H * W * 2;
        const size_t guard = pad ? align_up((size_t)(W + 2) * 16, 256) : 0;
        const size_t off = cur + guard;
        cur += guard + align_up(elems * sizeof(float), 256);
        return off;
    }
    static size_t p4_bytes(int nq, int H, int W) { return (size_t)nq * (H + 1) * (W + 1) * 16; }
    static size_t p4_guard(int W) { return align_up((size_t)(W + 2) * 16, 256); }

    Layout(int t_, int h_, int w_, int fh_, int fw_, int wph, int wpw) : t(t_), h(h_), w(w_), fh(fh_), fw(fw_) {
        wh8 = wph; ww8 = wpw; wh2 = wph / 4; ww2 = wpw / 4; whl = wph / 8; wwl = wpw / 8;
        const int nb = t > 1 ? t - 1 : 1;
        const int H2 = 2 * h, W2 = 2 * w, H8 = 8 * h, W8 = 8 * w;
        h1 = whl / 2; w1 = wwl / 2; h2 = h1 / 2; w2 = w1 / 2; h3 = h2 / 2; w3 = w2 / 2;
        status = take(1, 0, 1, 32, 1);
        state = take(1, 1, wh8, ww8, 0, 1);
        carry = take(1, 6, wh2, ww2, 0, 1);
        prev2 = take(1, 8, wh2, ww2, 0, 1);
        lr_q4 = take(t, 1, h, w);
        lrw_q4 = take(t, 1, whl, wwl);
        e_lr0 = take(t, 8, h, w);
        x_lr = take(t, 8, h, w);
        fvq = take(t, 1, fh, fw);
        eh0 = take(t, 1, fh, fw);
        x_hr = take(t, 1, fh, fw);
        flow_lr = take(nb, 1, whl, wwl);
        fa0 = take(nb, 8, whl, wwl); fa1 = take(nb, 8, whl, wwl); fp1 = take(nb, 8, h1, w1);
        fb0 = take(nb, 16, h1, w1); fb1 = take(nb, 16, h1, w1); fp2 = take(nb, 16, h2, w2);
        fc0 = take(nb, 32, h2, w2); fc1 = take(nb, 32, h2, w2); fp3 = take(nb, 32, h3, w3);
        fd0 = take(nb, 64, h3, w3); fd1 = take(nb, 64, h3, w3); fu1 = take(nb, 64, 2 * h3, 2 * w3);
        fe0 = take(nb, 32, 2 * h3, 2 * w3); fe1 = take(nb, 32, 2 * h3, 2 * w3); fu2 = take(nb, 32, 4 * h3, 4 * w3);
        ff0 = take(nb, 16, 4 * h3, 4 * w3); ff1 = take(nb, 16, 4 * h3, 4 * w3); fu3 = take(nb, 16, 8 * h3, 8 * w3);
        fg0 = take(nb, 8, 8 * h3, 8 * w3); fg1 = take(nb, 1, 8 * h3, 8 * w3);
        prop_a = take(1, 6, H2, W2); prop_b = take(1, 6, H2, W2);
        y0f = take(1, 8, H2, W2); y1f = take(1, 4, H2, W2); tmp8f = take(1, 2, H2, W2);
        for (int p = 0; p < 2; ++p) {
            prop0[p] = take(1, 6, H2, W2); win[p] = take(1, 6, wh2, ww2);
            flow2[p] = take(1, 0, wh2, ww2, 1); flow8[p] = take(1, 0, wh8, ww8, 1);
            up_full[p] = take(1, 1, H8, W8); upw[p] = take(1, 1, wh8, ww8); featf[p] = take(1, 1, H8, W8);
        }
        state_w = take(1, 1, wh8, ww8);
        prev2w = take(1, 8, wh2, ww2); carryw = take(1, 6, wh2, ww2);
        for (int l = 0; l < 3; ++l) {
            fa[l] = take(1, 8, wh2, ww2); fb[l] = take(1, 8, wh2, ww2); offfeat[l] = take(1, 8, wh2, ww2);
            aligned[l] = take(1, 8, wh2, ww2); y0[l] = take(1, 8, wh2, ww2); y1[l] = take(1, 4, wh2, ww2);
        }
        poff = take(1, 1, wh8, ww8);
        g0 = take(1, 1, wh8, ww8); g1 = take(1, 1, wh8, ww8); g2 = take(1, 1, wh8, ww8); al3 = take(1, 1, wh8, ww8);
        z1 = take(1, 1, H8, W8); feat2 = take(1, 1, H8, W8);
        tcrop = take(1, 1, fh, fw);
    }
    size_t bytes() const { return cur; }
};

// ------------------------------------------------------------------ small layout kernels (window <-> frame)
// dst[q][y][x] = src[q][y][x] for the top-left dH x dW pixels; either side may be a P4 tensor (pitch + 1, plane + 1 row)
__global__ void rt_copy_q4_kernel(const float4* __restrict__ src, int sH, int sW, int spad, float4* __restrict__ dst, int dH, int dW, int dpad,
                                  int cH, int cW) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6), q = blockIdx.z;
    if (x >= cW || y >= cH) return;
    dst[((long long)q * (dH + dpad) + y) * (dW + dpad) + x] = src[((long long)q * (sH + spad) + y) * (sW + spad) + x];
}
static int rt_copy_q4(const float* src, int sH, int sW, int spad, float* dst, int dH, int dW, int dpad, int nq, int cH, int cW, hipStream_t s) {
    ProfScope prof("rt_crop_paste_q4", s, (double)nq * cH * cW * 32.0, 0);
    rt_copy_q4_kernel<<<dim3((cW + 63) / 64, (cH + 3) / 4, nq), 256, 0, s>>>(reinterpret_cast<const float4*>(src), sH, sW, spad,
                                                                           reinterpret_cast<float4*>(dst), dH, dW, dpad, cH, cW);
    CRFP_CHECK_LAUNCH();
    return 0;
}
// top-left cH x cW pixels of NCHW [n][3][H][W] frames -> Q4 quads (r, g, b, 0)
__global__ void rt_crop_nchw3_kernel(const float* __restrict__ x, int H, int W, float4* __restrict__ out, int cH, int cW) {
    const int px = blockIdx.x * 64 + (threadIdx.x & 63), py = blockIdx.y * 4 + (threadIdx.x >> 6), n = blockIdx.z;
    if (px >= cW || py >= cH) return;
    const float* p = x + (long long)n * 3 * H * W + (long long)py * W + px;
    out[((long long)n * cH + py) * cW + px] = make_float4(p[0], p[(long long)H * W], p[2LL * H * W], 0.0f);
}
// feat = LeakyReLU(feat) outside the top-left fh x fw crop, = tcrop (already activated) inside it (:8645-8648)
// ... and the top-left sh x sw window of the result is the state the next frame carries (:8650): written to the P4 tensor `state` in the same pass
__global__ void rt_lrelu_paste_kernel(float4* __restrict__ feat, int H, int W, const float4* __restrict__ tcrop, int fh, int fw,
                                      float4* __restrict__ state, int sh, int sw, unsigned* __restrict__ ovf) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || y >= H) return;
    float4 v;
    if (x < fw && y < fh) v = tcrop[(long long)y * fw + x];
    else {
        v = feat[(long long)y * W + x];
        v.x = v.x > 0.0f ? v.x : 0.1f * v.x; v.y = v.y > 0.0f ? v.y : 0.1f * v.y;
        v.z = v.z > 0.0f ? v.z : 0.1f * v.z; v.w = v.w > 0.0f ? v.w : 0.1f * v.w;
    }
    feat[(long long)y * W + x] = v;
    if (x < sw && y < sh) state[(long long)y * (sw + 1) + x] = v;
    const float m = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));   // the state feeds split-fp16 convs
    if (ovf && !(m < 65504.0f)) atomicOr(ovf, 1u);
}

// ------------------------------------------------------------------ side streams
// Three non-blocking streams + an event pool per (host thread, device), created lazily by the first crfp_rt_forward_clip on that device
// and destroyed by crfp_shutdown(): the same contract as the CRFP_DSV engine's side stream (crfp_hip.h).  Every call joins all of them
// back into the caller's stream before it returns.
struct Lanes {
    hipStream_t s[3] = {nullptr, nullptr, nullptr};
    std::vector<hipEvent_t> ev;
    size_t used = 0;
    bool ok = true;
    hipEvent_t next_event() {
        if (used == ev.size()) {
            hipEvent_t e;
            if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { ok = false; return nullptr; }
            ev.push_back(e);
        }
        return ev[used++];
    }
    bool create() {
        for (int i = 0; i < 3 && ok; ++i)
            if (!s[i] && hipStreamCreateWithFlags(&s[i], hipStreamNonBlocking) != hipSuccess) ok = false;
        return ok;
    }
    void destroy() {
        for (hipEvent_t e : ev) (void)hipEventDestroy(e);
        ev.clear(); used = 0;
        for (int i = 0; i < 3; ++i) { if (s[i]) (void)hipStreamDestroy(s[i]); s[i] = nullptr; }
        ok = true;
    }
};
constexpr int kRtMaxDevices = 64;
// per-thread tables leased from a process-wide registry with a free list, as the CRFP_DSV side streams (engine.hip): crfp_shutdown()
// releases every thread's lanes, an exiting thread hands its table to the next new one without calling HIP
struct LaneTable { Lanes dev[kRtMaxDevices]; };
struct LaneRegistry { std::mutex mu; std::vector<LaneTable*> all, idle; };
static LaneRegistry& lane_registry() { static LaneRegistry* r = new LaneRegistry(); return *r; }
struct LaneLease {
    LaneTable* t = nullptr;
    ~LaneLease() {
        if (!t) return;
        LaneRegistry& r = lane_registry();
        std::lock_guard<std::mutex> lk(r.mu);
        r.idle.push_back(t);
    }
};
static thread_local LaneLease g_lanes_tl;
static Lanes* lane_table() {
    if (!g_lanes_tl.t) {
        LaneRegistry& r = lane_registry();
        std::lock_guard<std::mutex> lk(r.mu);
        if (!r.idle.empty()) { g_lanes_tl.t = r.idle.back(); r.idle.pop_back(); }
        else { g_lanes_tl.t = new LaneTable(); r.all.push_back(g_lanes_tl.t); }
    }
    return g_lanes_tl.t->dev;
}
static Lanes* lanes_for_current_device() {
    static const bool on = !(getenv("CRFP_SIDE_STREAM") && atoi(getenv("CRFP_SIDE_STREAM")) == 0);   // read once
    int dev = 0;
    if (!on || hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kRtMaxDevices) return nullptr;
    Lanes& l = lane_table()[dev];
    if (!l.create()) return nullptr;
    l.used = 0;
    return &l;
}

// ------------------------------------------------------------------ the schedule
struct Runner {
    const Model& M;
    const float* packed;
    char* ws;
    const Layout& L;
    hipStream_t s;
    int rc = 0;
#define RUN(expr) do { if (!rc) rc = (expr); } while (0)
    float* F(size_t off) const { return reinterpret_cast<float*>(ws + off); }
    unsigned* ovf() const { return reinterpret_cast<unsigned*>(ws + L.status); }
    struct SrcBind { const float* p; long long bs; int pad = 0; };
    struct DstBind { float* p; long long bs; int q0, q1; int pad = 0; };

    void mfma(int id, int N, int H, int W, std::vector<SrcBind> srcs, std::vector<DstBind> dsts, int dstH = 0, int dstW = 0,
              const float* resid = nullptr, const float* flow = nullptr, float* s3 = nullptr, int dst_f32 = 0) {
        if (rc) return;
        const Item& it = M.items[id];
        ConvArgs a = it.c;
        for (size_t i = 0; i < srcs.size(); ++i) { a.src[i].p = srcs[i].p; a.src[i].bstride = srcs[i].bs; a.src[i].pad = srcs[i].pad; }
        a.ndst = (int)dsts.size();
        for (size_t i = 0; i < dsts.size(); ++i) {
            a.dst[i].p = dsts[i].p; a.dst[i].bstride = dsts[i].bs; a.dst[i].q0 = dsts[i].q0; a.dst[i].q1 = dsts[i].q1; a.dst[i].pad = dsts[i].pad;
        }
        a.N = N; a.H = H; a.W = W; a.dstH = dstH; a.dstW = dstW;
        a.resid = resid; a.flow = flow; a.s3_dst = s3; a.dst_f32 = dst_f32;
        a.wpk = packed + it.off_w; a.bpk = packed + it.off_b; a.wsplit = packed + it.off_s;
        a.ovf = ovf();
        rc = launch_conv_mfma(a, it.name, s);
    }
    void mfma_q(int id, int N, const float* in, int nqi, float* out, int nqo, int H, int W) {
        mfma(id, N, H, W, {{in, (long long)nqi * H * W * 4}}, {{out, (long long)nqo * H * W * 4, 0, nqo}});
    }
    void narrow(int id, int H, int W, std::vector<const float*> srcs, float* dst, const float* resid = nullptr, const float* flow = nullptr,
                int src0_pad = 0, int dst_pad = 0, const float* base_lr = nullptr, long long src_bs = 0, int N = 1, long long dst_bs = 0) {
        if (rc) return;
        const Item& it = M.items[id];
        NarrowArgs a = it.nw;
        for (size_t i = 0; i < srcs.size(); ++i) { a.src[i].p = srcs[i]; a.src[i].bstride = src_bs; a.src[i].pad = 0; }
        a.src[0].pad = src0_pad; a.dst_pad = dst_pad;
        a.N = N; a.H = H; a.W = W;
        a.dst = dst; a.dst_bstride = dst_bs; a.resid = resid; a.flow = flow; a.base = nullptr; a.base_lr = base_lr; a.mask = nullptr;
        a.wpk = packed + it.off_w; a.bpk = packed + it.off_b;
        a.ovf = ovf();
        rc = launch_narrow(a, it.name, s);
    }

    // FNet on the LR window, nb pairs (engine.hip Runner::fnet, reference model/CRFP.py:797-814)
    void fnet(int nb, const float* cur, const float* prev) {
        const int h = L.whl, w = L.wwl;
        const long long fs = (long long)h * w * 4;
        auto bs = [](int nq, int H, int W) { return (long long)nq * H * W * 4; };
        mfma(RI_F0, nb, h, w, {{cur, fs}, {prev, fs}}, {{F(L.fa0), bs(8, h, w), 0, 8}});
        mfma_q(RI_F0 + 1, nb, F(L.fa0), 8, F(L.fa1), 8, h, w);
        RUN(launch_avgpool2_q4(F(L.fa1), bs(8, h, w), F(L.fp1), bs(8, L.h1, L.w1), nb, 8, h, w, s));
        mfma_q(RI_F0 + 2, nb, F(L.fp1), 8, F(L.fb0), 16, L.h1, L.w1);
        mfma_q(RI_F0 + 3, nb, F(L.fb0), 16, F(L.fb1), 16, L.h1, L.w1);
        RUN(launch_avgpool2_q4(F(L.fb1), bs(16, L.h1, L.w1), F(L.fp2), bs(16, L.h2, L.w2), nb, 16, L.h1, L.w1, s));
        mfma_q(RI_F0 + 4, nb, F(L.fp2), 16, F(L.fc0), 32, L.h2, L.w2);
        mfma_q(RI_F0 + 5, nb, F(L.fc0), 32, F(L.fc1), 32, L.h2, L.w2);
        RUN(launch_avgpool2_q4(F(L.fc1), bs(32, L.h2, L.w2), F(L.fp3), bs(32, L.h3, L.w3), nb, 32, L.h2, L.w2, s));
        mfma_q(RI_F0 + 6, nb, F(L.fp3), 32, F(L.fd0), 64, L.h3, L.w3);
        mfma_q(RI_F0 + 7, nb, F(L.fd0), 64, F(L.fd1), 64, L.h3, L.w3);
        const int H1 = 2 * L.h3, W1 = 2 * L.w3, H2_ = 4 * L.h3, W2_ = 4 * L.w3, H3 = 8 * L.h3, W3 = 8 * L.w3;
        RUN(launch_upsample_q4(F(L.fd1), bs(64, L.h3, L.w3), F(L.fu1), bs(64, H1, W1), nb, 64, L.h3, L.w3, H1, W1, 0.5f, 0.5f, 1.0f, s, 0));
        mfma_q(RI_F0 + 8, nb, F(L.fu1), 64, F(L.fe0), 32, H1, W1);
        mfma_q(RI_F0 + 9, nb, F(L.fe0), 32, F(L.fe1), 32, H1, W1);
        RUN(launch_upsample_q4(F(L.fe1), bs(32, H1, W1), F(L.fu2), bs(32, H2_, W2_), nb, 32, H1, W1, H2_, W2_, 0.5f, 0.5f, 1.0f, s, 0));
        mfma_q(RI_F0 + 10, nb, F(L.fu2), 32, F(L.ff0), 16, H2_, W2_);
        mfma_q(RI_F0 + 11, nb, F(L.ff0), 16, F(L.ff1), 16, H2_, W2_);
        RUN(launch_upsample_q4(F(L.ff1), bs(16, H2_, W2_), F(L.fu3), bs(16, H3, W3), nb, 16, H2_, W2_, H3, W3, 0.5f, 0.5f, 1.0f, s, 0));
        mfma_q(RI_F0 + 12, nb, F(L.fu3), 16, F(L.fg0), 8, H3, W3);
        mfma(RI_F0 + 13, nb, H3, W3, {{F(L.fg0), bs(8, H3, W3)}}, {{F(L.fg1), bs(1, H3, W3), 0, 1}}, 0, 0, nullptr, nullptr, nullptr, 1);
        RUN(launch_upsample_q4(F(L.fg1), bs(1, H3, W3), F(L.flow_lr), fs, nb, 1, H3, W3, h, w, (float)H3 / (float)h, (float)W3 / (float)w, 1.0f, s, 0));
    }

    void zero(size_t off, size_t bytes, size_t guard = 0) {
        if (!rc && hipMemsetAsync(ws + off - guard, 0, bytes + guard, s) != hipSuccess) { set_error("rt: hipMemsetAsync failed"); rc = 1; }
    }

    // ---- streams.  Lane 0 is the caller's stream; lanes 1..3 are the library's side streams (null pool: every lane is the caller's
    // stream and rec() / wait() do nothing -- the enqueue order below is a valid single-stream order as it stands)
    Lanes* lanes = nullptr;
    hipStream_t main_s = nullptr;
    void on(int lane) { s = (lanes && lane) ? lanes->s[lane - 1] : main_s; }
    hipEvent_t rec() {
        if (!lanes || rc) return nullptr;
        hipEvent_t e = lanes->next_event();
        if (!e || hipEventRecord(e, s) != hipSuccess) { set_error("rt: hipEventRecord failed"); rc = 1; return nullptr; }
        return e;
    }
    void wait(hipEvent_t e) {
        if (!lanes || rc || !e) return;
        if (hipStreamWaitEvent(s, e, 0) != hipSuccess) { set_error("rt: hipStreamWaitEvent failed"); rc = 1; }
    }
    // every lane drains into the caller's stream (also after an error: the caller may reuse its buffers once `stream` has drained)
    void join_all() {
        if (!lanes) return;
        for (int l = 0; l < 3; ++l) {
            hipEvent_t e = lanes->next_event();
            if (e && hipEventRecord(e, lanes->s[l]) == hipSuccess) (void)hipStreamWaitEvent(main_s, e, 0);
        }
        s = main_s;
    }

    // state-independent work of frame i >= 1 into parity set i & 1 (current stream): the flow fields at 2x / 8x, upsample(x_lr) and its
    // window (:8524, :8548), lrelu(upsample_post(.)) and its window (:8602), forward_resblocks_3.conv2 on the full frame (:8607-8609)
    void frame_pre(int i) {
        const int par = i & 1, h = L.h, w = L.w, H2 = 2 * h, W2 = 2 * w, H8 = 8 * h, W8 = 8 * w;
        const float* x_lr_i = F(L.x_lr) + (long long)i * 8 * h * w * 4;
        const float* flow = F(L.flow_lr) + (long long)(i - 1) * L.whl * L.wwl * 4;
        RUN(launch_upflow(flow, 0, F(L.flow2[par]), 0, 1, L.whl, L.wwl, 2, s));                       // :8531-8532
        RUN(launch_upflow(flow, 0, F(L.flow8[par]), 0, 1, L.whl, L.wwl, 8, s));
        mfma(RI_UPS, 1, h, w, {{x_lr_i, 0}}, {{F(L.prop0[par]), 0, 0, 6}}, H2, W2);
        RUN(rt_copy_q4(F(L.prop0[par]), H2, W2, 0, F(L.win[par]), L.wh2, L.ww2, 0, 6, L.wh2, L.ww2, s));
        mfma(RI_UPP, 1, H2, W2, {{F(L.prop0[par]), 0}}, {{F(L.up_full[par]), 0, 0, 1}}, H8, W8);
        RUN(rt_copy_q4(F(L.up_full[par]), H8, W8, 0, F(L.upw[par]), L.wh8, L.ww8, 0, 1, L.wh8, L.ww8, s));
        narrow(RI_R3_C2, H8, W8, {F(L.up_full[par])}, F(L.featf[par]));
    }

    // level l of frame i >= 1 up to its offset feature (current stream): dcn_block.0 -> .2 (-> conv_fuse with the previous level's feature)
    void level_head(int l, int par, const float* offprev) {
        const int wh2 = L.wh2, ww2 = L.ww2;
        const float* cw = F(L.carryw) + 2 * l * (long long)wh2 * ww2 * 4;
        mfma(it_lvl(l, L_DB0), 1, wh2, ww2, {{F(L.win[par]), 0}, {cw, 0}, {F(L.prev2w), 0}, {F(L.flow2[par]), 0}, {nullptr, 0}}, {{F(L.fa[l]), 0, 0, 8}});
        if (l == 0) mfma(it_lvl(l, L_DB1), 1, wh2, ww2, {{F(L.fa[l]), 0}}, {}, 0, 0, nullptr, nullptr, F(L.offfeat[l]));
        else mfma(it_lvl(l, L_DB1), 1, wh2, ww2, {{F(L.fa[l]), 0}}, {{F(L.fb[l]), 0, 0, 8}});
        (void)offprev;
    }
    void level_fuse(int l, const float* offprev) {
        mfma(it_lvl(l, L_FUSE), 1, L.wh2, L.ww2, {{F(L.fb[l]), 0}, {offprev, 0}}, {}, 0, 0, nullptr, nullptr, F(L.offfeat[l]));
    }
    // ... and from the offset feature to the carried 8 channels: fused offset head + DCN, forward_resblocks_l([cur | carry | aligned], .) of
    // which only channels 24..31 leave the level (:8571-8599)
    void level_tail(int l, int par) {
        const int wh2 = L.wh2, ww2 = L.ww2;
        const long long P2w = (long long)wh2 * ww2 * 4, P2wp = (long long)(wh2 + 1) * (ww2 + 1) * 4;
        const float* cw = F(L.carryw) + 2 * l * P2w;
        const Item& om = M.items[it_lvl(l, L_OMF)];
        const Item& dw = M.items[it_lvl(l, L_DCNW)];
        DcnFuseArgs fa;
        memset(&fa, 0, sizeof(fa));
        fa.feat = F(L.offfeat[l]); fa.flow = F(L.flow2[par]);
        fa.wconv = (const char*)(packed + om.off_s) + conv_split16_offset_bytes(om.c); fa.bconv = packed + om.off_b;
        fa.x = F(L.prev2); fa.wdcn = packed + dw.off_w + 36 * 2 * 32 * 4; fa.bdcn = packed + dw.off_b;
        fa.out = F(L.aligned[l]); fa.N = 1; fa.H = wh2; fa.W = ww2; fa.ovf = ovf();
        RUN(launch_dcn_fused(fa, s));
        mfma(it_lvl(l, L_C1), 1, wh2, ww2, {{F(L.win[par]), 0}, {cw, 0}, {F(L.aligned[l]), 0}}, {{F(L.y0[l]), 0, 0, 8}});
        mfma(it_lvl(l, L_B1), 1, wh2, ww2, {{F(L.y0[l]), 0}}, {{F(L.y1[l]), 0, 0, 4}});
        mfma(it_lvl(l, L_B2), 1, wh2, ww2, {{F(L.y1[l]), 0}}, {{F(L.carry) + 2 * l * P2wp, 0, 6, 8, 1}}, 0, 0, F(L.y0[l]));
    }

    hipEvent_t pre_done[2] = {nullptr, nullptr}, frame_done[2] = {nullptr, nullptr};

    void frame(int i, int t, const float* lr_nchw, float* out) {
        const int par = i & 1, h = L.h, w = L.w, H2 = 2 * h, W2 = 2 * w, H8 = 8 * h, W8 = 8 * w;
        const int wh2 = L.wh2, ww2 = L.ww2, wh8 = L.wh8, ww8 = L.ww8;
        const long long P2wp = (long long)(wh2 + 1) * (ww2 + 1) * 4;
        float* feat = F(L.feat2);
        hipEvent_t e_l[3] = {nullptr, nullptr, nullptr};
        on(0);
        if (i > 0) {
            wait(pre_done[par]);
            RUN(launch_flow_warp_q4(F(L.state), 0, F(L.flow8[par]), 0, F(L.state_w), 0, 1, 1, wh8, ww8, 0, 1, s));   // :8534-8535
            mfma(RI_DOWN, 1, wh2, ww2, {{F(L.state_w), 0, 0}}, {{F(L.prev2w), 0, 0, 8}});               // prev2_w = downsample(state_w) (:8536)
            mfma(RI_DOWN, 1, wh2, ww2, {{F(L.state), 0, 1}}, {{F(L.prev2), 0, 0, 8, 1}});               // prev2 = downsample(state) (:8537)
            RUN(launch_flow_warp_q4(F(L.carry), 0, F(L.flow2[par]), 0, F(L.carryw), 0, 1, 6, wh2, ww2, 0, 1, s));   // :8538-8547
            // (running downsample(state) and the carry warp on two other lanes beside these: 0.50 vs 0.49 ms per frame -- every cross-stream
            // dependency costs a few microseconds of its own, so only long independent chains are worth a lane)
            hipEvent_t e_base = rec();
            // The three levels read the same window features and only hand the offset feature down (:8549-8599): their dcn_block convs
            // run side by side, conv_fuse of level l waits for level l - 1's feature, and everything behind a level's feature is its own.
            level_head(0, par, nullptr);
            hipEvent_t e_off0 = rec();
            on(2); wait(e_base);
            level_head(1, par, nullptr);
            wait(e_off0);
            level_fuse(1, F(L.offfeat[0]));
            hipEvent_t e_off1 = rec();
            level_tail(1, par);
            e_l[1] = rec();
            on(3); wait(e_base);
            level_head(2, par, nullptr);
            wait(e_off1);
            level_fuse(2, F(L.offfeat[1]));
            mfma(RI_POFF, 1, wh2, ww2, {{F(L.offfeat[2]), 0}}, {{F(L.poff), 0, 0, 1}}, wh8, ww8);
            hipEvent_t e_poff = rec();
            level_tail(2, par);
            e_l[2] = rec();
            on(1); wait(e_off0);
            level_tail(0, par);
            e_l[0] = rec();
            if (i + 1 < t) {   // the next frame's pre-work, into the set frame i - 1 has finished with
                wait(frame_done[par ^ 1]);
                frame_pre(i + 1);
                pre_done[par ^ 1] = rec();
            }
            on(0);
            narrow(RI_D3B0, wh8, ww8, {F(L.upw[par]), F(L.state_w), F(L.flow8[par])}, F(L.g0));
            narrow(RI_D3B1, wh8, ww8, {F(L.g0)}, F(L.g1));
            wait(e_poff);
            narrow(RI_D3FUSE, wh8, ww8, {F(L.g1), F(L.poff)}, F(L.g2));
            const Item& om3 = M.items[RI_D3OM];
            const Item& d3 = M.items[RI_D3W];
            RUN(launch_dcn3_fused(F(L.state), 0, F(L.g2), 0, F(L.flow8[par]), packed + om3.off_w, packed + om3.off_b, packed + d3.off_w, packed + d3.off_b,
                                  F(L.al3), 0, 1, wh8, ww8, s));
            // forward_resblocks_3([upw | aligned], up): conv1's window result pasted over conv2 of the full frame (:8607-8609)
            // (the window result goes straight into the full-frame tensor: destination pitch W8)
            narrow(RI_R3_C1, wh8, ww8, {F(L.upw[par]), F(L.al3)}, F(L.featf[par]), nullptr, nullptr, 0, W8 - ww8);
            narrow(RI_R3_B1, H8, W8, {F(L.featf[par])}, F(L.z1));
            narrow(RI_R3_B2, H8, W8, {F(L.z1)}, feat, F(L.featf[par]));
        } else {
            const float* x_lr_0 = F(L.x_lr);
            mfma(RI_UPS, 1, h, w, {{x_lr_0, 0}}, {{F(L.prop0[0]), 0, 0, 6}}, H2, W2);                   // feat_prop_lv0 = upsample(x_lr) (:8524)
            float* prop = F(L.prop0[0]);
            float* nxt[3] = {F(L.prop_a), F(L.prop_b), F(L.prop_a)};
            for (int l = 0; l < 3; ++l) {   // forward_resblocks_l_(prop) on the full frame (:8617-8633)
                mfma(it_first(l, 0), 1, H2, W2, {{prop, 0}, {nullptr, 0}}, {{F(L.y0f), 0, 0, 8}});
                mfma(it_first(l, 1), 1, H2, W2, {{F(L.y0f), 0}}, {{F(L.y1f), 0, 0, 4}});
                mfma(it_first(l, 2), 1, H2, W2, {{F(L.y1f), 0}}, {{nxt[l], 0, 0, 6}, {F(L.tmp8f), 0, 6, 8}}, 0, 0, F(L.y0f));
                RUN(rt_copy_q4(F(L.tmp8f), H2, W2, 0, F(L.carry) + 2 * l * P2wp, wh2, ww2, 1, 2, wh2, ww2, s));
                prop = nxt[l];
            }
            mfma(RI_UPP, 1, H2, W2, {{prop, 0}}, {{F(L.up_full[0]), 0, 0, 1}}, H8, W8);                 // :8636
            narrow(RI_R3F_C1, H8, W8, {F(L.up_full[0])}, F(L.featf[0]));                                // forward_resblocks_3_(up) (:8637)
            narrow(RI_R3F_B1, H8, W8, {F(L.featf[0])}, F(L.z1));
            narrow(RI_R3F_B2, H8, W8, {F(L.z1)}, feat, F(L.featf[0]));
        }
        // conv_tttf on the fovea crop, pasted back, LeakyReLU everywhere (:8645-8648)
        // (the crop is read in place: source pitch W8, zero padding at the crop's own border comes from the kernel's validity mask)
        narrow(RI_TTTF, L.fh, L.fw, {feat, F(L.x_hr) + (long long)i * L.fh * L.fw * 4}, F(L.tcrop), nullptr, nullptr, W8 - L.fw);
        if (!rc) {
            ProfScope prof("rt_lrelu_paste", s, (double)H8 * W8 * 32.0, 0);
            rt_lrelu_paste_kernel<<<dim3((W8 + 63) / 64, (H8 + 3) / 4, 1), 256, 0, s>>>(reinterpret_cast<float4*>(feat), H8, W8,
                                                                                       reinterpret_cast<const float4*>(F(L.tcrop)), L.fh, L.fw,
                                                                                       reinterpret_cast<float4*>(F(L.state)), wh8, ww8, ovf());
            if (hipGetLastError() != hipSuccess) { set_error("rt: lrelu_paste launch failed"); rc = 1; }
        }
        narrow(RI_LAST, H8, W8, {feat}, out, nullptr, nullptr, 0, 0, lr_nchw);                           // conv_last + x8 bilinear LR (:8652-8654)
        for (int l = 0; l < 3; ++l) wait(e_l[l]);                                                        // the carried features of this frame are complete
        frame_done[par] = rec();
    }
#undef RUN
};

}  // namespace rt
void rt_shutdown_streams() {
    rt::LaneRegistry& r = rt::lane_registry();
    std::lock_guard<std::mutex> lk(r.mu);
    for (rt::LaneTable* t : r.all)
        for (int d = 0; d < rt::kRtMaxDevices; ++d) t->dev[d].destroy();
}
}  // namespace crfp

using namespace crfp;
using namespace crfp::rt;

extern "C" {

const char* crfp_rt_param_name(int index) {
    static thread_local std::string s;
    if (index < 0 || index >= CRFP_RT_NUM_PARAMS) return nullptr;
    s = std::string(kRt[index / 2].stem) + (index % 2 ? ".bias" : ".weight");
    return s.c_str();
}

int crfp_rt_param_numel(int index, int y_only) {
    if (index < 0 || index >= CRFP_RT_NUM_PARAMS) return CRFP_E_BADARG;
    const int ci = index / 2, co = rt_cout(ci, y_only);
    return index % 2 ? co : co * kRt[ci].cin * 9;
}

size_t crfp_rt_packed_weight_bytes(int y_only) { return model_for(y_only).total_floats * sizeof(float); }

int crfp_rt_pack_weights(const float* const* params, int y_only, void* packed, size_t packed_bytes, void* stream) {
    const Model& M = model_for(y_only);
    if (!params || !packed) { set_error("rt_pack_weights: null argument"); return CRFP_E_BADARG; }
    if (packed_bytes < M.total_floats * sizeof(float)) { set_error("rt_pack_weights: packed buffer too small"); return CRFP_E_WORKSPACE; }
    for (int i = 0; i < CRFP_RT_NUM_PARAMS; ++i)
        if (!params[i]) { set_error("rt_pack_weights: parameter %d (%s) is null", i, crfp_rt_param_name(i)); return CRFP_E_BADARG; }
    hipStream_t s = (hipStream_t)stream;
    float* pk = (float*)packed;
    for (int i = 0; i < RI_COUNT; ++i) {
        const Item& it = M.items[i];
        if (it.w1 < 0) continue;
        const float* w = params[2 * it.w1];
        const float* b = params[2 * it.w1 + 1];
        const float* w2 = it.w2 >= 0 ? params[2 * it.w2] : nullptr;
        const float* b2 = it.w2 >= 0 ? params[2 * it.w2 + 1] : nullptr;
        const int split = rt_cout(it.w1, y_only);
        int rc = 0;
        switch (it.type) {
            case T_MFMA:
                rc = launch_conv_pack(it.c, w, b, w2, b2, split, pk + it.off_w, pk + it.off_b, s);
                if (!rc) rc = launch_conv_pack_split(it.c, w, w2, split, pk + it.off_s, s);
                break;
            case T_NARROW: rc = launch_narrow_pack(it.nw, w, b, w2, b2, split, pk + it.off_w, pk + it.off_b, s); break;
            case T_DCN8:
                rc = launch_dcn_g8_pack(w, pk + it.off_w, s, false);
                if (!rc) rc = launch_dcn_g8_pack(w, pk + it.off_w + 36 * 2 * 32 * 4, s, true);
                if (!rc && hipMemcpyAsync(pk + it.off_b, b, 32 * sizeof(float), hipMemcpyDeviceToDevice, s) != hipSuccess) rc = 1;
                break;
            default:
                if (hipMemcpyAsync(pk + it.off_w, w, it.n_w * sizeof(float), hipMemcpyDeviceToDevice, s) != hipSuccess) rc = 1;
                if (!rc && hipMemcpyAsync(pk + it.off_b, b, it.n_b * sizeof(float), hipMemcpyDeviceToDevice, s) != hipSuccess) rc = 1;
        }
        if (rc) return rc;
    }
    return 0;
}

static int rt_check_dims(int t, int h, int w, int fh, int fw, int wph, int wpw) {
    if (t < 1 || h < 8 || w < 8 || fh < 1 || fw < 1 || fh > 8 * h || fw > 8 * w) { set_error("rt: bad clip / fovea size (t=%d h=%d w=%d fv=%dx%d)", t, h, w, fh, fw); return CRFP_E_BADARG; }
    if (wph < 64 || wpw < 64 || (wph & 7) || (wpw & 7) || wph > 8 * h || wpw > 8 * w) {
        set_error("rt: warp_size (%d, %d) must be a multiple of 8, at least 64 and inside the %d x %d frame", wph, wpw, 8 * h, 8 * w);
        return CRFP_E_BADARG;
    }
    return 0;
}

size_t crfp_rt_workspace_bytes(int t, int h, int w, int fh, int fw, int wph, int wpw) {
    if (rt_check_dims(t, h, w, fh, fw, wph, wpw)) return 0;
    return Layout(t, h, w, fh, fw, wph, wpw).bytes();
}

int crfp_rt_forward_clip(const void* packed, int flags, const float* lrs, const float* fvs, float* out, int t, int h, int w, int fh, int fw,
                         int wph, int wpw, void* workspace, size_t workspace_bytes, void* stream) {
    const int y_only = flags & CRFP_DSV_Y_ONLY;
    if (flags & CRFP_DSV_STRICT_F32) { set_error("rt_forward_clip: the regional wiring runs in the default precision only"); return CRFP_E_UNSUPPORTED; }
    if (!conv_s3_supported() || !dcn_fused_enabled()) { set_error("rt_forward_clip: needs the default conv / fused-DCN kernels (no CRFP_PRECISION / CRFP_DCN_FUSED overrides)"); return CRFP_E_UNSUPPORTED; }
    int rc = rt_check_dims(t, h, w, fh, fw, wph, wpw);
    if (rc) return rc;
    if (!packed || !workspace || !lrs || !fvs || !out) { set_error("rt_forward_clip: null argument"); return CRFP_E_BADARG; }
    Layout L(t, h, w, fh, fw, wph, wpw);
    if (workspace_bytes < L.bytes()) { set_error("rt_forward_clip: workspace %zu < required %zu bytes", workspace_bytes, L.bytes()); return CRFP_E_WORKSPACE; }
    Runner R{model_for(y_only), (const float*)packed, (char*)workspace, L, (hipStream_t)stream};
    hipStream_t s = (hipStream_t)stream;
    R.main_s = s;
    // per-kernel timing brackets launches per stream and CRFP_DSV_SINGLE_STREAM asks for it: everything on the caller's stream
    R.lanes = (prof_enabled() || (flags & CRFP_DSV_SINGLE_STREAM)) ? nullptr : lanes_for_current_device();
    // pads of the P4 tensors must read as zero; the carried features start at zero
    R.zero(L.status, 256);
    R.zero(L.state, Layout::p4_bytes(1, L.wh8, L.ww8), Layout::p4_guard(L.ww8));
    R.zero(L.carry, Layout::p4_bytes(6, L.wh2, L.ww2), Layout::p4_guard(L.ww2));
    R.zero(L.prev2, Layout::p4_bytes(8, L.wh2, L.ww2), Layout::p4_guard(L.ww2));
    if (R.rc) return R.rc;
    rc = launch_nchw_to_q4(lrs, R.F(L.lr_q4), t, 3, h, w, 0, s);
    if (rc) return rc;
    rt_crop_nchw3_kernel<<<dim3((L.wwl + 63) / 64, (L.whl + 3) / 4, t), 256, 0, s>>>(lrs, h, w, reinterpret_cast<float4*>(R.F(L.lrw_q4)), L.whl, L.wwl);
    CRFP_CHECK_LAUNCH();
    rc = launch_nchw_to_q4(fvs, R.F(L.fvq), t, 3, fh, fw, 0, s);
    if (rc) return rc;
    // from here on other streams may hold work of this call: every exit goes through join_all()
    hipEvent_t e_in = R.rec();
    const long long lf = (long long)h * w * 4, ff = (long long)fh * fw * 4, lw = (long long)L.whl * L.wwl * 4;
    R.mfma(RI_ENC_LR0, t, h, w, {{R.F(L.lr_q4), lf}}, {{R.F(L.e_lr0), 8 * lf, 0, 8}});
    R.mfma_q(RI_ENC_LR1, t, R.F(L.e_lr0), 8, R.F(L.x_lr), 8, h, w);
    hipEvent_t e_enc = R.rec();
    if (t > 1) {   // lane 1: flows on the LR window (:8487), then frame 1's state-independent work -- beside frame 0 on the caller's stream
        R.on(1); R.wait(e_in);
        R.fnet(t - 1, R.F(L.lrw_q4) + lw, R.F(L.lrw_q4));
        R.wait(e_enc);
        R.frame_pre(1);
        R.pre_done[1] = R.rec();
        R.on(0);
    }
    R.narrow(RI_EH0, fh, fw, {R.F(L.fvq), R.F(L.fvq)}, R.F(L.eh0), nullptr, nullptr, 0, 0, nullptr, ff, t, ff);   // encoder_hr(cat(fv, fv)) (:8507)
    R.narrow(RI_EH1, fh, fw, {R.F(L.eh0)}, R.F(L.x_hr), nullptr, nullptr, 0, 0, nullptr, ff, t, ff);
    const int co = y_only ? 1 : 3;
    for (int i = 0; i < t && !R.rc; ++i) R.frame(i, t, lrs + (long long)i * 3 * h * w, out + (long long)i * co * 64 * h * w);
    R.join_all();
    return R.rc;
}

}  // extern "C"

#endif  // !CRFP_ACT_BF16
