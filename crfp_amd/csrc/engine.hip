// CRFP_DSV engine: host-side schedule of the recurrent per-frame inference loop
// (reference model/CRFP.py:1387-1706, mid_channels=32, hr_dcn=True, offset_prop=True) over the
// HIP kernels of this library.  One call = one clip (or one streamed frame): ~50 asynchronous
// launches per frame on the caller's stream, no host synchronisation, all intermediates in the
// caller-provided workspace, weights pre-packed once (crfp_dsv_pack_weights).
//
// What is fused away relative to the reference's op-by-op graph:
//   * every torch.cat (:331,336,1547,1573,1586,1589,1629,1672) -> multi-source conv staging
//   * torch.chunk / split_ratio plumbing (:1592-1596) -> two destination channel ranges of one conv
//   * F.pixel_shuffle (:192) / pixel_unshuffle (:28-42) -> weight-row / K permutation + address math
//   * dcn_offset and dcn_mask convs (:337,339) -> one 32->216 conv with the 10*tanh(+flow.flip) /
//     sigmoid epilogue (:338,340,349)
//   * dcn_3's 9x replication of offset and mask (:343-347) -> never materialised
//   * the i == 0 branch's zero tensors (:1637,1666) -> K-restricted weight packs (0*w == 0 exactly)
//   * fovea blend + LeakyReLU (:1674-1675) and conv_last + bilinear base (:1678-1683) -> epilogues
//   * everything the fovea select discards (:1543-1547 -> :1672-1675 away from the mask) -> mask-gated launches (Runner::mask_gate_enabled)
#include "crfp_common.h"

#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

namespace CRFP_NS {

struct ConvDef { const char* stem; int cout, cin; };
// order == reference state_dict order (weight, bias per entry); checked against the imported
// reference by tests/golden/make_golden.py through crfp_amd/synth.py
constexpr int kNumDsvConvs = 59, kNumCraConvs = 72;
static const ConvDef kConvs[kNumCraConvs] = {
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
    {"forward_resblocks_0.main.0", 32, 64}, {"forward_resblocks_0.main.2.0.conv1", 32, 32},
    {"forward_resblocks_0.main.2.0.conv2", 32, 32},
    {"forward_resblocks_1.main.0", 32, 64}, {"forward_resblocks_1.main.2.0.conv1", 32, 32},
    {"forward_resblocks_1.main.2.0.conv2", 32, 32},
    {"forward_resblocks_2.main.0", 32, 64}, {"forward_resblocks_2.main.2.0.conv1", 32, 32},
    {"forward_resblocks_2.main.2.0.conv2", 32, 32},
    {"forward_resblocks_3.main.0", 4, 8}, {"forward_resblocks_3.main.2.0.conv1", 4, 4},
    {"forward_resblocks_3.main.2.0.conv2", 4, 4},
    {"downsample.downsample_conv", 32, 64}, {"upsample.upsample_conv", 96, 32},
    {"upsample_post.upsample_conv", 64, 24}, {"conv_last", 3, 4},
    // CRFP_DSV_CRA only (reference model/CRFP.py:2314-2664): the deeper levels of its LTE_simple_hr_ps fovea encoder (:156-166) and the
    // three per-level fusion convs
    {"encoder_hr.slice2.1", 16, 64}, {"encoder_hr.slice2.3", 16, 16}, {"encoder_hr.slice3.0", 16, 16}, {"encoder_hr.slice3.2", 16, 16},
    {"encoder_hr.slice4.0", 16, 16}, {"encoder_hr.slice4.2", 16, 16}, {"encoder_hr.conv_lv0", 16, 16}, {"encoder_hr.conv_lv1", 16, 16},
    {"encoder_hr.conv_lv2", 16, 16}, {"encoder_hr.conv_lv3", 4, 4}, {"conv_tttf_0", 32, 48}, {"conv_tttf_1", 32, 48},
    {"conv_tttf_2", 32, 48}};
// CRFP_DSV_CRA's state_dict order (the reference registers encoder_hr's levels and the fusion convs between encoder_lr and the
// propagation branches, :2347-2353), as indices into kConvs
static const int kCraOrder[kNumCraConvs] = {
    0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31, 32, 33, 34, 35, 36, 37,
    38, 39, 40, 41, 59, 60, 61, 62, 63, 64, 65, 66, 67, 68, 42, 69, 70, 71, 43, 44, 45, 46, 47, 48, 49, 50, 51, 52, 53, 54, 55, 56, 57, 58};

enum { CI_D3_UPS = 31, CI_D3_FUSE = 32, CI_D3_B0 = 33, CI_D3_B2 = 34, CI_D3_OFF = 35, CI_D3_MASK = 36, CI_D3_DCN = 37,
       CI_ENC_LR0 = 38, CI_ENC_LR1 = 39, CI_ENC_HR0 = 40, CI_ENC_HR1 = 41, CI_TTTF = 42, CI_RB3 = 52, CI_DOWN = 55,
       CI_UPS = 56, CI_UPP = 57, CI_LAST = 58,
       CI_C_S2A = 59, CI_C_S2B, CI_C_S3A, CI_C_S3B, CI_C_S4A, CI_C_S4B, CI_C_LV0, CI_C_LV1, CI_C_LV2, CI_C_LV3, CI_C_T0 };
static inline int ci_dcn(int lvl, int which) {  // which: 0 fuse, 1 block.0, 2 block.2, 3 offset, 4 mask, 5 dcn
    static const int base[3] = {13, 19, 25};     // lvl 0 has no conv_fuse: block.0 is 14
    return base[lvl] + which;
}
static inline int ci_rb(int lvl, int which) { return 43 + 3 * lvl + which; }

// Wirings of the recurrent chain one engine schedule exists for.  W_SIMPLE / W_DENSE: the reference's ablation models in front of CRFP_DSV --
// CRFP_simple ("v13", model/CRFP.py:816-1099) and CRFP ("v15", :1101-1385) with mid_channels = 32, hr_dcn and offset_prop on: the same 59
// convs under the same state_dict keys, four of them with other channel counts (`upsample` keeps all 32 features, nothing is carried
// beside a level; the dense variant hands every residual block the warped previous state as a third input).
enum Wiring { W_DSV = 0, W_CRA = 1, W_SIMPLE = 2, W_DENSE = 3 };
static int conv_cin(int ci, int wiring = W_DSV) {
    if (wiring >= W_SIMPLE) {
        if (ci == CI_UPP) return 32;                                               // upsample_post reads all mid_channels (:877)
        if (wiring == W_DENSE && (ci == 43 || ci == 46 || ci == 49)) return 96;   // forward_resblocks_k.main.0: mid_channels * 3 (:1138-1142)
        if (wiring == W_DENSE && ci == CI_RB3) return 12;                         // forward_resblocks_3.main.0: last_channels * 3 (:1148)
    }
    return kConvs[ci].cin;
}
static int conv_cout(int ci, int y_only, int wiring = W_DSV) {
    if (ci == CI_LAST && y_only) return 1;
    if (wiring >= W_SIMPLE && ci == CI_UPS) return 128;                           // upsample: PixelShufflePack(mid, mid, 2) (:875)
    return kConvs[ci].cout;
}

// ------------------------------------------------------------------ packed items
enum ItemType { T_MFMA = 0, T_NARROW = 1, T_DCN8 = 2, T_RAW = 3 };
struct Item {
    int type = T_MFMA;
    ConvArgs c;
    NarrowArgs nw;
    int w1 = -1, w2 = -1;
    size_t off_w = 0, off_b = 0, n_w = 0, n_b = 0;  // float offsets / counts inside the packed buffer
    size_t off_s = 0, n_s = 0;                       // split-bf16 weight image (T_MFMA only), in floats
    const char* name = "";
};

enum ItemId {
    IT_F0 = 0,  // .. IT_F0+13 : FNet
    IT_ENC_LR0 = 14, IT_ENC_LR1, IT_UPS, IT_DOWN,
    IT_LVL0,  // per level 10 items: FUSE, DB0, DB1, OM, DCNW, RB0, RB0F, RB1, RB2, OMF
    IT_UPP = IT_LVL0 + 30, IT_POFF,
    IT_EH0, IT_EH1, IT_D3B0, IT_D3B1, IT_D3FUSE, IT_D3OM, IT_D3W, IT_R3_0, IT_R3_0F, IT_R3_1, IT_R3_2, IT_TTTF,
    IT_LAST,
    // CRFP_DSV_CRA only
    IT_C_LV3, IT_C_S2A, IT_C_S2B, IT_C_LV2, IT_C_S3A, IT_C_S3B, IT_C_LV1, IT_C_S4A, IT_C_S4B, IT_C_LV0, IT_C_T0, IT_C_T1, IT_C_T2,
    IT_COUNT
};
enum { L_FUSE = 0, L_DB0, L_DB1, L_OM, L_DCNW, L_RB0, L_RB0F, L_RB1, L_RB2, L_OMF };
static inline int it_lvl(int lvl, int which) { return IT_LVL0 + 10 * lvl + which; }

struct SrcSpec { int kind, nch; };

static ConvArgs make_mfma(int y_only, int wiring, int ci, int ci2, std::vector<SrcSpec> srcs, int store, int ps_r, int act,
                          float post_scale, int cbase_override = -1) {
    ConvArgs a;
    memset(&a, 0, sizeof(a));
    int kq = 0, cbase = 0;
    a.nsrc = 0;
    for (auto& s : srcs) {
        ConvSrc& d = a.src[a.nsrc++];
        d.kind = s.kind;
        d.nch = s.nch;
        d.nq = src_quads(s.kind, s.nch);
        d.cbase = cbase;
        if (s.kind != SRC_ZERO) cbase += s.nch;
        kq += d.nq;
    }
    (void)cbase_override;
    if (kq & 3) {  // K is consumed in chunks of 4 quads (16 channels) by the split-bf16 main loop
        ConvSrc& d = a.src[a.nsrc++];
        d.kind = SRC_ZERO;
        d.nq = 4 - (kq & 3);
        d.nch = d.nq;
        d.cbase = cbase;
        kq += d.nq;
    }
    a.kq = kq;
    a.cin_total = conv_cin(ci, wiring);
    a.cout = conv_cout(ci, y_only, wiring) + (ci2 >= 0 ? conv_cout(ci2, y_only, wiring) : 0);
    a.store = store;
    a.ps_r = ps_r;
    a.act = act;
    a.post_scale = post_scale;
    a.ctiles = (conv_packed_rows(a.cout, store, ps_r) + 31) / 32;
    return a;
}

static NarrowArgs make_narrow(int y_only, int wiring, int ci, int ci2, std::vector<SrcSpec> srcs, int act, int epi) {
    NarrowArgs a;
    memset(&a, 0, sizeof(a));
    int kq = 0, cbase = 0;
    for (auto& s : srcs) {
        ConvSrc& d = a.src[a.nsrc++];
        d.kind = s.kind;
        d.nch = s.nch;
        d.nq = src_quads(s.kind, s.nch);
        d.cbase = cbase;
        cbase += s.nch;
        kq += d.nq;
    }
    a.kq = kq;
    a.cin_total = conv_cin(ci, wiring);
    a.cout = conv_cout(ci, y_only, wiring) + (ci2 >= 0 ? conv_cout(ci2, y_only, wiring) : 0);
    a.act = act;
    a.epi = epi;
    a.y_only = y_only;
    a.post_scale = 1.0f;
    return a;
}

struct Model {
    Item items[IT_COUNT];
    size_t total_floats = 0;
    int y_only = 0;
    bool use_s3 = false;   // producer-split SRC_S3 edges (default precision only); packed weights are identical either way
    bool cra = false;      // the CRFP_DSV_CRA wiring: 13 more convs behind the CRFP_DSV ones
    int wiring = W_DSV;
    bool abl() const { return wiring >= W_SIMPLE; }     // CRFP_simple / CRFP: all 32 features travel through the levels, no carried ones
    bool dense() const { return wiring == W_DENSE; }

    void add_mfma(int id, const char* name, int ci, int ci2, std::vector<SrcSpec> srcs, int store, int ps_r, int act,
                  float post_scale = 1.0f) {
        Item& it = items[id];
        it.type = T_MFMA;
        it.name = name;
        it.w1 = ci;
        it.w2 = ci2;
        it.c = make_mfma(y_only, wiring, ci, ci2, srcs, store, ps_r, act, post_scale);
        it.n_w = conv_packed_weight_floats(it.c);
        it.n_b = (size_t)it.c.ctiles * 32;
        it.n_s = conv_split_weight_bytes(it.c) / sizeof(float);
    }
    void add_narrow(int id, const char* name, int ci, int ci2, std::vector<SrcSpec> srcs, int act, int epi) {
        Item& it = items[id];
        it.type = T_NARROW;
        it.name = name;
        it.w1 = ci;
        it.w2 = ci2;
        it.nw = make_narrow(y_only, wiring, ci, ci2, srcs, act, epi);
        it.n_w = narrow_packed_weight_floats(it.nw);
        it.n_b = 4;
    }

    Model(int y_only_, bool use_s3_, int wiring_ = W_DSV) : y_only(y_only_), use_s3(use_s3_), cra(wiring_ == W_CRA), wiring(wiring_) {
        const int Q = SRC_Q4;
        typedef std::vector<SrcSpec> Srcs;
        static const char* fn[14] = {"conv_mfma:fnet.enc1a", "conv_mfma:fnet.enc1b", "conv_mfma:fnet.enc2a",
                                     "conv_mfma:fnet.enc2b", "conv_mfma:fnet.enc3a", "conv_mfma:fnet.enc3b",
                                     "conv_mfma:fnet.dec1a", "conv_mfma:fnet.dec1b", "conv_mfma:fnet.dec2a",
                                     "conv_mfma:fnet.dec2b", "conv_mfma:fnet.dec3a", "conv_mfma:fnet.dec3b",
                                     "conv_mfma:fnet.flow0", "conv_mfma:fnet.flow2"};
        // The LR frames enter as Q4 quads of 3 channels (fp32 build: one tiny NCHW -> Q4 pass per call, Runner::lr_to_q4): a cin = 3 / 6
        // NCHW source kept these two first layers on the fp32 MFMA (64 + 42 us per clip against ~20 + ~15 on the split-fp16 kernels).
        // The bf16 build reads the fp32 NCHW frames directly: a Q4 copy there would be bf16, i.e. a rounded INPUT.
        const int LRK = kActBf16 ? (int)SRC_NCHW : (int)Q;
        add_mfma(IT_F0, fn[0], 0, -1, {{LRK, 3}, {LRK, 3}}, ST_Q4, 0, CRFP_ACT_RELU);
        for (int i = 1; i < 13; ++i) add_mfma(IT_F0 + i, fn[i], i, -1, {{Q, kConvs[i].cin}}, ST_Q4, 0, CRFP_ACT_RELU);
        add_mfma(IT_F0 + 13, fn[13], 13, -1, {{Q, 32}}, ST_Q4, 0, CRFP_ACT_TANH, 256.0f);
        add_mfma(IT_ENC_LR0, "conv_mfma:enc_lr0", CI_ENC_LR0, -1, {{LRK, 3}}, ST_Q4, 0, CRFP_ACT_LRELU01);
        add_mfma(IT_ENC_LR1, "conv_mfma:enc_lr1", CI_ENC_LR1, -1, {{Q, 32}}, ST_Q4, 0, CRFP_ACT_LRELU01);
        add_mfma(IT_UPS, "conv_mfma:upsample_ps2", CI_UPS, -1, {{Q, 32}}, ST_PS, 2, CRFP_ACT_NONE);
        add_mfma(IT_DOWN, "conv_mfma:downsample_unshuf4", CI_DOWN, -1, {{SRC_UNSHUF4, 64}}, ST_Q4, 0, CRFP_ACT_NONE);
        // The DCN offset feature of a level (dcn_block.2 / conv_fuse output, model/CRFP.py:331-336) only ever feeds convs
        // (offset/mask head, next level's conv_fuse, dcn_3's pre-offset conv): it is stored as the producer-split SRC_S3
        // image instead of fp32 Q4 (same bytes), so the 216-channel head no longer converts the same tile 7 times.
        const int FS = use_s3 ? (int)SRC_S3 : (int)Q;
        for (int l = 0; l < 3; ++l) {
            if (l > 0)
                add_mfma(it_lvl(l, L_FUSE), "conv_mfma:dcn.conv_fuse", ci_dcn(l, 0), -1, {{Q, 32}, {FS, 32}}, ST_Q4, 0,
                         CRFP_ACT_LRELU01);
            // dcn_block.0 input = [cur = prop(24) | carry(8)] | warped prev(32) | flow(2)   (:331,1586)
            // CRFP_simple / CRFP: cur(32) | warped prev(32) | flow(2)   (:1029)
            add_mfma(it_lvl(l, L_DB0), "conv_mfma:dcn.block0", ci_dcn(l, 1), -1,
                     abl() ? Srcs{{Q, 32}, {Q, 32}, {SRC_FLOW2, 2}} : Srcs{{Q, 24}, {Q, 8}, {Q, 32}, {SRC_FLOW2, 2}}, ST_Q4, 0, CRFP_ACT_LRELU01);
            add_mfma(it_lvl(l, L_DB1), "conv_mfma:dcn.block2", ci_dcn(l, 2), -1, {{Q, 32}}, ST_Q4, 0, CRFP_ACT_LRELU01);
            add_mfma(it_lvl(l, L_OM), "conv_mfma:dcn.offset_mask", ci_dcn(l, 3), ci_dcn(l, 4), {{FS, 32}}, ST_OFFMASK, 0,
                     CRFP_ACT_NONE);
            items[it_lvl(l, L_OM)].c.n_off_quads = 36;
            // the same head packed in the register order of dcn_fused_kernel (offset / mask conv + dcn_g8 in one launch)
            add_mfma(it_lvl(l, L_OMF), "conv_mfma:dcn.offset_mask_fused", ci_dcn(l, 3), ci_dcn(l, 4), {{FS, 32}}, ST_DCNFUSE, 0,
                     CRFP_ACT_NONE);
            Item& dw = items[it_lvl(l, L_DCNW)];
            dw.type = T_DCN8;
            dw.name = "dcn_g8_weights";
            dw.w1 = ci_dcn(l, 5);
            dw.n_w = 2 * 36 * 2 * 32 * 4;   // fp32 MFMA image (strict), then the split-fp16 image (default), 36 KB each
            dw.n_b = 32;
            // resblocks input = [prop(24) | carry(8)] | aligned(32)   (:1589); first frame: prop only (:1637)
            // CRFP_simple: cur(32) | aligned(32) (:1034); CRFP: + the warped previous state (32) (:1311); first frame: cur only (:967, zeros behind it)
            add_mfma(it_lvl(l, L_RB0), "conv_mfma:res.main0", ci_rb(l, 0), -1,
                     dense() ? Srcs{{Q, 32}, {Q, 32}, {Q, 32}} : abl() ? Srcs{{Q, 32}, {Q, 32}} : Srcs{{Q, 24}, {Q, 8}, {Q, 32}}, ST_Q4, 0, CRFP_ACT_LRELU01);
            add_mfma(it_lvl(l, L_RB0F), "conv_mfma:res.main0_first", ci_rb(l, 0), -1, abl() ? Srcs{{Q, 32}} : Srcs{{Q, 24}}, ST_Q4, 0,
                     CRFP_ACT_LRELU01);
            add_mfma(it_lvl(l, L_RB1), "conv_mfma:res.conv1", ci_rb(l, 1), -1, {{Q, 32}}, ST_Q4, 0, CRFP_ACT_RELU);
            add_mfma(it_lvl(l, L_RB2), "conv_mfma:res.conv2_add", ci_rb(l, 2), -1, {{Q, 32}}, ST_Q4, 0, CRFP_ACT_NONE);
        }
        add_mfma(IT_UPP, "conv_mfma:upsample_post_ps4", CI_UPP, -1, abl() ? Srcs{{Q, 32}} : Srcs{{Q, 24}}, ST_PS, 4, CRFP_ACT_LRELU01);
        add_mfma(IT_POFF, "conv_mfma:dcn3.preoffset_ps4", CI_D3_UPS, -1, {{FS, 32}}, ST_PS, 4, CRFP_ACT_NONE, 2.0f);
        add_narrow(IT_EH0, "conv_narrow:enc_hr0", CI_ENC_HR0, -1, {{Q, 3}, {Q, 3}}, CRFP_ACT_LRELU01, NE_PLAIN);
        add_narrow(IT_EH1, "conv_narrow:enc_hr1", CI_ENC_HR1, -1, {{Q, 4}}, CRFP_ACT_LRELU01, NE_PLAIN);
        add_narrow(IT_D3B0, "conv_narrow:dcn3.block0", CI_D3_B0, -1, {{Q, 4}, {Q, 4}, {SRC_FLOW2, 2}}, CRFP_ACT_LRELU01,
                   NE_PLAIN);
        add_narrow(IT_D3B1, "conv_narrow:dcn3.block2", CI_D3_B2, -1, {{Q, 4}}, CRFP_ACT_LRELU01, NE_PLAIN);
        add_narrow(IT_D3FUSE, "conv_narrow:dcn3.conv_fuse", CI_D3_FUSE, -1, {{Q, 4}, {Q, 4}}, CRFP_ACT_LRELU01, NE_PLAIN);
        add_narrow(IT_D3OM, "conv_narrow:dcn3.offset_mask", CI_D3_OFF, CI_D3_MASK, {{Q, 4}}, CRFP_ACT_NONE, NE_OFFMASK3);
        Item& d3 = items[IT_D3W];
        d3.type = T_RAW;
        d3.name = "dcn3_weights";
        d3.w1 = CI_D3_DCN;
        d3.n_w = 4 * 4 * 9;
        d3.n_b = 4;
        add_narrow(IT_R3_0, "conv_narrow:res3.main0", CI_RB3, -1, dense() ? Srcs{{Q, 4}, {Q, 4}, {Q, 4}} : Srcs{{Q, 4}, {Q, 4}}, CRFP_ACT_LRELU01, NE_PLAIN);
        add_narrow(IT_R3_0F, "conv_narrow:res3.main0_first", CI_RB3, -1, {{Q, 4}}, CRFP_ACT_LRELU01, NE_PLAIN);
        add_narrow(IT_R3_1, "conv_narrow:res3.conv1", CI_RB3 + 1, -1, {{Q, 4}}, CRFP_ACT_RELU, NE_PLAIN);
        add_narrow(IT_R3_2, "conv_narrow:res3.conv2_add", CI_RB3 + 2, -1, {{Q, 4}}, CRFP_ACT_NONE, NE_PLAIN);
        add_narrow(IT_TTTF, "conv_narrow:tttf_blend", CI_TTTF, -1, {{Q, 4}, {Q, 4}}, CRFP_ACT_NONE, NE_BLEND);
        add_narrow(IT_LAST, "conv_narrow:last_plus_base", CI_LAST, -1, {{Q, 4}}, CRFP_ACT_NONE, NE_LAST);
        if (cra) {
            // LTE_simple_hr_ps (:156-166): slice1 is CRFP_DSV's encoder_hr pair; conv_lv3 (8x) gives the map conv_tttf blends in; slice2
            // opens with PixelUnshuffle(4) (rides in the conv's load, like `downsample`), slice2-4 and conv_lv0-2 are 16-channel convs at 2x
            add_narrow(IT_C_LV3, "conv_narrow:cra.lv3", CI_C_LV3, -1, {{Q, 4}}, CRFP_ACT_LRELU01, NE_PLAIN);
            add_mfma(IT_C_S2A, "conv_mfma:cra.slice2a_unshuf4", CI_C_S2A, -1, {{SRC_UNSHUF4, 64}}, ST_Q4, 0, CRFP_ACT_LRELU01);
            static const struct { int id, ci; const char* name; } c16[] = {
                {IT_C_S2B, CI_C_S2B, "conv_mfma:cra.slice2b"}, {IT_C_LV2, CI_C_LV2, "conv_mfma:cra.lv2"}, {IT_C_S3A, CI_C_S3A, "conv_mfma:cra.slice3a"},
                {IT_C_S3B, CI_C_S3B, "conv_mfma:cra.slice3b"}, {IT_C_LV1, CI_C_LV1, "conv_mfma:cra.lv1"}, {IT_C_S4A, CI_C_S4A, "conv_mfma:cra.slice4a"},
                {IT_C_S4B, CI_C_S4B, "conv_mfma:cra.slice4b"}, {IT_C_LV0, CI_C_LV0, "conv_mfma:cra.lv0"}};
            for (auto& c : c16) add_mfma(c.id, c.name, c.ci, -1, {{Q, 16}}, ST_Q4, 0, CRFP_ACT_LRELU01);
            // conv_tttf_k(cat(level features (32), fovea level (16))) (:2533): the blend with the resampled mask is its own small pass
            for (int k = 0; k < 3; ++k) add_mfma(IT_C_T0 + k, "conv_mfma:cra.tttf_level", CI_C_T0 + k, -1, {{Q, 32}, {Q, 16}}, ST_Q4, 0, CRFP_ACT_NONE);
        }
        size_t cur = 0;
        for (int i = 0; i < IT_COUNT; ++i) {
            Item& it = items[i];
            if (it.w1 < 0) continue;
            it.off_w = cur;
            cur += (it.n_w + 63) / 64 * 64;
            it.off_b = cur;
            cur += (it.n_b + 63) / 64 * 64;
            it.off_s = cur;
            cur += (it.n_s + 63) / 64 * 64;
        }
        total_floats = cur;
    }
};

// strict: the fp32-MFMA wiring (no SRC_S3 edges); same packed-weight layout as the default wiring
template <int WIRING>
static const Model& model_of(int y_only, bool strict) {
    static const Model m0(0, false, WIRING), m1(1, false, WIRING), s0(0, true, WIRING), s1(1, true, WIRING);
    if (strict || !conv_s3_supported()) return y_only ? m1 : m0;
    return y_only ? s1 : s0;
}
static const Model& model_for(int y_only, bool strict = false, int wiring = W_DSV) {
    switch (wiring) {
        case W_CRA: return model_of<W_CRA>(y_only, strict);
        case W_SIMPLE: return model_of<W_SIMPLE>(y_only, strict);
        case W_DENSE: return model_of<W_DENSE>(y_only, strict);
        default: return model_of<W_DSV>(y_only, strict);
    }
}

// ------------------------------------------------------------------ workspace arena
struct Buf { std::string name; size_t off; int N, nq, H, W, kind, pad; size_t bytes, guard; bool f32; };  // kind 0 = Q4, 1 = NHW2 (always float)
// a workspace tensor: byte offset + elements (of the tensor's own type) between two batch items
struct Ten {
    size_t off = 0;
    long long bs = 0;
    operator size_t() const { return off; }
};
struct Arena {
    size_t cur = 0;
    std::vector<Buf> bufs;
    // pad = 1: "P4" planes of (H+1) x (W+1) with zero pad row / column (sources of the gather kernels)
    // f32 = true: a Q4 tensor that stays float in the bf16 build as well (flow, DCN offsets / masks: coordinates)
    // The N items are contiguous, so item n > 0 of a P4 tensor finds its guard (one pad row + one element of zeros in front of
    // plane 0) in the tail of item n - 1: that item's last pad row, preceded by the pad column of its last data row.
    Ten take(const char* name, int N, int nq, int H, int W, int kind = 0, int pad = 0, bool f32 = false) {
        const size_t per = kind == 0 ? (size_t)nq * (H + pad) * (W + pad) * 4 : (size_t)H * W * 2;
        const size_t elems = (size_t)N * per;
        const size_t esz = (kind == 0 && !f32) ? sizeof(act_t) : sizeof(float);
        // P4 tensors carry a zeroed guard (>= one pad row + one element) in front of plane 0
        const size_t guard = pad ? align_up((size_t)(W + 2) * 16, 256) : 0;
        const size_t off = cur + guard;
        cur += guard + align_up(elems * esz, 256);
        bufs.push_back({name, off, N, nq, H, W, kind, pad, elems * esz, guard, kind != 0 || f32});
        Ten t;
        t.off = off;
        t.bs = (long long)per;
        return t;
    }
    const Buf* find(size_t off) const {
        for (auto& b : bufs) if (b.off == off) return &b;
        return nullptr;
    }
};

struct Q4 {
    float* p = nullptr;
    int nq = 0, H = 0, W = 0;
    long long bs() const { return (long long)nq * H * W * 4; }
};

// Frames per launch of the clip-level stages (FNet, encoder_lr).  A batch of B clips of t frames whose B * t frames fit kFlatFrames
// runs them ONCE over the flattened [B * t] frame sequence -- the API tensors' own memory order; the B - 1 "pairs" that straddle two
// clips are computed and never read.  Longer jobs walk every clip in chunks of kChunkFrames frames, so the workspace does not
// grow with t (BASELINE config 2-5 shapes are all flat; a 100-frame clip takes 13 chunks of 8).
constexpr int kFlatFrames = 32, kChunkFrames = 8;

struct Layout {
    Arena A;
    int B, t, h, w;
    bool flat;   // clip-level stages over all B * t frames at once
    int TC;      // frames per clip held by the clip-level stores (x_lr, flow_lr): t when flat, else kChunkFrames; slot of (clip b, frame i) = b * TC + i % TC
    // status word (fp16-operand overflow flag), then the persistent recurrent state (stable offsets for streaming)
    Ten status, state_hr, carry;
    // clip-level
    Ten flow_lr, e_lr0, x_lr, lr_q4, lr_keep[2];
    // FNet
    Ten fa0, fa1, fp1, fb0, fb1, fp2, fc0, fc1, fp3, fd0, fd1, fu1, fe0, fe1, fu2, ff0, ff1, fu3, fg0, fg1;
    // frame-level
    Ten xin8[2], eh[2], x_hr[2], prop0[2], prop_a, prop_b, flow2[2], flow8[2], prev2, prev2w, prevhrw, carryw, fa, fb, offfeat[3], offmask,
        aligned, y0, y1, up, poff, g0, g1, g2, om3, al3, z0, z1, feat, fg2, sc_prop, sc_cw, sc_al, sc_up, sc_al3;
    // CRFP_DSV_CRA: slice1's output, the 2x chain's two temporaries, the three fovea levels per buffer set, a level's features and their fused twin
    Ten c_s1, c_a, c_b, c_lv[2][3], c_y, c_f;
    bool cra;
    int pq;   // quads of the features a level passes on: 6 (CRFP_DSV: 24 + 8 carried beside the level) or 8 (CRFP_simple / CRFP)
    // mask gate of each buffer set: 4 flag bytes per 64 x 16 tile of the 8x map and clip (launch_mask_gate); gate_b = bytes per clip
    Ten gate[2];
    long long gate_b;
    int h1, w1, h2, w2, h3, w3;
    int fnet_cap;   // pairs one FNet pass can hold
    // FNet's small maps run their convs in K slices (conv_auto_ksplit): the float partial tensors of the layer in flight (the largest layer's ks * cout * H * W
    // floats per pair)
    Ten fpart;

    Layout(int B_, int t_, int h_, int w_, int wiring = W_DSV) : B(B_), t(t_), h(h_), w(w_), cra(wiring == W_CRA), pq(wiring >= W_SIMPLE ? 8 : 6) {
        flat = (long long)B * t <= kFlatFrames;
        TC = flat ? t : kChunkFrames;
        // pairs one FNet pass holds: all of a flat job's (the B - 1 straddling ones included); one per sequence for the one-frame-per-call layout
        const int nb = t == 1 ? B : (flat ? B * t - 1 : TC);
        fnet_cap = nb;
        const int H2 = 2 * h, W2 = 2 * w, H8 = 8 * h, W8 = 8 * w;
        h1 = h / 2; w1 = w / 2; h2 = h1 / 2; w2 = w1 / 2; h3 = h2 / 2; w3 = w2 / 2;
        status = A.take("status", 1, 0, 1, B > 64 ? (B + 1) / 2 : 32, 1);   // >= 256 bytes; word b = overflow flag of clip b
        state_hr = A.take("state_hr", B, 1, H8, W8, 0, 1);
        carry = A.take("carry", B, 6, H2, W2, 0, 1);
        // one-frame-per-call layout (t == 1): two flow slots and (bf16 build) two kept fp32 copies of the LR frame, indexed by call parity
        // (CRFP_DSV_INPUTS_RESIDENT: the flow network of call i runs while frame i - 1 still reads the other slot)
        flow_lr = A.take("flow_lr", t == 1 ? 2 * B : B * TC, 1, h, w, 0, 0, true);
        if (kActBf16 && t == 1)
            for (int p = 0; p < 2; ++p) lr_keep[p] = A.take(p ? "lr_keep.1" : "lr_keep", 3 * B, 0, h, (w + 1) / 2, 1);   // >= B * 3 * h * w floats
        lr_q4 = A.take("lr_q4", 2 * B * t, 1, h, w);   // the LR frames as quads: [0, B t) current frames, [B t, 2 B t) previous frames when they are not the same tensor
        e_lr0 = A.take("enc_lr0", B * TC, 8, h, w);
        x_lr = A.take("x_lr", B * TC, 8, h, w);
        fa0 = A.take("fnet.a0", nb, 8, h, w);
        fa1 = A.take("fnet.a1", nb, 8, h, w);
        fp1 = A.take("fnet.p1", nb, 8, h1, w1);
        fb0 = A.take("fnet.b0", nb, 16, h1, w1);
        fb1 = A.take("fnet.b1", nb, 16, h1, w1);
        fp2 = A.take("fnet.p2", nb, 16, h2, w2);
        fc0 = A.take("fnet.c0", nb, 32, h2, w2);
        fc1 = A.take("fnet.c1", nb, 32, h2, w2);
        fp3 = A.take("fnet.p3", nb, 32, h3, w3);
        fd0 = A.take("fnet.d0", nb, 64, h3, w3);
        fd1 = A.take("fnet.d1", nb, 64, h3, w3);
        fu1 = A.take("fnet.u1", nb, 64, 2 * h3, 2 * w3);
        fe0 = A.take("fnet.e0", nb, 32, 2 * h3, 2 * w3);
        fe1 = A.take("fnet.e1", nb, 32, 2 * h3, 2 * w3);
        fu2 = A.take("fnet.u2", nb, 32, 4 * h3, 4 * w3);
        ff0 = A.take("fnet.f0", nb, 16, 4 * h3, 4 * w3);
        ff1 = A.take("fnet.f1", nb, 16, 4 * h3, 4 * w3);
        fu3 = A.take("fnet.u3", nb, 16, 8 * h3, 8 * w3);
        fg0 = A.take("fnet.g0", nb, 8, 8 * h3, 8 * w3);
        fg1 = A.take("fnet.g1", nb, 1, 8 * h3, 8 * w3, 0, 0, true);
        {
            const int lh[7] = {h, h1, h2, h3, 2 * h3, 4 * h3, 8 * h3}, lw[7] = {w, w1, w2, w3, 2 * w3, 4 * w3, 8 * w3};
            long long most = 0;
            for (int i = 0; i < 14; ++i) {
                const int kq = (kConvs[i].cin / 4 + 3) / 4 * 4, H = lh[i / 2], W = lw[i / 2];
                const int ks = H > 0 && W > 0 ? conv_auto_ksplit(H, W, (kConvs[i].cout + 31) / 32, kq) : 1;
                if (ks > 1) most = std::max(most, (long long)ks * kConvs[i].cout * H * W);
            }
            const long long per = (long long)h * w * 4;
            fpart = A.take("fnet.part", nb, (int)((most + per - 1) / per), h, w, 0, 0, true);
        }
        // state-independent per-frame work (fovea blend, encoder_hr, upsample conv, flow upsampling) is
        // produced one or two frames ahead on a side stream -> two buffer sets, indexed by frame parity
        for (int p = 0; p < 2; ++p) {
            xin8[p] = A.take(p ? "xin8.1" : "xin8", B, 2, H8, W8);
            eh[p] = A.take(p ? "enc_hr0.1" : "enc_hr0", B, 1, H8, W8);
            x_hr[p] = A.take(p ? "x_hr.1" : "x_hr", B, 1, H8, W8);
            prop0[p] = A.take(p ? "prop0.1" : "prop0", B, pq, H2, W2);
            flow2[p] = A.take(p ? "flow2.1" : "flow2", B, 0, H2, W2, 1);
            flow8[p] = A.take(p ? "flow8.1" : "flow8", B, 0, H8, W8, 1);
            const int gtiles = ((W8 + 63) / 64) * ((H8 + 15) / 16);
            gate_b = ((long long)gtiles * 4 + 7) / 8 * 8;
            gate[p] = A.take(p ? "gate.1" : "gate", B, 0, 1, (int)(gate_b / 8), 1);
        }
        prop_a = A.take("prop_a", B, pq, H2, W2);
        prop_b = A.take("prop_b", B, pq, H2, W2);
        prev2 = A.take("prev2", B, 8, H2, W2, 0, 1);
        prev2w = A.take("prev2w", B, 8, H2, W2);
        prevhrw = A.take("prevhrw", B, 1, H8, W8);
        carryw = A.take("carryw", B, 6, H2, W2);
        fa = A.take("dcn.fa", B, 8, H2, W2);
        fb = A.take("dcn.fb", B, 8, H2, W2);
        for (int l = 0; l < 3; ++l) offfeat[l] = A.take(l == 0 ? "offfeat0" : (l == 1 ? "offfeat1" : "offfeat2"), B, 8, H2, W2);
        offmask = A.take("offmask", B, 54, H2, W2, 0, 0, true);
        aligned = A.take("aligned", B, 8, H2, W2);
        y0 = A.take("res.y0", B, 8, H2, W2);
        y1 = A.take("res.y1", B, 8, H2, W2);
        up = A.take("up", B, 1, H8, W8);
        poff = A.take("poff", B, 1, H8, W8);
        g0 = A.take("dcn3.g0", B, 1, H8, W8);
        g1 = A.take("dcn3.g1", B, 1, H8, W8);
        g2 = A.take("dcn3.g2", B, 1, H8, W8);
        om3 = A.take("om3", B, 1, H8, W8, 0, 0, true);
        al3 = A.take("aligned3", B, 1, H8, W8);
        z0 = A.take("res3.z0", B, 1, H8, W8);
        z1 = A.take("res3.z1", B, 1, H8, W8);
        feat = A.take("feat", B, 1, H8, W8);
        if (cra) {
            static const char* lvn[2][3] = {{"cra.lv0", "cra.lv1", "cra.lv2"}, {"cra.lv0.1", "cra.lv1.1", "cra.lv2.1"}};
            c_s1 = A.take("cra.s1", B, 1, H8, W8);
            c_a = A.take("cra.a", B, 4, H2, W2);
            c_b = A.take("cra.b", B, 4, H2, W2);
            for (int p = 0; p < 2; ++p)
                for (int k = 0; k < 3; ++k) c_lv[p][k] = A.take(lvn[p][k], B, 4, H2, W2);
            c_y = A.take("cra.y", B, 8, H2, W2);
            c_f = A.take("cra.fused", B, 8, H2, W2);
        }
        // regional-mask (fgs) copies of the streaming variant (one frame per call, one sequence per workspace)
        if (t == 1 && B == 1) {
            fg2 = A.take("fg2", 1, 0, H2 / 2, W2, 1);   // H2*W2 floats (kind 1 stores 2 floats per element)
            sc_prop = A.take("fg.prop", 1, 6, H2, W2);
            sc_cw = A.take("fg.carry", 1, 2, H2, W2);
            sc_al = A.take("fg.aligned", 1, 8, H2, W2);
            sc_up = A.take("fg.up", 1, 1, H8, W8);
            sc_al3 = A.take("fg.aligned3", 1, 1, H8, W8);
        }
    }
    size_t bytes() const { return A.cur; }
    // slot of (clip b, frame i) in the clip-level stores
    long long slot(int b, int i) const { return (long long)b * TC + i % TC; }
};

// One non-blocking side stream + an event pool per (host thread, device): state-independent work of upcoming frames is
// forked onto it (fork/join through events recorded on the caller's stream, so the call stays ordered on `stream` and
// remains graph-capturable).  This is the ONLY state the library keeps between calls (documented in crfp_hip.h);
// crfp_shutdown() destroys the calling thread's streams and events.
// host-side note per streamed sequence (keyed by its workspace): what CRFP_DSV_INPUTS_RESIDENT calls need to know about the previous call
struct StreamCtx {
    unsigned n = 0;        // calls since the sequence started: parity of the buffer sets
    bool kept = false;     // the previous call left its LR frame in the workspace
    bool chained = false;  // the previous call was a two-stream resident call: its side work waited for everything before it on the caller's stream
};
struct SideStream {
    hipStream_t s = nullptr;
    std::vector<hipEvent_t> ev;
    std::unordered_map<const void*, StreamCtx> ctx;
    bool ok = true;
    hipEvent_t event(size_t i) {
        while (ev.size() <= i) {
            hipEvent_t e;
            if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { ok = false; return nullptr; }
            ev.push_back(e);
        }
        return ev[i];
    }
    void destroy() {
        for (hipEvent_t e : ev) (void)hipEventDestroy(e);
        ev.clear();
        ctx.clear();
        if (s) (void)hipStreamDestroy(s);
        s = nullptr;
        ok = true;
    }
};
constexpr int kMaxDevices = 64;
// One table per host thread, taken on first use from a process-wide registry: crfp_shutdown() walks ALL tables, so the streams and
// events of worker threads that have exited are released as well (a thread_local destructor would have to call into the HIP
// runtime while the process may already be tearing it down).  Round 5: a thread that exits hands its table back to a
// free list -- its destructor makes NO HIP call, it only forgets the per-sequence notes -- and the next new thread reuses it with its
// stream and events, so a thread-per-request host no longer grows the registry by one table (64 slots, a stream, an event pool) per
// thread.  The registry itself is heap-allocated and never destroyed: worker threads may outlive the static destructors.
struct SideTable { SideStream dev[kMaxDevices]; };
struct SideRegistry { std::mutex mu; std::vector<SideTable*> all, idle; };
static SideRegistry& side_registry() { static SideRegistry* r = new SideRegistry(); return *r; }
struct SideLease {
    SideTable* t = nullptr;
    ~SideLease() {
        if (!t) return;
        for (int d = 0; d < kMaxDevices; ++d) t->dev[d].ctx.clear();   // sequences of the exiting thread: the next owner starts clean
        SideRegistry& r = side_registry();
        std::lock_guard<std::mutex> lk(r.mu);
        r.idle.push_back(t);
    }
};
static thread_local SideLease g_side_tl;
static SideStream* side_table() {
    if (!g_side_tl.t) {
        SideRegistry& r = side_registry();
        std::lock_guard<std::mutex> lk(r.mu);
        if (!r.idle.empty()) { g_side_tl.t = r.idle.back(); r.idle.pop_back(); }
        else { g_side_tl.t = new SideTable(); r.all.push_back(g_side_tl.t); }
    }
    return g_side_tl.t->dev;
}
static void destroy_all_side_streams() {   // caller: no crfp_dsv_* call in flight on any thread
    SideRegistry& r = side_registry();
    std::lock_guard<std::mutex> lk(r.mu);
    for (SideTable* t : r.all)
        for (int d = 0; d < kMaxDevices; ++d) t->dev[d].destroy();
}
#ifndef CRFP_ACT_BF16
extern "C" int crfp_debug_side_tables(void) {   // tests: how many per-thread tables the fp32 engine's registry holds
    SideRegistry& r = side_registry();
    std::lock_guard<std::mutex> lk(r.mu);
    return (int)r.all.size();
}
#endif
// streams and events belong to the device that was current when they were created; a device index outside the table
// gets no side stream (the caller then runs the single-stream schedule) instead of aliasing another device's slot
static SideStream* side_slot() {   // the calling thread's table entry for the current device; creates nothing
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return nullptr;
    return &side_table()[dev];
}
static SideStream* side_stream() {
    SideStream* ss = side_slot();
    if (!ss) return nullptr;
    if (!ss->s && ss->ok && hipStreamCreateWithFlags(&ss->s, hipStreamNonBlocking) != hipSuccess) ss->ok = false;
    return ss->ok ? ss : nullptr;
}
static bool side_stream_enabled() {
    static const bool on = !(getenv("CRFP_SIDE_STREAM") && atoi(getenv("CRFP_SIDE_STREAM")) == 0);   // read once
    return on;
}

struct Runner {
    const Model& M;
    const float* packed;
    char* ws;
    const Layout& L;
    hipStream_t s;
    int rc = 0;
    int strict = 0;     // CRFP_DSV_STRICT_F32: fp32 MFMA for every conv and for dcn_g8's GEMM
    bool down_done = false;   // one-frame-per-call path: downsample(state) already ran on the side stream, beside FNet
    int flow_slot = 0;        // one-frame-per-call path: which of the two flow_lr slots FNet writes
    float* flow_lr_slot() const { return F(L.flow_lr) + (long long)flow_slot * L.B * L.h * L.w * 4; }   // B flows per call parity

    // Status words: one per clip of the call.  Batch item n of a launch raises word ovf_off + (n + ovf_add) / ovf_div (ovf_word(),
    // crfp_common.h): the per-frame launches carry one item per clip (div 1); the clip-level stages set their own mapping (clip_stage).
    int ovf_div = 1, ovf_add = 0, ovf_off = 0, ovf_skip0 = 0;
    // null in strict mode: no kernel forms fp16 operands, the guard has nothing to watch (the words stay 0)
    unsigned* ovf() const {
        static const bool env_strict = precision_env_strict(0);
        return strict || env_strict ? nullptr : reinterpret_cast<unsigned*>(ws + L.status) + ovf_off;
    }
    float* F(size_t off) const { return reinterpret_cast<float*>(ws + off); }
    // activation pointers are opaque float* at this level: advance by ELEMENTS of the storage type
    static float* adv(float* p, long long elems) { return reinterpret_cast<float*>(reinterpret_cast<char*>(p) + elems * (long long)sizeof(act_t)); }
    static const float* adv(const float* p, long long elems) { return adv(const_cast<float*>(p), elems); }
    Q4 q(size_t off, int nq, int H, int W) const { Q4 r; r.p = F(off); r.nq = nq; r.H = H; r.W = W; return r; }

    struct SrcBind { const float* p; long long bs; int pad = 0; };
    struct DstBind { float* p; long long bs; int q0, q1; int pad = 0; };

    // s3: the output goes (only, when dsts is empty) to an SRC_S3 image
    void mfma(int id, int N, int H, int W, std::vector<SrcBind> srcs, std::vector<DstBind> dsts, int dstH = 0, int dstW = 0,
              const float* resid = nullptr, long long resid_bs = 0, const float* flow = nullptr, long long flow_bs = 0,
              float* s3 = nullptr, long long s3_bs = 0, int dst_f32 = 0, int src_bgroup = 0) {
        if (rc) return;
        const Item& it = M.items[id];
        ConvArgs a = it.c;
        a.src_bgroup = src_bgroup;
        for (size_t i = 0; i < srcs.size(); ++i) { a.src[i].p = srcs[i].p; a.src[i].bstride = srcs[i].bs; a.src[i].pad = srcs[i].pad; }
        a.ndst = (int)dsts.size();
        for (size_t i = 0; i < dsts.size(); ++i) {
            a.dst[i].p = dsts[i].p; a.dst[i].bstride = dsts[i].bs; a.dst[i].q0 = dsts[i].q0; a.dst[i].q1 = dsts[i].q1;
            a.dst[i].pad = dsts[i].pad;
        }
        a.N = N; a.H = H; a.W = W; a.dstH = dstH; a.dstW = dstW;
        a.resid = resid; a.resid_bstride = resid_bs; a.flow = flow; a.flow_bstride = flow_bs;
        a.s3_dst = s3; a.s3_bstride = s3_bs; a.dst_f32 = dst_f32;
        a.wpk = packed + it.off_w;
        a.bpk = packed + it.off_b;
        a.wsplit = packed + it.off_s;
        a.ovf = ovf(); a.ovf_div = ovf_div; a.ovf_add = ovf_add; a.ovf_skip0 = ovf_skip0;
        a.strict = strict;
        rc = launch_conv_mfma(a, it.name, s);
    }
    // the launch-time plan of item id as mfma() builds it
    ConvArgs plan(int id, int N, int H, int W, const std::vector<SrcBind>& srcs, const std::vector<DstBind>& dsts, int dstH, int dstW) const {
        const Item& it = M.items[id];
        ConvArgs a = it.c;
        for (size_t i = 0; i < srcs.size(); ++i) { a.src[i].p = srcs[i].p; a.src[i].bstride = srcs[i].bs; a.src[i].pad = srcs[i].pad; }
        a.ndst = (int)dsts.size();
        for (size_t i = 0; i < dsts.size(); ++i) {
            a.dst[i].p = dsts[i].p; a.dst[i].bstride = dsts[i].bs; a.dst[i].q0 = dsts[i].q0; a.dst[i].q1 = dsts[i].q1;
            a.dst[i].pad = dsts[i].pad;
        }
        a.N = N; a.H = H; a.W = W; a.dstH = dstH; a.dstW = dstW;
        a.wpk = packed + it.off_w; a.bpk = packed + it.off_b; a.wsplit = packed + it.off_s;
        a.ovf = ovf(); a.ovf_div = ovf_div; a.ovf_add = ovf_add; a.ovf_skip0 = ovf_skip0;
        a.strict = strict;
        return a;
    }
    // two independent convs of the same shape as ONE launch where the build can (conv_mfma.hip, launch_conv_mfma_dual)
    void mfma_dual(int idA, std::vector<SrcBind> srcsA, std::vector<DstBind> dstsA, int idB, std::vector<SrcBind> srcsB, std::vector<DstBind> dstsB,
                   int N, int H, int W, int dstH, int dstW, const char* name_both) {
        if (rc) return;
        ConvArgs a = plan(idA, N, H, W, srcsA, dstsA, dstH, dstW), b = plan(idB, N, H, W, srcsB, dstsB, dstH, dstW);
        rc = launch_conv_mfma_dual(a, M.items[idA].name, b, M.items[idB].name, name_both, s);
    }
    // fills the per-launch fields of a packed MFMA item's plan (what mfma() passes to launch_conv_mfma)
    ConvArgs bind(int id, int N, int H, int W, const std::vector<SrcBind>& srcs, const std::vector<DstBind>& dsts, const float* resid = nullptr,
                  long long resid_bs = 0) const {
        const Item& it = M.items[id];
        ConvArgs a = it.c;
        for (size_t i = 0; i < srcs.size(); ++i) { a.src[i].p = srcs[i].p; a.src[i].bstride = srcs[i].bs; a.src[i].pad = srcs[i].pad; }
        a.ndst = (int)dsts.size();
        for (size_t i = 0; i < dsts.size(); ++i) {
            a.dst[i].p = dsts[i].p; a.dst[i].bstride = dsts[i].bs; a.dst[i].q0 = dsts[i].q0; a.dst[i].q1 = dsts[i].q1;
            a.dst[i].pad = dsts[i].pad;
        }
        a.N = N; a.H = H; a.W = W;
        a.resid = resid; a.resid_bstride = resid_bs;
        a.wpk = packed + it.off_w; a.bpk = packed + it.off_b; a.wsplit = packed + it.off_s;
        a.ovf = ovf(); a.ovf_div = ovf_div; a.ovf_add = ovf_add;
        a.strict = strict;
        return a;
    }
#ifdef CRFP_ACT_BF16
    // conv idA -> conv idB in one launch (conv3x3_bf16_pair_kernel): idA's output has no other reader and is never written
    void mfma_pair(int idA, int idB, const char* name, int N, int H, int W, std::vector<SrcBind> srcsA, std::vector<DstBind> dstsB,
                   const float* residB = nullptr, long long residB_bs = 0) {
        if (rc) return;
        const ConvArgs a = bind(idA, N, H, W, srcsA, {}), b = bind(idB, N, H, W, {{nullptr, 0}}, dstsB, residB, residB_bs);
        rc = launch_conv_pair(a, b, name, s);
    }
#endif
    // which 2x-resolution conv pairs run as one launch: bf16 storage only (the fp32 build's intermediate does not fit LDS, conv_mfma.hip),
    // and one clip per call only: the pair kernel (two 77 KB workgroups per CU, 1.25x the MFMAs for conv A's halo) wins where a launch is a
    // single round of workgroups (39.5 -> 34.5 and 31.9 -> 25.3 us per clip, round 3); in a lock-step batch the two single convs on the
    // 4-wave kernel take 26.0 and 19.9 us per clip against the pairs' 29.0 and 19.9 (round 4, 4 clips, same box), and so do the 540 x 960 maps of
    // one 4K clip (1 088 pair tiles: 572 vs 562 frames/s).  So: pairs while the launch is at most one round of the 512 slots.
    // CRFP_CONV_PAIR=0 / 2: never / always
    bool pair_convs() const {
        static const int mode = !kActBf16 ? 0 : (getenv("CRFP_CONV_PAIR") ? atoi(getenv("CRFP_CONV_PAIR")) : 1);   // read once
        const long long pair_tiles = (long long)L.B * ((2 * L.w + 61) / 62) * ((2 * L.h + 7) / 8);   // 8 x 62 output tiles of the pair kernel
        return mode == 2 || (mode == 1 && pair_tiles <= 512);
    }
    // plain Q4 -> Q4 conv on whole tensors
    void mfma_q(int id, int N, const Q4& in, const Q4& out) {
        mfma(id, N, in.H, in.W, {{in.p, in.bs()}}, {{out.p, out.bs(), 0, out.nq}});
    }
    // 8x-resolution stencils over the B clips of the call.  NB: tensors are {pointer, batch stride in elements of their own type}
    struct NB { const float* p = nullptr; long long bs = 0; NB() {} NB(const float* p_, long long bs_) : p(p_), bs(bs_) {} };
    // gate_par >= 0: the mask-gated form over buffer set gate_par's flags, reading the "within gate_h pixels" byte
    void narrow(int id, int H, int W, std::vector<NB> srcs, NB dst, NB resid = NB(), NB flow = NB(), const uint8_t* mask = nullptr,
                long long mask_bs = 0, int src0_pad = 0, int dst_pad = 0, NB base_lr = NB(), int gate_par = -1, int gate_h = 0, NB dst2 = NB()) {
        if (rc) return;
        const Item& it = M.items[id];
        NarrowArgs a = it.nw;
        for (size_t i = 0; i < srcs.size(); ++i) { a.src[i].p = srcs[i].p; a.src[i].bstride = srcs[i].bs; a.src[i].pad = 0; }
        a.src[0].pad = src0_pad;
        a.dst_pad = dst_pad;
        a.N = L.B; a.H = H; a.W = W;
        a.dst = const_cast<float*>(dst.p); a.dst_bstride = dst.bs;
        a.resid = resid.p; a.resid_bstride = resid.bs;
        a.flow = flow.p; a.flow_bstride = flow.bs;
        a.base = nullptr; a.base_lr = base_lr.p; a.base_bstride = base_lr.bs;
        a.mask = mask; a.mask_bstride = mask_bs;
        a.wpk = packed + it.off_w;
        a.bpk = packed + it.off_b;
        a.ovf = ovf(); a.ovf_div = ovf_div; a.ovf_add = ovf_add;
        a.dst2 = const_cast<float*>(dst2.p); a.dst2_bstride = dst2.bs;
        if (gate_par >= 0 && mask_gate_enabled()) { a.gate = gate_ptr(gate_par); a.gate_bstride = L.gate_b; a.gate_h = gate_h; }
        rc = launch_narrow(a, it.name, s);
    }
    // two stencils in one pass: item idA (its output has no other reader) feeding item idB (conv_narrow.hip, launch_narrow_pair)
    void narrow_pair(int idA, int idB, const char* name, int H, int W, std::vector<NB> srcsA, NB dst, NB residB = NB(), NB flowB = NB(), NB dst2B = NB()) {
        if (rc) return;
        NarrowArgs a = M.items[idA].nw, b = M.items[idB].nw;
        for (size_t i = 0; i < srcsA.size(); ++i) { a.src[i].p = srcsA[i].p; a.src[i].bstride = srcsA[i].bs; a.src[i].pad = 0; }
        a.N = b.N = L.B; a.H = b.H = H; a.W = b.W = W;
        a.wpk = packed + M.items[idA].off_w; a.bpk = packed + M.items[idA].off_b;
        b.wpk = packed + M.items[idB].off_w; b.bpk = packed + M.items[idB].off_b;
        b.dst = const_cast<float*>(dst.p); b.dst_bstride = dst.bs;
        b.resid = residB.p; b.resid_bstride = residB.bs;
        b.flow = flowB.p; b.flow_bstride = flowB.bs;
        b.dst_pad = 0;
        b.dst2 = const_cast<float*>(dst2B.p); b.dst2_bstride = dst2B.bs;
        b.ovf = ovf(); b.ovf_div = ovf_div; b.ovf_add = ovf_add;
        rc = launch_narrow_pair(a, b, name, s);
    }
    // three stencils in one pass (fp32 build; conv_narrow.hip, launch_narrow_chain): items idA -> idB -> idC, the two tensors between them never
    // reach HBM.  gC: C's second source (its first is B's output); res: C adds A's output; dst2C: C's second destination (the new state)
    void narrow_chain(int idA, int idB, int idC, const char* name, int H, int W, std::vector<NB> srcsA, NB gC, NB dst, bool res, NB dst2C = NB()) {
        if (rc) return;
        NarrowArgs a = M.items[idA].nw, b = M.items[idB].nw, c = M.items[idC].nw;
        for (size_t i = 0; i < srcsA.size(); ++i) { a.src[i].p = srcsA[i].p; a.src[i].bstride = srcsA[i].bs; a.src[i].pad = 0; }
        if (gC.p) { c.src[1].p = gC.p; c.src[1].bstride = gC.bs; c.src[1].pad = 0; }
        a.N = b.N = c.N = L.B; a.H = b.H = c.H = H; a.W = b.W = c.W = W;
        a.wpk = packed + M.items[idA].off_w; a.bpk = packed + M.items[idA].off_b;
        b.wpk = packed + M.items[idB].off_w; b.bpk = packed + M.items[idB].off_b;
        c.wpk = packed + M.items[idC].off_w; c.bpk = packed + M.items[idC].off_b;
        c.dst = const_cast<float*>(dst.p); c.dst_bstride = dst.bs; c.dst_pad = 0;
        c.dst2 = const_cast<float*>(dst2C.p); c.dst2_bstride = dst2C.bs;
        c.ovf = ovf(); c.ovf_div = ovf_div; c.ovf_add = ovf_add;
        rc = launch_narrow_chain(a, b, c, res, name, s);
    }
#ifndef CRFP_NARROW_CHAIN
// bit 0: dcn_3's dcn_block.0 -> .2 -> conv_fuse, bit 1: forward_resblocks_3's main.0 -> conv1 -> conv2 (+ x).  Shipped: 2.  Same box, fp32 @A
// (profiles/r06_narrow_chain_ab.txt): the residual chain 87.1 us against 41.1 + 50.8; dcn_3's chain 148.4 us against 59.5 + 28.0 + 42.2 -- its A
// stage (three input quads on 1.33x the pixels) makes it issue-bound at three workgroups per CU, so it stays three launches.
#define CRFP_NARROW_CHAIN 2
#endif
#ifndef CRFP_NARROW_CHAIN_BF16
#define CRFP_NARROW_CHAIN_BF16 3   // bf16 build: both chains (its block-diagonal MFMA makes the stages' extra pixels cheap: profiles/r06_narrow_chain_ab.txt)
#endif
#ifdef CRFP_LAB   // lab library: CRFP_NARROW_CHAIN=<mask> at run time (tests/test_gpu_round6.py compares the chains with the launches they replace)
    static int chain_mask() {
        static const int m = getenv("CRFP_NARROW_CHAIN") ? atoi(getenv("CRFP_NARROW_CHAIN")) : (kActBf16 ? CRFP_NARROW_CHAIN_BF16 : CRFP_NARROW_CHAIN);
        return m;
    }
#else
    static constexpr int chain_mask() { return kActBf16 ? CRFP_NARROW_CHAIN_BF16 : CRFP_NARROW_CHAIN; }
#endif
    // which 8x-resolution conv pairs run fused (bit 0: encoder_hr.0->.2, 1: dcn_3 conv_fuse->offset/mask, 2: res3 conv1->conv2,
    // 3: dcn_3 block.0->.2).  Measured @A fp32, pair vs the two single kernels: res3 47.2 vs 57.8 us (bf16 46.7 vs 56.1);
    // encoder_hr 71.5 vs 67.6; conv_fuse->offset/mask 80 vs 76.8; dcn_3 block (3 input quads, one workgroup per CU) 126 vs 93.7.
    // The stencils are bound by 4x4x1-MFMA issue and LDS reads, not by HBM, and the fused form evaluates conv A on 1.16x the
    // pixels with 9 instead of 4.5 LDS reads per pixel and tap row -- so only the pair whose A has ONE input quad wins.
    // narrow conv pairs fused into one launch: res3.conv1 -> conv2(+x) wins in the fp32 build; in the bf16 build the
    // single convs run on the bf16 MFMA and two of them beat the pair kernel (41 vs 47.5 us)
    static constexpr int pair_mask() { return kActBf16 ? 0 : 4; }
#ifndef CRFP_STATE_FROM_EPILOGUE
#define CRFP_STATE_FROM_EPILOGUE 1   // A/B builds: 0 = the separate lrelu pass
#endif
#ifdef CRFP_LAB   // lab library: CRFP_STATE_FROM_EPILOGUE=0 at run time keeps the separate lrelu pass
    static bool state_from_epilogue() {
        static const bool on = getenv("CRFP_STATE_FROM_EPILOGUE") ? atoi(getenv("CRFP_STATE_FROM_EPILOGUE")) != 0 : CRFP_STATE_FROM_EPILOGUE != 0;
        return on;
    }
#else
    static constexpr bool state_from_epilogue() { return CRFP_STATE_FROM_EPILOGUE != 0; }
#endif
#define RUN(expr) do { if (!rc) rc = (expr); } while (0)

    // The fovea blend is a select under the mask (model/CRFP.py:1543-1544,1674-1675): what is computed only to be deselected -- the x8
    // frame stack, encoder_hr and conv_tttf away from the fovea -- is skipped tile by tile, same output bits.  CRFP_MASK_GATE=0: dense launches
    // (the skipped tiles of xin8 / enc_hr0 / x_hr then hold values again: tools/bisect_engine.py).
    static bool mask_gate_enabled() {
        static const bool on = !(getenv("CRFP_MASK_GATE") && atoi(getenv("CRFP_MASK_GATE")) == 0);   // read once
        return on;
    }
    const uint8_t* gate_ptr(int par) const { return reinterpret_cast<const uint8_t*>(ws + L.gate[par]); }

    // fp32 build: n NCHW LR frames -> Q4 quads in slot s0.. of lr_q4 (see the Model); the bf16 build reads the fp32 frames directly
    // check_div: the frames are the call's own LR input, frame f of clip f / check_div (0: one sequence): values an fp16 operand cannot
    // hold (>= 65504, inf, NaN) raise that clip's status word here -- the producer-side guard of the kernels never sees an API tensor
    const float* lr_to_q4(const float* lrs, int n, int s0, int check_div = -1) {
        if (kActBf16) return lrs;
        float* dst = adv(F(L.lr_q4), (long long)s0 * L.h * L.w * 4);
        RUN(launch_nchw_to_q4(lrs, dst, n, 3, L.h, L.w, 0, s, check_div >= 0 ? ovf() : nullptr, check_div));
        return dst;
    }
    // floats between two consecutive frames of what lr_to_q4 returns (fp32 build: a quad plane; bf16 build: the NCHW frame)
    long long lr_frame_floats() const { return kActBf16 ? 3LL * L.h * L.w : (long long)L.h * L.w * 4 * (long long)sizeof(act_t) / 4; }
    long long lr_bs(long long nchw_bs) const { return kActBf16 ? nchw_bs : (nchw_bs ? (long long)L.h * L.w * 4 : 0); }

    // FNet over nb pairs (reference model/CRFP.py:797-814); cur/prev: 3-channel frames as lr_to_q4 returns them (bs: NCHW batch stride);
    // flow_out: nb consecutive fp32 flow quads [h][w][4] (component 0 = dx, 1 = dy)
    // group > 0: pair n reads frames (and writes the flow of) item n + n / group of the cur / prev / flow_out sequences -- the B * (t - 1)
    // pairs of a lock-step batch out of its flattened [B * t] frame sequence, without the B - 1 pairs that straddle two clips (round 6)
    void fnet(int nb, const float* cur, long long cur_bs, const float* prev, long long prev_bs, float* flow_out, int group = 0) {
        const int h = L.h, w = L.w;
        if (!rc && nb > L.fnet_cap) { set_error("dsv: FNet pass of %d pairs exceeds the workspace's %d", nb, L.fnet_cap); rc = CRFP_E_WORKSPACE; }
        if (rc || nb < 1) return;
        cur_bs = lr_bs(cur_bs); prev_bs = lr_bs(prev_bs);
        Q4 a0 = q(L.fa0, 8, h, w), a1 = q(L.fa1, 8, h, w), p1 = q(L.fp1, 8, L.h1, L.w1);
        Q4 b0 = q(L.fb0, 16, L.h1, L.w1), b1 = q(L.fb1, 16, L.h1, L.w1), p2 = q(L.fp2, 16, L.h2, L.w2);
        Q4 c0 = q(L.fc0, 32, L.h2, L.w2), c1 = q(L.fc1, 32, L.h2, L.w2), p3 = q(L.fp3, 32, L.h3, L.w3);
        Q4 d0 = q(L.fd0, 64, L.h3, L.w3), d1 = q(L.fd1, 64, L.h3, L.w3), u1 = q(L.fu1, 64, 2 * L.h3, 2 * L.w3);
        Q4 e0 = q(L.fe0, 32, 2 * L.h3, 2 * L.w3), e1 = q(L.fe1, 32, 2 * L.h3, 2 * L.w3), u2 = q(L.fu2, 32, 4 * L.h3, 4 * L.w3);
        Q4 f0 = q(L.ff0, 16, 4 * L.h3, 4 * L.w3), f1 = q(L.ff1, 16, 4 * L.h3, 4 * L.w3), u3 = q(L.fu3, 16, 8 * L.h3, 8 * L.w3);
        Q4 g0 = q(L.fg0, 8, 8 * L.h3, 8 * L.w3), g1 = q(L.fg1, 1, 8 * L.h3, 8 * L.w3), fl = q(L.flow_lr, 1, h, w);
        fl.p = flow_out;
        mfma(IT_F0, nb, h, w, {{cur, cur_bs}, {prev, prev_bs}}, {{a0.p, a0.bs(), 0, 8}}, 0, 0, nullptr, 0, nullptr, 0, nullptr, 0, 0, group);
        // Round 6: a layer over a small map runs in K slices (conv_auto_ksplit: the layer's geometry decides, never the number of pairs, so one pair per
        // call and a clip's pairs compute the same bits); its partial tensors are added by the pool / resize pass behind it -- which reads the layer
        // anyway -- or by launch_ksplit_reduce.  conv(i, in, out) returns the slices (1: `out` holds the layer as before).
        KsIn ki;
        auto conv = [&](int i, const Q4& in, const Q4& out) {
            const Item& it = M.items[IT_F0 + i];
            static const bool env_strict = precision_env_strict(0);
            const int ks = (strict || env_strict || rc) ? 1 : conv_auto_ksplit(in.H, in.W, it.c.ctiles, it.c.kq);
            if (ks < 2) { mfma_q(IT_F0 + i, nb, in, out); return 1; }
            const long long pb = out.bs();   // floats of one (pair, slice): the layer's quads
            ConvArgs a = plan(IT_F0 + i, nb, in.H, in.W, {{in.p, in.bs()}}, {{F(L.fpart), pb, 0, out.nq}}, 0, 0);
            a.ksplit = ks;
            ki.part = F(L.fpart); ki.pb = pb; ki.ks = ks; ki.act = it.c.act;
            ki.ovf = ovf(); ki.ovf_div = ovf_div; ki.ovf_add = ovf_add;
            rc = launch_conv_ksplit(a, it.name, s);
            return ks;
        };
        auto reduce = [&](int ks, const Q4& out) {   // the layer as an ordinary tensor (its reader is another conv)
            if (ks > 1) RUN(launch_ksplit_reduce(ki.part, ki.pb, ks, out.p, out.bs(), nb, out.nq, out.H, out.W, ki.act, ki.ovf, ki.ovf_div, ki.ovf_add, s));
        };
        auto pool = [&](int ks, const Q4& in, const Q4& out) {
            if (ks > 1) RUN(launch_avgpool2_q4_ks(ki, out.p, out.bs(), nb, in.nq, in.H, in.W, s));
            else RUN(launch_avgpool2_q4(in.p, in.bs(), out.p, out.bs(), nb, in.nq, in.H, in.W, s));
        };
        auto resize2 = [&](int ks, const Q4& in, const Q4& out) {
            if (ks > 1) RUN(launch_upsample_q4_ks(ki, out.p, out.bs(), nb, in.nq, in.H, in.W, out.H, out.W, 0.5f, 0.5f, 1.0f, s));
            else RUN(launch_upsample_q4(in.p, in.bs(), out.p, out.bs(), nb, in.nq, in.H, in.W, out.H, out.W, 0.5f, 0.5f, 1.0f, s, 0));
        };
        pool(conv(1, a0, a1), a1, p1);
        reduce(conv(2, p1, b0), b0);
        pool(conv(3, b0, b1), b1, p2);
        reduce(conv(4, p2, c0), c0);
        pool(conv(5, c0, c1), c1, p3);
        reduce(conv(6, p3, d0), d0);
        resize2(conv(7, d0, d1), d1, u1);
        reduce(conv(8, u1, e0), e0);
        resize2(conv(9, e0, e1), e1, u2);
        reduce(conv(10, u2, f0), f0);
        resize2(conv(11, f0, f1), f1, u3);
        reduce(conv(12, u3, g0), g0);
        // tanh * 256 flow and its resize to (h, w) stay float in both builds (coordinates)
        mfma(IT_F0 + 13, nb, g0.H, g0.W, {{g0.p, g0.bs()}}, {{g1.p, g1.bs(), 0, g1.nq}}, 0, 0, nullptr, 0, nullptr, 0, nullptr, 0, 1);
        RUN(crfp::launch_upsample_q4(g1.p, g1.bs(), fl.p, fl.bs(), nb, 1, g1.H, g1.W, h, w, (float)g1.H / (float)h,
                                     (float)g1.W / (float)w, 1.0f, s, group));
    }

    // zero the P4 buffers (pads must read as 0; also gives the zero initial state)
    void reset_state() {
        if (!rc && hipMemsetAsync(ws + L.status, 0, L.A.find(L.status.off)->bytes, s) != hipSuccess) { set_error("dsv: hipMemsetAsync of the status words failed"); rc = 1; }
        for (size_t off : {L.state_hr.off, L.carry.off, L.prev2.off}) {
            const Buf* b = L.A.find(off);
            if (!rc && hipMemsetAsync(ws + off - b->guard, 0, b->bytes + b->guard, s) != hipSuccess) {
                set_error("dsv: hipMemsetAsync of recurrent state failed");
                rc = 1;
            }
        }
    }

    // n consecutive LR frames (as lr_to_q4 returns them; bs: NCHW batch stride) -> encoder_lr features in n consecutive x_lr slots from slot0
    void encode_lr(int n, const float* lrs, long long bs, long long slot0 = 0) {
        if (rc || n < 1) return;
        const long long xq = 8LL * L.h * L.w * 4;
        Q4 e0 = q(L.e_lr0, 8, L.h, L.w), x = q(L.x_lr, 8, L.h, L.w);
        e0.p = adv(e0.p, slot0 * xq); x.p = adv(x.p, slot0 * xq);
        mfma(IT_ENC_LR0, n, L.h, L.w, {{lrs, lr_bs(bs)}}, {{e0.p, e0.bs(), 0, 8}});
        mfma_q(IT_ENC_LR1, n, e0, x);
    }

    // The clip-level stages for frames [i0, i1) of every clip (Layout::flat: called once with [0, t)): encoder_lr of those frames and
    // FNet(frame i, frame i - 1) for those of them with i >= 1 (reference model/CRFP.py:1483-1508,1536).  lq: the call's LR frames as
    // lr_to_q4 returns them, [B][t] in the API tensor's order.  parts: 1 = encoder_lr, 2 = FNet
    void clip_stage(const float* lq, int i0, int i1, int parts) {
        const long long lqf = lr_frame_floats(), lr_f = 3LL * L.h * L.w, fq = (long long)L.h * L.w * 4;
        const int t = L.t, B = L.B;
        struct Restore { Runner& r; int d, a, o; ~Restore() { r.ovf_div = d; r.ovf_add = a; r.ovf_off = o; r.ovf_skip0 = 0; } } restore{*this, ovf_div, ovf_add, ovf_off};
        if (L.flat) {   // i0 == 0, i1 == t: one pass over the B * t frames / the B * (t - 1) frame pairs of the clips
            ovf_div = t;           // item n of encoder_lr's launches is frame n of the flattened sequence
            ovf_add = 0;
            if (parts & 1) encode_lr(B * t, lq, lr_f, 0);
            // Round 6: FNet's pair n is (frame n % (t - 1) + 1, frame n % (t - 1)) of clip n / (t - 1) = items n + n / (t - 1) of the cur / prev / flow
            // sequences: the B - 1 pairs that straddle two clips (11 % of FNet at B = 4, t = 7, 48 % at t = 2) are no longer computed
            ovf_div = t - 1;       // ... and pair n belongs to clip n / (t - 1)
            if ((parts & 2) && t > 1) fnet(B * (t - 1), lq + lqf, lr_f, lq, lr_f, F(L.flow_lr) + fq, t - 1);   // one-frame clips need no flow
            return;
        }
        ovf_div = 0;               // per-clip passes: every item of a launch belongs to clip b
        for (int b = 0; b < B && !rc; ++b) {
            ovf_off = b;
            const float* f0 = lq + ((long long)b * t + i0) * lqf;
            if (parts & 1) encode_lr(i1 - i0, f0, lr_f, L.slot(b, i0));
            const int j0 = i0 > 0 ? i0 : 1;
            if ((parts & 2) && i1 > j0)
                fnet(i1 - j0, lq + ((long long)b * t + j0) * lqf, lr_f, lq + ((long long)b * t + j0 - 1) * lqf, lr_f, F(L.flow_lr) + L.slot(b, j0) * fq);
        }
    }

    // ResidualBlockNoBN of level l (model/CRFP.py:449-481): y0 + conv2(relu(conv1(y0))), the 32 output channels going to the
    // propagated features (24) and the carried ones (8).  bf16 build: one launch, conv1's output stays in LDS.
    // CRFP_DSV_CRA (M.cra): the block's 32 channels stay together (cra.y); the level's fusion conv and the masked blend then write the
    // same two destinations (model/CRFP.py:2533-2535): [prop | carry] = mk2 * conv_tttf_l(cat(y, fovea level l)) + (1 - mk2) * y
    void res_block(int l, float* prop_next, float* carry_l, int H2, int W2, int par = 0, const uint8_t* mk = nullptr, long long mk_b = 0) {
        const int B = L.B;
        const long long P2q = (long long)H2 * W2 * 4, P2qp = (long long)(H2 + 1) * (W2 + 1) * 4;
        std::vector<DstBind> dsts = {{prop_next, 6 * P2q, 0, 6}, {carry_l, 6 * P2qp, 6, 8, 1}};
        if (M.cra) dsts = {{F(L.c_y), L.c_y.bs, 0, 8}};
        if (M.abl()) dsts = {{prop_next, 8 * P2q, 0, 8}};   // CRFP_simple / CRFP: all 32 channels are the next level's features
#ifdef CRFP_ACT_BF16
        if (pair_convs())
            mfma_pair(it_lvl(l, L_RB1), it_lvl(l, L_RB2), "conv_mfma_pair:res.conv1_conv2_add", B, H2, W2, {{F(L.y0), L.y0.bs}}, dsts, F(L.y0), L.y0.bs);
        else
#endif
        {
            mfma(it_lvl(l, L_RB1), B, H2, W2, {{F(L.y0), L.y0.bs}}, {{F(L.y1), L.y1.bs, 0, 8}});
            mfma(it_lvl(l, L_RB2), B, H2, W2, {{F(L.y1), L.y1.bs}}, dsts, 0, 0, F(L.y0), L.y0.bs);
        }
        if (!M.cra) return;
        const Ten& lv = L.c_lv[par][l];
        mfma(IT_C_T0 + l, B, H2, W2, {{F(L.c_y), L.c_y.bs}, {F(lv), lv.bs}}, {{F(L.c_f), L.c_f.bs, 0, 8}});
        RUN(launch_cra_blend(F(L.c_y), L.c_y.bs, F(L.c_f), L.c_f.bs, mk, mk_b, prop_next, 6 * P2q, carry_l, 6 * P2qp, B, H2, W2, s));
    }

    // What one frame step reads and writes outside the workspace: frame i of each of the B clips, so every batch stride is
    // "one clip" of the API tensor (t frames); the clip-level stores hand over their slot of frame i likewise.
    struct FrameIO {
        const float* lr = nullptr; long long lr_b = 0;       // [3][h][w] fp32
        const float* fv = nullptr; long long fv_b = 0;       // [3][8h][8w] fp32
        const uint8_t* mk = nullptr; long long mk_b = 0;     // [8h][8w] u8
        float* out = nullptr; long long out_b = 0;           // [3|1][8h][8w] fp32
        const float* flow_lr = nullptr; long long flow_b = 0;   // FNet's flow quads of this frame (null: first frame)
        const float* x_lr = nullptr; long long x_b = 0;         // encoder_lr features of this frame
    };
    FrameIO frame_io(int i, const float* lrs, const float* fvs, const uint8_t* mks, float* out, int y_only) const {
        const long long lr_f = 3LL * L.h * L.w, hr_px = 64LL * L.h * L.w, xq = 8LL * L.h * L.w * 4, fq = (long long)L.h * L.w * 4;
        const int co = y_only ? 1 : 3, t = L.t;
        FrameIO f;
        f.lr = lrs + i * lr_f; f.lr_b = t * lr_f;
        f.fv = fvs + i * 3 * hr_px; f.fv_b = t * 3 * hr_px;
        f.mk = mks + i * hr_px; f.mk_b = t * hr_px;
        f.out = out + (long long)i * co * hr_px; f.out_b = (long long)t * co * hr_px;
        f.flow_lr = i > 0 ? F(L.flow_lr) + L.slot(0, i) * fq : nullptr; f.flow_b = L.TC * fq;
        f.x_lr = adv(F(L.x_lr), L.slot(0, i) * xq); f.x_b = L.TC * xq;
        return f;
    }

    // state-independent part of a frame (reference model/CRFP.py:1538-1547,1560,1565-1566): buffer set `par`
    // before_ups: event this stream waits for right before the upsample conv, the only consumer of x_lr here (clip
    // schedule: the fovea blend and encoder_hr of frame 0 then run beside encoder_lr instead of behind it)
    // parts: 1 = fovea blend + encoder_hr + upsample conv (need no flow), 2 = the two flow up-samplings (need FNet)
    void frame_pre(int par, bool first, const FrameIO& io, hipEvent_t before_ups = nullptr, int parts = 3) {
        const int h = L.h, w = L.w, H2 = 2 * h, W2 = 2 * w, H8 = 8 * h, W8 = 8 * w, B = L.B;
        const long long P8q = (long long)H8 * W8 * 4;
        if (parts & 1) {
            const Ten& xin = L.xin8[par];
            const bool gated = mask_gate_enabled();
            if (gated) RUN(launch_mask_gate(io.mk, io.mk_b, const_cast<uint8_t*>(gate_ptr(par)), L.gate_b, B, H8, W8, s));
            // conv_tttf works on the tiles that hold mask pixels and reads x_hr one pixel into their neighbours: encoder_hr's second conv runs on
            // that ring of tiles too, its first conv on two rings, the frame stack on three -- no launch reads a tile nobody wrote.
            // CRFP_DSV_CRA: slice1's output also feeds the three 2x levels, whose reach is wider: only conv_lv3 is gated there.
            const int gp = (gated && !M.cra && !(pair_mask() & 1)) ? par : -1;
            RUN(launch_hr_prep(io.lr, io.fv, io.mk, F(xin), h, w, s, B, io.lr_b, io.fv_b, io.mk_b, xin.bs, gp >= 0 ? gate_ptr(par) : nullptr, L.gate_b));
            // CRFP_DSV_CRA: slice1's output feeds conv_lv3 (-> x_hr, what conv_tttf blends in) and the three 2x levels
            const Ten& s1 = M.cra ? L.c_s1 : L.x_hr[par];
            if (pair_mask() & 1)
                narrow_pair(IT_EH0, IT_EH1, "conv_narrow_pair:enc_hr", H8, W8, {{F(xin), xin.bs}, {adv(F(xin), P8q), xin.bs}}, {F(s1), s1.bs});
            else {
                narrow(IT_EH0, H8, W8, {{F(xin), xin.bs}, {adv(F(xin), P8q), xin.bs}}, {F(L.eh[par]), L.eh[par].bs}, NB(), NB(), nullptr, 0, 0, 0, NB(), gp, 2);
                narrow(IT_EH1, H8, W8, {{F(L.eh[par]), L.eh[par].bs}}, {F(s1), s1.bs}, NB(), NB(), nullptr, 0, 0, 0, NB(), gp, 1);
            }
            if (M.cra) {
                narrow(IT_C_LV3, H8, W8, {{F(s1), s1.bs}}, {F(L.x_hr[par]), L.x_hr[par].bs}, NB(), NB(), nullptr, 0, 0, 0, NB(), par, 1);
                const Q4 a = q(L.c_a, 4, H2, W2), b = q(L.c_b, 4, H2, W2);
                auto lv = [&](int k) { return q(L.c_lv[par][k], 4, H2, W2); };
                mfma(IT_C_S2A, B, H2, W2, {{F(s1), s1.bs, 0}}, {{a.p, a.bs(), 0, 4}});
                mfma_q(IT_C_S2B, B, a, b);
                mfma_q(IT_C_LV2, B, b, lv(2));
                mfma_q(IT_C_S3A, B, b, a);
                mfma_q(IT_C_S3B, B, a, b);
                mfma_q(IT_C_LV1, B, b, lv(1));
                mfma_q(IT_C_S4A, B, b, a);
                mfma_q(IT_C_S4B, B, a, b);
                mfma_q(IT_C_LV0, B, b, lv(0));
            }
            if (before_ups && !rc && hipStreamWaitEvent(s, before_ups, 0) != hipSuccess) { set_error("dsv: hipStreamWaitEvent failed"); rc = 1; }
            mfma(IT_UPS, B, h, w, {{io.x_lr, io.x_b}}, {{F(L.prop0[par]), L.prop0[par].bs, 0, L.pq}}, H2, W2);
        }
        if (!first && (parts & 2)) {
            RUN(crfp::launch_upflow(io.flow_lr, io.flow_b, F(L.flow2[par]), L.flow2[par].bs, B, h, w, 2, s));
            RUN(crfp::launch_upflow(io.flow_lr, io.flow_b, F(L.flow8[par]), L.flow8[par].bs, B, h, w, 8, s));
        }
    }

    // downsample(state) -> prev2 (model/CRFP.py:1569)
    void downsample_state() {
        mfma(IT_DOWN, L.B, 2 * L.h, 2 * L.w, {{F(L.state_hr), L.state_hr.bs, 1}}, {{F(L.prev2), L.prev2.bs, 0, 8, 1}});
    }

    // recurrent part of a frame (reference model/CRFP.py:1562-1684), all B clips of the call in lock-step: one launch per layer
    void frame(int par, bool first, const FrameIO& io, const uint8_t* fg = nullptr) {
        const int h = L.h, w = L.w, H2 = 2 * h, W2 = 2 * w, H8 = 8 * h, W8 = 8 * w, B = L.B;
        const long long P2q = (long long)H2 * W2 * 4;
        const long long P2qp = (long long)(H2 + 1) * (W2 + 1) * 4;   // padded (P4) plane at 2x resolution
        const long long bs8 = 8 * P2q;
        const long long bs6 = L.pq * P2q;   // batch stride of the features a level passes on (prop*): 6 quads, CRFP_simple / CRFP 8
        const bool abl = M.abl(), dense = M.dense();
        float* prop = F(L.prop0[par]);
        float* prop_next = F(L.prop_a);
        float* prop_other = F(L.prop_b);
        float* carry = F(L.carry);
        auto nb = [&](const Ten& tn) { return NB(F(tn), tn.bs); };
        std::vector<NB> r3_srcs;   // inputs of forward_resblocks_3.main.0
        // forward_resblocks_3 as one launch; the bf16 chain kernel takes at most two input quads (CRFP's three: three launches there)
        const bool r3_chain = (chain_mask() & 2) && !(kActBf16 && dense && !first);
        if (!first) {
            float* flow2 = F(L.flow2[par]);
            float* flow8 = F(L.flow8[par]);
            const long long f2b = L.flow2[par].bs, f8b = L.flow8[par].bs;
            if (!down_done) downsample_state();
            if (abl) {
                // CRFP_simple / CRFP (model/CRFP.py:1021-1026): the state is warped at 8x FIRST and both versions are brought to 2x by `downsample`
                RUN(launch_flow_warp_q4(F(L.state_hr), L.state_hr.bs, flow8, f8b, F(L.prevhrw), L.prevhrw.bs, B, 1, H8, W8, 0, 1, s));
                mfma(IT_DOWN, B, H2, W2, {{F(L.prevhrw), L.prevhrw.bs, 0}}, {{F(L.prev2w), L.prev2w.bs, 0, 8}});
            } else {
            WarpDualStrides wb;
            wb.xa = L.prev2.bs; wb.xb = L.carry.bs; wb.flow = f2b; wb.outa = L.prev2w.bs; wb.outb = L.carryw.bs;
            RUN(launch_flow_warp_p4_dual_8_6(F(L.prev2), carry, flow2, F(L.prev2w), F(L.carryw), H2, W2, s, B, wb));
            RUN(launch_flow_warp_q4(F(L.state_hr), L.state_hr.bs, flow8, f8b, F(L.prevhrw), L.prevhrw.bs, B, 1, H8, W8, 0, 1, s));
            }
            const float* offprev = nullptr;
            if (fg) RUN(crfp::launch_fg_prep(fg, F(L.fg2), H8, W8, s));
            for (int l = 0; l < 3; ++l) {
                const float* cw = adv(F(L.carryw), 2 * l * P2q);
                float* f = F(L.offfeat[l]);
                const bool s3 = M.use_s3;   // f holds the SRC_S3 image (same size) instead of fp32 Q4
                // dcn_block.0's inputs: [features | carried] | warped previous state | flow; CRFP_simple / CRFP: features (32) | warped | flow
                const std::vector<SrcBind> db0 = abl ? std::vector<SrcBind>{{prop, bs6}, {F(L.prev2w), bs8}, {flow2, f2b}, {nullptr, 0}}
                                                     : std::vector<SrcBind>{{prop, bs6}, {cw, bs6}, {F(L.prev2w), bs8}, {flow2, f2b}, {nullptr, 0}};
#ifdef CRFP_ACT_BF16
                if (pair_convs())   // dcn_block.0 -> .2 in one launch, the 32-channel tensor between them stays in LDS
                    mfma_pair(it_lvl(l, L_DB0), it_lvl(l, L_DB1), "conv_mfma_pair:dcn.block0_block2", B, H2, W2, db0, {{l == 0 ? f : F(L.fb), bs8, 0, 8}});
                else
#endif
                {
                mfma(it_lvl(l, L_DB0), B, H2, W2, db0, {{F(L.fa), bs8, 0, 8}});
                if (l == 0) {
                    if (s3) mfma(it_lvl(l, L_DB1), B, H2, W2, {{F(L.fa), bs8}}, {}, 0, 0, nullptr, 0, nullptr, 0, f, bs8);
                    else mfma(it_lvl(l, L_DB1), B, H2, W2, {{F(L.fa), bs8}}, {{f, bs8, 0, 8}});
                } else
                    mfma(it_lvl(l, L_DB1), B, H2, W2, {{F(L.fa), bs8}}, {{F(L.fb), bs8, 0, 8}});
                }
                if (l > 0) {
                    if (s3) mfma(it_lvl(l, L_FUSE), B, H2, W2, {{F(L.fb), bs8}, {offprev, bs8}}, {}, 0, 0, nullptr, 0, nullptr, 0, f, bs8);
                    else mfma(it_lvl(l, L_FUSE), B, H2, W2, {{F(L.fb), bs8}, {offprev, bs8}}, {{f, bs8, 0, 8}});
                }
                const Item& dw = M.items[it_lvl(l, L_DCNW)];
                const bool f16 = !strict && dcn_g8_use_f16();
                if ((s3 || kActBf16) && f16 && dcn_fused_enabled()) {   // offset / mask head + dcn_g8 in one launch: offsets stay in registers
                    const Item& om = M.items[it_lvl(l, L_OMF)];
                    DcnFuseArgs fa;
                    memset(&fa, 0, sizeof(fa));
                    fa.feat = f; fa.feat_b = bs8; fa.flow = flow2; fa.flow_b = f2b;
                    fa.wconv = (const char*)(packed + om.off_s) + conv_split16_offset_bytes(om.c); fa.bconv = packed + om.off_b;
                    fa.x = F(L.prev2); fa.xb = L.prev2.bs;
                    fa.wdcn = packed + dw.off_w + 36 * 2 * 32 * 4; fa.bdcn = packed + dw.off_b;
                    fa.out = F(L.aligned); fa.ob = bs8; fa.N = B; fa.H = H2; fa.W = W2; fa.ovf = ovf(); fa.ovf_div = ovf_div;
                    RUN(launch_dcn_fused(fa, s));
                } else {
                    mfma(it_lvl(l, L_OM), B, H2, W2, {{f, bs8}}, {{F(L.offmask), L.offmask.bs, 0, 54}}, 0, 0, nullptr, 0, flow2, f2b);
                    RUN(launch_dcn_g8(F(L.prev2), L.prev2.bs, F(L.offmask), L.offmask.bs, packed + dw.off_w + (f16 ? 36 * 2 * 32 * 4 : 0), packed + dw.off_b,
                                      F(L.aligned), bs8, B, H2, W2, s, f16, ovf(), ovf_div));
                }
                if (fg && l > 0) {  // model/CRFP_test.py:2361,2375: resblock input * fg (x0.25) for levels 1, 2   (one sequence per workspace: B == 1)
                    RUN(launch_scale_q4(prop, 0, F(L.sc_prop), 6, H2, W2, F(L.fg2), nullptr, s));
                    RUN(launch_scale_q4(cw, 0, F(L.sc_cw), 2, H2, W2, F(L.fg2), nullptr, s));
                    RUN(launch_scale_q4(F(L.aligned), 0, F(L.sc_al), 8, H2, W2, F(L.fg2), nullptr, s));
                    mfma(it_lvl(l, L_RB0), 1, H2, W2, {{F(L.sc_prop), 0}, {F(L.sc_cw), 0}, {F(L.sc_al), 0}}, {{F(L.y0), 0, 0, 8}});
                } else if (abl) {   // cat(cur, aligned) (:1034); CRFP: + the warped previous state (:1311)
                    std::vector<SrcBind> rb = {{prop, bs6}, {F(L.aligned), bs8}};
                    if (dense) rb.push_back({F(L.prev2w), bs8});
                    mfma(it_lvl(l, L_RB0), B, H2, W2, rb, {{F(L.y0), bs8, 0, 8}});
                } else
                    mfma(it_lvl(l, L_RB0), B, H2, W2, {{prop, bs6}, {cw, bs6}, {F(L.aligned), bs8}}, {{F(L.y0), bs8, 0, 8}});
                res_block(l, prop_next, adv(carry, 2 * l * P2qp), H2, W2, par, io.mk, io.mk_b);
                prop = prop_next;
                std::swap(prop_next, prop_other);
                offprev = f;
            }
            // the two 32 -> 64 pixel-shuffle convs behind level 2 are independent: one launch (round 6; fp32 build)
            mfma_dual(IT_UPP, {{prop, bs6}}, {{F(L.up), L.up.bs, 0, 1}}, IT_POFF, {{offprev, bs8}}, {{F(L.poff), L.poff.bs, 0, 1}}, B, H2, W2, H8, W8,
                      "conv_mfma:upsample_post_ps4+dcn3.preoffset_ps4");
            const NB fl8(flow8, f8b);
            if (chain_mask() & 1)   // dcn_block.0 -> .2 -> conv_fuse in one launch (round 6)
                narrow_chain(IT_D3B0, IT_D3B1, IT_D3FUSE, "conv_narrow_chain:dcn3.block0_block2_fuse", H8, W8, {nb(L.up), nb(L.prevhrw), fl8}, nb(L.poff),
                             nb(L.g2), false);
            else {
            if (pair_mask() & 8)
                narrow_pair(IT_D3B0, IT_D3B1, "conv_narrow_pair:dcn3.block", H8, W8, {nb(L.up), nb(L.prevhrw), fl8}, nb(L.g1));
            else {
                narrow(IT_D3B0, H8, W8, {nb(L.up), nb(L.prevhrw), fl8}, nb(L.g0));
                narrow(IT_D3B1, H8, W8, {nb(L.g0)}, nb(L.g1));
            }
            if (pair_mask() & 2)
                narrow_pair(IT_D3FUSE, IT_D3OM, "conv_narrow_pair:dcn3.fuse_offmask", H8, W8, {nb(L.g1), nb(L.poff)}, nb(L.om3), NB(), fl8);
            else
                narrow(IT_D3FUSE, H8, W8, {nb(L.g1), nb(L.poff)}, nb(L.g2));
            }
            const Item& d3 = M.items[IT_D3W];
            // dcn_3 with its offset / mask conv inside (gather.hip dcn3_kernel<true>): fp32 build 94.0 vs 73.7 + 37.3 us same-box, bit-identical;
            // the bf16 build keeps two kernels (its stand-alone conv runs on the bf16 MFMA: 84.8 vs 59.8 + 28.4 us, not worth the changed bits)
            if (!kActBf16 && !(pair_mask() & 2) && dcn_fused_enabled()) {
                const Item& om = M.items[IT_D3OM];
                RUN(launch_dcn3_fused(F(L.state_hr), L.state_hr.bs, F(L.g2), L.g2.bs, flow8, packed + om.off_w, packed + om.off_b, packed + d3.off_w,
                                      packed + d3.off_b, F(L.al3), L.al3.bs, B, H8, W8, s, f8b));
            } else {
                if (!(pair_mask() & 2)) narrow(IT_D3OM, H8, W8, {nb(L.g2)}, nb(L.om3), NB(), fl8);
                RUN(launch_dcn3(F(L.state_hr), L.state_hr.bs, F(L.om3), L.om3.bs, packed + d3.off_w, packed + d3.off_b, F(L.al3), L.al3.bs, B, H8, W8, s));
            }
            if (fg) {           // model/CRFP_test.py:2389
                RUN(launch_scale_q4(F(L.up), 0, F(L.sc_up), 1, H8, W8, nullptr, fg, s));
                RUN(launch_scale_q4(F(L.al3), 0, F(L.sc_al3), 1, H8, W8, nullptr, fg, s));
                r3_srcs = {nb(L.sc_up), nb(L.sc_al3)};
            } else
                r3_srcs = {nb(L.up), nb(L.al3)};
            if (dense) r3_srcs.push_back(nb(L.prevhrw));   // CRFP (:1316): + the warped previous state
            if (!r3_chain) narrow(IT_R3_0, H8, W8, r3_srcs, nb(L.z0));
        } else {
            for (int l = 0; l < 3; ++l) {
                mfma(it_lvl(l, L_RB0F), B, H2, W2, {{prop, bs6}, {nullptr, 0}}, {{F(L.y0), bs8, 0, 8}});
                res_block(l, prop_next, adv(carry, 2 * l * P2qp), H2, W2, par, io.mk, io.mk_b);
                prop = prop_next;
                std::swap(prop_next, prop_other);
            }
            mfma(IT_UPP, B, H2, W2, {{prop, bs6}}, {{F(L.up), L.up.bs, 0, 1}}, H8, W8);
            r3_srcs = {nb(L.up)};
            if (!r3_chain) narrow(IT_R3_0F, H8, W8, r3_srcs, nb(L.z0));
        }
        // gated: the state is lrelu(feat) wherever the mask is clear -- written by the epilogue of forward_resblocks_3's last conv (round 6: its second
        // destination; it was a separate streaming pass over the 8x map, 21 us per frame) -- and the blend kernel rewrites the tiles with mask pixels
        const NB st2 = mask_gate_enabled() && state_from_epilogue() ? nb(L.state_hr) : NB();
        if (r3_chain)   // main.0 -> conv1 -> conv2 (+ x) in one launch (round 6)
            narrow_chain(first ? IT_R3_0F : IT_R3_0, IT_R3_1, IT_R3_2, "conv_narrow_chain:res3.main0_conv1_conv2_add", H8, W8, r3_srcs, NB(), nb(L.feat), true, st2);
        else if (pair_mask() & 4)
            narrow_pair(IT_R3_1, IT_R3_2, "conv_narrow_pair:res3.conv1_conv2_add", H8, W8, {nb(L.z0)}, nb(L.feat), nb(L.z0), NB(), st2);
        else {
            narrow(IT_R3_1, H8, W8, {nb(L.z0)}, nb(L.z1));
            narrow(IT_R3_2, H8, W8, {nb(L.z1)}, nb(L.feat), nb(L.z0), NB(), nullptr, 0, 0, 0, NB(), -1, 0, st2);
        }
        if (mask_gate_enabled() && !st2.p) RUN(launch_lrelu_q4_to_p4(F(L.feat), L.feat.bs, F(L.state_hr), L.state_hr.bs, B, H8, W8, ovf(), ovf_div, s));
        narrow(IT_TTTF, H8, W8, {nb(L.feat), nb(L.x_hr[par])}, nb(L.state_hr), NB(), NB(), io.mk, io.mk_b, 0, 1, NB(), par, 0);
        // output head: conv_last(state) + x8 bilinear LR;
        // the x8 bilinear base is recomputed from the LR frame in both builds (fp32: 39.0 vs 41.3 us against reading the quad hr_prep
        // stored -- identical values, 59 MB less traffic; bf16: a stored bf16 base would cost 2^-9 of the output range)
        narrow(IT_LAST, H8, W8, {nb(L.state_hr)}, NB(io.out, io.out_b), NB(), NB(), nullptr, 0, 1, 0, NB(io.lr, io.lr_b));
    }
};

}  // namespace CRFP_NS

using namespace CRFP_NS;

#ifdef CRFP_ACT_BF16
// dcn_3's raw OIHW weights: rounded to bf16 values on the way into the packed buffer (every conv weight of the bf16 engine is one)
__global__ void round_bf16_copy_kernel(const float* __restrict__ src, float* __restrict__ dst, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = (float)(__bf16)src[i];
}
namespace crfp_bf16 { void shutdown_side_streams() { destroy_all_side_streams(); } }
#else
namespace crfp_bf16 { void shutdown_side_streams(); }
namespace crfp { void rt_shutdown_streams(); }   // engine_rt.hip
#endif

extern "C" {

#ifndef CRFP_ACT_BF16
const char* crfp_dsv_param_name(int index) {
    static thread_local std::string s;
    if (index < 0 || index >= CRFP_DSV_NUM_PARAMS) return nullptr;
    s = std::string(kConvs[index / 2].stem) + (index % 2 ? ".bias" : ".weight");
    return s.c_str();
}

int crfp_dsv_param_numel(int index, int y_only) {
    if (index < 0 || index >= CRFP_DSV_NUM_PARAMS) return CRFP_E_BADARG;
    const int ci = index / 2, co = conv_cout(ci, y_only);
    return index % 2 ? co : co * kConvs[ci].cin * 9;
}

const char* crfp_cra_param_name(int index) {
    static thread_local std::string s;
    if (index < 0 || index >= CRFP_CRA_NUM_PARAMS) return nullptr;
    s = std::string(kConvs[kCraOrder[index / 2]].stem) + (index % 2 ? ".bias" : ".weight");
    return s.c_str();
}

int crfp_cra_param_numel(int index, int y_only) {
    if (index < 0 || index >= CRFP_CRA_NUM_PARAMS) return CRFP_E_BADARG;
    const int ci = kCraOrder[index / 2], co = conv_cout(ci, y_only);
    return index % 2 ? co : co * kConvs[ci].cin * 9;
}

// CRFP_simple / CRFP: the state_dict keys and their order are CRFP_DSV's (crfp_dsv_param_name); four weights have other shapes
static int abl_param_numel(int wiring, int index, int y_only) {
    if (index < 0 || index >= CRFP_DSV_NUM_PARAMS) return CRFP_E_BADARG;
    const int ci = index / 2, co = conv_cout(ci, y_only, wiring);
    return index % 2 ? co : co * conv_cin(ci, wiring) * 9;
}
int crfp_simple_param_numel(int index, int y_only) { return abl_param_numel(W_SIMPLE, index, y_only); }
int crfp_dense_param_numel(int index, int y_only) { return abl_param_numel(W_DENSE, index, y_only); }

#endif  // parameter tables: exported once

}  // extern "C"

// params: 2 * kNumCraConvs pointers in kConvs order (the CRFP_DSV wiring reads the first 2 * kNumDsvConvs)
static int pack_weights_impl(const Model& M, const float* const* params, int y_only, void* packed, size_t packed_bytes, void* stream) {
    if (packed_bytes < M.total_floats * sizeof(float)) { set_error("pack_weights: packed buffer too small"); return CRFP_E_WORKSPACE; }
    hipStream_t s = (hipStream_t)stream;
    float* pk = (float*)packed;
    for (int i = 0; i < IT_COUNT; ++i) {
        const Item& it = M.items[i];
        if (it.w1 < 0) continue;
        const float* w = params[2 * it.w1];
        const float* b = params[2 * it.w1 + 1];
        const float* w2 = it.w2 >= 0 ? params[2 * it.w2] : nullptr;
        const float* b2 = it.w2 >= 0 ? params[2 * it.w2 + 1] : nullptr;
        const int split = conv_cout(it.w1, y_only, M.wiring);
        int rc = 0;
        switch (it.type) {
            case T_MFMA:
                rc = launch_conv_pack(it.c, w, b, w2, b2, split, pk + it.off_w, pk + it.off_b, s);
                if (!rc) rc = launch_conv_pack_split(it.c, w, w2, split, pk + it.off_s, s);
                break;
            case T_NARROW: rc = launch_narrow_pack(it.nw, w, b, w2, b2, split, pk + it.off_w, pk + it.off_b, s); break;
            case T_DCN8:
                if (!kActBf16) rc = launch_dcn_g8_pack(w, pk + it.off_w, s, false);   // fp32 MFMA image (strict mode, fp32 build)
                if (!rc) rc = launch_dcn_g8_pack(w, pk + it.off_w + 36 * 2 * 32 * 4, s, true);
                if (!rc && hipMemcpyAsync(pk + it.off_b, b, 32 * sizeof(float), hipMemcpyDeviceToDevice, s) != hipSuccess) rc = 1;
                break;
            default:
#ifdef CRFP_ACT_BF16
                round_bf16_copy_kernel<<<((int)it.n_w + 255) / 256, 256, 0, s>>>(w, pk + it.off_w, (int)it.n_w);
                if (hipGetLastError() != hipSuccess) rc = 1;
#else
                if (hipMemcpyAsync(pk + it.off_w, w, it.n_w * sizeof(float), hipMemcpyDeviceToDevice, s) != hipSuccess) rc = 1;
#endif
                if (!rc && hipMemcpyAsync(pk + it.off_b, b, it.n_b * sizeof(float), hipMemcpyDeviceToDevice, s) != hipSuccess) rc = 1;
        }
        if (rc) return rc;
    }
    return 0;
}

extern "C" {

size_t CRFP_API(crfp_dsv_packed_weight_bytes)(int y_only) { return model_for(y_only).total_floats * sizeof(float); }
size_t CRFP_API(crfp_cra_packed_weight_bytes)(int y_only) { return model_for(y_only, false, W_CRA).total_floats * sizeof(float); }
size_t CRFP_API(crfp_simple_packed_weight_bytes)(int y_only) { return model_for(y_only, false, W_SIMPLE).total_floats * sizeof(float); }
size_t CRFP_API(crfp_dense_packed_weight_bytes)(int y_only) { return model_for(y_only, false, W_DENSE).total_floats * sizeof(float); }

int CRFP_API(crfp_dsv_pack_weights)(const float* const* params, int y_only, void* packed, size_t packed_bytes, void* stream) {
    if (!params || !packed) { set_error("pack_weights: null argument"); return CRFP_E_BADARG; }
    for (int i = 0; i < CRFP_DSV_NUM_PARAMS; ++i)
        if (!params[i]) { set_error("pack_weights: parameter %d (%s) is null", i, crfp_dsv_param_name(i)); return CRFP_E_BADARG; }
    return pack_weights_impl(model_for(y_only), params, y_only, packed, packed_bytes, stream);
}

// params: CRFP_CRA_NUM_PARAMS device pointers in the order of the reference's CRFP_DSV_CRA state_dict (crfp_cra_param_name)
int CRFP_API(crfp_cra_pack_weights)(const float* const* params, int y_only, void* packed, size_t packed_bytes, void* stream) {
    if (!params || !packed) { set_error("pack_weights: null argument"); return CRFP_E_BADARG; }
    const float* eng[2 * kNumCraConvs];
    for (int j = 0; j < kNumCraConvs; ++j)
        for (int k = 0; k < 2; ++k) {
            if (!params[2 * j + k]) { set_error("pack_weights: parameter %d (%s) is null", 2 * j + k, crfp_cra_param_name(2 * j + k)); return CRFP_E_BADARG; }
            eng[2 * kCraOrder[j] + k] = params[2 * j + k];
        }
    return pack_weights_impl(model_for(y_only, false, W_CRA), eng, y_only, packed, packed_bytes, stream);
}

// params: CRFP_DSV_NUM_PARAMS device pointers in the order of the reference's CRFP_simple / CRFP state_dict (= crfp_dsv_param_name)
static int abl_pack_weights(int wiring, const float* const* params, int y_only, void* packed, size_t packed_bytes, void* stream) {
    if (!params || !packed) { set_error("pack_weights: null argument"); return CRFP_E_BADARG; }
    for (int i = 0; i < CRFP_DSV_NUM_PARAMS; ++i)
        if (!params[i]) { set_error("pack_weights: parameter %d (%s) is null", i, crfp_dsv_param_name(i)); return CRFP_E_BADARG; }
    return pack_weights_impl(model_for(y_only, false, wiring), params, y_only, packed, packed_bytes, stream);
}
int CRFP_API(crfp_simple_pack_weights)(const float* const* params, int y_only, void* packed, size_t packed_bytes, void* stream) {
    return abl_pack_weights(W_SIMPLE, params, y_only, packed, packed_bytes, stream);
}
int CRFP_API(crfp_dense_pack_weights)(const float* const* params, int y_only, void* packed, size_t packed_bytes, void* stream) {
    return abl_pack_weights(W_DENSE, params, y_only, packed, packed_bytes, stream);
}

static bool dims_ok(int n, int t, int h, int w) { return n >= 1 && t >= 1 && h >= 8 && w >= 8 && (long long)n * t <= (1 << 20); }

size_t CRFP_API(crfp_dsv_batch_workspace_bytes)(int n, int t, int h, int w) {
    if (!dims_ok(n, t, h, w)) return 0;
    return Layout(n, t, h, w).bytes();
}
size_t CRFP_API(crfp_dsv_workspace_bytes)(int t, int h, int w) { return CRFP_API(crfp_dsv_batch_workspace_bytes)(1, t, h, w); }
size_t CRFP_API(crfp_cra_batch_workspace_bytes)(int n, int t, int h, int w) {
    if (!dims_ok(n, t, h, w)) return 0;
    return Layout(n, t, h, w, W_CRA).bytes();
}
size_t CRFP_API(crfp_simple_batch_workspace_bytes)(int n, int t, int h, int w) { return dims_ok(n, t, h, w) ? Layout(n, t, h, w, W_SIMPLE).bytes() : 0; }
size_t CRFP_API(crfp_dense_batch_workspace_bytes)(int n, int t, int h, int w) { return dims_ok(n, t, h, w) ? Layout(n, t, h, w, W_DENSE).bytes() : 0; }

static int check_common(const void* packed, int n, int t, int h, int w, void* ws, size_t ws_bytes, const Layout& L) {
    if (!packed || !ws) { set_error("dsv: null packed weights or workspace"); return CRFP_E_BADARG; }
    if (!dims_ok(n, t, h, w)) { set_error("dsv: need n,t>=1, h,w>=8 (got %d,%d,%d,%d)", n, t, h, w); return CRFP_E_BADARG; }
    if (ws_bytes < L.bytes()) { set_error("dsv: workspace %zu < required %zu bytes", ws_bytes, L.bytes()); return CRFP_E_WORKSPACE; }
    return 0;
}

size_t CRFP_API(crfp_dsv_batch_status_offset)(int n, int t, int h, int w) {
    if (!dims_ok(n, t, h, w)) return 0;
    return Layout(n, t, h, w).status;
}
size_t CRFP_API(crfp_dsv_status_offset)(int t, int h, int w) { return CRFP_API(crfp_dsv_batch_status_offset)(1, t, h, w); }
size_t CRFP_API(crfp_cra_batch_status_offset)(int n, int t, int h, int w) {
    if (!dims_ok(n, t, h, w)) return 0;
    return Layout(n, t, h, w, W_CRA).status;
}
size_t CRFP_API(crfp_simple_batch_status_offset)(int n, int t, int h, int w) { return dims_ok(n, t, h, w) ? (size_t)Layout(n, t, h, w, W_SIMPLE).status : 0; }
size_t CRFP_API(crfp_dense_batch_status_offset)(int n, int t, int h, int w) { return dims_ok(n, t, h, w) ? (size_t)Layout(n, t, h, w, W_DENSE).status : 0; }

}  // extern "C"

// n clips in lock-step through the recurrent chain (reference model/CRFP.py:1510-1535: every op of forward() carries the batch
// axis n): ONE launch per layer and frame step over all n clips, so a 360 x 640 map that is a single round of workgroups for one
// clip becomes n rounds whose load / MFMA / store phases overlap.  Per clip the arithmetic is that of a one-clip call, bit for bit.
static int forward_batch_impl(int wiring, const void* packed, int flags, const float* lrs, const float* fvs, const uint8_t* mks,
                              float* out, int n, int t, int h, int w, void* workspace, size_t workspace_bytes, void* stream) {
    const int y_only = flags & CRFP_DSV_Y_ONLY;
    if (kActBf16 && (flags & CRFP_DSV_STRICT_F32)) { set_error("dsv (bf16 storage): CRFP_DSV_STRICT_F32 belongs to the fp32 entry points"); return CRFP_E_UNSUPPORTED; }
    if (!dims_ok(n, t, h, w)) { set_error("dsv: need n,t>=1, h,w>=8 (got %d,%d,%d,%d)", n, t, h, w); return CRFP_E_BADARG; }
    Layout L(n, t, h, w, wiring);
    int rc = check_common(packed, n, t, h, w, workspace, workspace_bytes, L);
    if (rc) return rc;
    if (!lrs || !fvs || !mks || !out) { set_error("dsv_forward_clip: null tensor"); return CRFP_E_BADARG; }
    Runner R{model_for(y_only, flags & CRFP_DSV_STRICT_F32, wiring), (const float*)packed, (char*)workspace, L, (hipStream_t)stream};
    R.strict = (flags & CRFP_DSV_STRICT_F32) ? 1 : 0;
    const int TC = L.TC;
    SideStream* ssp = side_stream_enabled() && !prof_enabled() && !(flags & CRFP_DSV_SINGLE_STREAM) ? side_stream() : nullptr;
    hipStream_t main_s = (hipStream_t)stream;
    if (!ssp) {
        // single-stream schedule (also used while per-kernel timing is on: events bracket launches per stream)
        R.reset_state();
        const float* lq = R.lr_to_q4(lrs, n * t, 0, t);
        for (int i = 0; i < t && !R.rc; ++i) {
            if (i % TC == 0) {   // flat: once, FNet first (the order of the one-clip engine of rounds 1-3)
                const int i1 = i + TC < t ? i + TC : t;
                R.clip_stage(lq, i, i1, 2);
                R.clip_stage(lq, i, i1, 1);
            }
            const Runner::FrameIO io = R.frame_io(i, lrs, fvs, mks, out, y_only);
            R.frame_pre(i & 1, i == 0, io);
            R.frame(i & 1, i == 0, io);
        }
        return R.rc;
    }
    // two-stream schedule: the side stream runs FNet and the state-independent part of every frame up to two
    // frames ahead of the recurrent chain on the caller's stream
    SideStream& ss = *ssp;
    bool forked = false;
    // every exit after the fork joins the side stream back into the caller's stream: the caller may free or reuse
    // lrs / fvs / mks / workspace as soon as its stream has drained, also after an error
    auto join = [&]() {
        hipEvent_t ej = ss.event(0);
        if (forked && ej && hipEventRecord(ej, ss.s) == hipSuccess) (void)hipStreamWaitEvent(main_s, ej, 0);
    };
    auto fail = [&](const char* what) { join(); set_error("dsv_forward_clip: %s failed", what); return 1; };
    hipEvent_t ev_start = ss.event(0), ev_xlr = ss.event(1);
    if (!ss.ok) return fail("hipEventCreate");
    // the status word and the recurrent state are cleared BEFORE the fork: the side stream's first kernel (frame 0's fovea
    // blend) may raise the overflow bit, and a memset racing with it on the other stream could wipe that
    R.reset_state();
    const float* lq = R.lr_to_q4(lrs, n * t, 0, t);   // before the fork: FNet on the side stream reads it as well
    if (R.rc) return R.rc;
    if (hipEventRecord(ev_start, main_s) != hipSuccess || hipStreamWaitEvent(ss.s, ev_start, 0) != hipSuccess) return fail("fork");
    forked = true;
    R.clip_stage(lq, 0, TC < t ? TC : t, 1);   // encoder_lr of the first chunk (flat: of every frame) on the caller's stream, beside frame 0's fovea blend
    if (hipEventRecord(ev_xlr, main_s) != hipSuccess) return fail("record");
    for (int i = 0; i < t && !R.rc; ++i) {
        hipEvent_t pre_done = ss.event(2 + 2 * i), main_done = ss.event(3 + 2 * i);
        if (!ss.ok) return fail("hipEventCreate");
        // side: pre-work of frame i into set i&1 (free once the recurrent part of frame i-2 is done)
        R.s = ss.s;
        // FNet (all pairs, 0.7 ms) goes BEHIND frame 0's pre-work: frame 0 needs no flow, and with FNet first the caller's
        // stream sat idle for FNet + pre(0) at the start of every clip.  It now runs beside frame 0's recurrent part.
        // (running the one pair frame 1 needs first and the other pairs beside frame 1 changes nothing: 10.62 vs 10.57 ms fp32, 6.31 vs 6.29 bf16)
        if (i == 1) R.clip_stage(lq, 0, TC < t ? TC : t, 2);
        // later chunks of a long clip: both clip-level stages on the side stream, in front of the chunk's first frame (the stores'
        // previous contents were last read by this stream's own earlier pre-work)
        if (i > 0 && i % TC == 0) R.clip_stage(lq, i, i + TC < t ? i + TC : t, 3);
        if (i >= 2 && hipStreamWaitEvent(ss.s, ss.event(3 + 2 * (i - 2)), 0) != hipSuccess) return fail("wait");
        const Runner::FrameIO io = R.frame_io(i, lrs, fvs, mks, out, y_only);
        R.frame_pre(i & 1, i == 0, io, i == 0 ? ev_xlr : nullptr);
        if (hipEventRecord(pre_done, ss.s) != hipSuccess) return fail("record");
        // main: recurrent part of frame i
        R.s = main_s;
        if (hipStreamWaitEvent(main_s, pre_done, 0) != hipSuccess) return fail("wait");
        R.frame(i & 1, i == 0, io);
        if (hipEventRecord(main_done, main_s) != hipSuccess) return fail("record");
    }
    if (R.rc) join();   // a launch failed mid-clip: the last pre_done wait may not have been enqueued
    return R.rc;
}

extern "C" {

int CRFP_API(crfp_dsv_forward_batch)(const void* packed, int flags, const float* lrs, const float* fvs, const uint8_t* mks,
                           float* out, int n, int t, int h, int w, void* workspace, size_t workspace_bytes, void* stream) {
    return forward_batch_impl(W_DSV, packed, flags, lrs, fvs, mks, out, n, t, h, w, workspace, workspace_bytes, stream);
}

// The same call for the reference's CRFP_DSV_CRA wiring (model/CRFP.py:2314-2664): packed = crfp_cra_pack_weights' buffer, workspace sized by
// crfp_cra_batch_workspace_bytes.  Schedule, status words and flags as crfp_dsv_forward_batch; the four-level fovea encoder is
// state-independent and runs with the rest of a frame's pre-work on the side stream.
int CRFP_API(crfp_cra_forward_batch)(const void* packed, int flags, const float* lrs, const float* fvs, const uint8_t* mks,
                           float* out, int n, int t, int h, int w, void* workspace, size_t workspace_bytes, void* stream) {
    return forward_batch_impl(W_CRA, packed, flags, lrs, fvs, mks, out, n, t, h, w, workspace, workspace_bytes, stream);
}

// The same call for the reference's ablation wirings CRFP_simple (model/CRFP.py:816-1099) and CRFP (:1101-1385) at mid_channels = 32 with
// hr_dcn and offset_prop on: packed = crfp_simple_pack_weights' / crfp_dense_pack_weights' buffer, workspace sized by the wiring's own query.
// Schedule, flags and status words as crfp_dsv_forward_batch.
int CRFP_API(crfp_simple_forward_batch)(const void* packed, int flags, const float* lrs, const float* fvs, const uint8_t* mks,
                           float* out, int n, int t, int h, int w, void* workspace, size_t workspace_bytes, void* stream) {
    return forward_batch_impl(W_SIMPLE, packed, flags, lrs, fvs, mks, out, n, t, h, w, workspace, workspace_bytes, stream);
}
int CRFP_API(crfp_dense_forward_batch)(const void* packed, int flags, const float* lrs, const float* fvs, const uint8_t* mks,
                           float* out, int n, int t, int h, int w, void* workspace, size_t workspace_bytes, void* stream) {
    return forward_batch_impl(W_DENSE, packed, flags, lrs, fvs, mks, out, n, t, h, w, workspace, workspace_bytes, stream);
}

int CRFP_API(crfp_dsv_forward_clip)(const void* packed, int flags, const float* lrs, const float* fvs, const uint8_t* mks,
                          float* out, int t, int h, int w, void* workspace, size_t workspace_bytes, void* stream) {
    return CRFP_API(crfp_dsv_forward_batch)(packed, flags, lrs, fvs, mks, out, 1, t, h, w, workspace, workspace_bytes, stream);
}

// One frame of n independent sequences in lock-step (the reference's streaming forward carries the batch axis too, model/CRFP_test.py:2250-2451):
// lr / lr_prev [n,3,h,w], fv [n,3,8h,8w], mk [n,1,8h,8w], out [n,3|1,8h,8w]; the workspace holds the n recurrent states.
int CRFP_API(crfp_dsv_stream_batch)(const void* packed, int flags, const float* lr, const float* lr_prev, const float* fv,
                          const uint8_t* mk, const uint8_t* fg, float* out, int first, int n, int h, int w, void* workspace,
                          size_t workspace_bytes, void* stream) {
    const int y_only = flags & CRFP_DSV_Y_ONLY;
    if (n < 1) { set_error("dsv_stream_batch: n = %d", n); return CRFP_E_BADARG; }
    if (fg && n > 1) { set_error("dsv_stream_batch: the regional mask `fg` is supported for one sequence per call (n = 1)"); return CRFP_E_UNSUPPORTED; }
    if (n > kFlatFrames) { set_error("dsv_stream_batch: at most %d sequences per call (got %d)", kFlatFrames, n); return CRFP_E_UNSUPPORTED; }
    const int B = n;
    const long long lr_f = 3LL * h * w, fq = (long long)h * w * 4;
    if (kActBf16 && (flags & CRFP_DSV_STRICT_F32)) { set_error("dsv (bf16 storage): CRFP_DSV_STRICT_F32 belongs to the fp32 entry points"); return CRFP_E_UNSUPPORTED; }
    Layout L(B, 1, h, w);
    int rc = check_common(packed, B, 1, h, w, workspace, workspace_bytes, L);
    if (rc) return rc;
    const bool resident = (flags & CRFP_DSV_INPUTS_RESIDENT) != 0;
    if (!lr || !fv || !mk || !out || (!first && !resident && !lr_prev)) { set_error("dsv_stream_batch: null tensor"); return CRFP_E_BADARG; }
    Runner R{model_for(y_only, flags & CRFP_DSV_STRICT_F32), (const float*)packed, (char*)workspace, L, (hipStream_t)stream};
    R.strict = (flags & CRFP_DSV_STRICT_F32) ? 1 : 0;
    // one frame of every sequence: the call's arguments ARE the frames (batch stride = one frame); flows and encoder_lr features sit in the
    // first B slots / the parity's B slots of their stores
    Runner::FrameIO io;
    io.lr = lr; io.fv = fv; io.mk = mk; io.out = out; io.x_lr = R.F(L.x_lr);
    io.lr_b = lr_f; io.fv_b = 3LL * 64 * h * w; io.mk_b = 64LL * h * w; io.out_b = (long long)(y_only ? 1 : 3) * 64 * h * w;
    io.x_b = 8LL * h * w * 4; io.flow_b = fq;
    SideStream* ssp = (!first && side_stream_enabled() && !prof_enabled() && !(flags & CRFP_DSV_SINGLE_STREAM)) ? side_stream() : nullptr;
    hipStream_t main_s = (hipStream_t)stream;
    // per-sequence host note, kept next to the side stream (also when this call runs on one stream)
    SideStream* tab = side_slot();
    StreamCtx* ctx = nullptr;
    if (tab && (resident || tab->ctx.count(workspace))) {
        if (tab->ctx.size() > 1024)   // workspaces come and go: drop the notes no resident sequence depends on
            for (auto it = tab->ctx.begin(); it != tab->ctx.end();) it = (!it->second.kept && it->first != workspace) ? tab->ctx.erase(it) : std::next(it);
        ctx = &tab->ctx[workspace];
    }
    if (resident) {
        // CRFP_DSV_INPUTS_RESIDENT: lr / fv / mk already hold their final values, and the previous frame is the copy the previous call
        // left in the workspace (the reference keeps `pre_lrs = lrs.clone()` on the model, model/CRFP_test.py:2234-2238).  Buffer sets
        // alternate with the call parity, so everything of call i that depends on neither the state nor earlier work on `stream` --
        // FNet, encoder_lr, the fovea blend, encoder_hr, the upsample conv -- is enqueued on the side stream WITHOUT waiting for the
        // caller's stream: it runs beside frame i - 1's recurrent chain.
        if (!ctx) { set_error("dsv_stream_batch: CRFP_DSV_INPUTS_RESIDENT needs the per-thread stream table (device index out of range)"); return CRFP_E_UNSUPPORTED; }
        if (first) *ctx = StreamCtx();
        else if (!ctx->kept) {
            set_error("dsv_stream_batch: CRFP_DSV_INPUTS_RESIDENT without a kept previous frame -- start the sequence with first != 0 and the "
                      "same flag, on this host thread");
            return CRFP_E_BADARG;
        }
        const int par = first ? 0 : (int)(++ctx->n & 1u);
        const size_t lr_bytes = (size_t)B * 3 * (size_t)h * w * sizeof(float);
        // the kept copy: fp32 build = the Q4 quads FNet reads anyway (slot par of lr_q4); bf16 build = an fp32 NCHW copy (a Q4 copy would be bf16)
        auto keep_and_get = [&](const float** cur, const float** prev) {
            if (kActBf16) {
                if (!R.rc && hipMemcpyAsync(R.F(L.lr_keep[par]), lr, lr_bytes, hipMemcpyDeviceToDevice, R.s) != hipSuccess) { set_error("dsv_stream_batch: hipMemcpyAsync failed"); R.rc = 1; }
                *cur = lr;
                *prev = R.F(L.lr_keep[par ^ 1]);
            } else {
                *cur = R.lr_to_q4(lr, B, par * B, 1);
                *prev = R.adv(R.F(L.lr_q4), (long long)(par ^ 1) * B * h * w * 4);
            }
        };
        R.flow_slot = par;
        const float *cur = nullptr, *prev = nullptr;
        if (!ssp) {   // one stream (first frame, profiling, CRFP_DSV_SINGLE_STREAM, CRFP_SIDE_STREAM=0): same order, same bits
            if (first) R.reset_state();
            keep_and_get(&cur, &prev);
            if (!first) R.fnet(B, cur, lr_f, prev, lr_f, R.flow_lr_slot());
            R.encode_lr(B, cur, lr_f);
            io.flow_lr = first ? nullptr : R.flow_lr_slot();
            R.frame_pre(par, first != 0, io);
            R.frame(par, first != 0, io, fg);
            ctx->kept = R.rc == 0;
            ctx->chained = false;
            return R.rc;
        }
        SideStream& ss = *ssp;
        bool forked = false;
        auto join = [&]() {
            hipEvent_t ej = ss.event(1);
            if (forked && ej && hipEventRecord(ej, ss.s) == hipSuccess) (void)hipStreamWaitEvent(main_s, ej, 0);
        };
        auto fail = [&](const char* what) { join(); ctx->kept = ctx->chained = false; set_error("dsv_stream_batch: %s failed", what); return 1; };
        hipEvent_t ev_start = ss.event(0), ev_side = ss.event(1);
        if (!ss.ok) return fail("hipEventCreate");
        if (hipEventRecord(ev_start, main_s) != hipSuccess) return fail("record");
        // Early start is safe only behind a call whose side work waited for ITS fork event: set par was last read by frame i - 2 on the
        // caller's stream, which that wait ordered in front of everything the side stream did since.
        const bool early = ctx->chained;
        R.s = ss.s;
        forked = true;
        if (!early && hipStreamWaitEvent(ss.s, ev_start, 0) != hipSuccess) return fail("fork");
        keep_and_get(&cur, &prev);
        R.fnet(B, cur, lr_f, prev, lr_f, R.flow_lr_slot());
        R.encode_lr(B, cur, lr_f);
        io.flow_lr = R.flow_lr_slot();
        R.frame_pre(par, false, io, nullptr, 3);   // incl. the two flow up-samplings (set par as well)
        if (early && hipStreamWaitEvent(ss.s, ev_start, 0) != hipSuccess) return fail("fork");
        R.downsample_state();   // needs the state frame i - 1 wrote
        R.down_done = true;
        if (hipEventRecord(ev_side, ss.s) != hipSuccess) return fail("record");
        R.s = main_s;
        if (hipStreamWaitEvent(main_s, ev_side, 0) != hipSuccess) return fail("join");
        R.frame(par, false, io, fg);
        ctx->kept = ctx->chained = R.rc == 0;
        return R.rc;
    }
    if (ctx) ctx->kept = ctx->chained = false;   // a call without the flag keeps no frame and orders nothing for a later resident call
    if (!ssp) {
        if (first) R.reset_state();
        const float* lq = R.lr_to_q4(lr, B, 0, 1);
        if (!first) R.fnet(B, lq, lr_f, R.lr_to_q4(lr_prev, B, B, 1), lr_f, R.F(L.flow_lr));
        R.encode_lr(B, lq, lr_f);
        io.flow_lr = first ? nullptr : R.F(L.flow_lr);
        R.frame_pre(0, first != 0, io);
        R.frame(0, first != 0, io, fg);
        return R.rc;
    }
    // two streams: FNet (one pair, small launch-latency-bound kernels, 0.45 ms) and the flow up-samplings stay on the
    // caller's stream; encoder_lr, the fovea blend, encoder_hr and the upsample conv run beside them on the side stream.
    // The caller's kernels are enqueued first (enqueue order across streams matters).
    SideStream& ss = *ssp;
    bool forked = false;
    auto join = [&]() {
        hipEvent_t ej = ss.event(1);
        if (forked && ej && hipEventRecord(ej, ss.s) == hipSuccess) (void)hipStreamWaitEvent(main_s, ej, 0);
    };
    auto fail = [&](const char* what) { join(); set_error("dsv_stream_batch: %s failed", what); return 1; };
    hipEvent_t ev_start = ss.event(0), ev_side = ss.event(1);
    if (!ss.ok) return fail("hipEventCreate");
    const float* lq = R.lr_to_q4(lr, B, 0, 1);   // before the fork: read on both streams
    const float* lqp = R.lr_to_q4(lr_prev, B, B, 1);
    if (hipEventRecord(ev_start, main_s) != hipSuccess) return fail("record");
    R.fnet(B, lq, lr_f, lqp, lr_f, R.F(L.flow_lr));
    io.flow_lr = R.F(L.flow_lr);
    R.frame_pre(0, false, io, nullptr, 2);
    R.s = ss.s;
    if (hipStreamWaitEvent(ss.s, ev_start, 0) != hipSuccess) return fail("fork");
    forked = true;
    R.encode_lr(B, lq, lr_f);
    R.frame_pre(0, false, io, nullptr, 1);
    // downsample(state) needs nothing of this call: it runs here, beside FNet's small kernels, instead of in front of the recurrent chain
    // (same box, 24 calls: bf16 27.74 -> 27.40 ms, fp32 45.81 -> 45.23 ms, same bits; profiles/r03_stream_down_side_ab.txt)
    R.downsample_state();
    R.down_done = true;
    if (hipEventRecord(ev_side, ss.s) != hipSuccess) return fail("record");
    R.s = main_s;
    if (hipStreamWaitEvent(main_s, ev_side, 0) != hipSuccess) return fail("join");
    R.frame(0, false, io, fg);
    return R.rc;
}

int CRFP_API(crfp_dsv_stream_frame)(const void* packed, int flags, const float* lr, const float* lr_prev, const float* fv,
                          const uint8_t* mk, const uint8_t* fg, float* out, int first, int h, int w, void* workspace,
                          size_t workspace_bytes, void* stream) {
    return CRFP_API(crfp_dsv_stream_batch)(packed, flags, lr, lr_prev, fv, mk, fg, out, first, 1, h, w, workspace, workspace_bytes, stream);
}

#ifndef CRFP_ACT_BF16
int crfp_shutdown(void) {
    destroy_all_side_streams();
    crfp_bf16::shutdown_side_streams();
    crfp::rt_shutdown_streams();
    return 0;
}
#endif

int CRFP_API(crfp_fnet_forward)(const void* packed, const float* cur, const float* prev, float* flow, int n, int h, int w,
                      void* workspace, size_t workspace_bytes, void* stream) {
    if (n < 1) { set_error("fnet_forward: n = %d", n); return CRFP_E_BADARG; }
    Layout L(1, n + 1, h, w);
    int rc = check_common(packed, 1, n + 1, h, w, workspace, workspace_bytes, L);
    if (rc) return rc;
    if (!cur || !prev || !flow) { set_error("fnet_forward: null tensor"); return CRFP_E_BADARG; }
    Runner R{model_for(0), (const float*)packed, (char*)workspace, L, (hipStream_t)stream};
    R.ovf_div = 0;   // pairs, not clips: one status word
    const long long lr_f = 3LL * h * w, lqf = R.lr_frame_floats();
    const float* cq = R.lr_to_q4(cur, n, 0);
    const float* pq = R.lr_to_q4(prev, n, n + 1);
    for (int p0 = 0; p0 < n && !R.rc; p0 += L.fnet_cap) {   // the workspace holds fnet_cap pairs at a time
        const int cnt = n - p0 < L.fnet_cap ? n - p0 : L.fnet_cap;
        R.fnet(cnt, cq + p0 * lqf, lr_f, pq + p0 * lqf, lr_f, R.F(L.flow_lr));
        if (!R.rc) R.rc = crfp::launch_q4_to_nchw(R.F(L.flow_lr), flow + (long long)p0 * 2 * h * w, cnt, 2, h, w, 0, (hipStream_t)stream);
    }
    return R.rc;
}

int CRFP_API(crfp_dsv_debug_fetch)(const char* name, int t, int h, int w, const void* workspace, float* out_nchw, int* c_out,
                         int* h_out, int* w_out, void* stream) {
    if (!name || !workspace) return CRFP_E_BADARG;
    Layout L(1, t, h, w);
    for (auto& b : L.A.bufs)
        if (b.name == name) {
            const float* p = reinterpret_cast<const float*>((const char*)workspace + b.off);
            int bN = b.N;
            // the flow of frame i lives in slot i of the store (slot 0 belongs to frame 0, which has none): present frames 1 .. t - 1
            if (b.name == "flow_lr" && t > 1 && L.flat) { p += (size_t)b.H * b.W * 4; bN = t - 1; }
            if (c_out) *c_out = b.kind == 0 ? b.nq * 4 : 2;
            if (h_out) *h_out = b.H;
            if (w_out) *w_out = b.W;
            if (!out_nchw) return bN;
            if (b.kind == 0 && b.f32) return crfp::launch_q4_to_nchw(p, out_nchw, bN, b.nq * 4, b.H, b.W, b.pad, (hipStream_t)stream);
            if (b.kind == 0) return launch_q4_to_nchw(p, out_nchw, bN, b.nq * 4, b.H, b.W, b.pad, (hipStream_t)stream);
            return hipMemcpyAsync(out_nchw, p, (size_t)bN * b.H * b.W * 2 * sizeof(float), hipMemcpyDeviceToDevice,
                                  (hipStream_t)stream) == hipSuccess ? 0 : 1;
        }
    set_error("debug_fetch: unknown buffer '%s'", name);
    return CRFP_E_BADARG;
}

}  // extern "C"
