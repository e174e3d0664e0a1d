/* libcrfp_hip.so -- C ABI of the MI355X-native (gfx950) CRFP recurrent x8 foveated-VSR path.
 *
 * This is the drop-in boundary: plain pointers and sizes, no torch types.  Every pointer is a
 * DEVICE pointer owned by the caller (e.g. PyTorch's caching allocator) unless the argument is
 * documented as host memory; the library never allocates, frees or retains device memory --
 * scratch arrives through `workspace` (size from the matching *_workspace_bytes query).  Kernels
 * are enqueued asynchronously on `stream` (a hipStream_t passed as void*); no entry point
 * synchronises the device.
 *
 * State kept between calls (all of it): per (host thread, device) ONE non-blocking HIP stream and a
 * small pool of timing-disabled events, created lazily by the first crfp_dsv_forward_clip /
 * crfp_dsv_stream_frame on that device.  The engine forks the state-independent part of each frame
 * onto that stream and joins it back with events before returning control of `stream`'s order, on
 * success AND on every error path, so for the caller each call behaves as if it ran on `stream`
 * alone (buffers may be reused once `stream` has drained).  Clips issued by one host thread on
 * several caller streams share that one side stream (their side work serialises; results are
 * unaffected).  CRFP_SIDE_STREAM=0 in the environment (read once) selects a single-stream
 * schedule that creates nothing.  A host thread that exits hands its table (stream + events) to the next
 * new thread -- without calling HIP -- so short-lived worker threads do not accumulate streams.
 * crfp_shutdown() destroys EVERY thread's side streams and events (call it when no library call is in
 * flight on any thread and the streams have drained); the next call re-creates them lazily.  Return value: 0 = success, negative = CRFP_E_* argument error,
 * positive = hipError_t of a failed launch.  crfp_last_error_string() describes the last failure
 * on the calling thread.  No exceptions cross the boundary.
 *
 * Reference interfaces replaced (paths relative to the reference repo eugenelet/CRFP):
 *   crfp_flow_warp_f32        model/CRFP.py:90-130   flow_warp()  (ATen grid_sampler_2d underneath)
 *   crfp_dcnv2_forward_f32    model/CRFP.py:6,318-320,350  third-party dcn_v2.DCNv2.forward
 *   crfp_conv3x3_f32          every nn.Conv2d(k=3,s=1,p=1) on the path (e.g. model/CRFP.py:48-50,303-317)
 *   crfp_upsample_bilinear_f32  nn.Upsample / F.interpolate(align_corners=False) (model/CRFP.py:808-812,1471-1478)
 *   crfp_fnet_*               model/CRFP.py:743-814  FNet.forward, :1483-1508 compute_flow
 *   crfp_dsv_*                model/CRFP.py:1387-1706 CRFP_DSV (ctor weights, forward) and the
 *                             one-frame-per-call variant model/CRFP_test.py:2114-2478
 *   crfp_psnr_partial_f32     utils.py:166-185,242-254,328-330 (psnr_cuda / bgr2ycbcr(y_only))
 *   crfp_spynet_forward       model/CRFP.py:554-741 SPyNet.forward (+ SPyNetBasicModule, `conv` :145-152)
 *   crfp_convkxk_f32          model/CRFP.py:145-152 `conv`: ReLU -> nn.Conv2d(k, stride 1, pad k/2)
 *   crfp_upsample_bilinear_ac_f32  F.interpolate(..., bilinear, align_corners=True) (model/CRFP.py:647-651)
 */
#ifndef CRFP_HIP_H
#define CRFP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CRFP_VERSION 201 /* major*10000 + minor*100 + patch */

/* error codes (negative) */
#define CRFP_E_BADARG (-1)      /* null pointer / non-positive size / unsupported combination */
#define CRFP_E_WORKSPACE (-2)   /* workspace missing or too small */
#define CRFP_E_UNSUPPORTED (-3) /* shape or mode outside what the kernels implement */
#define CRFP_E_STATE (-4)       /* engine used before weights were packed */

/* activations for crfp_conv3x3_f32: out = post_scale * act(conv + bias) */
#define CRFP_ACT_NONE 0
#define CRFP_ACT_RELU 1
#define CRFP_ACT_LRELU01 2 /* LeakyReLU(0.1) */
#define CRFP_ACT_TANH 3
#define CRFP_ACT_SIGMOID 4

#define CRFP_PAD_ZEROS 0
#define CRFP_PAD_BORDER 1

int crfp_version(void);
const char* crfp_last_error_string(void);
/* destroy every host thread's side streams / events (see "State kept between calls"); always 0 */
int crfp_shutdown(void);

/* ---- flow_warp: x[n,c,h,w] (NCHW f32), flow[n,h,w,2] = (dx,dy) pixels, out[n,c,h,w].
 * Bilinear backward warp, align_corners=True semantics of the reference (sample position =
 * pixel index + flow, routed through the [-1,1] normalisation in the same float32 order). */
size_t crfp_flow_warp_workspace_bytes(int n, int c, int h, int w);
int crfp_flow_warp_f32(const float* x, const float* flow, float* out, int n, int c, int h, int w,
                       int padding_mode, void* workspace, size_t workspace_bytes, void* stream);

/* ---- DCNv2 forward (stride 1): x[n,cin,h,w], offset[n,2*dg*k*k,h,w] ((dy,dx) interleaved per
 * tap per group), mask[n,dg*k*k,h,w], weight[cout,cin,k,k], bias[cout] -> out[n,cout,h,w].
 * k=3, pad=1, dil=1 only (every call site of the reference).  Fast paths: (cin=cout=32, dg=8)
 * on MFMA and (cin=cout=4, dg=1); other channel counts run a generic HIP kernel. */
size_t crfp_dcnv2_workspace_bytes(int n, int cin, int cout, int h, int w, int k, int dg);
int crfp_dcnv2_forward_f32(const float* x, const float* offset, const float* mask, const float* weight,
                           const float* bias, float* out, int n, int cin, int cout, int h, int w, int k,
                           int pad, int dil, int dg, void* workspace, size_t workspace_bytes, void* stream);

/* ---- DCNv2 with ONE (dy, dx) and ONE mask per pixel shared by the 9 taps (cin = cout = 4, deformable_groups = 1): what
 * DCN_module(repeat=True) computes after tiling its 2 + 1 channels 9x (model/CRFP.py:341-347,350); the tiled tensors never exist.
 * offset [n,2,h,w] = (dy, dx), mask [n,1,h,w]. */
size_t crfp_dcnv2_shared_workspace_bytes(int n, int c, int h, int w);
int crfp_dcnv2_shared_f32(const float* x, const float* offset, const float* mask, const float* weight, const float* bias, float* out,
                          int n, int cin, int cout, int h, int w, void* workspace, size_t workspace_bytes, void* stream);

/* ---- 3x3 stride-1 pad-1 convolution, NCHW f32, fused bias + activation (fp32 MFMA). */
size_t crfp_conv3x3_workspace_bytes(int n, int cin, int cout, int h, int w);
int crfp_conv3x3_f32(const float* x, const float* weight, const float* bias, float* out, int n, int cin,
                     int cout, int h, int w, int act, float post_scale, void* workspace,
                     size_t workspace_bytes, void* stream);

/* ---- the conv operator in full: up to two inputs (= conv(torch.cat([x, x2], 1)), e.g. model/CRFP.py:331,1589), optional
 * residual added after activation (model/CRFP.py:48-50 ResidualBlockNoBN), load through pixel_unshuffle(4) (PixelUnShufflePack_v2,
 * model/CRFP.py:28-42: x is [n, cin/16, 4h, 4w]), store into the channel slice [out_c0, out_c0 + cout) of an [n, out_ctotal, h, w]
 * tensor (a fused torch.cat on the output side) or through pixel_shuffle(r), r in {2, 4} (PixelShufflePack, model/CRFP.py:184-193:
 * out is [n, cout/r^2, h r, w r]; no residual, activation none / relu / lrelu).  Pass 0 / 1 for "no shuffle".  fp32 MFMA. */
size_t crfp_conv3x3_ex_workspace_bytes(int n, int cin, int cin2, int cout, int h, int w, int unshuffle_r, int shuffle_r, int has_residual);
int crfp_conv3x3_ex_f32(const float* x, int cin, const float* x2, int cin2, const float* weight, const float* bias, const float* residual,
                        float* out, int n, int cout, int h, int w, int act, float post_scale, int unshuffle_r, int shuffle_r,
                        int out_c0, int out_ctotal, void* workspace, size_t workspace_bytes, void* stream);

/* The same two operators with the weight repack hoisted out of the call (pack once per weight update):
 *   crfp_conv3x3_pack_f32 -> crfp_conv3x3_packed_f32           (any cin / cout)
 *   crfp_dcnv2_g8_pack_f32 -> crfp_dcnv2_g8_packed_f32         (cin = cout = 32, deformable_groups = 8, k = 3) */
size_t crfp_conv3x3_packed_bytes(int cin, int cout);
int crfp_conv3x3_pack_f32(const float* weight, const float* bias, int cin, int cout, void* packed, size_t packed_bytes, void* stream);
int crfp_conv3x3_packed_f32(const float* x, const void* packed, float* out, int n, int cin, int cout, int h, int w, int act,
                            float post_scale, void* stream);
size_t crfp_dcnv2_g8_packed_bytes(void);
int crfp_dcnv2_g8_pack_f32(const float* weight, void* packed, size_t packed_bytes, void* stream);
int crfp_dcnv2_g8_packed_f32(const float* x, const float* offset, const float* mask, const void* packed, const float* bias, float* out,
                             int n, int h, int w, void* workspace, size_t workspace_bytes, void* stream);

/* ---- bilinear resize, align_corners=False, NCHW f32; out = mul * resize(x).  scale_h/scale_w
 * are the source-index scales PyTorch uses (1/scale_factor for nn.Upsample, in/out for size=). */
int crfp_upsample_bilinear_f32(const float* x, float* out, int n, int c, int h, int w, int oh, int ow,
                               float scale_h, float scale_w, float mul, void* stream);

/* bilinear resize with align_corners=True (src = dst * (in - 1) / (out - 1)), NCHW f32; out = mul * resize(x). */
int crfp_upsample_bilinear_ac_f32(const float* x, float* out, int n, int c, int h, int w, int oh, int ow, float mul, void* stream);

/* k x k stride-1 pad-k/2 convolution (k in {3, 5, 7}), NCHW f32, direct fp32 FMA; pre_relu != 0 applies ReLU to the INPUT
 * (the reference's `conv` module, model/CRFP.py:145-152: self.conv(self.act(x))). */
int crfp_convkxk_f32(const float* x, const float* weight, const float* bias, float* out, int n, int cin, int cout, int h, int w,
                     int k, int pre_relu, void* stream);

/* ---- SPyNet.forward(ref, supp) -> flow[n,2,h,w] (model/CRFP.py:698-741), one call.  params: the 60 device pointers of
 * the reference's state_dict order, basic_module.{L}.basic_module.{j}.conv.{weight,bias} for L = 0..5 (coarsest first),
 * j = 0..4 (7x7 convs 8->32->64->32->16->2); the mean / std buffers are the ImageNet constants of :586-591. */
#define CRFP_SPYNET_NUM_PARAMS 60
size_t crfp_spynet_workspace_bytes(int n, int h, int w);
int crfp_spynet_forward(const float* const* params, const float* ref, const float* supp, float* flow, int n, int h, int w,
                        void* workspace, size_t workspace_bytes, void* stream);

/* ---- sum of squared differences for PSNR: acc[0] += sum((a-b)^2) over [n,c,h,w];
 * acc[1] += the same on the luma the reference's eval computes (24.966*c0+128.553*c1+65.481*c2+16,
 * utils.py:328-330), c==3 only.  acc is 2 doubles (device), zeroed by the caller. */
int crfp_psnr_partial_f32(const float* a, const float* b, double* acc, int n, int c, int h, int w, void* stream);

/* nn.AvgPool2d(2, 2), floor mode (FNet, model/CRFP.py:755): x [n,c,h,w] -> out [n,c,h/2,w/2]. */
int crfp_avgpool2_f32(const float* x, float* out, int n, int c, int h, int w, void* stream);

/* Fovea fusion + output head of one frame, fused (model/CRFP.py:1672-1684):
 *   f' = conv_tttf(cat(state, x_hr)); new_state = LeakyReLU_0.1(mask ? f' : state); out = conv_last(new_state) + bilinear_x8(lr)
 *   (y_only: + bilinear_x8(0.299 R + 0.587 G + 0.114 B), one channel).
 * state, x_hr, new_state: [n,4,8h,8w]; mask: [n,1,8h,8w] bytes; lr: [n,3,h,w]; out: [n,3|1,8h,8w];
 * w_tttf [4,8,3,3], b_tttf [4], w_last [3|1,4,3,3], b_last [3|1] in the reference's OIHW order. */
size_t crfp_fovea_head_workspace_bytes(int n, int h, int w);
int crfp_fovea_head_f32(const float* state, const float* x_hr, const unsigned char* mask, const float* lr, const float* w_tttf,
                        const float* b_tttf, const float* w_last, const float* b_last, float* new_state, float* out, int n,
                        int h, int w, int y_only, void* workspace, size_t workspace_bytes, void* stream);

/* Masked PSNR + SSIM raw sums in one pass: replaces utils.calc_psnr_and_ssim_cuda -> psnr_cuda / ssim_cuda / _ssim
 * (utils.py:166-185,187-240,242-254; callers trainer.py:349-369, test_video.py:357-369).  a, b: [n,c,h,w] fp32; mask:
 * [n,1,h,w] bytes (0 / non-0) or NULL (= all ones); x' = x*mul + add is the reference's range conversion (1/255, 0 when
 * max-min > 2; 0.5, 0.5 when > 1; else 1, 0).  acc (3 doubles, zeroed by the caller) receives
 * acc[0] += sum m (a'-b')^2 (all channels), acc[1] += sum m SSIM_map (all channels), acc[2] += sum m (per pixel). */
int crfp_psnr_ssim_partial_f32(const float* a, const float* b, const unsigned char* mask, double* acc, int n, int c, int h, int w,
                               float mul, float add, void* stream);

/* ---- CRFP_DSV engine (mid_channels=32, hr_dcn=True, offset_prop=True; y_only selectable).
 * Parameters arrive as CRFP_DSV_NUM_PARAMS device pointers in the order of the reference's
 * state_dict (weight then bias of each conv, list in crfp_dsv_param_name). */
#define CRFP_DSV_NUM_PARAMS 118
const char* crfp_dsv_param_name(int index);               /* state_dict key of parameter `index` */
int crfp_dsv_param_numel(int index, int y_only);          /* element count of that parameter */
size_t crfp_dsv_packed_weight_bytes(int y_only);
/* repack all parameters into the MFMA / stencil layouts (device -> device, async on stream) */
int crfp_dsv_pack_weights(const float* const* params, int y_only, void* packed, size_t packed_bytes, void* stream);

size_t crfp_dsv_workspace_bytes(int t, int h, int w);

/* `flags` of the two forward entry points (the round-1 ABI passed y_only in the same slot: 0 / 1 keep their meaning). */
#define CRFP_DSV_Y_ONLY 1     /* the model was built with y_only=True: one output channel */
#define CRFP_DSV_STRICT_F32 2 /* plain fp32 MFMA for every convolution and the DCN GEMM instead of the default split-fp16
                               * scheme (fp32-grade, 3 fp16 MFMAs per product, needs |activation|, |weight| < 65504) */
#define CRFP_DSV_SINGLE_STREAM 4 /* this call enqueues everything on `stream` itself (no fork onto the library's side stream);
                               * same bits, used to measure what the two-stream schedule hides */
#define CRFP_DSV_INPUTS_RESIDENT 8 /* crfp_dsv_stream_frame only.  The caller promises that lr, fv and mk hold their final values when the call
                               * is made (no earlier work on `stream` still writes them) and that they stay untouched until `stream` has
                               * drained this call.  `lr_prev` is ignored (may be NULL): the previous frame is the copy the previous call
                               * on this workspace left inside it, as the reference's model keeps `pre_lrs = lrs.clone()`
                               * (model/CRFP_test.py:2234-2238).  Every call of a sequence, from the `first` one on, must carry the flag and
                               * come from the same host thread (the library keeps a host-side call counter per workspace next to its side
                               * stream).  In return the state-independent part of frame i (FNet, encoder_lr, fovea blend, encoder_hr, the
                               * upsample conv) is enqueued on the side stream WITHOUT waiting for `stream` and runs beside frame i - 1's
                               * recurrent chain (buffer sets alternate with the call parity).  Same bits as without the flag.  Not for
                               * calls under HIP graph capture: the early side-stream work is, by design, not ordered behind `stream`. */

/* Numerics status: a 32-bit word inside the workspace at this byte offset.  Bit 0 is raised (sticky until the next clip
 * / the next `first` streamed frame) when a kernel of the split-fp16 scheme stores a value an fp16 operand cannot hold
 * (|v| >= 65504 or inf); from then on every output frame of the call is filled with NaN instead of plausible garbage.
 * Read it (after synchronising the stream) to tell "overflow" from "NaN inputs", and rerun with CRFP_DSV_STRICT_F32. */
size_t crfp_dsv_status_offset(int t, int h, int w);

/* One clip: lrs[t,3,h,w], fvs[t,3,8h,8w] f32, mks[t,1,8h,8w] u8 (bool), out[t,3|1,8h,8w].
 * Zero initial state; flows from FNet(frame i, frame i-1).  (= crfp_dsv_forward_batch with n = 1.)
 * mks is used as a select (0 / non-0), as the reference's 0 / 1 float mask acts in fvs * mk + x * (1 - mk): fvs is only ever read where mks is set, and the
 * work whose result the select discards (the x8 frame stack, encoder_hr and conv_tttf away from the mask) is skipped per 64 x 16 tile, bit-identically
 * (CRFP_MASK_GATE=0 in the environment launches it densely). */
int crfp_dsv_forward_clip(const void* packed, int flags, const float* lrs, const float* fvs, const uint8_t* mks,
                          float* out, int t, int h, int w, void* workspace, size_t workspace_bytes, void* stream);

/* A batch of n independent clips, the reference's own tensor shapes (model/CRFP.py:1510-1535: forward(lrs, fvs, mks) carries the
 * batch axis n through every op; main.py:37-38 scatters batches): lrs[n,t,3,h,w], fvs[n,t,3,8h,8w], mks[n,t,1,8h,8w] u8,
 * out[n,t,3|1,8h,8w].  The n clips walk the recurrent chain in lock-step -- ONE launch per layer and frame step over all n clips --
 * so a 2x-resolution map that is a single round of workgroups for one clip becomes n rounds whose load / MFMA / store phases
 * overlap.  Per clip the arithmetic is that of crfp_dsv_forward_clip: outputs are bit-identical to n one-clip calls.  The workspace
 * holds n recurrent states (query crfp_dsv_batch_workspace_bytes) and n status words at crfp_dsv_batch_status_offset, word b for clip
 * b: an fp16-operand overflow poisons the frames of the clip it happened in and no other, as with one call per clip.  The clip-level stages
 * (FNet, encoder_lr) run once over all n * t frames when n * t <= 32 and in chunks of 8 frames per clip otherwise, so the
 * workspace does not grow with t. */
size_t crfp_dsv_batch_workspace_bytes(int n, int t, int h, int w);
size_t crfp_dsv_batch_status_offset(int n, int t, int h, int w);
int crfp_dsv_forward_batch(const void* packed, int flags, const float* lrs, const float* fvs, const uint8_t* mks,
                           float* out, int n, int t, int h, int w, void* workspace, size_t workspace_bytes, void* stream);

/* Streaming: one frame per call, recurrent state kept inside `workspace` (same buffer every call).
 * `first` != 0 resets the state (clear_states of the reference's streaming model). lr_prev may be
 * NULL when first != 0, and is ignored under CRFP_DSV_INPUTS_RESIDENT (the library then keeps the previous frame in the
 * workspace and overlaps the state-independent part of this frame with the previous call's kernels).  fg: optional regional mask [8h,8w] u8 (the reference's `fgs`,
 * model/CRFP_test.py:2296-2298,2361,2375,2389), NULL = all ones. */
int crfp_dsv_stream_frame(const void* packed, int flags, const float* lr, const float* lr_prev, const float* fv,
                          const uint8_t* mk, const uint8_t* fg, float* out, int first, int h, int w, void* workspace,
                          size_t workspace_bytes, void* stream);

/* The same for n independent sequences in lock-step, one frame of each per call (the reference's streaming forward carries the batch axis too,
 * model/CRFP_test.py:2250-2451): lr / lr_prev [n,3,h,w], fv [n,3,8h,8w], mk [n,1,8h,8w], out [n,3|1,8h,8w]; the workspace
 * (crfp_dsv_batch_workspace_bytes(n, 1, h, w)) holds the n recurrent states and n status words.  n <= 32; `fg` needs n = 1.  Per sequence the
 * results are bit-identical to n one-sequence call chains.  crfp_dsv_stream_frame = this with n = 1. */
int crfp_dsv_stream_batch(const void* packed, int flags, const float* lr, const float* lr_prev, const float* fv,
                          const uint8_t* mk, const uint8_t* fg, float* out, int first, int n, int h, int w, void* workspace,
                          size_t workspace_bytes, void* stream);

/* ---- CRFP_DSV_CRA engine: the reference's cross-resolution-fusion wiring of the same model (model/CRFP.py:2314-2664; the `_cra` run of
 * eval.sh, factory line main.py:35): CRFP_DSV with the four-level fovea encoder LTE_simple_hr_ps (:156-166) and, after each 2x level's
 * residual block, features <- mk2 * conv_tttf_k(cat(features, fovea level k)) + (1 - mk2) * features with mk2 the x0.25 bilinear resample
 * of the fovea mask (:2501,2533-2535).  mid_channels = 32, hr_dcn = offset_prop = True.  Parameters: CRFP_CRA_NUM_PARAMS device pointers in
 * the order of the reference's CRFP_DSV_CRA state_dict (crfp_cra_param_name).  The forward call takes the tensors, flags, status words and
 * workspace rules of crfp_dsv_forward_batch (n clips in lock-step, bit-identical per clip to n one-clip calls); packed weights and
 * workspaces are this wiring's own (query the crfp_cra_* sizes). */
#define CRFP_CRA_NUM_PARAMS 144
const char* crfp_cra_param_name(int index);
int crfp_cra_param_numel(int index, int y_only);
size_t crfp_cra_packed_weight_bytes(int y_only);
int crfp_cra_pack_weights(const float* const* params, int y_only, void* packed, size_t packed_bytes, void* stream);
size_t crfp_cra_batch_workspace_bytes(int n, int t, int h, int w);
size_t crfp_cra_batch_status_offset(int n, int t, int h, int w);
int crfp_cra_forward_batch(const void* packed, int flags, const float* lrs, const float* fvs, const uint8_t* mks,
                           float* out, int n, int t, int h, int w, void* workspace, size_t workspace_bytes, void* stream);

/* ---- CRFP_simple / CRFP engines: the reference's two ablation wirings in front of CRFP_DSV -- CRFP_simple ("v13", model/CRFP.py:816-1099)
 * and CRFP ("v15", :1101-1385) -- as one-call schedules, for mid_channels = 32 with hr_dcn = offset_prop = True (round 6; every other
 * constructor combination stays a per-operator composition in the Python mirror).  Against CRFP_DSV: `upsample` keeps all 32 features
 * (:875) and nothing is carried beside a level; the previous state is warped at 8x FIRST and both versions are brought to 2x by
 * `downsample` (:1021-1026); `crfp_dense_*` (the reference's class CRFP) additionally feeds every residual block the warped previous state
 * as a third input (:1311,1316).  Parameters: CRFP_DSV_NUM_PARAMS device pointers under CRFP_DSV's state_dict keys, in its order
 * (crfp_dsv_param_name); `upsample`, `upsample_post` and -- dense -- the four `forward_resblocks_k.main.0` weights have other shapes
 * (crfp_simple_param_numel / crfp_dense_param_numel).  The forward calls take the tensors, flags, status words and workspace rules of
 * crfp_dsv_forward_batch; packed weights and workspaces are each wiring's own. */
int crfp_simple_param_numel(int index, int y_only);
size_t crfp_simple_packed_weight_bytes(int y_only);
int crfp_simple_pack_weights(const float* const* params, int y_only, void* packed, size_t packed_bytes, void* stream);
size_t crfp_simple_batch_workspace_bytes(int n, int t, int h, int w);
size_t crfp_simple_batch_status_offset(int n, int t, int h, int w);
int crfp_simple_forward_batch(const void* packed, int flags, const float* lrs, const float* fvs, const uint8_t* mks,
                              float* out, int n, int t, int h, int w, void* workspace, size_t workspace_bytes, void* stream);
int crfp_dense_param_numel(int index, int y_only);
size_t crfp_dense_packed_weight_bytes(int y_only);
int crfp_dense_pack_weights(const float* const* params, int y_only, void* packed, size_t packed_bytes, void* stream);
size_t crfp_dense_batch_workspace_bytes(int n, int t, int h, int w);
size_t crfp_dense_batch_status_offset(int n, int t, int h, int w);
int crfp_dense_forward_batch(const void* packed, int flags, const float* lrs, const float* fvs, const uint8_t* mks,
                             float* out, int n, int t, int h, int w, void* workspace, size_t workspace_bytes, void* stream);

/* ---- bf16 storage (BASELINE configs 3-5): the same engine with every activation tensor and the recurrent state held
 * as bf16 in HBM (half the traffic of the HBM-bound kernels, one bf16 MFMA per product instead of three fp16 ones).
 * What stays fp32: the API tensors (lrs, fvs, out), all accumulators and interpolation arithmetic, biases, and everything
 * that is a coordinate -- flow fields, DCN offsets and masks.  Conv / DCN weights are rounded to bf16 once at pack time.
 * Same arguments as the entry points above; `flags` takes CRFP_DSV_Y_ONLY only.  Packed weights and workspaces are NOT
 * interchangeable between the two builds (query the *_bf16 sizes).  Numerics: see DESIGN.md section 4 (oracle twin that
 * rounds at the same points). */
size_t crfp_dsv_packed_weight_bytes_bf16(int y_only);
int crfp_dsv_pack_weights_bf16(const float* const* params, int y_only, void* packed, size_t packed_bytes, void* stream);
size_t crfp_dsv_workspace_bytes_bf16(int t, int h, int w);
size_t crfp_dsv_status_offset_bf16(int t, int h, int w);
int crfp_dsv_forward_clip_bf16(const void* packed, int flags, const float* lrs, const float* fvs, const uint8_t* mks,
                               float* out, int t, int h, int w, void* workspace, size_t workspace_bytes, void* stream);
size_t crfp_dsv_batch_workspace_bytes_bf16(int n, int t, int h, int w);
size_t crfp_dsv_batch_status_offset_bf16(int n, int t, int h, int w);
int crfp_dsv_forward_batch_bf16(const void* packed, int flags, const float* lrs, const float* fvs, const uint8_t* mks,
                                float* out, int n, int t, int h, int w, void* workspace, size_t workspace_bytes, void* stream);
int crfp_dsv_stream_frame_bf16(const void* packed, int flags, const float* lr, const float* lr_prev, const float* fv,
                               const uint8_t* mk, const uint8_t* fg, float* out, int first, int h, int w, void* workspace,
                               size_t workspace_bytes, void* stream);
int crfp_dsv_stream_batch_bf16(const void* packed, int flags, const float* lr, const float* lr_prev, const float* fv,
                               const uint8_t* mk, const uint8_t* fg, float* out, int first, int n, int h, int w, void* workspace,
                               size_t workspace_bytes, void* stream);
size_t crfp_cra_packed_weight_bytes_bf16(int y_only);
int crfp_cra_pack_weights_bf16(const float* const* params, int y_only, void* packed, size_t packed_bytes, void* stream);
size_t crfp_cra_batch_workspace_bytes_bf16(int n, int t, int h, int w);
size_t crfp_cra_batch_status_offset_bf16(int n, int t, int h, int w);
int crfp_cra_forward_batch_bf16(const void* packed, int flags, const float* lrs, const float* fvs, const uint8_t* mks,
                                float* out, int n, int t, int h, int w, void* workspace, size_t workspace_bytes, void* stream);
size_t crfp_simple_packed_weight_bytes_bf16(int y_only);
int crfp_simple_pack_weights_bf16(const float* const* params, int y_only, void* packed, size_t packed_bytes, void* stream);
size_t crfp_simple_batch_workspace_bytes_bf16(int n, int t, int h, int w);
size_t crfp_simple_batch_status_offset_bf16(int n, int t, int h, int w);
int crfp_simple_forward_batch_bf16(const void* packed, int flags, const float* lrs, const float* fvs, const uint8_t* mks,
                                   float* out, int n, int t, int h, int w, void* workspace, size_t workspace_bytes, void* stream);
size_t crfp_dense_packed_weight_bytes_bf16(int y_only);
int crfp_dense_pack_weights_bf16(const float* const* params, int y_only, void* packed, size_t packed_bytes, void* stream);
size_t crfp_dense_batch_workspace_bytes_bf16(int n, int t, int h, int w);
size_t crfp_dense_batch_status_offset_bf16(int n, int t, int h, int w);
int crfp_dense_forward_batch_bf16(const void* packed, int flags, const float* lrs, const float* fvs, const uint8_t* mks,
                                  float* out, int n, int t, int h, int w, void* workspace, size_t workspace_bytes, void* stream);
int crfp_fnet_forward_bf16(const void* packed, const float* cur, const float* prev, float* flow, int n, int h, int w,
                           void* workspace, size_t workspace_bytes, void* stream);
int crfp_dsv_debug_fetch_bf16(const char* name, int t, int h, int w, const void* workspace, float* out_nchw,
                              int* c_out, int* h_out, int* w_out, void* stream);

/* FNet alone (compute_flow): pairs cur[n,3,h,w], prev[n,3,h,w] -> flow[n,2,h,w] (NCHW). */
int crfp_fnet_forward(const void* packed, const float* cur, const float* prev, float* flow, int n, int h, int w,
                      void* workspace, size_t workspace_bytes, void* stream);

/* ---- per-kernel timing for bench.py's roofline object: when enabled every kernel launch is
 * bracketed by hipEvents on its stream; crfp_prof_report synchronises those events and fills
 * up to `cap` records (name, launches, total ms, algorithmic bytes, algorithmic flops). */
typedef struct crfp_prof_record {
    char name[48];
    int launches;
    double total_ms;
    double bytes;
    double flops;
} crfp_prof_record;
int crfp_prof_enable(int on);
int crfp_prof_reset(void);
int crfp_prof_report(crfp_prof_record* out, int cap);
/* diagnostic: how many per-host-thread side-stream tables the fp32 CRFP_DSV engine holds (tables of exited threads are
 * reused by new threads, so a thread-per-request host keeps this at its peak thread count; crfp_shutdown() empties them) */
int crfp_debug_side_tables(void);

/* debug: copy a named intermediate of the last crfp_dsv_* call out of the workspace (Q4 layout
 * converted to NCHW).  Used by the parity tests to bisect; returns CRFP_E_BADARG for unknown names. */
int crfp_dsv_debug_fetch(const char* name, int t, int h, int w, const void* workspace, float* out_nchw,
                         int* c_out, int* h_out, int* w_out, void* stream);

/* ---- The benchmark-only regional wiring, model/CRFP_runtime.py::MRCF_simple_v18.forward(lrs, fvs, warp_size) (:8469-8664;
 * built and timed by the reference's test_runtime.py:41,142) as ONE call per clip.  mid_channels = 32, split_ratio = 3, offset_prop.
 * Same conventions as the CRFP_DSV engine above: parameters as CRFP_RT_NUM_PARAMS device pointers in state_dict order
 * (crfp_rt_param_name), packed once, one workspace, everything enqueued on `stream`.  FNet, the warps and the four DCNs only see the
 * top-left (wp_h, wp_w) window of the 8x frame (multiples of 8, >= 64, inside the frame); fvs is the (fh, fw) fovea crop the
 * reference feeds twice to encoder_hr (:8507) and fuses into the top-left (fh, fw) pixels (:8645-8648).
 * lrs[t,3,h,w], fvs[t,3,fh,fw], out[t,3|1,8h,8w] fp32.  flags: CRFP_DSV_Y_ONLY, CRFP_DSV_SINGLE_STREAM.  Default (split-fp16) precision
 * only: CRFP_DSV_STRICT_F32 is refused.  State kept between calls: three more non-blocking streams + events per (host thread, device)
 * under the same contract as the CRFP_DSV side stream (the three levels of a frame and the next frame's state-independent work run
 * beside the recurrent chain; joined before the call returns; CRFP_SIDE_STREAM=0 / CRFP_DSV_SINGLE_STREAM: none; crfp_shutdown() frees them);
 * the overflow word sits at byte 0 of the workspace and poisons the output like the CRFP_DSV engine's. */
#define CRFP_RT_NUM_PARAMS 158
const char* crfp_rt_param_name(int index);
int crfp_rt_param_numel(int index, int y_only);
size_t crfp_rt_packed_weight_bytes(int y_only);
int crfp_rt_pack_weights(const float* const* params, int y_only, void* packed, size_t packed_bytes, void* stream);
size_t crfp_rt_workspace_bytes(int t, int h, int w, int fh, int fw, int wp_h, int wp_w);   /* 0: bad geometry (crfp_last_error) */
int crfp_rt_forward_clip(const void* packed, int flags, const float* lrs, const float* fvs, float* out, int t, int h, int w,
                         int fh, int fw, int wp_h, int wp_w, void* workspace, size_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif
