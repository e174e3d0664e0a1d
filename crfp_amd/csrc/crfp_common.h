// Internal declarations shared by the HIP translation units of libcrfp_hip.so (gfx950 only).
//
// Activation layout ("Q4", channel-quad planes): a tensor with C channels is stored as
//   [N][ceil(C/4)][H][W][4] float32
// i.e. 4 consecutive channels of one pixel form one aligned 16-byte element.  Everything the
// irregular gathers (flow_warp, DCNv2) touch is then a single 16-B vector load per bilinear
// corner -- a DCNv2 deformable group of the 32-channel layers (32/8 = 4 channels) and the whole
// 4-channel 8x-resolution state are exactly one quad -- and the fp32 MFMA convolutions read their
// B operand (4 K-channels of a pixel) with one ds_read_b128 and store 4 output channels of a
// pixel with one global_store_dwordx4.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/crfp_hip.h"

// ------------------------------------------------------------------ storage type of activations in HBM
// conv_mfma / conv_narrow / gather / resample / engine are compiled TWICE (csrc/Makefile): as they are (namespace crfp,
// act_t = float: the fp32 path of round 1) and with -DCRFP_ACT_BF16 (namespace crfp_bf16, act_t = __bf16: BASELINE configs
// 3-5, "bf16 storage").  In the bf16 build every Q4 / P4 activation tensor and the recurrent state hold bf16 -- same
// element counts and index arithmetic, 8-byte instead of 16-byte pixel quads -- all arithmetic stays fp32 (MFMA
// accumulators, bilinear weights, activations), and everything that is a coordinate stays fp32 in memory: flow fields,
// DCN offsets and masks, the API tensors (lrs / fvs / out).  Pointers in the plan structs stay `const float*` for both
// builds (opaque: SRC_NCHW / SRC_FLOW2 sources really are float); kernels cast activation pointers to act_t.
#if defined(CRFP_ACT_BF16) && defined(CRFP_LAB)
#undef CRFP_LAB   // the lab experiments exist for the fp32 build only
#endif
#ifdef CRFP_ACT_BF16
#define CRFP_NS crfp_bf16
#define CRFP_API(name) name##_bf16
#else
#define CRFP_NS crfp
#define CRFP_API(name) name
#endif

namespace crfp {   // compiled once (runtime.hip): errors, per-kernel timing, environment
void set_error(const char* fmt, ...);
struct ProfScope {
    ProfScope(const char* name, hipStream_t s, double bytes, double flops);
    ~ProfScope();
    hipStream_t s_;
    int slot_;
};
bool prof_enabled();
bool precision_env_strict(int family);   // CRFP_PRECISION=f32 in the environment (lab library: or the round-1 knob of family 0 = conv, 1 = DCN)
inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
// float-tensor helpers of the fp32 build that the bf16 engine borrows for its float tensors (flow fields, masks)
int launch_upsample_q4(const float* x, long long xb, float* out, long long ob, int N, int nq, int H, int W, int OH,
                       int OW, float sh, float sw, float mul, hipStream_t s, int out_bgroup);
int launch_upflow(const float* flow_q4, long long fb, float* out_nhw2, long long ob, int N, int H, int W, int r,
                  hipStream_t s);
int launch_fg_prep(const uint8_t* fg, float* fg2, int H8, int W8, hipStream_t s);
int launch_q4_to_nchw(const float* x, float* out, int N, int C, int H, int W, int pad, hipStream_t s);
}  // namespace crfp

namespace CRFP_NS {
using crfp::set_error;
using crfp::ProfScope;
using crfp::prof_enabled;
using crfp::precision_env_strict;
using crfp::align_up;

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float cf32x4 __attribute__((ext_vector_type(4)));
typedef float cf32x2 __attribute__((ext_vector_type(2)));
typedef unsigned cu32x2 __attribute__((ext_vector_type(2)));

#ifdef CRFP_ACT_BF16
typedef __bf16 act_t;
typedef __bf16 cbf16x2 __attribute__((ext_vector_type(2)));
constexpr bool kActBf16 = true;
// 4 bf16 (one pixel quad, 8 bytes) <-> 4 fp32.  bf16 -> fp32 is a shift; fp32 -> bf16 rounds to nearest even (v_cvt_pk_bf16_f32)
__device__ __forceinline__ cf32x4 quad_from_bits(cu32x2 b) {
    return cf32x4{__builtin_bit_cast(float, b.x << 16), __builtin_bit_cast(float, b.x & 0xffff0000u),
                  __builtin_bit_cast(float, b.y << 16), __builtin_bit_cast(float, b.y & 0xffff0000u)};
}
__device__ __forceinline__ cu32x2 quad_to_bits(cf32x4 v) {
    const cbf16x2 lo = __builtin_convertvector(cf32x2{v.x, v.y}, cbf16x2), hi = __builtin_convertvector(cf32x2{v.z, v.w}, cbf16x2);
    return cu32x2{__builtin_bit_cast(unsigned, lo), __builtin_bit_cast(unsigned, hi)};
}
__device__ __forceinline__ cf32x4 ldq(const act_t* p) { return quad_from_bits(*reinterpret_cast<const cu32x2*>(p)); }
__device__ __forceinline__ void stq(act_t* p, cf32x4 v) { *reinterpret_cast<cu32x2*>(p) = quad_to_bits(v); }
#else
typedef float act_t;
constexpr bool kActBf16 = false;
__device__ __forceinline__ cf32x4 ldq(const act_t* p) { return *reinterpret_cast<const cf32x4*>(p); }
__device__ __forceinline__ void stq(act_t* p, cf32x4 v) { *reinterpret_cast<cf32x4*>(p) = v; }
#endif
constexpr int kQuadBytes = 4 * (int)sizeof(act_t);   // bytes of one pixel quad in HBM (16 / 8)
__host__ __device__ __forceinline__ const act_t* as_act(const float* p) { return reinterpret_cast<const act_t*>(p); }
__host__ __device__ __forceinline__ act_t* as_act(float* p) { return reinterpret_cast<act_t*>(p); }

// ------------------------------------------------------------------ conv plan
enum SrcKind : int {
    SRC_Q4 = 0,      // Q4 tensor, nch channels
    SRC_NCHW = 1,    // NCHW planes (API tensors / LR frames), nch channels
    SRC_UNSHUF4 = 2, // Q4 tensor at 4x resolution read through pixel_unshuffle(4): nch = 16 * C_hi
    SRC_FLOW2 = 3,   // [H][W][2] (dx,dy) pairs -> quad (dx,dy,0,0)
    SRC_ZERO = 4,    // padding quad(s)
    SRC_S3 = 5,      // pre-split fp16 pair image written by a producing conv (ConvArgs::s3_dst): 2 x nch/8 planes of
                     // [H][W] 16-byte elements = 8 channels of fp16; planes [0, nch/8) hold x0 = fp16(x), planes
                     // [nch/8, nch/4) hold x1s = fp16((x - x0) * 2^11).  Same bytes per pixel as the fp32 Q4 tensor it
                     // replaces; the consuming conv copies it to LDS instead of converting (nch % 16 == 0, chunk aligned)
    SRC_NCHW_SHIFT = 6 // NCHW planes read at (y + sy, x + sx), zero outside the image, optional ReLU on the way in; ConvSrc::rsv packs
                     // (sy + 8) | (sx + 8) << 4 | relu << 8.  Nine of them over one tensor turn a 7x7 convolution into a 3x3 one
                     // (spynet.hip); fp32-MFMA kernel only
};

enum StoreMode : int {
    ST_Q4 = 0,      // Q4 destination(s), optional channel-quad ranges to different tensors
    ST_PS = 1,      // pixel_shuffle(r) fused into the store, Q4 destination at (H*r, W*r)
    ST_NCHW = 2,    // NCHW planes
    ST_OFFMASK = 3, // DCN offset/mask epilogue: quads < n_off_quads: 10*tanh + flow(y,x); rest: sigmoid
    ST_DCNFUSE = 4  // pack-only mode: the 216 offset / mask rows in the register order of dcn_fused_kernel (gather.hip) -- each
                    // lane half receives (dy, dx, mask) of its 36 sampling positions as accumulator slot 3*p + c of 7 x 16
};

struct ConvSrc {
    const float* p;
    long long bstride;  // floats between batch items
    int kind;
    int nch;            // reference channels contributed to the concat
    int nq;             // K-quads occupied
    int cbase;          // first reference input channel of this source in the conv weight
    int pad;            // 1: planes are (H+1) x (W+1) with a zero pad row / column ("P4", gather sources)
    int rsv;
};

struct ConvDst {
    float* p;           // first destination plane for cout-quad q0
    long long bstride;
    int q0, q1;         // cout-quad range [q0,q1)
    int pad, rsv;       // 1: destination planes are (H+1) x (W+1) (pad row/column never written)
};

#define CRFP_MAX_SRC 9
#define CRFP_MAX_DST 3
#define CRFP_MAX_KQ 80   // K-quads per conv the per-quad descriptor table can hold (Cin <= 320)

// per-K-quad load descriptor (filled on the host at launch): element (gy,gx) of the quad is the 16 bytes
// at base + n*bstride + gy*rs + gx*cs (floats); mask selects the components that really exist
struct QuadDesc {
    const float* base;
    long long bstride;
    int rs, cs, mask, rsv;
};

struct ConvArgs {
    ConvSrc src[CRFP_MAX_SRC];
    ConvDst dst[CRFP_MAX_DST];
    const float* wpk;   // packed weights (see pack_index)
    const void* wsplit; // split-bf16 packed weights (3 bf16 images) or null -> fp32 MFMA path
    const void* wsplit16; // split-fp16 packed weights (2 fp16 images, second scaled by 2^11); set by the launcher
    const void* wsplit_sa; // single-accumulator split-fp16 weights (3 fp16 images A / B / C, conv3x3_split8_kernel); set by the launcher
    long long* stamps;  // diagnostic builds only: per-block phase cycle sums (null in production)
    QuadDesc qd[CRFP_MAX_KQ];
    const float* bpk;   // packed bias [ctiles*32]
    const float* resid; // Q4, same cout-quad indexing, added after activation
    const float* flow;  // [H][W][2] for ST_OFFMASK
    long long resid_bstride, flow_bstride;
    int nsrc, ndst;
    int kq;             // total K quads (even)
    int cin_total;      // Cin of the reference weight tensor
    int cout;           // Cout of the reference weight tensor
    int ctiles;         // ceil(packed rows / 32)
    int N, H, W;
    int act;
    float post_scale;
    int store, ps_r;
    int n_off_quads;
    int dstH, dstW;
    float* s3_dst;      // ST_Q4 only: also (or, with ndst == 0, only) write the output as an SRC_S3 image (cout % 8 == 0)
    long long s3_bstride;
    // fp16-operand range guard of the split-fp16 scheme: every kernel whose output can become an fp16 operand (split convs,
    // dcn_g8, the state-producing stencil) ORs 1 into *ovf when it stores |v| >= 65504; the output head turns the frame into
    // NaN when the word is set (sticky per clip / stream).  Null: not tracked (per-op API, strict fp32).
    unsigned* ovf;
    int ovf_div, ovf_add;   // which status word batch item n of the launch raises: ovf[(n + ovf_add) / ovf_div]; ovf_div == 0: ovf[0] (ovf_word())
    int ovf_skip0;          // 1: items with (n + ovf_add) % ovf_div == 0 raise nothing (rounds 4-5: FNet's never-read pairs that straddled two clips; the engine
                            // no longer launches those pairs -- src_bgroup below -- and leaves this 0)
    int strict;         // 1: plain fp32 MFMA for this launch (CRFP_DSV_STRICT_F32)
    int dst_f32;        // bf16 build, ST_Q4, one destination: store float quads (FNet's flow output stays fp32)
    int src_bgroup;     // > 0: batch item n of the launch READS source item n + n / src_bgroup (destinations, residual, flow and status words keep n):
                        // FNet's first conv over the B * (t - 1) frame pairs of a lock-step batch -- pair n = (clip n / (t - 1), frame n % (t - 1) + 1)
                        // of the flattened [B * t] frame sequence, skipping the B - 1 pairs that would straddle two clips (round 6)
    int ksplit;         // fp32-MFMA kernel only: > 0 = blockIdx.z enumerates (batch item, K slice): slice z % ksplit covers K-quads
                        // [slice * kq / ksplit, ...), reads batch item z / ksplit and writes "batch item" z of the destination (partial sums,
                        // added up by the caller).  SPyNet's coarse pyramid levels: one workgroup per shifted view instead of one for all nine
};

bool conv_s3_supported();   // the selected conv kernels consume SRC_S3 sources (default f16x3 path of the fp32 build only)

// packed row (0..ctiles*32) -> reference output channel, or -1 (padding)
__host__ __device__ inline int conv_row_to_cout(int row, int cout, int store, int ps_r) {
    if (store == ST_PS) {
        const int r2 = ps_r * ps_r, co_n = cout / r2;
        const int cq = row >> 2, rr = row & 3;
        const int Q = cq / r2, s = cq - Q * r2;
        const int co = 4 * Q + rr;
        return co < co_n ? co * r2 + s : -1;
    }
    if (store == ST_DCNFUSE) {   // MFMA row 8q + 4h + r of cout tile T = accumulator register 4q + r of lane half h
        const int T = row >> 5, r = row & 31, h = (r >> 2) & 1;
        const int s = 16 * T + 4 * (r >> 3) + (r & 3);       // slot: position p = s / 3 of the half, component s % 3
        if (s >= 108) return -1;
        const int p = s / 3, c = s - 3 * p, P = 36 * h + p;  // P = deformable group * 9 + tap
        return c == 2 ? 144 + P : 2 * P + c;                 // reference channel order [offset (dy, dx) x 72 | mask x 72]
    }
    return row < cout ? row : -1;
}

// number of packed rows needed
__host__ __device__ inline int conv_packed_rows(int cout, int store, int ps_r) {
    if (store == ST_PS) {
        const int r2 = ps_r * ps_r, co_n = cout / r2;
        return ((co_n + 3) / 4) * r2 * 4;
    }
    if (store == ST_DCNFUSE) return 224;
    return cout;
}

// (source-local K-quad, component) -> source-local reference channel, or -1
__host__ __device__ inline int conv_k_to_cin(int kind, int nch, int kql, int comp) {
    switch (kind) {
        case SRC_Q4:
        case SRC_S3:
        case SRC_NCHW_SHIFT:
        case SRC_NCHW: {
            const int c = 4 * kql + comp;
            return c < nch ? c : -1;
        }
        case SRC_UNSHUF4: {  // hi-res quad Qp, sub-position ij: unshuffled channel (4*Qp+comp)*16 + ij
            const int Qp = kql >> 4, ij = kql & 15;
            const int c = (4 * Qp + comp) * 16 + ij;
            return c < nch ? c : -1;
        }
        case SRC_FLOW2:
            return comp < 2 ? comp : -1;
        default:
            return -1;
    }
}

__host__ __device__ inline int src_quads(int kind, int nch) {
    switch (kind) {
        case SRC_Q4:
        case SRC_S3:
        case SRC_NCHW_SHIFT:
        case SRC_NCHW: return (nch + 3) / 4;
        case SRC_UNSHUF4: return ((nch / 16 + 3) / 4) * 16;
        case SRC_FLOW2: return 1;
        default: return nch;  // SRC_ZERO: nch carries the quad count
    }
}

// ------------------------------------------------------------------ narrow (VALU stencil) conv plan
enum NarrowEpi : int {
    NE_PLAIN = 0,    // Q4 quad out (cout <= 4), act, optional residual
    NE_BLEND = 1,    // state = lrelu(mask ? conv : centre value of source 0)  (conv_tttf + fovea blend)
    NE_LAST = 2,     // NCHW out (3 or 1 planes) = conv + base (x8 bilinear LR from a Q4 quad)
    NE_OFFMASK3 = 3  // (10*tanh(o0)+fy, 10*tanh(o1)+fx, sigmoid(o2), 0)  (dcn_3 shared offset/mask)
};

struct NarrowArgs {
    ConvSrc src[3];
    const float* wpk;  // [tap][kq][comp][4]
    const float* bpk;  // [4]
    float* dst;
    const float* resid;
    const float* flow;
    const float* base;     // NE_LAST: Q4 quad holding the x8 bilinear LR (null: recompute it from base_lr)
    const float* base_lr;  // NE_LAST with base == null: the LR frame [3][H/8][W/8] fp32 (base_bstride applies)
    const uint8_t* mask;
    long long dst_bstride, resid_bstride, flow_bstride, base_bstride, mask_bstride;
    int nsrc, kq, cin_total, cout;
    int N, H, W;
    int act, epi, y_only;
    float post_scale;
    int dst_pad, rsv;
    long long* stamps;  // diagnostic builds (-DCRFP_NARROW_STAMPS) only
    unsigned* ovf;      // fp16-operand range guard (see ConvArgs::ovf): NE_BLEND raises it, NE_LAST poisons the frame when set
    int ovf_div, ovf_add;   // see ConvArgs
    const uint8_t* gate;    // mask gate (launch_mask_gate): 4 flag bytes per 64 x 16 tile of the map, null = dense launch
    long long gate_bstride; // bytes between batch items
    int gate_h;             // which flag this launch reads: "a mask pixel within gate_h tiles of the tile" (0 .. 3)
    // NE_PLAIN, round 6: optional second destination -- P4 planes ((H+1) x (W+1)) that receive lrelu_0.1 of what dst receives, under the range
    // guard `ovf`.  The new recurrent state wherever the fovea mask is clear (model/CRFP.py:1674-1675 with mk = 0), written by the epilogue of
    // forward_resblocks_3's last conv instead of by a separate streaming pass over the 8x map (launch_lrelu_q4_to_p4).
    float* dst2;
    long long dst2_bstride;
};

// Activations of the offset/mask heads (DCN modules, model/CRFP.py:338-340): hardware exp2/rcp, ~1 ulp each.
__device__ __forceinline__ float fast_sigmoid(float v) { return __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }
// 10*tanh(v) + f as (10 + f) - 20 / (1 + e^(2v)), c10f = 10 + f: mul, exp, add, rcp, fma.  e^(2v) = inf / 0 at the ends
// gives f + 10 / f - 10 exactly; absolute error ~2e-6 px.
__device__ __forceinline__ float tanh10_plus(float v, float c10f) {
    const float r = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v * 2.8853900817779268f));
    return __builtin_fmaf(r, -20.0f, c10f);
}

// One status word per clip of a lock-step batch (engine.hip): batch item n of a launch belongs to clip (n + add) / div -- div = 1 for the
// per-frame launches (n = clip), div = t for the clip-level stages that run over the flattened [B * t] frame sequence; div = 0 keeps
// the single word of a one-clip call.  A clip's overflow then poisons that clip's frames only, as with one call per clip.
__device__ __forceinline__ unsigned* ovf_word(unsigned* base, int div, int add, int n, int skip0 = 0) {
    if (!base) return nullptr;
    if (!div) return base;
    const int g = n + add, q = g / div;
    return (skip0 && g == q * div) ? nullptr : base + q;
}

// What a K-sliced conv (launch_conv_ksplit, conv_mfma.hip) would have stored at float offset `off` of batch item n: the sum of its ks partial
// tensors in slice order (the bias rides in slice 0), the activation (NONE / RELU / LRELU(0.1) as the conv epilogue's max(v, slope v)), rounded to
// the storage type; a value an fp16 operand cannot hold raises `ovfw` as the epilogue's guard would.  part: float Q4, pb floats per (item, slice).
__device__ __forceinline__ cf32x4 ksplit_load(const float* __restrict__ part, long long pb, int ks, int n, long long off, int act, unsigned* ovfw) {
    const float* p = part + (long long)n * ks * pb + off;
    cf32x4 v = *reinterpret_cast<const cf32x4*>(p);
    for (int k = 1; k < ks; ++k) v += *reinterpret_cast<const cf32x4*>(p + (long long)k * pb);
    const float slope = act == CRFP_ACT_RELU ? 0.0f : (act == CRFP_ACT_LRELU01 ? 0.1f : 1.0f);
    v = cf32x4{fmaxf(v.x, slope * v.x), fmaxf(v.y, slope * v.y), fmaxf(v.z, slope * v.z), fmaxf(v.w, slope * v.w)};
    const float vmax = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
    if (ovfw && !(vmax < 65504.0f)) atomicOr(ovfw, 1u);
#ifdef CRFP_ACT_BF16
    v = quad_from_bits(quad_to_bits(v));
#endif
    return v;
}
// the K slices a source tensor arrives in (null part: an ordinary Q4 tensor)
struct KsIn {
    const float* part = nullptr;   // float Q4 partial tensors, item n slice k at (n * ks + k) * pb
    long long pb = 0;
    int ks = 0, act = 0;
    unsigned* ovf = nullptr;
    int ovf_div = 0, ovf_add = 0;
};

// ------------------------------------------------------------------ XCD-aware tile order
// Workgroups are dealt round-robin to the 8 XCDs (linear id % 8), each with its own L2.  With the natural order two
// neighbouring tiles (which share halo rows / gathered lines) always sit on different XCDs and both fetch the shared lines
// from HBM.  This maps linear workgroup id b of `total` to a tile id such that XCD x walks the contiguous band
// [x*q + min(x,r), ...) of the tile list (q = total/8, r = total%8): a bijection on [0, total).
__device__ __forceinline__ int xcd_band_tile(int b, int total) {
#ifdef CRFP_NO_XCD_BAND
    return b;   // A/B builds only
#endif
    const int q = total >> 3, r = total & 7, x = b & 7, i = b >> 3;
    return x * q + min(x, r) + i;
}

// ------------------------------------------------------------------ errors
#define CRFP_CHECK_LAUNCH()                                     \
    do {                                                        \
        hipError_t e__ = hipGetLastError();                     \
        if (e__ != hipSuccess) {                                \
            set_error("%s:%d launch failed: %s", __FILE__, __LINE__, hipGetErrorString(e__)); \
            return (int)e__;                                    \
        }                                                       \
    } while (0)

// ------------------------------------------------------------------ launchers (one per kernel family)
// conv_mfma.hip
size_t conv_packed_weight_floats(const ConvArgs& a);
int launch_conv_pack(const ConvArgs& a, const float* w_oihw, const float* bias, const float* w2, const float* bias2,
                     int cout_split, float* wpk, float* bpk, hipStream_t s);
int launch_conv_mfma(const ConvArgs& a, const char* name, hipStream_t s);
// two convs of the same shape (N, H, W, cout tiles) in one launch where the build has the kernel for it (fp32: conv3x3_split_dual_kernel), else two launches
int launch_conv_mfma_dual(const ConvArgs& a0, const char* name0, const ConvArgs& a1, const char* name1, const char* name_both, hipStream_t s);
#ifdef CRFP_ACT_BF16
// bf16 build: conv a (32 couts, no other reader of its output) feeding conv b (32 -> 32) in ONE launch, the tensor between them in LDS
int launch_conv_pair(const ConvArgs& a, const ConvArgs& b, const char* name, hipStream_t s);
#endif
size_t conv_split_weight_bytes(const ConvArgs& a);
size_t conv_split16_offset_bytes(const ConvArgs& a);   // offset of the fp16 pair (bf16 build: the bf16) image inside the split weight block
int launch_conv_pack_split(const ConvArgs& a, const float* w_oihw, const float* w2, int cout_split, void* wsplit,
                           hipStream_t s);
// conv_narrow.hip
size_t narrow_packed_weight_floats(const NarrowArgs& a);
int launch_narrow_pack(const NarrowArgs& a, const float* w_oihw, const float* bias, const float* w2, const float* bias2,
                       int cout_split, float* wpk, float* bpk, hipStream_t s);
int launch_narrow(const NarrowArgs& a, const char* name, hipStream_t s);
// a (NE_PLAIN, no residual) feeding b (one input quad = a's output; NE_PLAIN [+ residual] or NE_OFFMASK3) in one pass
int launch_narrow_pair(const NarrowArgs& a, const NarrowArgs& b, const char* name, hipStream_t s);
// three stencils in one pass (conv_narrow.hip): a -> b -> c; res: c adds a's output (residual block; the only form of the bf16 build)
int launch_narrow_chain(const NarrowArgs& a, const NarrowArgs& b, const NarrowArgs& c, bool res, const char* name, hipStream_t s);
// gather.hip
// src_pad = 1: x is P4 (padded planes, zero pads): validity logic replaced by clamping + hardware range check
// N > 1: batch strides in elements of each tensor's own type (activations: act_t, flow: float)
struct WarpDualStrides { long long xa = 0, xb = 0, flow = 0, outa = 0, outb = 0; };
int launch_flow_warp_p4_dual_8_6(const float* xa, const float* xb, const float* flow, float* outa, float* outb, int H, int W,
                                 hipStream_t s, int N = 1, const WarpDualStrides& bs = WarpDualStrides());
int launch_flow_warp_q4(const float* x, long long xb, const float* flow, long long fb, float* out, long long ob,
                        int N, int nq, int H, int W, int border, int src_pad, hipStream_t s);
int launch_dcn_g8(const float* x, long long xb, const float* offmask, long long omb, const float* wpk,
                  const float* bias, float* out, long long ob, int N, int H, int W, hipStream_t s, bool f16 = false,
                  unsigned* ovf = nullptr, int ovf_div = 0);
int launch_dcn_g8_pack(const float* w_oihw, float* wpk, hipStream_t s, bool f16 = false);  // fp32 [36][2][32][4] or split-fp16 image
bool dcn_g8_use_f16();   // engine: split-fp16 DCN GEMM unless CRFP_DCN_MODE=f32
// offset / mask conv (32 -> 216, model/CRFP.py:337-340) + dcn_g8 in ONE kernel: the 199 MB offset / mask tensor never exists
struct DcnFuseArgs {
    const float* feat;      // the DCN offset feature (32 channels): SRC_S3 image in the fp32 build, bf16 Q4 in the bf16 build
    long long feat_b;       // floats (fp32 build) / elements (bf16 build) between batch items
    const float* flow;      // [H][W][2] (x, y) added to the offsets as (y, x)
    long long flow_b;
    const void* wconv;      // offset / mask conv packed with ST_DCNFUSE rows: fp16 pair image (fp32 build) / bf16 image
    const float* bconv;     // its 224 packed biases
    const float* x;         // P4 features to sample (32 channels = 8 deformable groups)
    long long xb;
    const float* wdcn;      // split-fp16 DCN weight image (launch_dcn_g8_pack(f16 = true))
    const float* bdcn;
    float* out;             // aligned features, Q4
    long long ob;
    int N, H, W;
    unsigned* ovf;
    int ovf_div;            // 0: one status word; 1: batch item n raises ovf[n]
    int probe;              // lab library only: timing experiments
};
bool dcn_fused_enabled();   // default on; CRFP_DCN_FUSED=0 keeps the two-kernel path
int launch_dcn_fused(const DcnFuseArgs& a, hipStream_t s);
int launch_dcn3(const float* x, long long xb, const float* offmask3, long long omb, const float* w_oihw,
                const float* bias, float* out, long long ob, int N, int H, int W, hipStream_t s);
int launch_dcn3_fused(const float* x, long long xb, const float* g2, long long gb, const float* flow, const float* wom, const float* bom,
                      const float* w_oihw, const float* bias, float* out, long long ob, int N, int H, int W, hipStream_t s,
                      long long flow_b = 0);
int launch_dcn_generic(const float* x, const float* offset, const float* mask, const float* w, const float* b,
                       float* out, int N, int cin, int cout, int H, int W, int dg, hipStream_t s);
// resample.hip
// ovf: the range guard on an API tensor that becomes an fp16 operand as it is (the LR frames): item n raises ovf[n / ovf_div] (ovf_div 0: ovf[0])
// when a value is >= 65504, inf or NaN -- a producer-side check cannot see those, and inf operands turn into NaN, which v_max drops
int launch_nchw_to_q4(const float* x, float* out, int N, int C, int H, int W, int pad, hipStream_t s, unsigned* ovf = nullptr, int ovf_div = 0);
int launch_q4_to_nchw(const float* x, float* out, int N, int C, int H, int W, int pad, hipStream_t s);
int launch_upsample_q4(const float* x, long long xb, float* out, long long ob, int N, int nq, int H, int W, int OH,
                       int OW, float sh, float sw, float mul, hipStream_t s, int out_bgroup);
int launch_upsample_nchw(const float* x, float* out, int N, int C, int H, int W, int OH, int OW, float sh, float sw,
                         float mul, hipStream_t s);
int launch_upflow(const float* flow_q4, long long fb, float* out_nhw2, long long ob, int N, int H, int W, int r,
                  hipStream_t s);
int launch_avgpool2_nchw(const float* x, float* out, int N, int C, int H, int W, hipStream_t s);
int launch_avgpool2_q4(const float* x, long long xb, float* out, long long ob, int N, int nq, int H, int W,
                       hipStream_t s);
// K slices for small maps (round 6, conv_mfma.hip / resample.hip): the geometry-only rule, the sliced conv, and the three passes that add the slices
int conv_auto_ksplit(int H, int W, int ctiles, int kq);
int launch_conv_ksplit(const ConvArgs& a, const char* name, hipStream_t s);
int launch_ksplit_reduce(const float* part, long long pb, int ks, float* out, long long ob, int N, int nq, int H, int W, int act, unsigned* ovf, int ovf_div,
                         int ovf_add, hipStream_t s);
int launch_avgpool2_q4_ks(const KsIn& in, float* out, long long ob, int N, int nq, int H, int W, hipStream_t s);
int launch_upsample_q4_ks(const KsIn& in, float* out, long long ob, int N, int nq, int H, int W, int OH, int OW, float sh, float sw, float mul,
                          hipStream_t s);
// N > 1: lr_b / fv_b / mk_b = elements between the batch items of the three API tensors (frames of different clips), out_b likewise
// gate (optional, launch_mask_gate): only tiles with a mask pixel within 3 tiles are produced (what encoder_hr's two convs then read)
int launch_hr_prep(const float* lr, const float* fv, const uint8_t* mk, float* out_q4, int h, int w, hipStream_t s, int N = 1,
                   long long lr_b = 0, long long fv_b = 0, long long mk_b = 0, long long out_b = 0, const uint8_t* gate = nullptr,
                   long long gate_b = 0);
int launch_offmask_nchw_to_q4(const float* offset, const float* mask, float* out, int N, int noff, int nmask, int H,
                              int W, hipStream_t s);
int launch_fg_prep(const uint8_t* fg, float* fg2, int H8, int W8, hipStream_t s);
// dst = src * scale(y,x); scale is float [H][W] (scale_f) or u8 [H][W] (scale_u8)
int launch_scale_q4(const float* src, int src_pad, float* dst, int nq, int H, int W, const float* scale_f,
                    const uint8_t* scale_u8, hipStream_t s);
// Per 64 x 16 tile of the [H8][W8] u8 mask (W8 a multiple of 4), 4 flag bytes: byte k != 0 iff a tile at most k tiles away (Chebyshev) holds
// a set mask pixel (k = 0 .. 3).  gate: [N] items of gate_b bytes each (>= 4 * tiles).  The tile grid is conv3x3_narrow_kernel's.
int launch_mask_gate(const uint8_t* mk, long long mk_b, uint8_t* gate, long long gate_b, int N, int H8, int W8, hipStream_t s);
// dst (one P4 quad per item) = lrelu_0.1(src (one Q4 quad)); raises the item's status word like the blend kernel when a value leaves the fp16 range
int launch_lrelu_q4_to_p4(const float* src, long long src_b, float* dst, long long dst_b, int N, int H, int W, unsigned* ovf, int ovf_div, hipStream_t s);
// CRFP_DSV_CRA level fusion: [prop | carry] = mk2 * fused + (1 - mk2) * y, mk2 = the x0.25 bilinear resample of the u8 mask [4H][4W]
int launch_cra_blend(const float* y, long long y_b, const float* fused, long long f_b, const uint8_t* mk, long long mk_b, float* prop,
                     long long prop_b, float* carry, long long carry_b, int N, int H, int W, hipStream_t s);
int launch_psnr_ssim_partial(const float* a, const float* b, const uint8_t* mask, double* acc, int N, int C, int H, int W,
                             float mul, float add, hipStream_t s);
int launch_psnr_partial(const float* a, const float* b, double* acc, int N, int C, int H, int W, hipStream_t s);

}  // namespace CRFP_NS
