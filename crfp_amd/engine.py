"""Python handle on the C++ CRFP_DSV engine (crfp_amd/csrc/engine.hip): owns the packed weights
and the workspace tensors, forwards clips / streamed frames with one C-ABI call each."""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from .ops import _dev, _stream


def param_names(wiring: str = "dsv"):
    """The reference's state_dict keys in the order the pack call takes them: ``dsv`` = CRFP_DSV, ``cra`` = CRFP_DSV_CRA."""
    L = _lib.lib()
    if wiring == "cra":
        return [L.crfp_cra_param_name(i).decode() for i in range(_lib.CRA_NUM_PARAMS)]
    return [L.crfp_dsv_param_name(i).decode() for i in range(_lib.NUM_PARAMS)]


def embed_mid32(state_dict, mid: int, wiring: str = "dsv", y_only: bool = False):
    """A CRFP_DSV / CRFP_DSV_CRA / CRFP_simple / CRFP state_dict of ``mid_channels = mid`` (16: the reference's constructor default, model/CRFP.py:1388,
    and the only width below 32 its `downsample` assertion admits, :206-222) -> the state_dict of the SAME function at mid_channels = 32, which the one-call engine schedules run: every
    tensor of the narrow model sits at fixed channel positions of its 32- (8x maps: 4-) channel twin, the other channels are exact
    zeros (zero weight rows and biases; lrelu / relu / the residual adds keep them zero), and zero weight columns ignore them.  The
    positions: identity, except (a) a residual block's output keeps its [features | carried] split -- the 3 mid / 4 features in
    channels [0, 24), the mid / 4 carried ones from 24 on (model/CRFP.py:1590-1591), and (b) `downsample`'s output, the tensor DCNv2
    samples with 8 deformable groups, puts group g's mid / 8 channels at [4 g, 4 g + mid / 8) -- the engine's one-quad-per-group
    layout.  Pixel-(un)shuffle channel indices c * r^2 + s survive as they are.  Products with the padding are exact zeros, so the
    results equal the narrow model's up to fp32 summation order (tests: the reference's own mid16 goldens)."""
    import torch as _t
    if mid != 16 or wiring not in ("dsv", "cra", "simple", "dense"):
        raise ValueError(f"embed_mid32: mid_channels {mid} / wiring {wiring!r}")
    m, l = mid, mid // 8
    abl = wiring in ("simple", "dense")
    p = m if abl else 3 * m // 4                     # features a level passes on
    ident = lambda n, base=0: [base + j for j in range(n)]   # noqa: E731
    feat = ident(m)
    cur = ident(m) if abl else ident(p) + ident(m - p, 24)
    grp = [4 * (j // l) + j % l for j in range(m)]
    lq = ident(l)
    off = lambda e, base: [base + j for j in e]      # noqa: E731
    ups_rows = [c * 4 + s_ for c in (ident(m) if abl else ident(p)) for s_ in range(4)]
    third2 = off(grp, 64) if wiring == "dense" else []          # CRFP: + the warped previous state (in `downsample`'s layout)
    third8 = off(lq, 8) if wiring == "dense" else []
    # conv stem -> (rows, cols, full cout, full cin)
    table = {}
    for k in range(3):
        d = f"dcn_{k}."
        table[d + "dcn_block.0"] = (feat, cur + off(grp, 32) + [64, 65], 32, 66)
        table[d + "dcn_block.2"] = (feat, feat, 32, 32)
        if k:
            table[d + "conv_fuse"] = (feat, feat + off(feat, 32), 32, 64)
        table[d + "dcn_offset"] = (ident(144), feat, 144, 32)
        table[d + "dcn_mask"] = (ident(72), feat, 72, 32)
        table[d + "dcn"] = (feat, grp, 32, 32)
        r = f"forward_resblocks_{k}."
        table[r + "main.0"] = (cur, cur + off(feat, 32) + third2, 32, 96 if wiring == "dense" else 64)
        table[r + "main.2.0.conv1"] = (feat, cur, 32, 32)
        table[r + "main.2.0.conv2"] = (cur, feat, 32, 32)
    table["dcn_3.upsample.upsample_conv"] = (ident(16 * l), feat, 64, 32)
    table["dcn_3.conv_fuse"] = (lq, lq + off(lq, 4), 4, 8)
    table["dcn_3.dcn_block.0"] = (lq, lq + off(lq, 4) + [8, 9], 4, 10)
    table["dcn_3.dcn_block.2"] = (lq, lq, 4, 4)
    table["dcn_3.dcn_offset"] = ([0, 1], lq, 2, 4)
    table["dcn_3.dcn_mask"] = ([0], lq, 1, 4)
    table["dcn_3.dcn"] = (lq, lq, 4, 4)
    table["encoder_lr.slice1.0"] = (feat, [0, 1, 2], 32, 3)
    table["encoder_lr.slice1.2"] = (feat, feat, 32, 32)
    table["encoder_hr.slice1.0"] = (lq, ident(6), 4, 6)
    table["encoder_hr.slice1.2"] = (lq, lq, 4, 4)
    table["conv_tttf"] = (lq, lq + off(lq, 4), 4, 8)
    table["forward_resblocks_3.main.0"] = (lq, lq + off(lq, 4) + third8, 4, 12 if wiring == "dense" else 8)
    table["forward_resblocks_3.main.2.0.conv1"] = (lq, lq, 4, 4)
    table["forward_resblocks_3.main.2.0.conv2"] = (lq, lq, 4, 4)
    table["downsample.downsample_conv"] = (grp, ident(16 * l), 32, 64)
    table["upsample.upsample_conv"] = (ups_rows, feat, 128 if abl else 96, 32)
    table["upsample_post.upsample_conv"] = (ident(16 * l), ident(p), 64, 32 if abl else 24)
    if wiring == "cra":   # CRFP_DSV_CRA's four-level fovea encoder LTE_simple_hr_ps(l) (4 l channels at 2x) and the per-level fusion convs (m + 4 l -> m)
        q4 = ident(4 * l)
        table["encoder_hr.slice2.1"] = (q4, ident(16 * l), 16, 64)
        for stem in ("slice2.3", "slice3.0", "slice3.2", "slice4.0", "slice4.2", "conv_lv0", "conv_lv1", "conv_lv2"):
            table["encoder_hr." + stem] = (q4, q4, 16, 16)
        table["encoder_hr.conv_lv3"] = (lq, lq, 4, 4)
        for k in range(3):
            table[f"conv_tttf_{k}"] = (cur, cur + off(q4, 32), 32, 48)
    co_last = 1 if y_only else 3
    table["conv_last"] = (ident(co_last), lq, co_last, 4)
    out = {}
    for key, v in state_dict.items():
        stem, _, kind = key.rpartition(".")
        if stem not in table or kind not in ("weight", "bias"):
            out[key] = v            # the flow network has no mid_channels in it
            continue
        rows, cols, co, ci = table[stem]
        v = v.detach().to(_t.float32)
        if kind == "bias":
            if v.numel() != len(rows):
                raise ValueError(f"parameter {key}: {v.numel()} elements, expected {len(rows)} at mid_channels = {mid}")
            b = v.new_zeros(co)
            b[_t.tensor(rows, device=v.device)] = v
            out[key] = b
            continue
        if tuple(v.shape) != (len(rows), len(cols), 3, 3):
            raise ValueError(f"parameter {key}: shape {tuple(v.shape)}, expected {(len(rows), len(cols), 3, 3)} at mid_channels = {mid}")
        w = v.new_zeros(co, ci, 3, 3)
        ri, cj = _t.tensor(rows, device=v.device), _t.tensor(cols, device=v.device)
        w[ri[:, None], cj[None, :]] = v
        out[key] = w
    return out


class DSVEngine:
    """precision: "split" (default: split-fp16 MFMA scheme, fp32-grade, operands must stay below 65504) or "f32" (strict
    fp32 MFMA, CRFP_DSV_STRICT_F32).  on_overflow: what a clip / streamed frame does when the split scheme's range guard
    fires -- "poison" (default, no host sync: the output frames are NaN and ``overflowed()`` tells why), "fallback"
    (synchronise, rerun the call in strict fp32) or "raise" (FloatingPointError)."""

    def __init__(self, state_dict, device, y_only: bool = False, precision: str = "split", on_overflow: str = "poison",
                 storage: str = "f32", mid_channels: int = 32):
        """state_dict: mapping with the reference's CRFP_DSV keys -> tensors (any device).
        mid_channels: 32, or 16 -- the narrower model runs embedded in the 32-channel schedule (``embed_mid32``).
        storage: "f32" (default) or "bf16" -- activations and recurrent state held as bf16 in HBM (crfp_dsv_*_bf16 entry
        points, BASELINE configs 3-5); API tensors, accumulators, flow / offsets / masks stay fp32."""
        if precision not in ("split", "f32") or on_overflow not in ("poison", "fallback", "raise") or storage not in ("f32", "bf16"):
            raise ValueError(f"precision {precision!r} / on_overflow {on_overflow!r} / storage {storage!r}")
        if storage == "bf16" and precision == "f32":
            raise ValueError("storage='bf16' has one precision (bf16 MFMA operands, fp32 accumulate)")
        self.precision, self.on_overflow, self.storage = precision, on_overflow, storage
        self._sfx = "_bf16" if storage == "bf16" else ""
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("crfp_amd.DSVEngine needs a CUDA/HIP device (no CPU path in the product)")
        self.y_only = int(bool(y_only))
        self.single_stream = False   # True: CRFP_DSV_SINGLE_STREAM on every call (no fork onto the library's side stream)
        # stream_frame only, CRFP_DSV_INPUTS_RESIDENT: the caller's lr / fv / mk tensors are complete when stream_frame() is called and are
        # not modified until the stream has drained that call (true for frames that already sit in HBM, e.g. a decoded video held on the
        # device; NOT true when an earlier kernel or copy on the same stream is still producing them).  The library then keeps the previous
        # frame itself and runs the state-independent part of each frame beside the previous frame's recurrent chain.  Same bits.
        self.inputs_resident = False
        # forward() on n > 1 clips: "lockstep" (default) = ONE crfp_dsv_forward_batch call, every layer launched once over all n clips;
        # "loop" = n one-clip calls in turn (rounds 1-3; what "lockstep" is bit-identical to, clip by clip)
        self.batch_mode = "lockstep"
        self._ws = {}
        self._ovf = None             # int32[1] on the device: status words of the last forward()'s clips, OR-ed (no host sync)
        self._stream_ws = None
        self._stream_prev = None
        self._stream_prev_buf = None
        self._stream_resident = False
        self._stream_hw = None
        self.mid_channels = int(mid_channels)
        if self.mid_channels != 32:
            state_dict = embed_mid32(state_dict, self.mid_channels, self.WIRING, bool(y_only))
        self.pack(state_dict)

    WIRING = "dsv"   # which parameter table / entry-point family this handle drives (CRAEngine: "cra", SimpleEngine: "simple", DenseEngine: "dense")
    MODEL_NAME = "CRFP_DSV"

    def _fn(self, name):
        return getattr(_lib.lib(), name + self._sfx)

    def pack(self, state_dict):
        L = _lib.lib()
        names = param_names(self.WIRING)
        missing = [k for k in names if k not in state_dict]
        if missing:
            raise KeyError(f"state_dict lacks {self.MODEL_NAME} parameters: {missing[:4]}{'...' if len(missing) > 4 else ''}")
        keep = []
        ptrs = (C.c_void_p * len(names))()
        numel = getattr(L, f"crfp_{self.WIRING}_param_numel")
        for i, k in enumerate(names):
            t = state_dict[k].detach().to(device=self.device, dtype=torch.float32).contiguous()
            want = numel(i, self.y_only)
            if t.numel() != want:
                raise ValueError(f"parameter {k}: {t.numel()} elements, expected {want}")
            keep.append(t)
            ptrs[i] = t.data_ptr()
        # range of the split-fp16 scheme (conv_mfma.hip, "f16x3s": the sum is kept scaled by 2^11, so 2^11 * w must stay an
        # fp16 value): |w| < 32.  Checked HERE, where the offending tensor can be named; without it an out-of-range weight
        # becomes an inf operand image and only shows up downstream as "an activation overflowed".
        wmax = torch.stack([t.abs().max() for t, k in zip(keep, names) if k.endswith(".weight")]).cpu()
        wnames = [k for k in names if k.endswith(".weight")]
        self._wmax = float(wmax.max())
        self._wmax_name = wnames[int(wmax.argmax())]
        if not torch.isfinite(wmax).all():
            raise ValueError(f"crfp_amd: parameter {wnames[int((~torch.isfinite(wmax)).nonzero()[0])]} holds inf / NaN")
        nbytes = self._fn("crfp_dsv_packed_weight_bytes")(self.y_only)
        self.packed = torch.zeros(nbytes, dtype=torch.uint8, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self._fn("crfp_dsv_pack_weights")(ptrs, self.y_only, self.packed.data_ptr(), nbytes, _stream()),
                       "crfp_dsv_pack_weights")
            torch.cuda.current_stream().synchronize()   # `keep` may be freed after this point

    def _workspace(self, t, h, w, n=1):
        key = (n, t, h, w)
        if key not in self._ws:
            nb = self._fn("crfp_dsv_batch_workspace_bytes")(n, t, h, w)
            if nb == 0:
                raise ValueError(f"unsupported clip shape n={n} t={t} h={h} w={w}")
            ws = torch.empty(nb, dtype=torch.uint8, device=self.device)
            off = self._fn("crfp_dsv_batch_status_offset")(n, t, h, w)
            # a fresh workspace starts with clear status words -- one per clip, so 4 n bytes (crfp_fnet_forward never writes them)
            ws[off:off + max(256, 4 * n)].zero_()
            self._ws = {key: ws}        # keep one shape alive
        return self._ws[key]

    @staticmethod
    def _mask_u8(mks):
        if mks.dtype == torch.bool:
            return mks.contiguous().view(torch.uint8)
        return (mks != 0).contiguous().view(torch.uint8)

    WEIGHT_LIMIT_SPLIT = 32.0   # |w| < 32 for the default split-fp16 convolution scheme (fp32 storage)

    def _flags(self, strict=None):
        strict = (self.precision == "f32") if strict is None else strict
        if not strict and self.storage == "f32" and self._wmax >= self.WEIGHT_LIMIT_SPLIT:
            if self.on_overflow == "fallback":
                return self.y_only | _lib.DSV_STRICT_F32 | (_lib.DSV_SINGLE_STREAM if self.single_stream else 0)
            raise ValueError(f"crfp_amd: weight {self._wmax_name} has max |w| = {self._wmax:.4g} >= {self.WEIGHT_LIMIT_SPLIT:g}, outside "
                             "the operand range of the split-fp16 convolution scheme; run this model with precision='f32' "
                             "(CRFP_DSV_STRICT_F32)")
        return self.y_only | (_lib.DSV_STRICT_F32 if strict else 0) | (_lib.DSV_SINGLE_STREAM if self.single_stream else 0)

    def _status(self, ws, t, h, w, n=1) -> int:
        """OR of the call's status words (one per clip of a lock-step batch).  Synchronises."""
        off = self._fn("crfp_dsv_batch_status_offset")(n, t, h, w)
        return int(ws[off:off + 4 * n].view(torch.int32).max().item())

    def overflowed(self, stream: bool = False) -> bool:
        """True when the split-fp16 range guard fired in the last clip forward (or, stream=True, in the running
        streamed sequence).  Synchronises the device."""
        if stream:
            return self._stream_ws is not None and bool(self._status(self._stream_ws, 1, *self._stream_hw[1:], self._stream_hw[0]) & 1)
        return self._ovf is not None and bool(int(self._ovf.item()) & 1)   # OR over every clip of the last forward()

    def _after(self, ws, key, rerun):
        """on_overflow policy after a call that used workspace `ws`; `rerun(strict=True)` repeats it."""
        if self.on_overflow == "poison" or self.precision == "f32":
            return None
        if not self._status(ws, *key) & 1:
            return None
        if self.on_overflow == "raise":
            raise FloatingPointError("crfp_amd: an activation reached the fp16 operand range (|v| >= 65504) of the split-fp16 "
                                     "convolution scheme; run with precision='f32' (CRFP_DSV_STRICT_F32)")
        return rerun()

    def forward(self, lrs, fvs, mks):
        """lrs[n,t,3,h,w], fvs[n,t,3,8h,8w], mks[n,t,1,8h,8w] (bool) -> [n,t,3|1,8h,8w]."""
        lrs, fvs = _dev(lrs, "lrs"), _dev(fvs, "fvs")
        if not mks.is_cuda:
            raise RuntimeError("crfp_amd: `mks` must be a CUDA/HIP tensor")
        mk8 = self._mask_u8(mks)
        n, t, c, h, w = lrs.shape
        assert c == 3 and tuple(fvs.shape) == (n, t, 3, 8 * h, 8 * w) and tuple(mk8.shape) == (n, t, 1, 8 * h, 8 * w)
        out = torch.empty((n, t, 1 if self.y_only else 3, 8 * h, 8 * w), dtype=torch.float32, device=self.device)
        if self.batch_mode not in ("lockstep", "loop"):
            raise ValueError(f"batch_mode {self.batch_mode!r}")
        lock = n > 1 and self.batch_mode == "lockstep"
        nb = n if lock else 1
        ws = self._workspace(t, h, w, nb)

        def run(b, strict=None):
            if lock:   # all n clips in one call: lrs / fvs / mks / out are the reference's [n, t, ...] tensors as they lie
                _lib.check(self._fn("crfp_dsv_forward_batch")(self.packed.data_ptr(), self._flags(strict), lrs.data_ptr(), fvs.data_ptr(),
                                                              mk8.data_ptr(), out.data_ptr(), n, t, h, w, ws.data_ptr(), ws.numel(),
                                                              _stream()), "crfp_dsv_forward_batch")
            else:
                _lib.check(self._fn("crfp_dsv_forward_clip")(self.packed.data_ptr(), self._flags(strict), lrs[b].data_ptr(),
                                                             fvs[b].data_ptr(), mk8[b].data_ptr(), out[b].data_ptr(), t, h, w,
                                                             ws.data_ptr(), ws.numel(), _stream()), "crfp_dsv_forward_clip")

        off = self._fn("crfp_dsv_batch_status_offset")(nb, t, h, w)
        words = ws[off:off + 4 * nb].view(torch.int32)   # one status word per clip of the call: a clip's overflow poisons that clip only
        with torch.cuda.device(self.device):
            if self._ovf is None:
                self._ovf = torch.zeros(1, dtype=torch.int32, device=self.device)
            for b in range(1 if lock else n):
                run(b)
                # "fallback" under lockstep reruns the whole batch in strict fp32
                self._after(ws, (t, h, w, nb), lambda b=b: run(b, strict=True))
                # every call resets the workspace's status word: keep the OR over the batch (stream-ordered, no host sync)
                self._ovf.copy_(words.max().reshape(1)) if b == 0 else self._ovf.bitwise_or_(words.max().reshape(1))
        return out

    # ---- streaming: one frame per call, state lives in a dedicated workspace
    def clear_states(self):
        self._stream_prev = None

    def stream_frame(self, lr, fv, mk, fg=None):
        """lr[3,h,w], fv[3,8h,8w], mk[1,8h,8w], optional regional mask fg[1,8h,8w] -> [3|1,8h,8w]; the first
        call after clear_states() starts a sequence.  With a leading batch axis -- lr[n,3,h,w], fv[n,3,8h,8w], mk[n,1,8h,8w] -> [n,3|1,8h,8w] --
        the n sequences advance in lock-step in ONE call (crfp_dsv_stream_batch: n <= 32, no fg), per sequence bit-identical to n engines.
        With ``inputs_resident = True`` the library reads lr / fv / mk on its own side stream WITHOUT waiting for the caller's stream
        (that is the point: frame i's flow network runs beside frame i - 1).  The tensors must therefore be complete when this
        method is called: nothing still queued on the current torch stream may be writing them (a non_blocking host-to-device copy, a
        decode or crop kernel, an in-place op) -- synchronise such producers first, or leave the flag off.  Only dtype and contiguity
        can be checked here."""
        if self.inputs_resident:
            # a conversion here would be a kernel on this stream that is still writing the input when the library starts reading it
            for name, t_, dt in (("lr", lr, (torch.float32,)), ("fv", fv, (torch.float32,)), ("mk", mk, (torch.bool, torch.uint8))):
                if not (isinstance(t_, torch.Tensor) and t_.is_cuda and t_.dtype in dt and t_.is_contiguous()):
                    raise ValueError(f"crfp_amd: inputs_resident needs `{name}` as a contiguous device tensor of dtype {dt} (no conversion may run)")
        lr, fv = _dev(lr, "lr"), _dev(fv, "fv")
        mk8 = mk.view(torch.uint8) if (self.inputs_resident and mk.dtype == torch.bool) else (mk if self.inputs_resident else self._mask_u8(mk))
        fg8 = None if fg is None else self._mask_u8(fg)
        batched = lr.dim() == 4
        n = lr.shape[0] if batched else 1
        h, w = lr.shape[-2:]
        if batched and (tuple(fv.shape) != (n, 3, 8 * h, 8 * w) or mk8.numel() != n * 64 * h * w):
            raise ValueError(f"stream_frame: lr {tuple(lr.shape)} / fv {tuple(fv.shape)} / mk {tuple(mk.shape)}: expected [n,3,h,w], [n,3,8h,8w], [n,1,8h,8w]")
        if self._stream_ws is None or self._stream_hw != (n, h, w):
            nb = self._fn("crfp_dsv_batch_workspace_bytes")(n, 1, h, w)
            if nb == 0:
                raise ValueError(f"unsupported streaming shape n={n} h={h} w={w}")
            self._stream_ws = torch.empty(nb, dtype=torch.uint8, device=self.device)
            self._stream_hw = (n, h, w)
            self._stream_prev = None
        first = self._stream_prev is None
        out = torch.empty(((n,) if batched else ()) + (1 if self.y_only else 3, 8 * h, 8 * w), dtype=torch.float32, device=self.device)
        if self.on_overflow == "fallback" and self.precision != "f32":
            raise NotImplementedError("on_overflow='fallback' cannot rewind a streamed sequence: use 'poison' or 'raise'")
        resident = bool(self.inputs_resident)
        if not first and resident != self._stream_resident:
            raise RuntimeError("crfp_amd: inputs_resident changed in the middle of a streamed sequence; call clear_states() first")
        self._stream_resident = resident
        with torch.cuda.device(self.device):
            _lib.check(self._fn("crfp_dsv_stream_batch")(
                self.packed.data_ptr(), self._flags() | (_lib.DSV_INPUTS_RESIDENT if resident else 0), lr.data_ptr(),
                None if (first or resident) else self._stream_prev.data_ptr(), fv.data_ptr(), mk8.data_ptr(),
                None if fg8 is None else fg8.data_ptr(), out.data_ptr(),
                1 if first else 0, n, h, w, self._stream_ws.data_ptr(), self._stream_ws.numel(), _stream()),
                "crfp_dsv_stream_batch")
        self._after(self._stream_ws, (1, h, w, n), None)
        # the reference keeps a COPY of the frame (model/CRFP_test.py:2234-2238, ``.clone()``): a caller that refills one
        # input buffer in place must not change what the next call sees as the previous frame
        if resident:            # the library kept the frame inside the workspace
            self._stream_prev = lr
            return out
        if self._stream_prev_buf is None or self._stream_prev_buf.shape != lr.shape:
            self._stream_prev_buf = torch.empty_like(lr)
        self._stream_prev_buf.copy_(lr)
        self._stream_prev = self._stream_prev_buf
        return out

    def compute_flow(self, cur, prev):
        """FNet(cur, prev): [n,3,h,w] x2 -> [n,2,h,w] (reference CRFP_DSV.compute_flow pairs)."""
        cur, prev = _dev(cur, "cur"), _dev(prev, "prev")
        n, _, h, w = cur.shape
        ws = self._workspace(n + 1, h, w)
        flow = torch.empty((n, 2, h, w), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self._fn("crfp_fnet_forward")(self.packed.data_ptr(), cur.data_ptr(), prev.data_ptr(),
                                                    flow.data_ptr(), n, h, w, ws.data_ptr(), ws.numel(), _stream()),
                       "crfp_fnet_forward")
        return flow

    def debug_fetch(self, name, t, h, w):
        """Copy a named workspace intermediate of the last clip forward to an NCHW tensor (tests only)."""
        L = _lib.lib()
        ws = self._workspace(t, h, w)
        c, hh, ww = C.c_int(), C.c_int(), C.c_int()
        n = self._fn("crfp_dsv_debug_fetch")(name.encode(), t, h, w, ws.data_ptr(), None, C.byref(c), C.byref(hh), C.byref(ww),
                                   _stream())
        if n < 0:
            raise KeyError(name)
        shape = (n, c.value, hh.value, ww.value) if c.value != 2 or "flow" not in name else (n, hh.value, ww.value, 2)
        out = torch.empty(shape, dtype=torch.float32, device=self.device)
        _lib.check(self._fn("crfp_dsv_debug_fetch")(name.encode(), t, h, w, ws.data_ptr(), out.data_ptr(), None, None, None,
                                          _stream()), "crfp_dsv_debug_fetch")
        return out


class RuntimeEngine:
    """Handle on the one-call schedule of the benchmark-only regional wiring (crfp_amd/csrc/engine_rt.hip; reference
    model/CRFP_runtime.py::MRCF_simple_v18.forward, :8469-8664): packed weights + one workspace per geometry, one
    ``crfp_rt_forward_clip`` per clip.  fp32 storage, default (split-fp16) precision; the range guard poisons the output
    frames with NaN like DSVEngine's and ``overflowed()`` tells why."""

    WEIGHT_LIMIT_SPLIT = DSVEngine.WEIGHT_LIMIT_SPLIT

    def __init__(self, state_dict, device, y_only: bool = False):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("crfp_amd.RuntimeEngine needs a CUDA/HIP device (no CPU path in the product)")
        self.y_only = int(bool(y_only))
        self.single_stream = False   # True: CRFP_DSV_SINGLE_STREAM (no work on the library's side streams; same bits)
        self._ws = {}
        self.pack(state_dict)

    @staticmethod
    def param_names():
        L = _lib.lib()
        return [L.crfp_rt_param_name(i).decode() for i in range(_lib.RT_NUM_PARAMS)]

    def pack(self, state_dict):
        L = _lib.lib()
        names = self.param_names()
        missing = [k for k in names if k not in state_dict]
        if missing:
            raise KeyError(f"state_dict lacks MRCF_simple_v18 parameters: {missing[:4]}{'...' if len(missing) > 4 else ''}")
        keep = []
        ptrs = (C.c_void_p * _lib.RT_NUM_PARAMS)()
        for i, k in enumerate(names):
            t = state_dict[k].detach().to(device=self.device, dtype=torch.float32).contiguous()
            want = L.crfp_rt_param_numel(i, self.y_only)
            if t.numel() != want:
                raise ValueError(f"parameter {k}: {t.numel()} elements, expected {want}")
            keep.append(t)
            ptrs[i] = t.data_ptr()
        wnames = [k for k in names if k.endswith(".weight")]
        wmax = torch.stack([t.abs().max() for t, k in zip(keep, names) if k.endswith(".weight")]).cpu()
        if not torch.isfinite(wmax).all():
            raise ValueError(f"crfp_amd: parameter {wnames[int((~torch.isfinite(wmax)).nonzero()[0])]} holds inf / NaN")
        if float(wmax.max()) >= self.WEIGHT_LIMIT_SPLIT:   # same operand range as DSVEngine's default precision; no strict mode here
            raise ValueError(f"crfp_amd: weight {wnames[int(wmax.argmax())]} has max |w| = {float(wmax.max()):.4g} >= "
                             f"{self.WEIGHT_LIMIT_SPLIT:g}, outside the operand range of the split-fp16 convolution scheme")
        nbytes = L.crfp_rt_packed_weight_bytes(self.y_only)
        self.packed = torch.zeros(nbytes, dtype=torch.uint8, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(L.crfp_rt_pack_weights(ptrs, self.y_only, self.packed.data_ptr(), nbytes, _stream()), "crfp_rt_pack_weights")
            torch.cuda.current_stream().synchronize()   # `keep` may be freed after this point

    def _workspace(self, key):
        if key not in self._ws:
            nb = _lib.lib().crfp_rt_workspace_bytes(*key)
            if nb == 0:
                raise ValueError(f"unsupported geometry (t, h, w, fh, fw, wp_h, wp_w) = {key}: {_lib.lib().crfp_last_error_string().decode(errors='replace')}")
            self._ws = {key: torch.empty(nb, dtype=torch.uint8, device=self.device)}   # keep one geometry alive
        return self._ws[key]

    def overflowed(self) -> bool:
        """True when the range guard fired in the last forward (any clip of the batch).  Synchronises the device."""
        return self._ovf is not None and bool(int(self._ovf.item()) & 1)

    _ovf = None

    def forward(self, lrs, fvs, warp_size):
        """lrs[n,t,3,h,w], fvs[n,t,3,fh,fw] (the fovea crop) -> [n,t,3|1,8h,8w]"""
        lrs, fvs = _dev(lrs, "lrs"), _dev(fvs, "fvs")
        n, t, c, h, w = lrs.shape
        fh, fw = fvs.shape[-2:]
        if c != 3 or tuple(fvs.shape[:3]) != (n, t, 3):
            raise ValueError(f"lrs {tuple(lrs.shape)} / fvs {tuple(fvs.shape)}: expected [n,t,3,h,w] and [n,t,3,fh,fw]")
        key = (t, h, w, fh, fw, int(warp_size[0]), int(warp_size[1]))
        ws = self._workspace(key)
        out = torch.empty(n, t, 1 if self.y_only else 3, 8 * h, 8 * w, dtype=torch.float32, device=self.device)
        L = _lib.lib()
        ovf = torch.zeros(1, dtype=torch.int32, device=self.device) if n > 1 else None
        flags = self.y_only | (_lib.DSV_SINGLE_STREAM if self.single_stream else 0)
        with torch.cuda.device(self.device):
            for b in range(n):
                _lib.check(L.crfp_rt_forward_clip(self.packed.data_ptr(), flags, lrs[b].data_ptr(), fvs[b].data_ptr(), out[b].data_ptr(),
                                                  *key, ws.data_ptr(), ws.numel(), _stream()), "crfp_rt_forward_clip")
                if ovf is not None:
                    ovf |= ws[:4].view(torch.int32)
        self._ovf = ovf if ovf is not None else ws[:4].view(torch.int32)
        return out


class CRAEngine(DSVEngine):
    """Handle on the one-call schedule of the reference's CRFP_DSV_CRA wiring (crfp_cra_* entry points, include/crfp_hip.h): the same
    clip / lock-step batch forward, flags, status words and overflow policies as DSVEngine, over this wiring's own packed weights and
    workspace.  Clip forward only: the reference's one-frame-per-call model (model/CRFP_test.py) is the plain CRFP_DSV."""

    WIRING = "cra"
    MODEL_NAME = "CRFP_DSV_CRA"
    _CALLS = ("packed_weight_bytes", "pack_weights", "batch_workspace_bytes", "batch_status_offset", "forward_batch")

    def _fn(self, name):
        fam = f"crfp_{self.WIRING}_"
        if name == "crfp_dsv_forward_clip":   # batch_mode "loop": the n = 1 form of the batch call
            f = getattr(_lib.lib(), fam + "forward_batch" + self._sfx)
            return lambda packed, flags, lrs, fvs, mks, out, t, h, w, ws, nb, stream: f(packed, flags, lrs, fvs, mks, out, 1, t, h, w, ws, nb, stream)
        if not name.startswith("crfp_dsv_") or name[len("crfp_dsv_"):] not in self._CALLS:
            raise NotImplementedError(f"crfp_amd: {name} has no {self.MODEL_NAME} counterpart (clip forward only)")
        return getattr(_lib.lib(), fam + name[len("crfp_dsv_"):] + self._sfx)

    def stream_frame(self, *a, **k):
        raise NotImplementedError("crfp_amd: the one-frame-per-call schedule exists for the plain CRFP_DSV wiring only")

    def compute_flow(self, cur, prev):
        raise NotImplementedError(f"crfp_amd: use the model's flow network modules ({self.MODEL_NAME}.compute_flow)")

    def debug_fetch(self, name, t, h, w):
        raise NotImplementedError("crfp_amd: debug_fetch reads the CRFP_DSV workspace layout")


class SimpleEngine(CRAEngine):
    """Handle on the one-call schedule of the reference's CRFP_simple wiring ("v13", model/CRFP.py:816-1099) at mid_channels = 32 with hr_dcn and
    offset_prop on (crfp_simple_* entry points, include/crfp_hip.h): CRFP_DSV's state_dict keys, clip / lock-step batch forward, flags, status
    words and overflow policies; its own packed weights and workspace."""

    WIRING = "simple"
    MODEL_NAME = "CRFP_simple"


class DenseEngine(SimpleEngine):
    """The same for the reference's class CRFP ("v15", model/CRFP.py:1101-1385; crfp_dense_* entry points): CRFP_simple with the warped previous
    state as a third input of every residual block."""

    WIRING = "dense"
    MODEL_NAME = "CRFP"
