"""Autograd operators of the hot path, each a thin host wrapper over the C-ABI launchers in
libo2m_hip.so (include/o2m_hip.h).  No op here has an eager/CPU fallback.

Internal activation format ("internal tensors"): contiguous ``[B, H, W, Cp]`` (NHWC), ``Cp``
a multiple of 8, dtype = the compute dtype (bf16, or fp32 in parity mode); channels beyond
the logical count are exactly zero everywhere (zero weight rows / zero bias keep them so).
Public tensors are logical NCHW views of those buffers (``to_public``), so the reference's
module API (builder.py) is met without copies.
"""

from __future__ import annotations

import contextlib
import math
import os as _os

import torch

from . import _hip as H
from . import resample as R

_STATE = {"dtype": torch.bfloat16, "epoch": 0}


def set_precision(precision: str) -> None:
    """"bf16": bf16 storage + bf16 MFMA (throughput mode, BASELINE config #2).
    "fp32": fp32 storage + bf16x3 split MFMA (parity mode, <=1e-3 of the CPU reference).
    "fp8" : BASELINE config #5 -- bf16 storage; the forward and data-gradient products of every
            convolution with a multiple of 128 input channels run on the fp8 MFMA: activations
            quantised per tensor to OCP e4m3 (gradients: e5m2), filters to e4m3, scale = format
            maximum / amax computed on the device, fp32 accumulation, bf16 results.  Weight
            gradients, normalisation, losses and the fp32 master weights / Adam stay as in "bf16"."""
    if precision not in ("bf16", "fp32", "fp8"):
        raise ValueError(precision)
    _STATE["dtype"] = torch.float32 if precision == "fp32" else torch.bfloat16
    _STATE["fp8"] = precision == "fp8"


FP8_EVERYWHERE = False  # tests: route every Ci % 128 == 0 convolution through the fp8 kernel, whatever its size


def fp8_enabled() -> bool:
    return bool(_STATE.get("fp8", False))


# O2M_FP8_DELAYED=0: every fp8 quantisation as amax pass + convert pass (dynamic scaling) instead of delayed scaling
_FP8_DELAYED = _os.environ.get("O2M_FP8_DELAYED", "1") == "1"


def _quantize(t: torch.Tensor, fmt: torch.dtype, deq_pair: torch.Tensor, site=None) -> torch.Tensor:
    """Per-tensor fp8 copy of ``t``; ``deq_pair`` (2 floats on the device) receives {1 / scale, amax}.  ``site``
    (PreparedWeight.fp8_site): delayed scaling -- from the site's second call on, ONE pass with the scale of the tensor it
    quantised last time (the activations / gradients of a layer change slowly from step to step; what outgrows the
    previous amax saturates).  Filters (no site) keep the exact two-pass form: they are quantised once per step."""
    q = torch.empty(t.shape, dtype=fmt, device=t.device)
    if site is not None and _FP8_DELAYED:
        H.quantize_fp8_site(t, q, deq_pair, site)
    else:
        H.quantize_fp8(t, q, deq_pair)
    return q


def set_deterministic(flag: bool) -> None:
    """The reference's ``deterministic_cuda_kernels`` switch (train.py:41-45, config.toml:9): when on,
    every kernel that sums across workgroups with float atomics takes its two-stage, fixed-order
    form instead, so two runs of a step are bitwise equal: the per-(sample, channel) sums of
    act_bwd_reduce and fold_scale_dot go through per-chunk rows added in chunk order (o2m_hip.h), and the
    weight gradient keeps its slab + ordered sum even under O2M_WGRAD_ATOMICS=1.  (The weight gradient's
    default form, InstanceNorm, the loss reductions and the style gradients are order-fixed in both modes.)"""
    _STATE["deterministic"] = bool(flag)
    H.DETERMINISTIC = bool(flag)


def deterministic() -> bool:
    return bool(_STATE.get("deterministic", False))


def compute_dtype() -> torch.dtype:
    return _STATE["dtype"]


def bump_weights_epoch(params=None) -> None:
    """Called by optimisers that update parameters behind autograd's back (fused Adam): ``params`` = the
    parameters just written (None: every cached filter form is stale)."""
    if params is None:
        _STATE["epoch"] += 1
        return
    for p in params:
        p._o2m_epoch = getattr(p, "_o2m_epoch", 0) + 1


def weights_epoch(p) -> tuple:
    return (_STATE["epoch"], getattr(p, "_o2m_epoch", 0))


def prepare_network(module) -> None:
    """The kernel-side forms of EVERY filter of ``module`` in one launch (o2m_prepare_weights_batched) when all of
    them are stale -- the state right after the network's optimiser step -- instead of one launch per layer on
    first use (54 launches per D+G step at 256 x 256).  Anything unusual (some layers fresh, parameters not fp32 /
    contiguous) is left to the per-layer path of PreparedWeight.get()."""
    import ctypes

    preps = [m._prepared() for m in module.modules() if hasattr(m, "_prepared")]
    if not preps or not preps[0].weight.is_cuda:
        return
    try:
        _prepare_network_bf16(module, preps)
    finally:
        if fp8_enabled():
            # the e4m3 filter copies too, HERE (on the stream the caller forks its groups from): built lazily they would be
            # written by whichever stream uses a layer first while the other group's stream may already read them
            for p in preps:  # (the channel conditions of PreparedWeight.fp8_ok)
                if p.cip % 128 == 0 and (FP8_EVERYWHERE or p.cop > 128):
                    p.get_fp8(False)
                if p.cop % 128 == 0 and (FP8_EVERYWHERE or p.cip > 128):
                    p.get_fp8(True)


def _prepare_network_bf16(module, preps) -> None:
    import ctypes

    keys = [p.version_key() for p in preps]
    if any(k == p._key for k, p in zip(keys, preps)):
        return  # (partly) fresh: the lazy path prepares what is left
    if any(p.weight.dtype != torch.float32 or not p.weight.is_contiguous() for p in preps):
        return
    bufs = [p.buffers() for p in preps]
    # the cached job table holds raw device pointers: it is valid only while EVERY tensor it names is the one it was
    # built for, and it lives on the module (a table keyed by id(module) outlived its module: a later network could
    # get the same id -- and, from the caching allocator, the same parameter addresses -- with other buffer addresses,
    # and the launch then wrote through stale pointers)
    ident = tuple((p.weight.data_ptr(),) + tuple(t.data_ptr() for t in b if t is not None)
                  for p, b in zip(preps, bufs)) + (compute_dtype(),)
    hit = module.__dict__.get("_o2m_prep_jobs")
    if hit is None or hit[0] != ident:
        jobs, first = (H.PrepJob * len(preps))(), 0
        for j, (p, (w_f, w_d, q, full, qt)) in enumerate(zip(preps, bufs)):
            jobs[j] = H.PrepJob(p.weight.data_ptr(), full.data_ptr(), w_f.data_ptr(), w_d.data_ptr(),
                                q.data_ptr() if q is not None else None, qt.data_ptr() if qt is not None else None,
                                p.co, p.ci, p.kh * p.kw, p.cop, p.cip, p.c, first, 0)
            first += (p.cop * p.cip + 255) // 256
        raw = torch.frombuffer(bytearray(ctypes.string_at(ctypes.addressof(jobs), ctypes.sizeof(jobs))), dtype=torch.uint8)
        hit = module.__dict__["_o2m_prep_jobs"] = (ident, raw.to(preps[0].weight.device), len(preps), first)
    outs = [t for b in bufs for t in b if t is not None]
    with torch.no_grad():
        H.prepare_weights_batched(hit[1], hit[2], hit[3], compute_dtype(), outs)
    for p, k, b in zip(preps, keys, bufs):
        p._val, p._key = b, k


def pad8(c: int) -> int:
    return (c + 7) // 8 * 8


# ------------------------------------------------------------------------ layout boundary


def _as_internal_view(x: torch.Tensor):
    """Zero-copy recovery of the internal buffer behind a public view, or None."""
    if x.dim() != 4 or x.dtype != compute_dtype() or not x.is_cuda:
        return None
    b, c, h, w = x.shape
    cp = pad8(c)
    if x.stride() != (h * w * cp, 1, w * cp, cp) or x.storage_offset() % 8:
        return None
    return x.as_strided((b, h, w, cp), (h * w * cp, w * cp, cp, 1))


def _pack(x: torch.Tensor) -> torch.Tensor:
    v = _as_internal_view(x)
    if v is not None:
        return v
    if not x.is_cuda:
        raise RuntimeError("the one-to-many GAN hot path runs on the GPU only (no CPU fallback)")
    b, c, h, w = x.shape
    out = torch.empty((b, h, w, pad8(c)), dtype=compute_dtype(), device=x.device)
    H.pack_nchw(x.detach().float().contiguous(), out)
    return out


class _ToInternal(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        ctx.c = x.shape[1]
        return _pack(x)

    @staticmethod
    def backward(ctx, g):
        return g.permute(0, 3, 1, 2)[:, : ctx.c]


class _ToPublic(torch.autograd.Function):
    @staticmethod
    def forward(ctx, t, c):
        return t.permute(0, 3, 1, 2)[:, :c]

    @staticmethod
    def backward(ctx, g):
        return _pack(g), None


def to_internal(x: torch.Tensor) -> torch.Tensor:
    """Logical NCHW tensor (any dtype/strides) -> internal NHWC buffer (zero-copy if it
    already is a public view of one)."""
    return _ToInternal.apply(x)


def to_public(t: torch.Tensor, c: int) -> torch.Tensor:
    """Internal NHWC buffer -> logical (B, c, H, W) view."""
    return _ToPublic.apply(t, c)


# ------------------------------------------------------------------------ prepared weights


class PreparedWeight:
    """Per-layer cache of the kernel-side forms of one equalised-LR filter
    (layers.py:12-24): W*c cast to the compute dtype in [Co][KH][KW][Ci] (forward/wgrad
    operand order), its flipped transpose [Ci][KH][KW][Co] (the data-gradient filter) and,
    for modulated convs, Q[o,i] = c^2 * sum_k W[o,i,k]^2 (layers.py:156-161 factored so the
    demodulation needs no per-sample weights).  Rebuilt when the parameter changes."""

    def __init__(self, weight: torch.nn.Parameter, need_q: bool):
        self.weight = weight
        self.need_q = need_q
        co, ci, kh, kw = weight.shape
        self.co, self.ci, self.kh, self.kw = co, ci, kh, kw
        self.cop, self.cip = pad8(co), pad8(ci)
        self.c = 1.0 / math.sqrt(ci * kh * kw)
        self._key = None
        self._val = None
        # end-of-backward gradient accumulators (kernel layout), see _finalize_weight_grads
        self.dw_acc = None
        self.gq_acc = None
        self.pending = False
        self.dw2_acc = None  # space-to-depth form of dw_acc (s2d_wgrad), folded at finalize
        self.dw2_used = False
        # forward applications recorded for a backward that wants this filter's gradient, and how
        # many of them the running backward has reduced: when they meet, the layer is finalised on
        # the spot (its gradient is complete) instead of at the end of backward
        self.fwd_uses = 0
        self.bwd_uses = 0

    def accumulators(self, device):
        if self.dw_acc is None or self.dw_acc.device != device:
            self.dw_acc = torch.zeros((self.cop, self.kh, self.kw, self.cip), dtype=torch.float32, device=device)
            self.gq_acc = (torch.zeros((self.cop, self.cip), dtype=torch.float32, device=device)
                           if self.need_q else None)
        _enter_backward_pass()
        if not self.pending:
            self.pending = True
            _PENDING.append(self)
        return self.dw_acc, self.gq_acc

    def note_forward_use(self, wants_weight_grad: bool) -> bool:
        """Forward side of the use count.  ``wants_weight_grad`` = ``ctx.needs_input_grad[<weight>]``
        of the calling autograd Function (grad mode itself reads as off inside ``forward``)."""
        if wants_weight_grad:
            self.fwd_uses += 1
        return bool(wants_weight_grad)

    def use_reduced(self, device):
        """Backward side: one counted application's weight gradient (and dL/dQ) has been added to
        the accumulators.  When it was the last one recorded, the gradient is complete NOW, not at
        the end of backward: finalise the layer, so the data-parallel reducer can start the bucket
        segment's all-reduce under the rest of backward (dist.BucketReducer)."""
        self.bwd_uses += 1
        if self.bwd_uses == self.fwd_uses and _early_finalize():
            if device.type != "cuda":  # (the gloo rehearsal of the reducer on CPU tensors)
                _finalize_layer(self)
                return
            cur = torch.cuda.current_stream(device)
            wst = _wgrad_stream(device)
            if wst is not None:
                cur.wait_stream(wst)
            # the layer's other uses may have been reduced on the other compute stream (generator_step runs the
            # decode group and the extraction group on two streams): their style-path accumulations must have
            # landed before the accumulators are converted and handed to the all-reduce
            for other in (torch.cuda.default_stream(device), _GSTREAM.get(device)):
                if other is not None and other != cur:
                    cur.wait_stream(other)
            _finalize_layer(self)

    def fp8_ok(self, data_grad: bool, rows: int) -> bool:
        """The fp8 kernel stages 128 reduction elements of ONE filter tap per K-tile and exists as the
        256 x 256-tile phase-pipelined kernel only: by default it takes the layers that kernel would take
        in bf16 (more than 128 output channels, >= 256 tiles); FP8_EVERYWHERE (parity tests) drops the
        size condition so that small cases reach it too."""
        k_ch, n_ch = (self.cop, self.cip) if data_grad else (self.cip, self.cop)
        if not fp8_enabled() or k_ch % 128 != 0:
            return False
        return FP8_EVERYWHERE or (n_ch > 128 and -(-rows // 256) * -(-n_ch // 256) >= 256)

    def fp8_site(self, kind: str, device):
        """Delayed-scaling state of one of this layer's quantisation sites ("x": forward input, "g": data-gradient input,
        "w": per-sample filters), per stream: the two groups of a generator step run a layer on two streams at once."""
        sites = self.__dict__.setdefault("_fp8_sites", {})
        key = (kind, torch.cuda.current_stream(device).cuda_stream)
        ent = sites.get(key)
        if ent is None:
            ent = sites[key] = [H.Fp8Site(), None]
        # delayed scaling for a site's FIRST tensor of a weights epoch only (one optimiser step ago the same site saw the
        # same kind of tensor); a second tensor through the same site in the same step (a decoder layer applied to
        # several batches one after the other) is scaled by its own amax -- returns None for it
        epoch = weights_epoch(self.weight)
        if ent[1] == epoch:
            return None
        ent[1] = epoch
        return ent[0]

    def get_fp8(self, data_grad: bool):
        """e4m3 copy of the forward (or data-gradient) filter + the layer's 4-float dequantisation record
        {1/scale_x, amax_x, 1/scale_w, amax_w}: the filter half is written here, once per weight version;
        the activation half by the quantisation of every call's input (same stream, so ordered)."""
        w_f, w_d = self.get()[:2]
        key = (self._key, data_grad)
        cache = getattr(self, "_fp8", None)
        if cache is None:
            cache = self._fp8 = {}
        hit = cache.get(data_grad)
        if hit is None or hit[0] != key:
            rec = torch.zeros(4, dtype=torch.float32, device=w_f.device)
            hit = cache[data_grad] = (key, _quantize(w_d if data_grad else w_f, torch.float8_e4m3fn, rec[2:4]), rec)
        return hit[1], hit[2]

    def discard_partial(self):
        """Drop what an aborted backward pass left behind (see _enter_backward_pass)."""
        for t in (self.dw_acc, self.gq_acc, self.dw2_acc):
            if t is not None:
                t.zero_()
        self.pending, self.dw2_used = False, False
        self.fwd_uses = self.bwd_uses = 0

    S2D = 4  # output pixels per side folded into channels by the space-to-depth form

    def s2d_ok(self, pad, pad_mode, hh, ww):
        """The few-output image conv (Co <= 8, large kernel) as a strided conv over 4x4 output
        blocks: out[4by+dy, 4bx+dx, co] = sum W[co, kh-dy, kw-dx] x[4by+kh-pad, 4bx+kw-pad]."""
        r = self.S2D
        return (self.cop == 8 and self.kh >= 5 and self.kh == self.kw and 2 * pad == self.kh - 1
                and pad_mode == H.PAD_REFLECT and hh % r == 0 and ww % r == 0 and self.ci % 64 == 0)

    def s2d_weights(self):
        key = self.version_key()
        if getattr(self, "_s2d_key", None) != key:
            full = self.get()[3]  # [cop, kh, kw, cip] fp32, W*c
            r, k = self.S2D, self.kh
            kk = k + r - 1
            nco = pad8(r * r * self.co)
            w2 = torch.zeros((nco, kk, kk, self.cip), dtype=torch.float32, device=full.device)
            # w2[(dy, dx, co), dy + a, dx + b, :] = full[co, a, b, :] for the 16 block offsets: ONE strided copy
            self._s2d_blocks(w2).copy_(full[: self.co].expand(r, r, self.co, k, k, self.cip))
            self._s2d_val = w2.to(compute_dtype()).contiguous()
            self._s2d_key = key
        return self._s2d_val

    def s2d_wgrad(self, x, gu, pad, pad_mode):
        """Weight gradient of the few-output image conv in the same 4x4-block form as its forward:
        a stride-4 reduction with 16*co outputs and (k+3)^2 taps -- 16x fewer GEMM rows, 4x fewer
        padded MFMA flops and 4x less re-reading of x than Co = 3 padded to a 32-wide tile."""
        r, k = self.S2D, self.kh
        kk = k + r - 1
        B, Hh, Ww, _ = x.shape
        nco = pad8(r * r * self.co)
        if self.dw2_acc is None or self.dw2_acc.device != x.device:
            self.dw2_acc = torch.zeros((nco, kk, kk, self.cip), dtype=torch.float32, device=x.device)
        g2 = torch.zeros((B, Hh // r, Ww // r, nco), dtype=gu.dtype, device=gu.device)
        g2[..., : r * r * self.co] = (gu[..., : self.co].view(B, Hh // r, r, Ww // r, r, self.co)
                                      .permute(0, 1, 3, 2, 4, 5).reshape(B, Hh // r, Ww // r, r * r * self.co))
        H.conv2d_wgrad(x, g2, self.dw2_acc, pad=pad, pad_mode=pad_mode, stride=r)
        self.dw2_used = True

    def fold_s2d(self):
        """dW[co, kh, kw] = sum over the 16 block offsets of dW2[(dy, dx, co), kh + dy, kw + dx]
        (the adjoint of s2d_weights)."""
        self.dw_acc[: self.co] += self._s2d_blocks(self.dw2_acc).sum(dim=(0, 1))
        self.dw2_acc.zero_()
        self.dw2_used = False

    def _s2d_blocks(self, w2):
        """View of a [(dy, dx, co), k + 3, k + 3, cip] space-to-depth filter as [dy, dx, co, a, b, cip] with
        element (dy, dx, co, a, b) at tap (dy + a, dx + b): the 16 shifted copies of the k x k filter."""
        r, k = self.S2D, self.kh
        kk = k + r - 1
        sa, sb = kk * self.cip, self.cip  # strides of the two tap dimensions
        sco = kk * kk * self.cip
        return w2.as_strided((r, r, self.co, k, k, self.cip), (r * self.co * sco + sa, self.co * sco + sb, sco, sa, sb, 1))

    def version_key(self):
        """Changes when the kernel-side forms are stale: the parameter was written through autograd's version
        counter, re-homed, updated by its fused optimiser (``weights_epoch`` of its bucket -- per network, so a
        step of the discriminator's optimiser does not invalidate the generator's filters) or the precision mode
        changed."""
        w = self.weight
        return (w._version, w.data_ptr(), weights_epoch(w), compute_dtype())

    def buffers(self):
        """The per-layer output buffers, allocated once per (device, dtype) and refilled in place: stable addresses
        are what lets one batched launch serve a whole network (prepare_network).  Every stream that reads them has
        been joined with the main one by the end of the previous backward pass (_finalize_weight_grads)."""
        dev, cd = self.weight.device, compute_dtype()
        if getattr(self, "_buf_key", None) != (dev, cd):
            full = torch.empty((self.cop, self.kh, self.kw, self.cip), dtype=torch.float32, device=dev)
            w_f = torch.empty((self.cop, self.kh, self.kw, self.cip), dtype=cd, device=dev)
            w_d = torch.empty((self.cip, self.kh, self.kw, self.cop), dtype=cd, device=dev)
            q = torch.empty((self.cop, self.cip), dtype=torch.float32, device=dev) if self.need_q else None
            qt = torch.empty((self.cip, self.cop), dtype=torch.float32, device=dev) if self.need_q else None
            self._bufs, self._buf_key = (w_f, w_d, q, full, qt), (dev, cd)
            self._key = None
        return self._bufs

    def get(self):
        key = self.version_key()
        if key != self._key:
            with torch.no_grad():
                wsrc = self.weight.detach().float().contiguous()  # no-ops for the fp32 parameters
                w_f, w_d, q, full, qt = self.buffers()
                H.prepare_weights(wsrc, full, w_f, w_d, q, qt, self.c)  # one launch for this layer
            self._val = (w_f, w_d, q, full, qt)
            self._key = key
        return self._val


import os as _os

# The style-path kernels are B x C sized: alone on the GPU they leave 250 CUs idle for 6-14 us each
# (~110 launches per generator backward).  O2M_SIDE_STREAM=1 runs them on a second HIP stream, ordered by
# events, so they overlap the neighbouring convolution kernels: worth 0.4 ms/step in round 1, but since the
# weight-gradient kernels have their own stream it LOSES 0.6-1.2 ms (three interleaved same-box pairs:
# 52.3 / 52.6 / 52.1 on vs 51.7 / 51.3 / 51.7 off -- the main chain then waits for the style kernels
# behind the weight-gradient stream's work).  Off by default.
_SIDE_STREAM = _os.environ.get("O2M_SIDE_STREAM", "0") == "1"
_SIDE: dict = {}
# O2M_DIRECT_STYLE_GRADS=1 (default): the style-gradient kernel adds the to_style gradients straight
# into the parameters' .grad when those are the fp32 slices of a FusedAdam bucket (no autograd
# additions for the five uses of a decoder layer per step: ~100 tiny launches less).
_DIRECT_STYLE_GRADS = _os.environ.get("O2M_DIRECT_STYLE_GRADS", "1") == "1"


# O2M_WGRAD_STREAM=1 (default): the weight-gradient reductions are off the critical path of backward
# (only the layer's finalisation needs them), so they run on their own stream: they fill the CUs the
# one-round data-gradient launches leave idle at their end (and their 128x128 tail launches) and overlap
# the HBM-bound pointwise kernels of the following layers.  Same-box A/B: 51.5 / 52.0 vs 51.8 / 52.6 ms.
_WGRAD_STREAM = _os.environ.get("O2M_WGRAD_STREAM", "1") == "1"
# O2M_FUSED_IN_STATS=0: InstanceNorm statistics by their own pass over the conv output (A/B runs)
_FUSED_IN_STATS = _os.environ.get("O2M_FUSED_IN_STATS", "1") == "1"
# Early finalisation (a filter's gradient converted the moment its last use of the pass is reduced) exists for
# the data-parallel reducer, which can then start a bucket segment's all-reduce inside backward; it makes the
# main stream wait for the weight-gradient stream at every such point, which costs ~1 ms/step on one GPU
# (same-box pairs: 50.4 / 50.8 with, 49.7 / 49.4 without).  Default: on exactly when a reducer has registered
# its readiness hooks (GRAD_READY_HOOKS); O2M_EARLY_FINALIZE=0|1 forces it.
_EARLY_FINALIZE = _os.environ.get("O2M_EARLY_FINALIZE")


def _early_finalize() -> bool:
    if _EARLY_FINALIZE is not None:
        return _EARLY_FINALIZE == "1"
    return bool(GRAD_READY_HOOKS)
_WSTREAM: dict = {}


# The phase-pipelined weight-gradient kernel (o2m_wgrad_desc.kernel_hint) holds a whole CU per block: alone on the
# chip it is 1.15-1.35x faster than the register-staged tiles (B = 16 / 32 / 48: 800 / 882 / 966 vs 580 / 700 / 826
# TFLOP/s on the 256 -> 256 layers).  Mid-round-3 it lost 0.3-0.4 ms per step on the weight-gradient stream BESIDE the
# main stream and was used only in-line; with its loader halves slimmed (conv_wgrad.hip) it gains 0.3 ms there too (four
# same-box pairs, gpurun_out/r03v/ab.log) and is the default wherever its geometry applies.  O2M_WGRAD_P8=0 switches
# it off (A/B runs).
_WGRAD_P8 = _os.environ.get("O2M_WGRAD_P8", "1") == "1"


def _wgrad_p8(wst) -> bool:
    return _WGRAD_P8


def _wgrad_stream(device):
    if not _WGRAD_STREAM or device.type != "cuda":
        return None
    st = _WSTREAM.get(device)
    if st is None:
        st = _WSTREAM[device] = torch.cuda.Stream(device=device)
    return st


# O2M_GROUP_STREAM=0: generator_step's extraction group on the main stream (default: its own stream, so that its
# HBM-bound kernels run beside the decode group's MFMA kernels and vice versa; off in deterministic mode -- two
# concurrent style backward launches of one layer add into the same accumulators)
_GROUP_STREAM = _os.environ.get("O2M_GROUP_STREAM", "1") == "1"
_GSTREAM: dict = {}


# O2M_ASYNC_H2D=0: host-drawn random numbers reach the device by a blocking copy from pageable memory (A/B)
_ASYNC_H2D = _os.environ.get("O2M_ASYNC_H2D", "1") == "1"


def host_to_device(t: torch.Tensor, device) -> torch.Tensor:
    """A small host tensor (the reference's CPU-generator draws: z, theta) onto the device WITHOUT making the host wait
    for the stream: through pinned memory and a non-blocking copy.  ``t.to(device)`` from pageable memory returns only
    when the copy has run, i.e. when everything queued before it has -- nine such waits per D+G step kept the host from
    running ahead, and the device starved in every phase made of small kernels (profiles/README.md, round 4)."""
    device = torch.device(device)
    if device.type != "cuda" or t.device.type != "cpu" or not _ASYNC_H2D:
        return t.to(device)
    return t.pin_memory().to(device, non_blocking=True)


# O2M_D_OVERLAP=0: discriminator_step on the caller's stream (the generator step's forward then waits for all of it)
_D_OVERLAP = _os.environ.get("O2M_D_OVERLAP", "1") == "1"
_DSTREAM: dict = {}
_D_EVENTS: dict = {}   # device -> {"prep": event behind the step's weight preparation, "done": event behind the whole step}


def d_step_stream(device):
    """Stream of ``discriminator_step`` (None: the current one).  The generator step that follows needs the updated
    discriminator only for its adversarial term, AFTER its own encoder and decoder passes: with the discriminator step
    on a stream of its own, those passes share the chip with the discriminator's backward and optimiser (a batch-16 /
    32 sequence of small launches that leaves CUs idle) instead of queuing behind them.  ``generator_step`` waits for
    ``d_step_event(device, "done")`` before it touches the discriminator.  Not under graph capture, not in
    deterministic mode (one stream there), not on CPU tensors."""
    device = torch.device(device)
    if not _D_OVERLAP or device.type != "cuda" or deterministic() or torch.cuda.is_current_stream_capturing():
        return None
    st = _DSTREAM.get(device)
    if st is None:
        st = _DSTREAM[device] = torch.cuda.Stream(device=device)
    return st


def _on_d_stream(device) -> bool:
    st = _DSTREAM.get(torch.device(device)) if torch.device(device).type == "cuda" else None
    return st is not None and torch.cuda.current_stream(device) == st


def d_step_mark(device, which: str):
    """Record the discriminator step's ``prep`` / ``done`` event on the current stream -- only when that is the
    discriminator step's own stream (a step on the caller's stream is ordered by the stream itself: events dropped)."""
    device = torch.device(device)
    if device.type != "cuda":
        return
    if not _on_d_stream(device):
        _D_EVENTS.pop(device, None)
        return
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream(device))
    _D_EVENTS.setdefault(device, {})[which] = ev


def d_step_wait(device, which: str):
    """Order the current stream behind the last discriminator step's ``prep`` / ``done`` event (no-op when that step ran
    on the caller's stream or none has run)."""
    device = torch.device(device)
    if device.type != "cuda":
        return
    ev = _D_EVENTS.get(device, {}).get(which)
    if ev is not None and not torch.cuda.is_current_stream_capturing():
        torch.cuda.current_stream(device).wait_event(ev)


class StepThrottle:
    """At most ``limit`` training steps queued on the device.  The host issues a 256 x 256 step in ~13 ms and the device
    takes ~34: unchecked, the host runs ahead until the HIP queues fill (6-8 steps), every tensor with a cross-stream
    use is held until the device catches up, and the caching allocator's pool keeps growing (107 GiB reserved, ~8
    hipMalloc calls per step for as long as the lead grows).  With two steps in flight the device never starves (same
    step time) and the pool settles within the first steps.  ``with throttle: <one step>``; O2M_MAX_STEPS_IN_FLIGHT=0:
    unbounded."""

    def __init__(self, device, limit=None):
        self.device = torch.device(device)
        self.limit = int(_os.environ.get("O2M_MAX_STEPS_IN_FLIGHT", "2")) if limit is None else limit
        self.queue, self.cur = [], None

    def __enter__(self):
        self.cur = None
        if self.device.type == "cuda" and self.limit > 0 and not torch.cuda.is_current_stream_capturing():
            if len(self.queue) >= self.limit:
                self.queue.pop(0).synchronize()
            self.cur = torch.cuda.Event()
        return self

    def __exit__(self, *exc):
        if self.cur is not None:
            self.cur.record(torch.cuda.current_stream(self.device))
            self.queue.append(self.cur)
        return False


def group_stream(device):
    """Second compute stream for an independent sub-graph of a step (None: run it on the current stream)."""
    if not _GROUP_STREAM or device.type != "cuda" or deterministic():
        return None
    st = _GSTREAM.get(device)
    if st is None:
        st = _GSTREAM[device] = torch.cuda.Stream(device=device, priority=int(_os.environ.get("O2M_PRIO_GROUP", "0")))
    return st


_MSTREAM: dict = {}


def mapping_stream(device):
    """A stream of its own for generator_step's mapping-network passes (tiny launches, forward and backward): on the
    group stream they would queue behind the style extractor's pass and hold up the decode group that waits for them
    (measured: +1.6 ms)."""
    if group_stream(device) is None:
        return None
    st = _MSTREAM.get(device)
    if st is None:
        st = _MSTREAM[device] = torch.cuda.Stream(device=device, priority=int(_os.environ.get("O2M_PRIO_MAP", "0")))
    return st


def _side_stream(device):
    if not _SIDE_STREAM or device.type != "cuda":
        return None
    st = _SIDE.get(device)
    if st is None:
        st = _SIDE[device] = torch.cuda.Stream(device=device)
    return st
_PENDING: list = []
_PASS = {"task": None}


def _enter_backward_pass():
    """Called by every weight-gradient producer.  One end-of-backward callback is queued per
    autograd graph task (identified by the engine's task id), not "whenever the pending list is
    empty": a backward that raised leaves the list non-empty and its callback is dropped by the
    engine, which in round 1 meant that no later backward ever finalised a filter gradient again.
    Entering a new task with leftovers from a dead one discards them first."""
    task = torch._C._current_graph_task_id()
    if task == _PASS["task"]:
        return
    if _PENDING:  # a previous backward died before its callback ran
        for prep in _PENDING:
            prep.discard_partial()
        _PENDING.clear()
    _PASS["task"] = task
    if task >= 0:
        torch.autograd.Variable._execution_engine.queue_callback(_finalize_weight_grads)


# parameter -> callable(param), invoked when that filter's gradient has been written by
# _finalize_weight_grads (the data-parallel reducer counts these like autograd's own
# post-accumulate hooks, which never fire for the filters)
GRAD_READY_HOOKS: dict = {}


def _finalize_layer(prep):
    """Kernel-layout accumulators of one layer -> ``weight.grad`` (+= like autograd), clears them,
    and tells the data-parallel reducer that this filter's gradient is complete."""
    prep.pending = False
    prep.fwd_uses = prep.bwd_uses = 0
    if prep.dw2_used:
        prep.fold_s2d()
    w = prep.weight
    if w.grad is None:
        w.grad = torch.zeros_like(w)
    grad = w.grad
    if not grad.is_contiguous() or grad.dtype != torch.float32:
        tmp = torch.zeros(w.shape, dtype=torch.float32, device=w.device)
        H.wgrad_finalize(prep.dw_acc, prep.gq_acc, prep.get()[3], tmp, prep.co, prep.ci, prep.c)
        grad.add_(tmp.to(grad.dtype))
    else:
        H.wgrad_finalize(prep.dw_acc, prep.gq_acc, prep.get()[3], grad, prep.co, prep.ci, prep.c)
    # the to_style parameters whose gradients style_bwd added straight into .grad never pass through
    # autograd's AccumulateGrad: they are complete when the layer's last use has been reduced, too
    for p in (w, *getattr(prep, "direct_style", ())):
        hook = GRAD_READY_HOOKS.get(p)
        if hook is not None:
            hook(p)


# O2M_BATCHED_FINALIZE=0: one o2m_wgrad_finalize launch (+ a clear launch for modulated layers) per pending layer at the
# end of backward (29 + 16 launches per D+G step) instead of one batched launch (+ one) per backward pass
_BATCHED_FINALIZE = _os.environ.get("O2M_BATCHED_FINALIZE", "1") == "1"
_FIN_JOBS: dict = {}


def _finalize_batched(live) -> bool:
    """Every pending layer's kernel-layout accumulators -> ``weight.grad`` in ONE launch (o2m_wgrad_finalize_batched).
    Falls back (False) when a layer is unusual: no fp32 contiguous .grad, CPU tensors."""
    import ctypes

    for prep in live:
        w = prep.weight
        if not w.is_cuda:
            return False
        if w.grad is None:
            w.grad = torch.zeros_like(w)
        if w.grad.dtype != torch.float32 or not w.grad.is_contiguous():
            return False
    for prep in live:
        if prep.dw2_used:
            prep.fold_s2d()
    w32s = [prep.get()[3] for prep in live]
    ident = tuple((prep.dw_acc.data_ptr(), prep.gq_acc.data_ptr() if prep.gq_acc is not None else 0, w.data_ptr(),
                   prep.weight.grad.data_ptr()) for prep, w in zip(live, w32s))
    hit = _FIN_JOBS.get(ident)
    if hit is None:
        if len(_FIN_JOBS) > 16:
            _FIN_JOBS.clear()
        jobs, first = (H.WfinJob * len(live))(), 0
        for j, (prep, w32) in enumerate(zip(live, w32s)):
            jobs[j] = H.WfinJob(prep.dw_acc.data_ptr(), prep.gq_acc.data_ptr() if prep.gq_acc is not None else None,
                                w32.data_ptr(), prep.weight.grad.data_ptr(), prep.co, prep.ci, prep.kh * prep.kw, prep.cop,
                                prep.cip, prep.c, first, 0)
            first += H.wgrad_finalize_blocks(prep.cop, prep.kh * prep.kw, prep.cip)
        raw = torch.frombuffer(bytearray(ctypes.string_at(ctypes.addressof(jobs), ctypes.sizeof(jobs))), dtype=torch.uint8)
        hit = _FIN_JOBS[ident] = (raw.to(live[0].weight.device), len(live), first,
                                  any(prep.gq_acc is not None for prep in live))
    touched = [t for prep in live for t in (prep.dw_acc, prep.gq_acc, prep.weight.grad) if t is not None]
    H.wgrad_finalize_batched(hit[0], hit[1], hit[2], hit[3], touched)
    for prep in live:
        prep.pending = False
        prep.fwd_uses = prep.bwd_uses = 0
        for p in (prep.weight, *getattr(prep, "direct_style", ())):
            hook = GRAD_READY_HOOKS.get(p)
            if hook is not None:
                hook(p)
    return True


def _finalize_weight_grads():
    """End of a backward pass (autograd engine callback): finalises every layer that was not
    already finalised when its last recorded use was reduced (``_ConvFn.backward``) -- e.g. a
    layer whose forward ran more often than its backward (part of the graph unused)."""
    pend = list(_PENDING)
    _PENDING.clear()
    _PASS["task"] = None
    for dev, wst in _WSTREAM.items():  # weight gradients reduced on their own stream
        torch.cuda.current_stream(dev).wait_stream(wst)
    for dev, gst in _GSTREAM.items():  # style gradients accumulated by a group that ran on its own stream
        torch.cuda.current_stream(dev).wait_stream(gst)
    for dev, mst in _MSTREAM.items():  # (the mapping network's backward: its parameters' gradients)
        torch.cuda.current_stream(dev).wait_stream(mst)
    live = [prep for prep in pend if prep.pending]
    if len(live) > 1 and _BATCHED_FINALIZE and _finalize_batched(live):
        return
    for prep in live:
        _finalize_layer(prep)


class _ZeroPool:
    """Pre-zeroed fp32 scratch for the atomically accumulated per-(sample, channel) tables of the conv
    backward (``sums`` / ``dots``): ~60 of them per generator step, each a few KB, each costing a
    ``torch.zeros`` fill launch.  Slices are handed out bump-pointer style from one buffer that is
    cleared by ONE memset per optimiser ``zero_grad`` (``reset``), when no backward temporaries are
    alive; a request that does not fit falls back to ``torch.zeros``."""

    CAPACITY = 4 << 20  # floats (16 MB)

    def __init__(self):
        self.buf, self.used, self.dirty = {}, {}, {}

    @staticmethod
    def _key(device):
        # the discriminator step may run on its own stream beside the next generator step's forward (d_step_stream):
        # its backward's slices must not be cleared by the generator step's zero_grad -- one pool per side
        return (device, _on_d_stream(device))

    def take(self, n: int, device) -> torch.Tensor:
        key = self._key(device)
        buf = self.buf.get(key)
        if buf is None:
            buf = self.buf[key] = torch.zeros(self.CAPACITY, dtype=torch.float32, device=device)
            self.used[key] = 0
        off = self.used[key]
        n4 = (n + 3) // 4 * 4  # 16-B aligned slices
        if off + n4 > self.CAPACITY:
            return torch.zeros(n, dtype=torch.float32, device=device)
        self.used[key] = off + n4
        return buf[off: off + n]

    def reset(self):
        for key, buf in self.buf.items():
            if self.used[key] and key == self._key(key[0]):  # (the pool of the side that is resetting)
                buf[: self.used[key]].zero_()
                self.used[key] = 0


ZERO_POOL = _ZeroPool()


def _pad_cols(t: torch.Tensor, n: int) -> torch.Tensor:
    t = t.float()
    if t.shape[1] == n:
        return t.contiguous()
    out = torch.zeros((t.shape[0], n), dtype=torch.float32, device=t.device)
    out[:, : t.shape[1]] = t
    return out


# ---------------------------------------------------------------------------------- conv


# The weight gradient of a modulated conv reduces gu x (x * s).  Default: x * s is a stored by-product of the pass
# that reads x anyway (fold_scale_dot / the data-gradient epilogue).  O2M_WGRAD_INSCALE=1: the weight-gradient
# kernel scales x while staging it instead (no tensor write; measured: +35 % kernel time and 16 more VGPRs, which
# keep small kernels of the main stream from sharing its CUs -- kept for A/B).
_WGRAD_XS = _os.environ.get("O2M_WGRAD_INSCALE", "0") != "1"
_WGRAD_HALO = _os.environ.get("O2M_WGRAD_HALO", "1") != "0" and _os.environ.get("O2M_WGRAD_ATOMICS", "0") != "1"


# O2M_BORDER_DGRAD=1 (experiment, off): the data gradient of a conv behind ReflectionPad2d(1) as the zero-padded conv on
# the CROPPED domain (a whole number of 256-row tiles per sample at 64 x 64: no 6 % more rows, no tail launch, no fold)
# + o2m_conv2d_reflect_border for the ring the crop leaves out.  Same-box A/B (gpurun_out/r04f, r04g, r04h): the tail
# launches shrink by 0.86 ms per step, the border launches cost 1.43 -> 1.18 -> 0.78 ms over three versions of the
# kernel (L2-traffic-bound: conv_direct.hip) and the step stays 0.5 ms SLOWER.  Default: the padded 66 x 66 domain + fold.
_BORDER_DGRAD = int(_os.environ.get("O2M_BORDER_DGRAD", "0"))  # 1: the plain (encoder) convs only; 2: the modulated ones too


def _border_dgrad_ok(prep, g, Hh, Ww, pad, pad_mode, modulated=False) -> bool:
    return (_BORDER_DGRAD >= (2 if modulated else 1) and pad_mode == H.PAD_REFLECT and pad == 1 and prep.kh == 3 and prep.kw == 3
            and g.dtype == torch.bfloat16 and not deterministic() and prep.cop % 64 == 0 and prep.cip % 64 == 0
            and min(Hh, Ww) >= 4)


def _wgrad_scales_in_kernel(prep, x, pad, pad_mode) -> bool:
    """Whether the weight gradient of this modulated conv will run on the halo-tile kernel (conv_wgrad.hip:
    3 x 3, zero padding 1, 64-channel multiples, rows of a multiple of 32 pixels, >= 64 tiles, bf16), which folds the
    style into its per-sample partials: the scaled input x * s is then not needed at all.  Mirrors wgrad_halo_ok; a
    mismatch only costs speed (o2m_conv2d_wgrad takes in_scale on every path)."""
    B, Hh, Ww, cip = x.shape
    return (_WGRAD_HALO and x.dtype == torch.bfloat16 and prep.kh == 3 and prep.kw == 3 and pad == 1 and pad_mode == H.PAD_ZERO
            and prep.cop % 64 == 0 and cip % 64 == 0 and Ww % 32 == 0 and Hh % 8 == 0 and B * (Hh // 8) * (Ww // 32) >= 64
            and (prep.cop // 64) * (cip // 64) <= 64)
# O2M_FUSED_DGRAD_DOT=0: zero-padded modulated convs run fold_scale_dot behind their data gradient (round-2 form)
# instead of taking the style scale and the style dot out of the data-gradient epilogue (O2M_STATS_DOT)
_FUSED_DGRAD_DOT = _os.environ.get("O2M_FUSED_DGRAD_DOT", "1") == "1"
# O2M_BLOCK_LINK=0: residual-block backward without the links below (autograd adds the residual gradient, every
# conv runs its own act_bwd_reduce and fold_scale_dot)
_BLOCK_LINK = _os.environ.get("O2M_BLOCK_LINK", "1") == "1"


class BlockLink:
    """Backward-time hand-offs inside a residual block ``x + conv_b(act(conv_a(x)))`` (blocks.py:36-68), shared by
    the two ``_ConvFn`` calls of one block application:

    * ``res_grad``: conv_b adds the residual x, so dL/d(out) also flows to x unchanged.  Instead of returning it
      to autograd (one elementwise add of two activation-sized tensors per block), conv_b parks it here and
      conv_a's fold_scale_dot adds it while writing its own data gradient (``gres``).
    * ``deferred``: conv_b's input IS conv_a's output u = act(...).  conv_b stops after its data-gradient GEMM
      and parks the padded gradient here; conv_a then runs ONE kernel that folds / scales it, forms conv_b's style
      dot, applies conv_a's activation backward and demodulation and adds conv_a's sums -- the gradient of u is
      never stored (4 tensor passes instead of 7).  conv_b's style backward (which needs that dot) runs right
      after, from the closure parked with it.

    Valid only while u has no other consumer than conv_b and x reaches conv_b's residual unchanged, which the
    block guarantees; a gradient for u from anywhere else makes conv_a raise instead of silently using it."""

    def __init__(self, fuse_act: bool = False):
        self.fuse_act = fuse_act  # conv_b may park its fold for conv_a (u has no other consumer)
        self.res_grad = None
        self.deferred = None
        self.lazy_ptr = None


class _ConvFn(torch.autograd.Function):
    """y = act(d[b,o] * conv(W*c, pad(x * s[b,i])) + bias) + residual with
    s = to_style(w_style) and d = demodulation (both computed by o2m_style_fwd)."""

    @staticmethod
    def forward(ctx, x, weight, bias, w_style, ts_weight, ts_bias, residual, prep, pad, pad_mode, act,
                demodulate, eps, stats_eps, link=None):
        w_f, w_d, q, w32, qt = prep.get()
        B, Hh, Ww, cip = x.shape
        if cip != prep.cip:
            raise RuntimeError(f"conv input has {cip} channels, layer expects {prep.cip} (padded)")
        ho, wo = Hh + 2 * pad - prep.kh + 1, Ww + 2 * pad - prep.kw + 1
        s = d = wv = ws = None
        if w_style is not None:
            wv = w_style.detach().float().contiguous()
            ws = ts_weight.detach().float().contiguous()
            s = torch.empty((B, prep.cip), dtype=torch.float32, device=x.device)
            d = torch.empty((B, prep.cop), dtype=torch.float32, device=x.device) if demodulate else None
            H.style_fwd(wv, ws, ts_bias.detach().float().contiguous(), qt if demodulate else None, s, d,
                        prep.ci, 1.0 / math.sqrt(wv.shape[1]), eps)
        bias_p = None
        if bias is not None:
            if prep.cop == prep.co and bias.dtype == torch.float32:
                bias_p = bias.detach()
            else:
                bias_p = torch.zeros(prep.cop, dtype=torch.float32, device=x.device)
                bias_p[: prep.co] = bias.detach().float()
        y = torch.empty((B, ho, wo, prep.cop), dtype=x.dtype, device=x.device)
        if s is None and residual is None and prep.s2d_ok(pad, pad_mode, Hh, Ww):
            # image conv with 3 (1) real outputs: 4x4 output pixels per GEMM row
            r = prep.S2D
            w2 = prep.s2d_weights()
            b2 = None
            if bias_p is not None:
                b2 = torch.zeros(w2.shape[0], dtype=torch.float32, device=x.device)
                b2[: r * r * prep.co] = bias_p[: prep.co].repeat(r * r)
            y2 = torch.empty((B, Hh // r, Ww // r, w2.shape[0]), dtype=x.dtype, device=x.device)
            H.conv2d_fwd(x, w2, y2, bias=b2, pad=pad, pad_mode=pad_mode, act=act, stride=r)
            y.zero_()
            y.view(B, Hh // r, r, Ww // r, r, prep.cop)[..., : prep.co] = (
                y2[..., : r * r * prep.co].view(B, Hh // r, Ww // r, r, r, prep.co).permute(0, 1, 3, 2, 4, 5))
        elif s is not None and (ho * wo) % 256 == 0:
            # style folded into per-sample filters (rounded after folding): the A-operand
            # loader stays a plain copy, like the unmodulated conv
            w_b = torch.empty((B, *w_f.shape), dtype=x.dtype, device=x.device)
            H.modulate_weights(w32, s, w_b)
            if prep.fp8_ok(False, B * ho * wo):  # config #5: e4m3 activations x e4m3 per-sample filters
                rec = torch.empty(4, dtype=torch.float32, device=x.device)
                x8 = _quantize(x, torch.float8_e4m3fn, rec[0:2], prep.fp8_site("x", x.device))
                w8 = _quantize(w_b, torch.float8_e4m3fn, rec[2:4], prep.fp8_site("w", x.device))
                H.conv2d_fwd(x8, w8, y, out_scale=d, bias=bias_p, residual=residual, pad=pad,
                             pad_mode=pad_mode, act=act, per_sample_w=True, deq=rec)
            else:
                H.conv2d_fwd(x, w_b, y, out_scale=d, bias=bias_p, residual=residual, pad=pad,
                             pad_mode=pad_mode, act=act, per_sample_w=True)
        elif stats_eps is not None:
            # conv feeding an InstanceNorm: the epilogue emits the per-(sample, channel) partial sums
            # of y on its way out, so the statistics pass over y is not run (odd-sized maps: fallback)
            if s is not None or residual is not None or act != H.ACT_NONE:
                raise RuntimeError("InstanceNorm statistics are emitted by plain convolutions only")
            mr = torch.empty((B, prep.cop, 2), dtype=torch.float32, device=x.device)
            xin, win, deq = x, w_f, None
            if prep.fp8_ok(False, B * ho * wo):
                win, deq = prep.get_fp8(False)
                deq = deq.clone()  # per call: the two groups of a generator step use a layer on two streams at once
                xin = _quantize(x, torch.float8_e4m3fn, deq[0:2], prep.fp8_site("x", x.device))
            # partial rows per sample the selected kernel's epilogue writes (0: none -- separate statistics pass)
            nchunks = H.conv2d_stats_chunks(xin, win, y, pad=pad) if _FUSED_IN_STATS else 0
            part = None
            if nchunks:
                part = torch.empty(B * nchunks * prep.cop * 2, dtype=torch.float32, device=x.device)
            H.conv2d_fwd(xin, win, y, bias=bias_p, pad=pad, pad_mode=pad_mode, act=act, stats=part, deq=deq)
            if nchunks:
                H.instnorm_finalize(part, mr, ho * wo, nchunks, stats_eps)
            else:
                ws = torch.empty(H.instnorm_ws_floats(B, ho * wo, prep.cop), dtype=torch.float32, device=x.device)
                H.instnorm_stats(y, ws, mr, stats_eps)
        elif s is None and prep.fp8_ok(False, B * ho * wo):
            w8, rec = prep.get_fp8(False)
            rec = rec.clone()  # (per call, see above)
            H.conv2d_fwd(_quantize(x, torch.float8_e4m3fn, rec[0:2], prep.fp8_site("x", x.device)), w8, y, out_scale=d, bias=bias_p,
                         residual=residual, pad=pad, pad_mode=pad_mode, act=act, deq=rec)
        else:
            H.conv2d_fwd(x, w_f, y, in_scale=s, out_scale=d, bias=bias_p, residual=residual,
                         pad=pad, pad_mode=pad_mode, act=act)
        ctx.prep, ctx.pad, ctx.pad_mode, ctx.act = prep, pad, pad_mode, act
        # the (mean, rstd) output never carries a gradient: do not let the engine build a zero tensor for it per call
        ctx.set_materialize_grads(False)
        ctx.link = link if _BLOCK_LINK else None
        ctx.norm_follows = stats_eps is not None
        ctx.counted = prep.note_forward_use(ctx.needs_input_grad[1])
        ctx.ts_params = (ts_weight, ts_bias)
        ctx.bias_param = bias
        ctx.has_bias, ctx.has_res = bias is not None, residual is not None
        ctx.save_for_backward(x, y, residual, s, d, weight, bias_p, wv, ws)
        if stats_eps is not None:
            ctx.mark_non_differentiable(mr)  # InstanceNorm's backward carries the dependence on y itself
            return y, mr
        return y

    @staticmethod
    def backward(ctx, g, *_unused):
        x, y, residual, s, d, weight, bias_p, wv, ws = ctx.saved_tensors
        prep, pad, pad_mode, act = ctx.prep, ctx.pad, ctx.pad_mode, ctx.act
        w_f, w_d, q, _, _ = prep.get()
        if g is None:  # (set_materialize_grads(False): y itself took no gradient -- nothing flows through this conv)
            g = torch.zeros_like(y)
        g = g.contiguous()
        B, Hh, Ww, cip = x.shape
        need_x, need_w, need_b = ctx.needs_input_grad[0:3]
        need_s = s is not None and any(ctx.needs_input_grad[3:6])
        need_res = ctx.needs_input_grad[6]
        dev = g.device
        link = ctx.link
        head = link is not None and not ctx.has_res   # first conv of a residual block
        tail = link is not None and ctx.has_res       # the conv that adds the block's residual
        deferred = None
        if head and link.deferred is not None:
            deferred, link.deferred = link.deferred, None
            if g.data_ptr() != link.lazy_ptr:
                raise RuntimeError("BlockLink: the gradient of a block's inner activation arrived from outside the "
                                   "block (the deferred fold would be lost)")

        want_sums = act != H.ACT_NONE or d is not None or (ctx.has_bias and need_b and not ctx.norm_follows)
        run_style = s is not None and (need_s or (need_w and d is not None))
        want_dots = s is not None and (need_x or need_s or (need_w and d is not None))
        # data gradient of a zero-padded modulated conv: style scale and style dot out of the GEMM's epilogue
        kpad = prep.kh - 1 - (0 if pad_mode == H.PAD_REFLECT else pad)
        hp = Hh + (2 * pad if pad_mode == H.PAD_REFLECT else 0)
        wp = Ww + (2 * pad if pad_mode == H.PAD_REFLECT else 0)
        fuse_dot = (_FUSED_DGRAD_DOT and s is not None and pad_mode == H.PAD_ZERO and (need_x or need_s)
                    and not prep.fp8_ok(True, B * hp * wp))
        dot_rows = 0
        if fuse_dot:
            dot_rows = H.conv2d_stats_rows(g, w_d, x, pad=kpad)
            fuse_dot = dot_rows > 0
        sums = dots = None
        if want_sums or (want_dots and not fuse_dot):  # one zero-fill for both atomically accumulated tables
            ns = B * 2 * prep.cop if want_sums else 0
            nd = B * cip if (want_dots and not fuse_dot) else 0
            z = ZERO_POOL.take(ns + nd, dev)
            sums = z[:ns].view(B, 2, prep.cop) if want_sums else None
            dots = z[ns:].view(B, cip) if nd else None

        # ---- 1. gradient of the GEMM result: activation backward, demodulation folded in -------------------
        if deferred is not None:
            # the block's second conv parked its padded data gradient: fold + its style scale + its style dot
            # + THIS conv's activation backward and sums in one pass over (padded gradient, u)
            gu = torch.empty_like(g)
            H.fold_scale_dot(deferred["gxp"], y, deferred["s"], gu, deferred["dots"], deferred["pad"],
                             xs=deferred["xs"], act=act, act_mul=d, act_sums=sums)
            deferred["finish"]()
        elif act != H.ACT_NONE or d is not None:
            gu = torch.empty_like(g)
            # u = y - residual = act(pre); the stored tensor is gu * d (demodulation folded in)
            H.act_bwd_reduce(g, y, residual, d, gu, sums, act)
        else:
            gu = g

        g_bias = None
        if ctx.has_bias and need_b and ctx.norm_follows:
            # A bias ahead of InstanceNorm is mathematically dead: the incoming gradient (InstanceNorm's
            # backward) has zero mean over the pixels of every (sample, channel), so its sum is exactly 0
            # -- the reference's value there is rounding noise.  No reduction pass.
            g_bias = None  # (no gradient = the zero the optimiser's zero_grad left in place)
            # no AccumulateGrad node runs for this parameter, so its post-accumulate hook never fires: tell the
            # data-parallel reducer here that the (zero) gradient is complete -- otherwise the bucket segment that
            # holds the bias, and every earlier one behind it, would wait for the end of backward (dist.BucketReducer
            # launches segments strictly last to first; its ``seen`` set ignores repeats of a multiply used layer)
            hook = GRAD_READY_HOOKS.get(ctx.bias_param)
            if hook is not None:
                hook(ctx.bias_param)
        elif ctx.has_bias and need_b:
            if act == H.ACT_NONE and d is None:  # reduce-only pass over g (one read, nothing stored)
                H.act_bwd_reduce(g, None, None, None, None, sums, H.ACT_NONE)
            tot = sums[:, 0].sum(0)
            g_bias = tot[: prep.co].to(weight.dtype)

        # ---- 2. data gradient ---------------------------------------------------------------------------
        res_in = None
        if head and link.res_grad is not None:
            res_in, link.res_grad = link.res_grad, None
        g_res = g if (ctx.has_res and need_res) else None
        if tail and g_res is not None and need_x:
            link.res_grad, g_res = g_res, None  # the block's first conv adds it to its data gradient

        # to_style gradients: a decoder layer is applied five times per generator step; when the parameters
        # already own fp32 .grad buffers (FusedAdam's flat bucket) the kernel adds straight into them instead of
        # autograd summing temporaries per parameter
        direct = make_direct = False
        if run_style:
            tsw, tsb = ctx.ts_params

            def grad_ok(p):
                return p.grad.dtype == torch.float32 and p.grad.is_contiguous() and p.grad.device == dev

            wanted = all(ctx.needs_input_grad[k] and p is not None for k, p in ((4, tsw), (5, tsb)))
            direct = _DIRECT_STYLE_GRADS and wanted and all(p.grad is not None and grad_ok(p) for p in (tsw, tsb))
            # a deferred style backward (below) fills its outputs AFTER this function has returned them: only the
            # in-place route into .grad is safe then (autograd would accumulate the returned temporaries at once)
            make_direct = wanted and all(p.grad is None or grad_ok(p) for p in (tsw, tsb))

        g_x = xs = None
        defer_fold = False
        if need_x or need_s:
            if fuse_dot:
                nchunks = hp * wp // dot_rows
                part = torch.empty(B * nchunks * cip * 2, dtype=torch.float32, device=dev)
                g_x = torch.empty_like(x)
                if need_w and _WGRAD_XS and not _wgrad_scales_in_kernel(prep, x, pad, pad_mode):
                    xs = torch.empty_like(x)  # x * s for the weight gradient, written while x is in registers
                H.conv2d_fwd(gu, w_d, g_x, out_scale=s, pad=kpad, pad_mode=H.PAD_ZERO, act=H.ACT_NONE,
                             stats=part, aux=x, aux_scaled=xs)
                dots = torch.empty((B, cip), dtype=torch.float32, device=dev)
                H.conv2d_dots_finalize(part, dots, nchunks)
            elif s is None and _border_dgrad_ok(prep, g, Hh, Ww, pad, pad_mode) and not prep.fp8_ok(True, B * hp * wp):
                # plain 3 x 3 conv behind ReflectionPad2d(1) (the encoder's residual blocks): cropped-domain conv (the
                # block's residual gradient added by its epilogue) + the border ring
                g_x = torch.empty_like(x)
                H.conv2d_fwd(gu, w_d, g_x, pad=1, pad_mode=H.PAD_ZERO, act=H.ACT_NONE, residual=res_in)
                H.conv2d_reflect_border(gu, w_d, g_x)
                res_in = None
            elif (_FOLD_EPILOGUE and s is None and pad_mode == H.PAD_REFLECT and pad == 1 and g.dtype == torch.bfloat16
                  and not deterministic() and not prep.fp8_ok(True, B * hp * wp)
                  and min(Hh, Ww) >= 2 * pad + 2):
                # plain 3 x 3 conv behind ReflectionPad2d(1) (the encoder's residual blocks): the GEMM's epilogue adds
                # every pixel of the padded domain at its mirror image of g_x (o2m_conv_desc.fold_pad) and the block's
                # residual gradient with it -- no padded gradient, no fold pass.  (Not the 7 x 7 head behind
                # ReflectionPad2d(3): 9 % of its 256 x 256 x 64 map would go through atomics -- 14 M of them per launch of
                # a 0.4 ms kernel, measured +1.3 ms -- profiles/README.md.)
                g_x = torch.empty_like(x)
                H.conv2d_fwd(gu, w_d, g_x, pad=kpad, pad_mode=H.PAD_ZERO, act=H.ACT_NONE, residual=res_in, fold_pad=pad)
                res_in = None
            else:
                fold = pad if pad_mode == H.PAD_REFLECT else 0  # what the pass behind the GEMM still has to fold
                border = (s is not None and _border_dgrad_ok(prep, g, Hh, Ww, pad, pad_mode, modulated=True)
                          and not prep.fp8_ok(True, B * hp * wp))
                if border:  # modulated conv behind ReflectionPad2d(1): cropped-domain conv + border ring, nothing left to fold
                    gxp = torch.empty((B, Hh, Ww, cip), dtype=g.dtype, device=dev)
                    H.conv2d_fwd(gu, w_d, gxp, pad=1, pad_mode=H.PAD_ZERO, act=H.ACT_NONE)
                    H.conv2d_reflect_border(gu, w_d, gxp)
                    fold = 0
                else:
                    gxp = torch.empty((B, hp, wp, cip), dtype=g.dtype, device=dev)
                if border:
                    pass
                elif prep.fp8_ok(True, B * hp * wp):  # e5m2 gradients x e4m3 filter
                    w8, rec = prep.get_fp8(True)
                    rec = rec.clone()  # (per call: the activation half is this call's)
                    H.conv2d_fwd(_quantize(gu, torch.float8_e5m2, rec[0:2], prep.fp8_site("g", dev)), w8, gxp, pad=kpad, pad_mode=H.PAD_ZERO,
                                 act=H.ACT_NONE, deq=rec)
                else:
                    H.conv2d_fwd(gu, w_d, gxp, pad=kpad, pad_mode=H.PAD_ZERO, act=H.ACT_NONE)
                if s is not None or pad_mode == H.PAD_REFLECT:
                    # the block's first conv finishes this (BlockLink.deferred) when it can fuse its activation
                    # backward: its output is this conv's input, and nothing else wants dL/dx
                    # (not in fp8 mode: the combination deferred fold + weight-gradient stream makes the open fp8 issue
                    # described in optim.FusedAdam.step -- a non-finite slice partial of the phase-pipelined weight
                    # gradient -- show up within a few steps on every run, against 2 runs in 6 over 40 steps without the
                    # deferral: tools/fp8_nan_probe.py; O2M_FP8_DEFER=1 brings the combination back for debugging)
                    defer_fold = (tail and link.fuse_act and s is not None and need_x
                                  and (not run_style or direct or make_direct)
                                  and (not fp8_enabled() or _os.environ.get("O2M_FP8_DEFER") == "1"))
                    if defer_fold and run_style and not direct:
                        for p in (tsw, tsb):
                            if p.grad is None:
                                p.grad = torch.zeros_like(p, dtype=torch.float32)
                        direct = True
                    if defer_fold:
                        g_x = torch.empty_like(x)  # never read: the head checks that it gets exactly this back
                        link.lazy_ptr = g_x.data_ptr()
                        if need_w and _WGRAD_XS:
                            xs = torch.empty_like(x)  # written by the head's fused kernel
                    else:
                        g_x = torch.empty_like(x)
                        if s is not None and need_w and _WGRAD_XS and not _wgrad_scales_in_kernel(prep, x, pad, pad_mode):
                            xs = torch.empty_like(x)  # x * s for the weight gradient, written while x is being read
                        H.fold_scale_dot(gxp, x if s is not None else None, s, g_x, dots, fold, xs=xs, gres=res_in)
                        res_in = None
                else:
                    g_x = gxp
            if res_in is not None:  # no fold pass to carry it (fused epilogue / plain conv)
                g_x = g_x + res_in
                res_in = None

        ev_pre = None
        if run_style:  # outputs allocated (and sums / dots complete) before the event the side stream waits on
            wd_ = wv.shape[1]
            e = torch.empty((B, prep.cop), dtype=torch.float32, device=dev) if d is not None else None
            gs = torch.empty((B, cip), dtype=torch.float32, device=dev)
            g_ws = torch.empty((B, wd_), dtype=torch.float32, device=dev)
            if direct:
                g_tw, g_tb = tsw.grad, tsb.grad
                prep.direct_style = (tsw, tsb)  # reported complete together with the filter (_finalize_layer)
            else:
                g_tw = torch.empty((prep.ci, wd_), dtype=torch.float32, device=dev)
                g_tb = torch.empty((prep.ci,), dtype=torch.float32, device=dev)
            gq_tmp = None
            if d is not None and not need_w:  # dL/dQ is discarded when the filter wants no gradient
                gq_tmp = torch.zeros((prep.cop, cip), dtype=torch.float32, device=dev)
            if _side_stream(dev) is not None and not defer_fold:
                ev_pre = torch.cuda.Event()
                ev_pre.record(torch.cuda.current_stream(dev))

        # ---- 3. weight gradient (own stream) ---------------------------------------------------------------
        gq_acc = dw_acc = None
        if need_w:
            # accumulated in the kernel layout across every use of the layer in this backward;
            # converted into weight.grad once, by _finalize_weight_grads
            dw_acc, gq_acc = prep.accumulators(dev)

        def launch_wgrad():
            wst = _wgrad_stream(dev)
            if wst is not None:
                ev_w = torch.cuda.Event()
                ev_w.record(torch.cuda.current_stream(dev))
                wst.wait_event(ev_w)
                for t in (x, xs, gu, s):  # keep the operands alive until the side stream is done
                    if t is not None:
                        t.record_stream(wst)
            with (torch.cuda.stream(wst) if wst is not None else contextlib.nullcontext()):
                if s is None and residual is None and prep.s2d_ok(pad, pad_mode, Hh, Ww):
                    prep.s2d_wgrad(x, gu, pad, pad_mode)
                elif xs is not None:
                    H.conv2d_wgrad(xs, gu, dw_acc, pad=pad, pad_mode=pad_mode, p8=_wgrad_p8(wst))
                else:  # plain conv (s None), or the modulated input formed while x is staged (in_scale)
                    H.conv2d_wgrad(x, gu, dw_acc, in_scale=s, pad=pad, pad_mode=pad_mode, p8=_wgrad_p8(wst))

        if need_w and not defer_fold:
            launch_wgrad()

        # ---- 4. style path (B x C sized) -------------------------------------------------------------------
        def style_and_count():
            if run_style:
                gq = (gq_acc if gq_acc is not None else gq_tmp) if d is not None else None
                side = None if defer_fold else _side_stream(dev)
                if side is not None:  # overlaps the weight-gradient kernel enqueued just above
                    main = torch.cuda.current_stream(dev)
                    side.wait_event(ev_pre)
                    with torch.cuda.stream(side):
                        H.style_bwd(sums, bias_p, dots, s, d, q, wv, ws, e, gs, g_ws, g_tw, g_tb, gq, prep.ci,
                                    1.0 / math.sqrt(wd_), accumulate=direct)
                        ev_done = torch.cuda.Event()
                        ev_done.record(side)
                    main.wait_event(ev_done)
                else:
                    H.style_bwd(sums, bias_p, dots, s, d, q, wv, ws, e, gs, g_ws, g_tw, g_tb, gq, prep.ci,
                                1.0 / math.sqrt(wd_), accumulate=direct)
            if need_w and ctx.counted:  # after style_bwd above: it adds this use's dL/dQ
                prep.use_reduced(dev)

        if defer_fold:
            # dots is written by the head's fused kernel; the style backward reads it: both run from there.
            # (g_ws is allocated now and filled then -- before anything downstream of this backward can read it:
            # the head's backward runs earlier in stream order than any consumer of the style gradients.)
            def finish():
                if need_w:
                    launch_wgrad()  # x * s has just been written by the head's fused kernel
                style_and_count()

            link.deferred = {"gxp": gxp, "s": s, "dots": dots, "pad": fold, "xs": xs, "finish": finish}
        else:
            style_and_count()
        # (separate names: style_and_count may run later, from the block's first conv, and reads g_tw / g_tb)
        r_ws, r_tw, r_tb = (g_ws, None if direct else g_tw, None if direct else g_tb) if run_style else (None,) * 3
        return (g_x if need_x else None, None, g_bias, r_ws, r_tw, r_tb, g_res,
                None, None, None, None, None, None, None, None)


# Debug tap (None in production): a list that receives, in execution order, the sign mask of every
# fused ReLU / LeakyReLU output as a CPU bool NCHW tensor (padded channels included).  The parity
# suite replays these masks in the fp64 oracle to separate kernel error from activation-mask flips.
ACT_TAP = None


def _tap_activation(y, act, residual):
    if ACT_TAP is not None and act in (H.ACT_RELU, H.ACT_LRELU):
        u = y.detach() if residual is None else y.detach().float() - residual.detach().float()
        ACT_TAP.append((u > 0).permute(0, 3, 1, 2).cpu())


def conv2d(x, weight, bias, prep, *, pad, pad_mode=H.PAD_ZERO, act=H.ACT_NONE, style=None,
           residual=None, demodulate=True, eps=1e-8, norm_eps=None, link=None):
    """``style`` = (w, to_style.weight, to_style.bias) for the modulated conv, else None.
    ``norm_eps``: the conv feeds an InstanceNorm with this eps; returns ``(y, mean_rstd)`` with the
    statistics of y ([B][C][2] fp32) for ``instance_norm_act(..., stats=mean_rstd)``.
    ``link``: the BlockLink shared by the two convs of one residual-block application."""
    w_style, ts_w, ts_b = style if style is not None else (None, None, None)
    out = _ConvFn.apply(x, weight, bias, w_style, ts_w, ts_b, residual, prep, pad, pad_mode, act,
                        demodulate, eps, norm_eps, link)
    if norm_eps is None:
        _tap_activation(out, act, residual)
    return out


# --------------------------------------------------------------------------- instance norm


class _InstNormFn(torch.autograd.Function):
    """y = act(InstanceNorm2d(x)) + residual  (eps 1e-5, biased variance, no affine)."""

    @staticmethod
    def forward(ctx, x, residual, act, eps, mr, link=None):
        ctx.link = link if (_BLOCK_LINK and residual is not None) else None
        B, Hh, Ww, Cn = x.shape
        if mr is None:
            ws = torch.empty(H.instnorm_ws_floats(B, Hh * Ww, Cn), dtype=torch.float32, device=x.device)
            mr = torch.empty((B, Cn, 2), dtype=torch.float32, device=x.device)
            H.instnorm_stats(x, ws, mr, eps)
        y = torch.empty_like(x)
        H.instnorm_apply(x, mr, residual, y, act)
        ctx.act = act
        ctx.save_for_backward(x, mr)
        return y

    @staticmethod
    def backward(ctx, g):
        x, mr = ctx.saved_tensors
        g = g.contiguous()
        B, Hh, Ww, Cn = x.shape
        gx = None
        if ctx.needs_input_grad[0]:
            ws = torch.empty(H.instnorm_ws_floats(B, Hh * Ww, Cn), dtype=torch.float32, device=x.device)
            gs = torch.empty((B, Cn, 2), dtype=torch.float32, device=x.device)
            gx = torch.empty_like(x)
            H.instnorm_bwd(g, x, mr, ws, gs, gx, ctx.act)
        g_res = g if ctx.needs_input_grad[1] else None
        if ctx.link is not None and g_res is not None:
            ctx.link.res_grad, g_res = g_res, None  # added by the block's first conv (BlockLink)
        return gx, g_res, None, None, None, None


def instance_norm_act(x, act=H.ACT_NONE, residual=None, eps=1e-5, stats=None, link=None):
    """``stats``: mean / rstd of x already computed (by the epilogue of the conv that produced x).
    ``link``: BlockLink of the residual block this closes (the residual's gradient goes to the block's first conv)."""
    y = _InstNormFn.apply(x, residual, act, eps, stats, link)
    _tap_activation(y, act, residual)
    return y


# ------------------------------------------------------------------- batch plumbing of the fused step
#
# generator_step runs its three decodes as ONE 3B pass and its two feature extractions as ONE 2B pass (every
# op of the decoder is per-sample: core/training.py).  These three functions do the batch bookkeeping on
# internal buffers without autograd's generic slice / cat machinery (which materialises a full-size zero
# tensor per slice in backward).


class _BatchGatherFn(torch.autograd.Function):
    """out = cat([t[c*n:(c+1)*n] for c in chunks]) along the batch; backward sums the chunks back."""

    @staticmethod
    def forward(ctx, t, n, chunks):
        ctx.n, ctx.chunks, ctx.shape = n, tuple(chunks), t.shape
        return torch.cat([t[c * n:(c + 1) * n] for c in chunks], 0)

    @staticmethod
    def backward(ctx, g):
        n, gt, seen = ctx.n, None, set()
        g = g.contiguous()
        gt = torch.empty(ctx.shape, dtype=g.dtype, device=g.device)
        for k, c in enumerate(ctx.chunks):
            src, dst = g[k * n:(k + 1) * n], gt[c * n:(c + 1) * n]
            if c in seen:
                dst.add_(src)
            else:
                dst.copy_(src)
                seen.add(c)
        for c in range(ctx.shape[0] // n):
            if c not in seen:
                gt[c * n:(c + 1) * n].zero_()
        return gt, None, None


def batch_gather(t, n, chunks):
    """Internal buffer [m*n, ...] -> [len(chunks)*n, ...] made of its n-sample chunks ``chunks`` (repeats allowed)."""
    return _BatchGatherFn.apply(t, n, chunks)


class _SplitBatchFn(torch.autograd.Function):
    """k equal, contiguous batch slices of an internal buffer (views); backward is one cat."""

    @staticmethod
    def forward(ctx, t, k):
        ctx.k, ctx.shape = k, t.shape
        return tuple(t.chunk(k, 0))

    @staticmethod
    def backward(ctx, *gs):
        ref = next(g for g in gs if g is not None)
        n = ctx.shape[0] // ctx.k
        parts = [g if g is not None else torch.zeros((n, *ctx.shape[1:]), dtype=ref.dtype, device=ref.device)
                 for g in gs]
        return torch.cat(parts, 0), None


def split_batch(t, k):
    return _SplitBatchFn.apply(t, k)


# O2M_FOLD_EPILOGUE=0: the data gradient of a reflection-padded plain conv as padded GEMM output + fold pass (rounds 1-3;
# always so in deterministic mode: the epilogue form adds the mirrored rows / columns with bf16 atomics)
_FOLD_EPILOGUE = _os.environ.get("O2M_FOLD_EPILOGUE", "1") == "1"

# O2M_FUSED_NORM_DOWN=0: InstanceNorm + activation and the DownSample behind it as two passes (round-2 form)
_FUSED_NORM_DOWN = _os.environ.get("O2M_FUSED_NORM_DOWN", "1") == "1"


class _InstNormDownFn(torch.autograd.Function):
    """y = DownSample(act(InstanceNorm2d(x))) in one pass (builder.py:170-173,272-282); the normalised map is
    neither stored nor needed: the backward gathers the fine gradient D^T g inside both InstanceNorm passes."""

    @staticmethod
    def forward(ctx, x, mr, act, kind):
        B, Hh, Ww, Cn = x.shape
        sy, wy, sx, wx, T, ho, wo = R.taps(kind, Hh, Ww, False, x.device)
        y = torch.empty((B, ho, wo, Cn), dtype=x.dtype, device=x.device)
        H.instnorm_act_resample2d(x, mr, y, sy, wy, sx, wx, int(T), T.span_y, T.span_x, act)
        ctx.act, ctx.kind = act, kind
        ctx.save_for_backward(x, mr)
        return y

    @staticmethod
    def backward(ctx, g):
        x, mr = ctx.saved_tensors
        g = g.contiguous()
        B, Hh, Ww, Cn = x.shape
        sy, wy, sx, wx, T, _, _ = R.taps(ctx.kind, Hh, Ww, True, x.device)  # the transposed operator: fine <- coarse
        ws = torch.empty(H.instnorm_ws_floats(B, Hh * Ww, Cn), dtype=torch.float32, device=x.device)
        gs = torch.empty((B, Cn, 2), dtype=torch.float32, device=x.device)
        gx = torch.empty_like(x)
        H.instnorm_resample_bwd(g, x, mr, ws, gs, gx, sy, wy, sx, wx, int(T), ctx.act)
        return gx, None, None, None


def norm_down_fusable(x, kind, stats) -> bool:
    """Whether instance_norm_act + resample(kind) can run as the fused pair: statistics in hand, a DownSample
    operator the kernels cover (4 taps forward / 2 taps transposed), no activation tap being recorded."""
    if not _FUSED_NORM_DOWN or stats is None or ACT_TAP is not None or kind not in ("down", "down_nosmooth"):
        return False
    Hh, Ww = x.shape[1], x.shape[2]
    T = R.taps(kind, Hh, Ww, False, x.device)[4]
    Tt = R.taps(kind, Hh, Ww, True, x.device)[4]
    return int(T) == 4 and T.span_y in (2, 3) and T.span_x in (2, 3) and int(Tt) == 2


def instance_norm_act_down(x, act, kind, stats):
    return _InstNormDownFn.apply(x, stats, act, kind)


# -------------------------------------------------------------------------------- resample


# O2M_SPLIT_WIDE_RESAMPLE=1: the 6-tap transposed upsample as a vertical and a horizontal 1-D launch (the round-2
# form, kept for A/B) instead of the one-launch LDS tile kernel
_SPLIT_WIDE_RESAMPLE = _os.environ.get("O2M_SPLIT_WIDE_RESAMPLE", "0") == "1"


def _apply_taps(x, taps):
    """One banded 2-D operator on an NHWC buffer (the 6-tap transposed upsample -- 36 taps per output -- runs
    as the separable LDS tile kernel of o2m_resample2d)."""
    sy, wy, sx, wx, T, ho, wo = taps
    B, Hh, Ww, Cn = x.shape
    y = torch.empty((B, ho, wo, Cn), dtype=x.dtype, device=x.device)
    if int(T) == 6 and T.span_y == 2 and T.span_x == 2 and _SPLIT_WIDE_RESAMPLE:  # A/B: two 1-D passes
        ix, iwx = R.identity_taps(Ww, x.device)
        iy, iwy = R.identity_taps(ho, x.device)
        mid = torch.empty((B, ho, Ww, Cn), dtype=x.dtype, device=x.device)
        H.resample2d(x, mid, sy, wy, ix, iwx, int(T), 1, T.span_y, 1)
        H.resample2d(mid, y, iy, iwy, sx, wx, 1, int(T), 1, T.span_x)
    else:
        H.resample2d(x, y, sy, wy, sx, wx, int(T), int(T), T.span_y, T.span_x)
    return y


# ---- functional operators registered with the dispatcher ------------------------------------------------------------
# the reference's operator API is torch.nn.functional; the build's is namespace o2m::.  The
# launchers are out-variant ops (csrc/torch_ops.cpp: TORCH_LIBRARY(o2m, ...)); the FUNCTIONAL forms below -- the ones with a
# derivative and no Python-side state -- are defined as fragment ops of the same namespace with their autograd formula
# attached by torch.library.register_autograd, so torch.ops.o2m.resample / pair_sum / moments / instance_norm_act
# differentiate like any ATen op (and carry a fake kernel for shape inference).  The conv functions keep a
# torch.autograd.Function: their backward hands tensors between the two convs of a block (BlockLink), defers work to
# other streams and accumulates weight gradients in kernel layout across uses -- state an operator schema cannot carry.
_RESAMPLE_KINDS = ("blur", "up", "down")


def _resample_out_hw(kind, Hh, Ww, transposed):
    if kind == "blur":
        return Hh, Ww
    if (kind == "up") != transposed:
        return 2 * Hh, 2 * Ww
    return Hh // 2, Ww // 2


@torch.library.custom_op("o2m::resample", mutates_args=())
def _resample_op(x: torch.Tensor, kind: str) -> torch.Tensor:
    B, Hh, Ww, Cn = x.shape
    return _apply_taps(x.contiguous(), R.taps(kind, Hh, Ww, False, x.device))


@_resample_op.register_fake
def _(x, kind):
    ho, wo = _resample_out_hw(kind, x.shape[1], x.shape[2], False)
    return x.new_empty((x.shape[0], ho, wo, x.shape[3]))


@torch.library.custom_op("o2m::resample_transposed", mutates_args=())
def _resample_t_op(g: torch.Tensor, kind: str, height: int, width: int) -> torch.Tensor:
    """Adjoint of o2m::resample on an input of height x width."""
    return _apply_taps(g.contiguous(), R.taps(kind, height, width, True, g.device))


@_resample_t_op.register_fake
def _(g, kind, height, width):
    return g.new_empty((g.shape[0], height, width, g.shape[3]))


def _resample_setup(ctx, inputs, output):
    x, kind = inputs
    ctx.kind, ctx.hw = kind, (x.shape[1], x.shape[2])


def _resample_bwd(ctx, g):
    return torch.ops.o2m.resample_transposed(g, ctx.kind, ctx.hw[0], ctx.hw[1]), None


def _resample_t_setup(ctx, inputs, output):
    ctx.kind = inputs[1]


def _resample_t_bwd(ctx, gg):  # the adjoint of the adjoint is the operator
    return torch.ops.o2m.resample(gg, ctx.kind), None, None, None


torch.library.register_autograd("o2m::resample", _resample_bwd, setup_context=_resample_setup)
torch.library.register_autograd("o2m::resample_transposed", _resample_t_bwd, setup_context=_resample_t_setup)


def resample(x, kind):
    """kind in {"blur", "up", "down"} (Smooth / UpSample / DownSample of layers.py)."""
    if kind not in _RESAMPLE_KINDS:
        raise ValueError(kind)
    return torch.ops.o2m.resample(x, kind)


# ---------------------------------------------------------------------------------- losses


def _partials(n_elems, n_out, device):
    return torch.empty(n_out * H.reduce_blocks(n_elems), dtype=torch.float32, device=device)


@torch.library.custom_op("o2m::pair_sum", mutates_args=())
def _pair_sum_op(a: torch.Tensor, b: torch.Tensor | None, w: torch.Tensor | None, mode: int) -> torch.Tensor:
    """sum over all elements of |a-b| (mode RED_L1) or w[b]*(a-b)^2 (RED_SQ); an fp32 scalar."""
    part = _partials(a.numel(), 1, a.device)
    H.reduce_fwd(a, b, w, part, mode)
    return part.sum()


@_pair_sum_op.register_fake
def _(a, b, w, mode):
    return a.new_empty((), dtype=torch.float32)


@torch.library.custom_op("o2m::pair_sum_grad", mutates_args=())
def _pair_sum_grad_op(a: torch.Tensor, b: torch.Tensor | None, w: torch.Tensor | None, coef: torch.Tensor,
                      mode: int) -> torch.Tensor:
    """d pair_sum / d a scaled by the device scalar coef (the gradient of b is its negative)."""
    ga = torch.empty_like(a)
    H.reduce_bwd(a, b, w, coef, ga, mode)
    return ga


@_pair_sum_grad_op.register_fake
def _(a, b, w, coef, mode):
    return torch.empty_like(a)


def _pair_sum_setup(ctx, inputs, output):
    a, b, w, mode = inputs
    ctx.mode = mode
    ctx.save_for_backward(a, b, w)


def _pair_sum_bwd(ctx, g):
    a, b, w = ctx.saved_tensors
    coef = (g.float() * (2.0 if ctx.mode == H.RED_SQ else 1.0)).reshape(1).contiguous()
    ga = torch.ops.o2m.pair_sum_grad(a, b, w, coef, ctx.mode)
    gb = -ga if (b is not None and ctx.needs_input_grad[1]) else None
    return (ga if ctx.needs_input_grad[0] else None), gb, None, None


torch.library.register_autograd("o2m::pair_sum", _pair_sum_bwd, setup_context=_pair_sum_setup)


@torch.library.custom_op("o2m::moments", mutates_args=())
def _moments_op(a: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor]:
    """(sum a, sum a^2) as fp32 scalars (kl_loss_func, loss.py:86-87)."""
    nb = H.reduce_blocks(a.numel())
    part = _partials(a.numel(), 2, a.device)
    H.reduce_fwd(a, None, None, part, H.RED_MOM)
    return part[:nb].sum(), part[nb:].sum()


@_moments_op.register_fake
def _(a):
    return a.new_empty((), dtype=torch.float32), a.new_empty((), dtype=torch.float32)


def _moments_setup(ctx, inputs, output):
    ctx.save_for_backward(inputs[0])


def _moments_bwd(ctx, g1, g2):
    (a,) = ctx.saved_tensors
    z = torch.zeros((), dtype=torch.float32, device=a.device)
    coef = torch.stack([(g1 if g1 is not None else z).float(), 2.0 * (g2 if g2 is not None else z).float()]).contiguous()
    return torch.ops.o2m.pair_sum_grad(a, None, None, coef, H.RED_MOM)


torch.library.register_autograd("o2m::moments", _moments_bwd, setup_context=_moments_setup)


class _HalvesSqFn(torch.autograd.Function):
    """sum_b w[b] * sum (t[b] - t[b + n])^2 over the two halves of a 2n batch (path_loss_func on the two
    extraction passes that generator_step runs as one batch); returns an fp32 scalar."""

    @staticmethod
    def forward(ctx, t, w):
        n = t.shape[0] // 2
        a, b = t[:n], t[n:]
        part = _partials(a.numel(), 1, t.device)
        H.reduce_fwd(a, b, w, part, H.RED_SQ)
        ctx.save_for_backward(t, w)
        return part.sum()

    @staticmethod
    def backward(ctx, g):
        t, w = ctx.saved_tensors
        n = t.shape[0] // 2
        coef = (g.float() * 2.0).reshape(1).contiguous()
        gt = torch.empty_like(t)
        H.reduce_bwd(t[:n], t[n:], w, coef, gt[:n], H.RED_SQ)
        torch.neg(gt[:n], out=gt[n:])
        return gt, None


def halves_sq_sum(t, w):
    return _HalvesSqFn.apply(t, w)


class _HalvesTapFn(torch.autograd.Function):
    """(t, sum_b w[b] * sum (t[b] - t[b + n])^2): the path-loss term of a decoder feature map that ALSO goes on to
    the next decoder layer (loss.py:98-111 on the maps Generator.extract returns, builder.py:226-253).  Taking the map
    THROUGH this function makes it the map's only consumer, so backward sees the next layer's gradient and the term's
    upstream scalar together and forms the map's gradient in one pass (o2m_pair_grad) -- instead of reduce_bwd + a
    negation pass for the second half + autograd's accumulation add of two activation-sized tensors."""

    @staticmethod
    def forward(ctx, t, w):
        n = t.shape[0] // 2
        part = _partials(t[:n].numel(), 1, t.device)
        H.reduce_fwd(t[:n], t[n:], w, part, H.RED_SQ)
        ctx.save_for_backward(t, w)
        return t.view_as(t), part.sum()

    @staticmethod
    def backward(ctx, g_t, g_s):
        t, w = ctx.saved_tensors
        n = t.shape[0] // 2
        if g_s is None:  # the term is unused: pass the map's gradient through
            return g_t, None
        coef = (g_s.float() * 2.0).reshape(1).contiguous()
        gt = torch.empty_like(t)
        gin = g_t.contiguous() if g_t is not None else None
        H.pair_grad(t[:n], t[n:], w, coef, gin[:n] if gin is not None else None, gin[n:] if gin is not None else None,
                    gt[:n], gt[n:])
        return gt, None


def halves_sq_tap(t, w):
    """``t`` (to be used in place of the argument from here on) and the pair term of its two batch halves."""
    return _HalvesTapFn.apply(t, w)


class _LsganFn(torch.autograd.Function):
    """[sum (s - t0)^2 over samples [0, n_first), sum (s - t1)^2 over the rest, sum sign(2 s - 1) over each half] of a
    discriminator patch map (internal [N][h][w][C], channel 0 = the score): the adversarial loss and the confidence of
    training.py:111-118,202 from one launch (o2m_lsgan_fwd); the confidences carry no gradient."""

    @staticmethod
    def forward(ctx, scores, n_first, t0, t1):
        out = torch.empty(4, dtype=torch.float32, device=scores.device)
        H.lsgan_fwd(scores, out, n_first, t0, t1)
        ctx.save_for_backward(scores)
        ctx.args = (n_first, t0, t1)
        return out

    @staticmethod
    def backward(ctx, g):
        (scores,) = ctx.saved_tensors
        gs = torch.empty_like(scores)
        H.lsgan_bwd(scores, g.contiguous(), gs, *ctx.args)  # (the kernel reads g[0], g[1]; the sign sums are constants)
        return gs, None, None, None


def lsgan_sums(scores, n_first, t0, t1):
    return _LsganFn.apply(scores, n_first, float(t0), float(t1))


_WSUM_COEFS: dict = {}


class _WeightedSumFn(torch.autograd.Function):
    """sum_i c_i * x_i over device scalars with host constants c_i: one stack + one dot forward, one scaling backward
    -- instead of a mul and an add per term and their mirror images from the autograd engine (the loss arithmetic of
    training.py:236-243 and loss.py:104-109 was ~60 single-element launches per step)."""

    @staticmethod
    def forward(ctx, coefs, *xs):
        ctx.coefs = coefs
        return torch.dot(torch.stack([x.reshape(()) for x in xs]), coefs)

    @staticmethod
    def backward(ctx, g):
        return (None, *(g * ctx.coefs).unbind(0))


def weighted_sum(xs, cs):
    """``xs``: fp32 device scalars; ``cs``: Python floats."""
    dev = xs[0].device
    key = (dev, tuple(float(c) for c in cs))
    coefs = _WSUM_COEFS.get(key)
    if coefs is None:
        if len(_WSUM_COEFS) > 256:
            _WSUM_COEFS.clear()
        coefs = _WSUM_COEFS[key] = torch.tensor(key[1], dtype=torch.float32, device=dev)
    return _WeightedSumFn.apply(coefs, *[x.float() if x.dtype != torch.float32 else x for x in xs])


def l1_sum(a, b):
    return torch.ops.o2m.pair_sum(a, b, None, H.RED_L1)


def sq_sum(a, b=None, w=None):
    return torch.ops.o2m.pair_sum(a, b, w, H.RED_SQ)


def moments(a):
    return torch.ops.o2m.moments(a)
