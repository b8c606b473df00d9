"""Host binding of the HIP kernels.

Two shared objects, both built in-tree by build.py:

* ``lib/libo2m_hip.so``   -- the kernels behind the C ABI of include/o2m_hip.h (torch-free);
* ``lib/libo2m_torch.so`` -- csrc/torch_ops.cpp: ``TORCH_LIBRARY(o2m, ...)``, one dispatcher-visible
  op per launcher (``torch.ops.o2m.conv2d_fwd`` ...) with CUDA and Meta kernels.  Each op checks its
  tensors, takes the current HIP stream of their device under a device guard and calls the C ABI.

The wrappers below are what ops.py calls: they only pick the op and label the launch for
bench.py's HIP-event profile.  ``SIGNATURES`` (ctypes) mirrors the header one for one and is used
to verify at load time that the C library exports every declared symbol.

There is NO fallback: if a library is missing or a call is made without a GPU tensor the
import / call raises.  The product path never routes through PyTorch eager kernels for the
ops declared in the header, and never through the CPU oracle.
"""

from __future__ import annotations

import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# O2M_HIP_LIB: alternative build of the same ABI (kernel A/B experiments only)
LIB_PATH = os.environ.get("O2M_HIP_LIB") or os.path.join(_HERE, "lib", "libo2m_hip.so")
TORCH_LIB_PATH = os.path.join(_HERE, "lib", "libo2m_torch.so")

BF16, F32, FP8_E4M3, BF8_E5M2 = 0, 1, 2, 3
ACT_NONE, ACT_RELU, ACT_LRELU, ACT_TANH = 0, 1, 2, 3
PAD_ZERO, PAD_REFLECT = 0, 1
RED_L1, RED_SQ, RED_MOM = 0, 1, 2
STATS_MOMENTS, STATS_DOT = 0, 1
ABI_VERSION = 22

_vp, _i32, _i64, _f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float


class ConvDesc(C.Structure):
    _fields_ = [("x", _vp), ("w", _vp), ("y", _vp), ("in_scale", _vp), ("out_scale", _vp),
                ("bias", _vp), ("residual", _vp),
                ("B", _i32), ("H", _i32), ("W", _i32), ("Ci", _i32), ("Co", _i32), ("KH", _i32),
                ("KW", _i32), ("pad", _i32), ("pad_mode", _i32), ("act", _i32), ("dtype", _i32),
                ("w_batch_stride", _i32), ("stride", _i32), ("stats_mode", _i32), ("fold_pad", _i32), ("reserved1", _i32),
                ("stats", _vp),
                ("deq_scale", _vp), ("aux", _vp), ("aux_scaled", _vp)]


class PrepJob(C.Structure):
    """o2m_prep_job (include/o2m_hip.h)."""

    _fields_ = [("w", _vp), ("full", _vp), ("w_f", _vp), ("w_d", _vp), ("q", _vp), ("qt", _vp),
                ("Co", _i32), ("Ci", _i32), ("KK", _i32), ("Cop", _i32), ("Cip", _i32), ("c", _f32),
                ("first_block", _i32), ("reserved", _i32)]


class WfinJob(C.Structure):
    """o2m_wfin_job (include/o2m_hip.h)."""

    _fields_ = [("acc", _vp), ("gq", _vp), ("w32", _vp), ("grad", _vp),
                ("Co", _i32), ("Ci", _i32), ("KK", _i32), ("Cop", _i32), ("Cip", _i32), ("c", _f32),
                ("first_block", _i32), ("reserved", _i32)]


class WgradDesc(C.Structure):
    _fields_ = [("x", _vp), ("gy", _vp), ("dw", _vp), ("in_scale", _vp), ("gy_scale", _vp),
                ("B", _i32), ("H", _i32), ("W", _i32), ("Ci", _i32), ("Co", _i32), ("KH", _i32),
                ("KW", _i32), ("pad", _i32), ("pad_mode", _i32), ("dtype", _i32), ("splits", _i32),
                ("reserved0", _i32), ("stride", _i32), ("kernel_hint", _i32), ("slabs", _vp)]


# name -> (restype, argtypes); mirrors include/o2m_hip.h one for one
SIGNATURES = {
    "o2m_abi_version": (_i32, []),
    "o2m_launch_timing": (_i32, [_i32]),
    "o2m_launch_timing_read": (_i32, [_vp, _i32]),
    "o2m_debug_fill_blocks": (_i32, [_i32]),
    "o2m_conv2d_fwd": (_i32, [C.POINTER(ConvDesc), _vp]),
    "o2m_conv2d_reflect_border": (_i32, [C.POINTER(ConvDesc), _vp]),
    "o2m_conv2d_stats_rows": (_i32, [C.POINTER(ConvDesc)]),
    "o2m_conv2d_stats_chunks": (_i32, [C.POINTER(ConvDesc)]),
    "o2m_conv2d_dots_finalize": (_i32, [_vp, _vp, _i32, _i32, _i32, _vp]),
    "o2m_amax": (_i32, [_vp, _vp, _i64, _i32, _vp]),
    "o2m_quantize_fp8": (_i32, [_vp, _vp, _vp, _vp, _i64, _i32, _i32, _vp]),
    "o2m_quantize_fp8_delayed": (_i32, [_vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _vp]),
    "o2m_instnorm_finalize": (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _f32, _vp]),
    "o2m_conv2d_wgrad": (_i32, [C.POINTER(WgradDesc), _vp]),
    "o2m_conv2d_wgrad_slab_floats": (C.c_size_t, [C.POINTER(WgradDesc)]),
    "o2m_wgrad_finalize": (_i32, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _f32, _vp]),
    "o2m_style_fwd": (_i32, [_vp] * 6 + [_i32] * 5 + [_f32, _f32, _vp]),
    "o2m_style_bwd": (_i32, [_vp] * 14 + [_i32] * 5 + [_f32, _i32, _vp]),
    "o2m_chan_partials_floats": (C.c_size_t, [_i32, _i32, _i32, _i32]),
    "o2m_act_bwd_reduce": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp]),
    "o2m_prepare_weights": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, C.c_float,
                                   _i32, _vp]),
    "o2m_prepare_weights_batched": (_i32, [_vp, _i32, _i32, _i32, _vp]),
    "o2m_modulate_weights": (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp]),
    "o2m_fold_scale_dot": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32,
                                  _vp, _vp]),
    "o2m_instnorm_ws_floats": (C.c_size_t, [_i32, _i32, _i32]),
    "o2m_instnorm_stats": (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _f32, _i32, _vp]),
    "o2m_instnorm_apply": (_i32, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp]),
    "o2m_instnorm_bwd": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp]),
    "o2m_instnorm_act_resample2d": (_i32, [_vp] * 7 + [_i32] * 11 + [_vp]),
    "o2m_instnorm_resample_bwd": (_i32, [_vp] * 10 + [_i32] * 9 + [_vp]),
    "o2m_resample2d": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32,
                              _i32, _i32, _i32, _i32, _vp]),
    "o2m_ada_grid_sample": (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "o2m_ada_grid_sample_bwd": (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "o2m_reflect_fold": (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "o2m_ada_colour": (_i32, [_vp, _vp, _vp, _i32, C.c_int64, _i32, _i32, _i32, _vp]),
    "o2m_gather_images": (_i32, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "o2m_pack_nchw": (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "o2m_unpack_nhwc": (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "o2m_lsgan_fwd": (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _f32, _f32, _i32, _vp]),
    "o2m_lsgan_bwd": (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _f32, _f32, _i32, _vp]),
    "o2m_wgrad_finalize_blocks": (_i32, [_i32, _i32, _i32]),
    "o2m_wgrad_finalize_batched": (_i32, [_vp, _i32, _i32, _i32, _vp]),
    "o2m_reduce_blocks": (_i32, [_i64]),
    "o2m_reduce_fwd": (_i32, [_vp, _vp, _vp, _vp, _i32, _i64, _i32, _i32, _vp]),
    "o2m_reduce_bwd": (_i32, [_vp, _vp, _vp, _vp, _vp, _i32, _i64, _i32, _i32, _vp]),
    "o2m_pair_grad": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i64, _i32, _vp]),
    "o2m_adam_step": (_i32, [_vp, _vp, _vp, _vp, _vp, _i64, _f32, _f32, _f32, _f32, _f32, _vp]),
}

_lib = None
MEASUREMENT_ONLY = {"o2m_launch_timing", "o2m_launch_timing_read", "o2m_debug_fill_blocks"}  # C-ABI entries without a torch.ops.o2m twin


class LaunchStat(C.Structure):
    """o2m_launch_stat (include/o2m_hip.h)."""

    _fields_ = [("kernel", C.c_char * 64), ("launches", C.c_int32), ("ms", C.c_float), ("flops", C.c_double)]


def launch_timing(enable: bool) -> bool:
    """Switch the library's per-kernel launch timing (HIP-event pairs inside the launch sites, on the
    launch stream) on or off; returns the previous state.  bench.py's roofline object uses it."""
    return bool(lib().o2m_launch_timing(int(bool(enable))))


def debug_fill_blocks(n: int) -> int:
    """Test hook (include/o2m_hip.h): the tile count o2m_conv2d_fwd's kernel selection takes for "one workgroup per
    CU"; n <= 0 restores 256.  Returns the previous value."""
    return int(lib().o2m_debug_fill_blocks(int(n)))


def launch_timing_read(capacity: int = 64):
    """{kernel name: (launches, seconds, algorithmic flops)} of the launches recorded since the last
    read; waits for them to finish."""
    table = (LaunchStat * capacity)()
    n = lib().o2m_launch_timing_read(table, capacity)
    if n < 0:
        raise RuntimeError(f"o2m_launch_timing_read failed with code {n}")
    return {table[i].kernel.decode(): (table[i].launches, table[i].ms * 1e-3, table[i].flops) for i in range(n)}


def lib():
    """Load the C-ABI library (once) and check its exports.  Raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: the HIP extension has not been built "
                "(run `python -m one_to_many_gan_amd.build` or __graft_entry__.build()). "
                "There is no CPU or eager fallback."
            )
        handle = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)  # the torch shim binds to these symbols
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError if the symbol is not exported
            fn.restype, fn.argtypes = res, args
        if handle.o2m_abi_version() != ABI_VERSION:
            raise RuntimeError("libo2m_hip.so ABI version mismatch: rebuild the extension")
        _lib = handle
    return _lib


_ops = None


def ops():
    """``torch.ops.o2m`` (loads libo2m_torch.so once).  Raises if it has not been built."""
    global _ops
    if _ops is None:
        lib()  # the C ABI first: the shim's undefined symbols resolve against it
        if not os.path.exists(TORCH_LIB_PATH):
            raise RuntimeError(f"{TORCH_LIB_PATH} is missing: run `python -m one_to_many_gan_amd.build`. "
                               "There is no CPU or eager fallback.")
        torch.ops.load_library(TORCH_LIB_PATH)
        if torch.ops.o2m.abi_version() != ABI_VERSION:
            raise RuntimeError("libo2m_torch.so was built against another libo2m_hip.so: rebuild")
        _ops = torch.ops.o2m
    return _ops


def dtype_code(dt: torch.dtype) -> int:
    if dt == torch.bfloat16:
        return BF16
    if dt == torch.float32:
        return F32
    raise TypeError(f"unsupported activation dtype {dt}")


def ptr(t):
    """Raw device pointer for direct ctypes calls of the C ABI (tests/test_kernels_gpu.py)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("o2m HIP op called with a non-GPU tensor: the hot path has no CPU fallback")
    if not t.is_contiguous():
        raise RuntimeError("o2m HIP op needs contiguous tensors")
    return C.c_void_p(t.data_ptr())


def _stream(t: torch.Tensor):
    return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def check(err: int, what: str):
    if err != 0:
        raise RuntimeError(f"{what} failed with code {err}")


# ------------------------------------------------------------------------------- wrappers


def conv2d_fwd(x, w, y, *, in_scale=None, out_scale=None, bias=None, residual=None, pad, pad_mode, act,
               per_sample_w=False, stride=1, stats=None, deq=None, aux=None, aux_scaled=None, fold_pad=0):
    """``stats``: fp32 workspace for the InstanceNorm partial sums of y (see o2m_conv_desc.stats) -- or, with
    ``aux`` (a tensor of y's shape), for the style-dot partials sum_p acc * aux (O2M_STATS_DOT); ``aux_scaled``
    then optionally receives aux * out_scale.
    ``deq``: fp8 operands (x float8_e4m3fn / float8_e5m2, w float8_e4m3fn): device tensor of the two
    dequantisation factors.
    ``fold_pad`` = f > 0: the conv is the data gradient behind ReflectionPad2d(f); y / residual are the CROPPED map
    and every output pixel of the padded domain is added at its mirror image (o2m_conv_desc.fold_pad)."""
    return ops().conv2d_fwd(x, w, y, in_scale, out_scale, bias, residual, pad, pad_mode, act, per_sample_w, stride,
                            stats, deq, aux, aux_scaled, fold_pad)


def conv2d_reflect_border(x, w, y):
    """Adds the border ring of the data gradient of a 3 x 3 conv behind ReflectionPad2d(1) to ``y``, which already
    holds the zero-padded (pad 1) data-gradient conv of ``x`` with ``w`` (o2m_conv2d_reflect_border)."""
    ops().conv2d_reflect_border(x, w, y)


def lsgan_fwd(scores, out, n_first, t0, t1):
    ops().lsgan_fwd(scores, out, n_first, float(t0), float(t1))


def lsgan_bwd(scores, coef, g_scores, n_first, t0, t1):
    ops().lsgan_bwd(scores, coef, g_scores, n_first, float(t0), float(t1))


def conv2d_dots_finalize(partial, dots, nchunks):
    ops().conv2d_dots_finalize(partial, dots, nchunks)


AMAX_PARTIALS = 1024


class _StreamScratch:
    """fp32 scratch buffers keyed by (device, stream).  The weight gradient runs on its own HIP stream beside the
    main one, so a workspace shared by "whatever stream is current" would have two writers, and a regrow would
    hand the old block back to the allocator while a kernel on the OTHER stream still used it (the round-2
    multi-segment fault).  Each stream owns its buffer: every use and every regrow is ordered on that stream,
    and the caching allocator reuses a freed block only behind the work of the stream it was allocated on."""

    def __init__(self, floor: int):
        self.floor, self.buf = floor, {}

    def get(self, device, need: int) -> torch.Tensor:
        key = (device, torch.cuda.current_stream(device).cuda_stream)
        ws = self.buf.get(key)
        if ws is None or ws.numel() < need:
            ws = self.buf[key] = torch.empty(max(int(need), self.floor), dtype=torch.float32, device=device)
        return ws


_AMAX_WS = _StreamScratch(AMAX_PARTIALS)


def quantize_fp8(x, y, deq):
    """Per-tensor fp8 quantisation of ``x`` (bf16 / fp32) into ``y`` (float8_e4m3fn or float8_e5m2, same shape):
    amax pass + scale-and-convert pass; ``deq`` (2 floats) receives {1 / scale, amax}."""
    ws = _AMAX_WS.get(x.device, AMAX_PARTIALS)
    ops().amax(x, ws)
    ops().quantize_fp8(x, ws, y, deq)


class Fp8Site:
    """Delayed-scaling state of ONE quantisation call site (a layer's activations or gradients, per stream): two amax
    workspaces used alternately -- the scale of a call comes from the tensor the site quantised last time."""

    __slots__ = ("bufs", "cur")

    def __init__(self):
        self.bufs, self.cur = None, -1  # cur: index of the workspace holding the LAST tensor's partial maxima


def quantize_fp8_site(x, y, deq, site: Fp8Site):
    """``quantize_fp8`` with delayed scaling: the first call of a site runs the two-pass form (and keeps its partial
    maxima); every later call is ONE pass -- scale from the previous call's maxima, this tensor's maxima recorded for
    the next (o2m_quantize_fp8_delayed).  Values that outgrow the previous amax saturate."""
    if site.bufs is None or site.bufs[0].device != x.device:
        site.bufs = [torch.zeros(AMAX_PARTIALS, dtype=torch.float32, device=x.device) for _ in range(2)]
        site.cur = -1
    if site.cur < 0:
        ops().amax(x, site.bufs[0])
        ops().quantize_fp8(x, site.bufs[0], y, deq)
        site.cur = 0
        return
    prev, nxt = site.bufs[site.cur], site.bufs[1 - site.cur]
    ops().quantize_fp8_delayed(x, prev, y, deq, nxt)
    site.cur = 1 - site.cur


def conv2d_stats_chunks(x, w, y, *, pad, stride=1):
    """InstanceNorm partial rows per sample the conv's epilogue writes (0: it cannot emit them): the stats workspace is
    B * chunks * Co * 2 floats and o2m_instnorm_finalize takes ``chunks`` (o2m_conv2d_stats_chunks)."""
    return ops().conv2d_stats_chunks(x, w, y, pad, stride)


def conv2d_stats_rows(x, w, y, *, pad, stride=1):
    """Rows per InstanceNorm partial the epilogue of this conv would emit (0: not available)."""
    return ops().conv2d_stats_rows(x, w, y, pad, stride)


_SLABS = _StreamScratch(16 << 20)  # (device, stream) -> fp32 workspace of the slice partials (grown on demand)
WGRAD_ATOMICS = os.environ.get("O2M_WGRAD_ATOMICS", "0") == "1"  # A/B: float atomics instead of slab + reduce


def _slab_workspace(x, gy, dw, pad, pad_mode, splits, stride, hint):
    if WGRAD_ATOMICS and not DETERMINISTIC:
        return None
    need = ops().conv2d_wgrad_slab_floats(x, gy, dw, pad, pad_mode, splits, stride, hint)
    return _SLABS.get(x.device, need)


def conv2d_wgrad(x, gy, dw, *, in_scale=None, gy_scale=None, pad, pad_mode, splits=0, stride=1, p8=False):
    """The pixel slices go through a slab workspace + ordered second-stage sum (deterministic; see
    o2m_wgrad_desc.slabs) unless O2M_WGRAD_ATOMICS=1.  ``p8``: prefer the phase-pipelined kernel where it applies
    (o2m_wgrad_desc.kernel_hint)."""
    hint = 1 if p8 else 0
    slabs = _slab_workspace(x, gy, dw, pad, pad_mode, splits, stride, hint)
    return ops().conv2d_wgrad(x, gy, dw, in_scale, gy_scale, pad, pad_mode, splits, stride, slabs, hint)


DETERMINISTIC = False  # ops.set_deterministic(): ordered two-stage sums instead of fp32 atomics
_PARTIALS = _StreamScratch(1 << 20)  # (device, stream) -> fp32 workspace of the per-chunk rows


def _chan_partials(g, P, nv):
    """Workspace for the ordered channel sums (None outside deterministic mode: fp32 atomics)."""
    if not DETERMINISTIC:
        return None
    need = ops().chan_partials_floats(g.shape[0], P, g.shape[-1], nv)
    return _PARTIALS.get(g.device, need)


def act_bwd_reduce(g, y, residual, out_mul, gu, sums, act):
    partials = _chan_partials(g, g.shape[1] * g.shape[2], 2) if sums is not None else None
    ops().act_bwd_reduce(g, y, residual, out_mul, gu, sums, act, partials)


def style_fwd(w, ws, bs, qt, s, d, ci, cs, eps):
    ops().style_fwd(w, ws, bs, qt, s, d, ci, cs, eps)


def style_bwd(sums, bias, dots, s, d, q, w, ws, e, gs, gw, gws, gbs, gq, ci, cs, accumulate=False):
    ops().style_bwd(sums, bias, dots, s, d, q, w, ws, e, gs, gw, gws, gbs, gq, ci, cs, bool(accumulate))


def wgrad_finalize(acc, gq, w32, grad, co, ci, c):
    ops().wgrad_finalize(acc, gq, w32, grad, co, ci, c)


def prepare_weights(w, full, w_f, w_d, q, qt, c):
    """W*c in the kernel layouts (see o2m_prepare_weights); ``w`` is the raw (Co,Ci,KH,KW) parameter."""
    ops().prepare_weights(w, full, w_f, w_d, q, qt, c)


def wgrad_finalize_blocks(cop, kk, cip):
    return int(ops().wgrad_finalize_blocks(cop, kk, cip))


def wgrad_finalize_batched(jobs, n_jobs, total_blocks, any_gq, touched):
    """Every pending layer's accumulators -> .grad in one launch; ``jobs``: uint8 device tensor of WfinJob records,
    ``touched``: the tensors the jobs name (accumulators, dL/dQ tables, gradients) for the schema's mutation list."""
    ops().wgrad_finalize_batched(jobs, n_jobs, total_blocks, bool(any_gq), touched)


def prepare_weights_batched(jobs, n_jobs, total_blocks, dtype, outs):
    """Every filter of a network in one launch; ``jobs``: uint8 device tensor of PrepJob records."""
    ops().prepare_weights_batched(jobs, n_jobs, total_blocks, dtype_code(dtype), outs)


def modulate_weights(w32, s, out):
    """out[b,o,kh,kw,i] = w32[o,kh,kw,i] * s[b,i] in the dtype of ``out``."""
    ops().modulate_weights(w32, s, out)


def fold_scale_dot(gpad, x, scale, gx, dots, pad, xs=None, *, gres=None, act=ACT_NONE, act_mul=None, act_sums=None):
    """``gres``: residual-path gradient added to gx.  ``act_sums`` (with ``act`` / ``act_mul``): fused backward of
    the activation whose output is x (see o2m_fold_scale_dot)."""
    nv = 3 if act_sums is not None else 1
    partials = (_chan_partials(gx, gx.shape[1] * gx.shape[2], nv)
                if (dots is not None or act_sums is not None) else None)
    ops().fold_scale_dot(gpad, x, scale, gx, dots, pad, xs, gres, act, act_mul, act_sums, partials)


def instnorm_ws_floats(B, P, Cn):
    return ops().instnorm_ws_floats(B, P, Cn)


def instnorm_stats(x, partial, mean_rstd, eps):
    ops().instnorm_stats(x, partial, mean_rstd, eps)


def instnorm_finalize(partial, mean_rstd, P, nchunks, eps):
    """Second stage of the statistics alone: conv-epilogue partials [B][nchunks][C][2] -> mean / rstd."""
    ops().instnorm_finalize(partial, mean_rstd, P, nchunks, eps)


def instnorm_apply(x, mean_rstd, residual, y, act):
    ops().instnorm_apply(x, mean_rstd, residual, y, act)


def instnorm_bwd(g, x, mean_rstd, partial, gsums, gx, act):
    ops().instnorm_bwd(g, x, mean_rstd, partial, gsums, gx, act)


def instnorm_act_resample2d(x, mean_rstd, y, sy, wy, sx, wx, t, span_y, span_x, act):
    """y = D(act(InstanceNorm(x))) for the DownSample operators (t == 4, spans 2 or 3)."""
    ops().instnorm_act_resample2d(x, mean_rstd, y, sy, wy, sx, wx, t, span_y, span_x, act)


def instnorm_resample_bwd(g_coarse, x, mean_rstd, partial, gsums, gx, sy, wy, sx, wx, t, act):
    ops().instnorm_resample_bwd(g_coarse, x, mean_rstd, partial, gsums, gx, sy, wy, sx, wx, t, act)


def resample2d(x, y, sy, wy, sx, wx, ty, tx=None, span_y=0, span_x=0):
    """``ty`` / ``tx``: taps per axis (wy is [Ho][ty], wx is [Wo][tx]); spans: see resample.Taps."""
    ops().resample2d(x, y, sy, wy, sx, wx, ty, ty if tx is None else tx, span_y, span_x)


def ada_grid_sample(x, theta, y):
    ops().ada_grid_sample(x, theta, y)


def ada_grid_sample_bwd(gy, theta, gx):
    ops().ada_grid_sample_bwd(gy, theta, gx)


def reflect_fold(gpad, gx, pad_top, pad_left):
    ops().reflect_fold(gpad, gx, pad_top, pad_left)


def ada_colour(x, m, y, c):
    ops().ada_colour(x, m, y, c)


def gather_images(pool, index, flip, out):
    """Batch from the HBM-resident uint8 pool (see o2m_gather_images)."""
    ops().gather_images(pool, index, flip, out)


def pack_nchw(src, dst):
    ops().pack_nchw(src, dst)


def unpack_nhwc(src, dst):
    ops().unpack_nhwc(src, dst)


def reduce_blocks(n):
    return ops().reduce_blocks(n)


def reduce_fwd(a, b, w, partials, mode):
    ops().reduce_fwd(a, b, w, partials, mode)


def reduce_bwd(a, b, w, coef, ga, mode):
    ops().reduce_bwd(a, b, w, coef, ga, mode)


def pair_grad(a, b, w, coef, gin_a, gin_b, ga, gb):
    """ga = gin_a + coef * w[b] * (a - b), gb = gin_b - coef * w[b] * (a - b) in one pass (o2m_pair_grad)."""
    ops().pair_grad(a, b, w, coef, gin_a, gin_b, ga, gb)


def adam_step(p, g, m, v, step, lr, beta1, beta2, eps, grad_scale):
    ops().adam_step(p, g, m, v, step, lr, beta1, beta2, eps, grad_scale)
