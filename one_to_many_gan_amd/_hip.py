"""ctypes binding of libo2m_hip.so (the C ABI declared in include/o2m_hip.h).

There is NO fallback: if the library is missing or a call is made without a GPU tensor the
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

BF16, F32 = 0, 1
ACT_NONE, ACT_RELU, ACT_LRELU, ACT_TANH = 0, 1, 2, 3
PAD_ZERO, PAD_REFLECT = 0, 1
RED_L1, RED_SQ, RED_MOM = 0, 1, 2
ABI_VERSION = 12

_vp, _i32, _i64, _f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float


class ConvDesc(C.Structure):
    _fields_ = [("x", _vp), ("w", _vp), ("y", _vp), ("in_scale", _vp), ("out_scale", _vp),
                ("bias", _vp), ("residual", _vp),
                ("B", _i32), ("H", _i32), ("W", _i32), ("Ci", _i32), ("Co", _i32), ("KH", _i32),
                ("KW", _i32), ("pad", _i32), ("pad_mode", _i32), ("act", _i32), ("dtype", _i32),
                ("w_batch_stride", _i32), ("stride", _i32), ("reserved", _i32 * 3)]


class WgradDesc(C.Structure):
    _fields_ = [("x", _vp), ("gy", _vp), ("dw", _vp), ("in_scale", _vp), ("gy_scale", _vp),
                ("B", _i32), ("H", _i32), ("W", _i32), ("Ci", _i32), ("Co", _i32), ("KH", _i32),
                ("KW", _i32), ("pad", _i32), ("pad_mode", _i32), ("dtype", _i32), ("splits", _i32),
                ("nseg", _i32), ("stride", _i32), ("reserved", _i32 * 3), ("x_seg", _vp * 8), ("gy_seg", _vp * 8)]


# name -> (restype, argtypes); mirrors include/o2m_hip.h one for one
SIGNATURES = {
    "o2m_abi_version": (_i32, []),
    "o2m_conv2d_fwd": (_i32, [C.POINTER(ConvDesc), _vp]),
    "o2m_conv2d_wgrad": (_i32, [C.POINTER(WgradDesc), _vp]),
    "o2m_wgrad_finalize": (_i32, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _f32, _vp]),
    "o2m_style_fwd": (_i32, [_vp] * 6 + [_i32] * 5 + [_f32, _f32, _vp]),
    "o2m_style_bwd": (_i32, [_vp] * 14 + [_i32] * 5 + [_f32, _i32, _vp]),
    "o2m_act_bwd_reduce": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp]),
    "o2m_prepare_weights": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, C.c_float,
                                   _i32, _vp]),
    "o2m_modulate_weights": (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp]),
    "o2m_fold_scale_dot": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "o2m_instnorm_ws_floats": (C.c_size_t, [_i32, _i32, _i32]),
    "o2m_instnorm_stats": (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _f32, _i32, _vp]),
    "o2m_instnorm_apply": (_i32, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp]),
    "o2m_instnorm_bwd": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp]),
    "o2m_resample2d": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32,
                              _i32, _i32, _i32, _i32, _vp]),
    "o2m_ada_grid_sample": (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "o2m_ada_grid_sample_bwd": (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "o2m_reflect_fold": (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "o2m_ada_colour": (_i32, [_vp, _vp, _vp, _i32, C.c_int64, _i32, _i32, _i32, _vp]),
    "o2m_gather_images": (_i32, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "o2m_pack_nchw": (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "o2m_unpack_nhwc": (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "o2m_reduce_blocks": (_i32, [_i64]),
    "o2m_reduce_fwd": (_i32, [_vp, _vp, _vp, _vp, _i32, _i64, _i32, _i32, _vp]),
    "o2m_reduce_bwd": (_i32, [_vp, _vp, _vp, _vp, _vp, _i32, _i64, _i32, _i32, _vp]),
    "o2m_adam_step": (_i32, [_vp, _vp, _vp, _vp, _vp, _i64, _f32, _f32, _f32, _f32, _f32, _vp]),
}

_lib = None

# bench.py sets this to a list to time every MFMA conv launch with a HIP-event pair on the
# launch stream: entries are (kernel_name, algorithmic_flops, start_event, end_event).
PROFILE = None


def _igemm_name(dt, co, scaled, m=1 << 30, k=1 << 30):
    """Mirrors launch_dtype() in csrc/conv_igemm.hip, so the labels map one-to-one onto the
    template instantiations rocprofv3 reports."""
    t = "bf16" if dt == torch.bfloat16 else "f32x3"

    def tiles(bm, bn):
        return -(-m // bm) * -(-co // bn)

    if co > 128:
        tile = "256x256" if tiles(256, 256) >= 256 else "128x128"
    elif co > 64:
        if k <= 1152 and tiles(256, 64) >= 512:
            tile = "256x64"
        else:
            tile = "256x128" if tiles(256, 128) >= 256 else "128x128"
    elif co > 32:
        tile = "256x64" if tiles(256, 64) >= 256 else "128x64"
    else:
        tile = "256x32"
    return f"conv_igemm<{t},{tile},in_scale={int(scaled)}>"


def _wgrad_name(dt, co, m=0, k=0):
    """Mirrors launch_dtype() in csrc/conv_wgrad.hip."""
    t = "bf16" if dt == torch.bfloat16 else "f32x3"
    if co > 256 or (co > 128 and m >= 100000):
        tile = "co256xk128"
    elif 64 < co <= 128 and k % 256 == 0 and dt == torch.bfloat16:
        tile = "co128xk256"
    else:
        tile = ("co128" if co > 64 else ("co64" if co > 32 else "co32")) + "xk128"
    return f"conv_wgrad<{t},{tile}>"


def _timed(name, flops, tensor, launch):
    if PROFILE is None:
        return launch()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st = torch.cuda.current_stream(tensor.device)
    e0.record(st)
    launch()
    e1.record(st)
    PROFILE.append((name, flops, e0, e1))


def lib():
    """Load the shared library (once).  Raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: the HIP extension has not been built "
                "(run `python -m one_to_many_gan_amd.build` or __graft_entry__.build()). "
                "There is no CPU or eager fallback."
            )
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError if the symbol is not exported
            fn.restype, fn.argtypes = res, args
        if handle.o2m_abi_version() != ABI_VERSION:
            raise RuntimeError("libo2m_hip.so ABI version mismatch: rebuild the extension")
        _lib = handle
    return _lib


def dtype_code(dt: torch.dtype) -> int:
    if dt == torch.bfloat16:
        return BF16
    if dt == torch.float32:
        return F32
    raise TypeError(f"unsupported activation dtype {dt}")


def _stream(t: torch.Tensor):
    return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def ptr(t):
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("o2m HIP op called with a non-GPU tensor: the hot path has no CPU fallback")
    if not t.is_contiguous():
        raise RuntimeError("o2m HIP op needs contiguous tensors")
    return C.c_void_p(t.data_ptr())


def check(err: int, what: str):
    if err != 0:
        raise RuntimeError(f"{what} failed with code {err}")


# ------------------------------------------------------------------------------- wrappers


def conv2d_fwd(x, w, y, *, in_scale=None, out_scale=None, bias=None, residual=None, pad, pad_mode, act,
               per_sample_w=False, stride=1):
    B, H, W, Ci = x.shape
    Co, KH, KW, _ = w.shape[-4:]
    wstride = Co * KH * KW * Ci if per_sample_w else 0
    d = ConvDesc(ptr(x), ptr(w), ptr(y), ptr(in_scale), ptr(out_scale), ptr(bias), ptr(residual),
                 B, H, W, Ci, Co, KH, KW, pad, pad_mode, act, dtype_code(x.dtype), wstride, stride)
    flops = 2.0 * y.shape[0] * y.shape[1] * y.shape[2] * Co * KH * KW * Ci
    _timed(_igemm_name(x.dtype, Co, in_scale is not None, y.shape[0] * y.shape[1] * y.shape[2], KH * KW * Ci),
           flops, x,
           lambda: check(lib().o2m_conv2d_fwd(C.byref(d), _stream(x)), "o2m_conv2d_fwd"))


def conv2d_wgrad(x, gy, dw, *, in_scale=None, gy_scale=None, pad, pad_mode, splits=0, more=(), stride=1):
    """``more``: extra (x, gy) pairs of the same shape reduced by the same launch (<= 7)."""
    B, H, W, Ci = x.shape
    Co, KH, KW, _ = dw.shape
    d = WgradDesc(ptr(x), ptr(gy), ptr(dw), ptr(in_scale), ptr(gy_scale), B, H, W, Ci, Co, KH, KW,
                  pad, pad_mode, dtype_code(x.dtype), splits, 1 + len(more), stride)
    for i, (xi, gi) in enumerate(more, start=1):
        if xi.shape != x.shape or gi.shape != gy.shape or xi.dtype != x.dtype:
            raise RuntimeError("wgrad segments must share one shape")
        d.x_seg[i], d.gy_seg[i] = xi.data_ptr(), gi.data_ptr()
        ptr(xi), ptr(gi)  # contiguity / device checks
    flops = 2.0 * (1 + len(more)) * gy.shape[0] * gy.shape[1] * gy.shape[2] * Co * KH * KW * Ci
    _timed(_wgrad_name(x.dtype, Co, (1 + len(more)) * gy.shape[0] * gy.shape[1] * gy.shape[2], KH * KW * Ci), flops, x,
           lambda: check(lib().o2m_conv2d_wgrad(C.byref(d), _stream(x)), "o2m_conv2d_wgrad"))


def act_bwd_reduce(g, y, residual, out_mul, gu, sums, act):
    B, P, Cn = g.shape[0], g.shape[1] * g.shape[2], g.shape[3]
    check(lib().o2m_act_bwd_reduce(ptr(g), ptr(y), ptr(residual), ptr(out_mul), ptr(gu), ptr(sums), B, P,
                                   Cn, act, dtype_code(g.dtype), _stream(g)), "o2m_act_bwd_reduce")


def style_fwd(w, ws, bs, qt, s, d, ci, cs, eps):
    B, WD = w.shape
    cip = s.shape[1]
    cop = d.shape[1] if d is not None else 0
    check(lib().o2m_style_fwd(ptr(w), ptr(ws), ptr(bs), ptr(qt), ptr(s), ptr(d), B, WD, ci, cip, cop, cs,
                              eps, _stream(s)), "o2m_style_fwd")


def style_bwd(sums, bias, dots, s, d, q, w, ws, e, gs, gw, gws, gbs, gq, ci, cs, accumulate=False):
    B, WD = w.shape
    cip = s.shape[1]
    cop = d.shape[1] if d is not None else 8
    check(lib().o2m_style_bwd(ptr(sums), ptr(bias), ptr(dots), ptr(s), ptr(d), ptr(q), ptr(w), ptr(ws),
                              ptr(e), ptr(gs), ptr(gw), ptr(gws), ptr(gbs), ptr(gq), B, WD, ci, cip, cop, cs,
                              int(bool(accumulate)), _stream(s)), "o2m_style_bwd")


def wgrad_finalize(acc, gq, w32, grad, co, ci, c):
    cop, kh, kw, cip = acc.shape
    check(lib().o2m_wgrad_finalize(ptr(acc), ptr(gq), ptr(w32), ptr(grad), co, ci, kh * kw, cop, cip, c,
                                   _stream(acc)), "o2m_wgrad_finalize")


def prepare_weights(w, full, w_f, w_d, q, qt, c):
    """W*c in the kernel layouts (see o2m_prepare_weights); ``w`` is the raw (Co,Ci,KH,KW) parameter."""
    Co, Ci, KH, KW = w.shape
    Cop, _, _, Cip = full.shape
    check(lib().o2m_prepare_weights(ptr(w), ptr(full), ptr(w_f), ptr(w_d), ptr(q), ptr(qt), Co, Ci, KH * KW,
                                    Cop, Cip, c, dtype_code(w_f.dtype), _stream(w)), "o2m_prepare_weights")


def modulate_weights(w32, s, out):
    """out[b,o,kh,kw,i] = w32[o,kh,kw,i] * s[b,i] in the dtype of ``out``."""
    Co, KH, KW, Ci = w32.shape
    check(lib().o2m_modulate_weights(ptr(w32), ptr(s), ptr(out), s.shape[0], Co, KH * KW, Ci,
                                     dtype_code(out.dtype), _stream(out)), "o2m_modulate_weights")


def fold_scale_dot(gpad, x, scale, gx, dots, pad, xs=None):
    B, H, W, Cn = gx.shape
    check(lib().o2m_fold_scale_dot(ptr(gpad), ptr(x), ptr(scale), ptr(gx), ptr(dots), ptr(xs), B, H, W, Cn,
                                   pad, dtype_code(gx.dtype), _stream(gx)), "o2m_fold_scale_dot")


def instnorm_ws_floats(B, P, Cn):
    return int(lib().o2m_instnorm_ws_floats(B, P, Cn))


def instnorm_stats(x, partial, mean_rstd, eps):
    B, P, Cn = x.shape[0], x.shape[1] * x.shape[2], x.shape[3]
    check(lib().o2m_instnorm_stats(ptr(x), ptr(partial), ptr(mean_rstd), B, P, Cn, eps,
                                   dtype_code(x.dtype), _stream(x)), "o2m_instnorm_stats")


def instnorm_apply(x, mean_rstd, residual, y, act):
    B, P, Cn = x.shape[0], x.shape[1] * x.shape[2], x.shape[3]
    check(lib().o2m_instnorm_apply(ptr(x), ptr(mean_rstd), ptr(residual), ptr(y), B, P, Cn, act,
                                   dtype_code(x.dtype), _stream(x)), "o2m_instnorm_apply")


def instnorm_bwd(g, x, mean_rstd, partial, gsums, gx, act):
    B, P, Cn = x.shape[0], x.shape[1] * x.shape[2], x.shape[3]
    check(lib().o2m_instnorm_bwd(ptr(g), ptr(x), ptr(mean_rstd), ptr(partial), ptr(gsums), ptr(gx), B, P,
                                 Cn, act, dtype_code(x.dtype), _stream(x)), "o2m_instnorm_bwd")


def resample2d(x, y, sy, wy, sx, wx, ty, tx=None, span_y=0, span_x=0):
    """``ty`` / ``tx``: taps per axis (wy is [Ho][ty], wx is [Wo][tx]); spans: see resample.Taps."""
    B, H, W, Cn = x.shape
    _, Ho, Wo, _ = y.shape
    tx = ty if tx is None else tx
    check(lib().o2m_resample2d(ptr(x), ptr(y), ptr(sy), ptr(wy), ptr(sx), ptr(wx), B, H, W, Ho, Wo, Cn,
                               ty, tx, span_y, span_x, dtype_code(x.dtype), _stream(x)), "o2m_resample2d")


def ada_grid_sample(x, theta, y):
    B, Hs, Ws, Cp = x.shape
    _, Ho, Wo, _ = y.shape
    check(lib().o2m_ada_grid_sample(ptr(x), ptr(theta), ptr(y), B, Hs, Ws, Ho, Wo, Cp, dtype_code(x.dtype),
                                    _stream(x)), "o2m_ada_grid_sample")


def ada_grid_sample_bwd(gy, theta, gx):
    B, Ho, Wo, Cp = gy.shape
    _, Hs, Ws, _ = gx.shape
    if gx.dtype != gy.dtype:
        raise RuntimeError("ada_grid_sample_bwd: gx and gy share one dtype")
    check(lib().o2m_ada_grid_sample_bwd(ptr(gy), ptr(theta), ptr(gx), B, Hs, Ws, Ho, Wo, Cp, dtype_code(gy.dtype),
                                        _stream(gy)), "o2m_ada_grid_sample_bwd")


def reflect_fold(gpad, gx, pad_top, pad_left):
    B, Hp, Wp, Cp = gpad.shape
    _, Hh, Ww, _ = gx.shape
    check(lib().o2m_reflect_fold(ptr(gpad), ptr(gx), B, Hh, Ww, Hp, Wp, pad_top, pad_left, Cp,
                                 dtype_code(gpad.dtype), dtype_code(gx.dtype), _stream(gx)), "o2m_reflect_fold")


def ada_colour(x, m, y, c):
    B, Hh, Ww, Cp = x.shape
    check(lib().o2m_ada_colour(ptr(x), ptr(m), ptr(y), B, Hh * Ww, c, Cp, dtype_code(x.dtype), _stream(x)),
          "o2m_ada_colour")


def gather_images(pool, index, flip, out):
    """Batch from the HBM-resident uint8 pool (see o2m_gather_images)."""
    N, H, W, Cn = pool.shape
    B, _, _, Cp = out.shape
    if pool.dtype != torch.uint8 or index.dtype != torch.int32 or flip.dtype != torch.uint8:
        raise RuntimeError("gather_images: pool uint8, index int32, flip uint8")
    if index.shape[0] != B or flip.shape[0] != B or out.shape[1:3] != pool.shape[1:3]:
        raise RuntimeError("gather_images: shape mismatch")
    check(lib().o2m_gather_images(ptr(pool), ptr(index), ptr(flip), ptr(out), N, B, H, W, Cn, Cp,
                                  dtype_code(out.dtype), _stream(out)), "o2m_gather_images")


def pack_nchw(src, dst):
    B, Cn, H, W = src.shape
    check(lib().o2m_pack_nchw(ptr(src), ptr(dst), B, Cn, H, W, dst.shape[3], dtype_code(dst.dtype),
                              _stream(dst)), "o2m_pack_nchw")


def unpack_nhwc(src, dst):
    B, Cn, H, W = dst.shape
    check(lib().o2m_unpack_nhwc(ptr(src), ptr(dst), B, Cn, H, W, src.shape[3], dtype_code(src.dtype),
                                _stream(src)), "o2m_unpack_nhwc")


def reduce_blocks(n):
    return int(lib().o2m_reduce_blocks(n))


def reduce_fwd(a, b, w, partials, mode):
    B = a.shape[0]
    nps = a.numel() // B
    check(lib().o2m_reduce_fwd(ptr(a), ptr(b), ptr(w), ptr(partials), B, nps, mode, dtype_code(a.dtype),
                               _stream(a)), "o2m_reduce_fwd")


def reduce_bwd(a, b, w, coef, ga, mode):
    B = a.shape[0]
    nps = a.numel() // B
    check(lib().o2m_reduce_bwd(ptr(a), ptr(b), ptr(w), ptr(coef), ptr(ga), B, nps, mode,
                               dtype_code(a.dtype), _stream(a)), "o2m_reduce_bwd")


def adam_step(p, g, m, v, step, lr, beta1, beta2, eps, grad_scale):
    check(lib().o2m_adam_step(ptr(p), ptr(g), ptr(m), ptr(v), ptr(step), p.numel(), lr, beta1, beta2,
                              eps, grad_scale, _stream(p)), "o2m_adam_step")
