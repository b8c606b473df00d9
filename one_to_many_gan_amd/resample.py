"""Host-side construction of the 1-D banded operators behind Smooth / UpSample / DownSample.

All three reference layers (layers.py:191-247) are separable linear maps
``Y = A_h . X . A_w^T``:

* blur      : replicate-pad 1 then [1,2,1]/4 per axis             (layers.py:197-214)
* bilinear  : align_corners=False, src = (i+0.5)*in/out - 0.5      (layers.py:223-229,241-247)
* UpSample  = blur(2n) . bilinear(n -> 2n);  DownSample = bilinear(n -> n//2) . blur(n)
  (odd n gives the fractional taps of scale n / floor(n/2)).

The device kernel (o2m_resample2d) applies any such operator given per-output-row taps
``(start, weights[T])``; the backward pass is the same kernel with the transposed operator.
Host logic only (numpy); tested on CPU against F.interpolate / the oracle.
"""

from __future__ import annotations

import functools

import numpy as np
import torch


def blur_matrix(n: int) -> np.ndarray:
    a = np.zeros((n, n), dtype=np.float64)
    for i in range(n):
        a[i, max(i - 1, 0)] += 0.25
        a[i, i] += 0.5
        a[i, min(i + 1, n - 1)] += 0.25
    return a


def bilinear_matrix(n_in: int, n_out: int, scale: float | None = None) -> np.ndarray:
    """Rows follow ATen's upsample_bilinear2d (align_corners=False), evaluated in fp32."""
    sc = np.float32(n_in / n_out if scale is None else scale)
    a = np.zeros((n_out, n_in), dtype=np.float64)
    for i in range(n_out):
        src = sc * np.float32(i + 0.5) - np.float32(0.5)
        if src < 0:
            src = np.float32(0.0)
        i0 = min(int(np.floor(src)), n_in - 1)
        i1 = min(i0 + 1, n_in - 1)
        lam = np.float32(src - np.float32(i0))
        a[i, i0] += float(np.float32(1.0) - lam)
        a[i, i1] += float(lam)
    return a


def operator_matrix(kind: str, n: int) -> np.ndarray:
    if kind == "blur":
        return blur_matrix(n)
    if kind == "up":
        return blur_matrix(2 * n) @ bilinear_matrix(n, 2 * n, scale=0.5)
    if kind == "down":
        return bilinear_matrix(n, n // 2) @ blur_matrix(n)
    if kind == "down_nosmooth":  # DownSample(smooth=False), layers.py:235-247
        return bilinear_matrix(n, n // 2)
    raise ValueError(kind)


def banded(a: np.ndarray):
    """Dense operator -> (start int32 [rows], weights fp32 [rows][T], T)."""
    rows, cols = a.shape
    first = np.zeros(rows, dtype=np.int64)
    width = 1
    for r in range(rows):
        nz = np.nonzero(a[r])[0]
        if nz.size:
            first[r] = nz[0]
            width = max(width, int(nz[-1] - nz[0] + 1))
    width = min(width, cols)
    start = np.minimum(first, cols - width)
    w = np.zeros((rows, width), dtype=np.float32)
    for r in range(rows):
        w[r] = a[r, start[r]: start[r] + width]
    return start.astype(np.int32), w, width


@functools.lru_cache(maxsize=256)
def _taps_host(kind: str, n: int, transposed: bool):
    a = operator_matrix(kind, n)
    if transposed:
        a = a.T
    return banded(np.ascontiguousarray(a))


_dev_cache: dict = {}


class Taps(int):
    """Tap count of a banded operator pair.  ``span_y`` / ``span_x`` = largest step between
    consecutive tap starts of that axis when the starts are non-decreasing (lets the kernel share
    one input patch between neighbouring outputs), else 0."""

    def __new__(cls, t, span_y, span_x):
        obj = super().__new__(cls, t)
        obj.span_y, obj.span_x = span_y, span_x
        return obj


def _span(s):
    if s.shape[0] < 2:
        return 1
    d = np.diff(s)
    lo, hi = int(d.min()), int(d.max())
    return max(hi, 1) if lo >= 0 and hi <= 3 else 0


_id_cache: dict = {}


def identity_taps(n: int, device):
    """(start, weights) of the n x n identity as a 1-tap banded operator."""
    key = (n, str(device))
    hit = _id_cache.get(key)
    if hit is None:
        hit = (torch.arange(n, dtype=torch.int32, device=device), torch.ones((n, 1), dtype=torch.float32, device=device))
        _id_cache[key] = hit
    return hit


def fit_taps(s, w, t, T, n_src):
    """Widen a banded operator's rows from t to T taps, keeping every tap inside the source."""
    if t == T:
        return s, w
    s2 = np.minimum(s, max(n_src - T, 0)).astype(np.int32)
    w2 = np.zeros((w.shape[0], T), dtype=np.float32)
    for r in range(w.shape[0]):
        w2[r, s[r] - s2[r]: s[r] - s2[r] + t] = w[r]
    return s2, w2


def coo_width(rows, cols, n_rows):
    """Band width (max over rows of last - first + 1) of an operator given as COO triplets."""
    lo = np.full(n_rows, np.iinfo(np.int64).max, dtype=np.int64)
    hi = np.full(n_rows, -1, dtype=np.int64)
    np.minimum.at(lo, rows, cols)
    np.maximum.at(hi, rows, cols)
    used = hi >= 0
    return (int((hi[used] - lo[used]).max()) + 1 if used.any() else 1), np.where(used, lo, 0)


def banded_coo(rows, cols, vals, n_rows, n_cols, width=None):
    """COO triplets (duplicates add) -> (start int32 [rows], weights fp32 [rows][T], T), vectorised:
    the per-call operators of the augmentation pipe have ~1000 rows."""
    w0, first = coo_width(rows, cols, n_rows)
    T = min(max(w0, width or 1), n_cols)
    start = np.minimum(first, n_cols - T)
    w = np.zeros((n_rows, T), dtype=np.float64)
    np.add.at(w, (rows, cols - start[rows]), vals)
    return start.astype(np.int32), w.astype(np.float32), T


def taps_from_coo(op_h, op_w, device, min_width=1):
    """``taps`` tuple for two operators given as (rows, cols, vals, n_rows, n_cols)."""
    T = max(coo_width(op_h[0], op_h[1], op_h[3])[0], coo_width(op_w[0], op_w[1], op_w[3])[0], min_width)
    sy, wy, _ = banded_coo(*op_h, width=T)
    sx, wx, _ = banded_coo(*op_w, width=T)
    return (torch.from_numpy(sy).to(device), torch.from_numpy(wy).to(device), torch.from_numpy(sx).to(device),
            torch.from_numpy(wx).to(device), Taps(T, _span(sy), _span(sx)), wy.shape[0], wx.shape[0])


def taps_1d_coo(op, device):
    s, w, t = banded_coo(*op)
    return torch.from_numpy(s).to(device), torch.from_numpy(w).to(device), t, _span(s)


def taps_from_matrices(a_h: np.ndarray, a_w: np.ndarray, device):
    """The tuple ``taps`` returns, for two arbitrary banded operators [out][in] (vertical, horizontal)."""
    sy, wy, ty = banded(np.ascontiguousarray(a_h))
    sx, wx, tx = banded(np.ascontiguousarray(a_w))
    T = max(ty, tx)
    sy, wy = fit_taps(sy, wy, ty, T, a_h.shape[1])
    sx, wx = fit_taps(sx, wx, tx, T, a_w.shape[1])
    return (torch.from_numpy(sy).to(device), torch.from_numpy(np.ascontiguousarray(wy)).to(device),
            torch.from_numpy(sx).to(device), torch.from_numpy(np.ascontiguousarray(wx)).to(device),
            Taps(T, _span(sy), _span(sx)), wy.shape[0], wx.shape[0])


def taps_1d(a: np.ndarray, device):
    """(start, weights, T, span) of one banded operator [out][in] on the device."""
    s, w, t = banded(np.ascontiguousarray(a))
    return torch.from_numpy(s).to(device), torch.from_numpy(np.ascontiguousarray(w)).to(device), t, _span(s)


def taps(kind: str, n_h: int, n_w: int, transposed: bool, device):
    """Device-resident taps for both axes, padded to a common T."""
    key = (kind, n_h, n_w, transposed, str(device))
    hit = _dev_cache.get(key)
    if hit is not None:
        return hit
    sy, wy, ty = _taps_host(kind, n_h, transposed)
    sx, wx, tx = _taps_host(kind, n_w, transposed)
    T = max(ty, tx)

    def fit(s, w, t, n_src):
        if t == T:
            return s, w
        # widen to T taps, keeping every tap inside the source
        s2 = np.minimum(s, max(n_src - T, 0)).astype(np.int32)
        w2 = np.zeros((w.shape[0], T), dtype=np.float32)
        for r in range(w.shape[0]):
            w2[r, s[r] - s2[r]: s[r] - s2[r] + t] = w[r]
        return s2, w2

    a_h = operator_matrix(kind, n_h)
    a_w = operator_matrix(kind, n_w)
    src_h = a_h.shape[0] if transposed else a_h.shape[1]
    src_w = a_w.shape[0] if transposed else a_w.shape[1]
    sy, wy = fit(sy, wy, ty, src_h)
    sx, wx = fit(sx, wx, tx, src_w)
    out_h, out_w = wy.shape[0], wx.shape[0]
    res = (
        torch.from_numpy(sy).to(device), torch.from_numpy(np.ascontiguousarray(wy)).to(device),
        torch.from_numpy(sx).to(device), torch.from_numpy(np.ascontiguousarray(wx)).to(device),
        Taps(T, _span(sy), _span(sx)), out_h, out_w,
    )
    _dev_cache[key] = res
    return res
