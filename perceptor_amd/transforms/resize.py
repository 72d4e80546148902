"""ResizeRight-compatible resize as banded 1-D operators applied by pmi_resize_apply.

Drop-in for perceptor.transforms.resize.resize with its default arguments
(perceptor/transforms/resize/resize_right.py:34-189): antialiased, lanczos3 when both
dims shrink else bicubic (:102-108), zero ("constant") padding, dims processed in order of
increasing scale factor (:113-117).  The host builds, once per (in, out) size, the tap table
  idx[out][taps] (input row or -1), w[out][taps]
and its transpose (for the image gradient); the GPU applies them.

Table construction follows the reference's arithmetic step by step:
  projected grid       resize_right.py:198-207
  field of view        :210-219   (left = ceil(grid - support/2 - eps), `ceil(support - eps)` taps)
  antialiasing         :463-472   (kernel stretched by 1/scale, support / scale)
  weights              :275-285   (normalised to sum 1 over ALL taps, also those falling in the zero pad)
  kernels              interpolation_methods.py:38-60
"""
from __future__ import annotations

from functools import lru_cache
from math import ceil, pi
from typing import Tuple

import numpy as np
import torch

from .._hip import call, ptr

_EPS = float(np.finfo(np.float32).eps)


def _lanczos3(x):
    return ((torch.sin(pi * x) * torch.sin(pi * x / 3) + _EPS) / ((pi**2 * x**2 / 3) + _EPS)) * (x.abs() < 3).to(x.dtype)


def _cubic(x):
    a = x.abs()
    a2, a3 = a**2, a**3
    return (1.5 * a3 - 2.5 * a2 + 1.0) * (a <= 1.0).to(x.dtype) + \
        (-0.5 * a3 + 2.5 * a2 - 4.0 * a + 2.0) * ((1.0 < a) & (a <= 2.0)).to(x.dtype)


_METHODS = {"lanczos3": (_lanczos3, 6), "cubic": (_cubic, 4)}


@lru_cache(maxsize=64)
def band_tables(in_sz: int, out_sz: int, method: str):
    """(idx [out,taps] int32, w [out,taps] f32, idx_T [in,tapsT] int32, w_T [in,tapsT] f32) on CPU."""
    fn, support = _METHODS[method]
    scale = out_sz / in_sz
    grid = torch.arange(out_sz) / float(scale) + (in_sz - 1) / 2 - (out_sz - 1) / (2 * float(scale))
    if scale < 1.0:
        cur_support = support / scale
        f = lambda a: scale * fn(scale * a)
    else:
        cur_support, f = support, fn
    left = (grid - cur_support / 2 - _EPS).ceil().long()
    fov = left[:, None] + torch.arange(ceil(cur_support - _EPS))
    # the reference shifts grid and field of view by the left pad before evaluating the kernel
    # (calc_pad_sz, resize_right.py:222-234); same fp32 rounding here
    pad0 = -int(fov[0, 0])
    w = f((grid + pad0)[:, None] - (fov + pad0))
    s = w.sum(1, keepdim=True)
    s[s == 0] = 1
    w = (w / s).float()
    valid = (fov >= 0) & (fov < in_sz)
    idx = torch.where(valid, fov, torch.full_like(fov, -1)).int()
    w = torch.where(valid, w, torch.zeros_like(w))
    # transpose band: for each input row the (output row, weight) pairs
    rows = [[] for _ in range(in_sz)]
    for j in range(out_sz):
        for t in range(idx.shape[1]):
            r = int(idx[j, t])
            if r >= 0:
                rows[r].append((j, float(w[j, t])))
    tt = max(1, max(len(r) for r in rows))
    idx_t = torch.full((in_sz, tt), -1, dtype=torch.int32)
    w_t = torch.zeros((in_sz, tt), dtype=torch.float32)
    for r, lst in enumerate(rows):
        for k, (j, wv) in enumerate(lst):
            idx_t[r, k] = j
            w_t[r, k] = wv
    return idx.contiguous(), w.contiguous(), idx_t.contiguous(), w_t.contiguous()


_dev_cache = {}


def _tables_on(device, in_sz, out_sz, method):
    key = (str(device), in_sz, out_sz, method)
    if key not in _dev_cache:
        _dev_cache[key] = tuple(t.to(device) for t in band_tables(in_sz, out_sz, method))
    return _dev_cache[key]


def _apply(x: torch.Tensor, idx, w, axis: int, out_sz: int) -> torch.Tensor:
    """Apply a band along ``axis`` (2 = H, 3 = W) of an NCHW fp32 tensor."""
    n, c, h, wd = x.shape
    if axis == 2:
        outer, in_sz, inner = n * c, h, wd
        out = torch.empty((n, c, out_sz, wd), dtype=torch.float32, device=x.device)
    else:
        outer, in_sz, inner = n * c * h, wd, 1
        out = torch.empty((n, c, h, out_sz), dtype=torch.float32, device=x.device)
    call("pmi_resize_apply", ptr(x), ptr(out), ptr(idx), ptr(w), outer, in_sz, inner, out_sz, idx.shape[1], 0, 0)
    return out


def _plan(h: int, w: int, out_shape: Tuple[int, int]):
    oh, ow = out_shape
    method = "lanczos3" if (h >= oh and w >= ow) else "cubic"
    dims = sorted([(oh / h, 2, h, oh), (ow / w, 3, w, ow)], key=lambda z: z[0])
    return method, [d for d in dims if d[0] != 1.0]


def resize(images: torch.Tensor, out_shape: Tuple[int, int]) -> torch.Tensor:
    """Forward resize of NCHW fp32 images on the HIP device."""
    if not images.is_cuda:
        raise RuntimeError("perceptor_amd.transforms.resize runs on a HIP device only (no CPU fallback)")
    x = images.float().contiguous()
    method, dims = _plan(x.shape[2], x.shape[3], out_shape)
    for _, axis, i, o in dims:
        idx, w, _, _ = _tables_on(x.device, i, o, method)
        x = _apply(x, idx, w, axis, o)
    return x


def resize_backward(grad_out: torch.Tensor, in_hw: Tuple[int, int]) -> torch.Tensor:
    """Adjoint of ``resize`` (gradient w.r.t. the input images)."""
    g = grad_out.float().contiguous()
    method, dims = _plan(in_hw[0], in_hw[1], (g.shape[2], g.shape[3]))
    for _, axis, i, o in reversed(dims):
        _, _, idx_t, w_t = _tables_on(g.device, i, o, method)
        g = _apply(g, idx_t, w_t, axis, i)
    return g
