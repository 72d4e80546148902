"""Launch wrappers for the fused sampler-update kernels (fp32 NCHW, per-sample scalars)."""
from __future__ import annotations

from typing import Optional

import torch

from .._hip import call, ptr


def _prep(t: torch.Tensor) -> torch.Tensor:
    if not t.is_cuda:
        raise RuntimeError("perceptor_amd sampler updates run on a HIP device only (no CPU fallback)")
    return t.float().contiguous()


def _vec(v: torch.Tensor, n: int, device) -> torch.Tensor:
    v = torch.as_tensor(v, dtype=torch.float32, device=device).reshape(-1)
    if v.numel() == 1 and n > 1:
        v = v.expand(n)
    assert v.numel() == n, (v.shape, n)
    return v.contiguous()


def lincomb2(a, ca, b=None, cb=None, cc=None):
    """ca[n]*a + cb[n]*b + cc[n]"""
    a = _prep(a)
    n, chw = a.shape[0], a[0].numel()
    dev = a.device
    out = torch.empty_like(a)
    bb = _prep(b) if b is not None else None
    # the coefficient vectors must stay referenced until the launch is queued: an unnamed temporary goes back to the caching
    # allocator as soon as ptr() returns and the NEXT temporary reuses (and overwrites) its block
    va = _vec(ca, n, dev)
    vb = _vec(cb, n, dev) if b is not None else None
    vc = _vec(cc, n, dev) if cc is not None else None
    call("pmi_lincomb2", ptr(a), ptr(bb), ptr(va), ptr(vb), ptr(vc), ptr(out), n, chw)
    return out


def clamp(a, lo, hi):
    a = _prep(a)
    n, chw = a.shape[0], a[0].numel()
    out = torch.empty_like(a)
    vlo, vhi = _vec(lo, n, a.device), _vec(hi, n, a.device)      # named: see lincomb2
    call("pmi_clamp", ptr(a), ptr(vlo), ptr(vhi), ptr(out), n, chw)
    return out


def ddim_step(kind: str, images, pred, a_f, s_f, a_t=None, s_t=None, want_next=True, want_denoised=False):
    """kind 'eps' or 'v'. Returns (next_images, denoised_images) (None where not requested)."""
    images, pred = _prep(images), _prep(pred)
    n, chw = images.shape[0], images[0].numel()
    dev = images.device
    nxt = torch.empty_like(images) if want_next else None
    den = torch.empty_like(images) if want_denoised else None
    af, sf = _vec(a_f, n, dev), _vec(s_f, n, dev)
    at = _vec(a_t, n, dev) if want_next else None
    st = _vec(s_t, n, dev) if want_next else None
    call("pmi_ddim_eps_step" if kind == "eps" else "pmi_ddim_v_step", ptr(images), ptr(pred), ptr(af), ptr(sf), ptr(at), ptr(st),
         ptr(nxt), ptr(den), n, chw)
    return nxt, den


def guided_update(pred, grad, s_f, scale: float, clamp_value: float):
    pred, grad = _prep(pred), _prep(grad)
    n, chw = pred.shape[0], pred[0].numel()
    out = torch.empty_like(pred)
    vs = _vec(s_f, n, pred.device)
    call("pmi_guided_update", ptr(pred), ptr(grad), ptr(vs), float(scale), float(clamp_value), ptr(out), n, chw)
    return out
