"""Launch wrappers for the fused sampler-update kernels (fp32 NCHW, per-sample scalars)."""
from __future__ import annotations


import torch

from .._hip import call, ptr


def _prep(t: torch.Tensor) -> torch.Tensor:
    if not t.is_cuda:
        raise RuntimeError("perceptor_amd sampler updates run on a HIP device only (no CPU fallback)")
    return t.float().contiguous()


def _vec(v: torch.Tensor, n: int, device) -> torch.Tensor:
    if not torch.is_tensor(v):          # a Python scalar: a device-side fill (a host-to-device copy is not capturable in a HIP graph)
        return torch.full((n,), float(v), dtype=torch.float32, device=device)
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


# ---------------------------------------------------------------------------------------------------------------- device RNG
class DeviceRng:
    """Counter-based normal noise for the stochastic Predictions variants (csrc/sampling.hip: Philox4x32-10 + Box-Muller).

    Seeding contract: every draw takes a fresh 128-bit (seed, stream) pair -- two int64 from torch's default CPU generator (or from the
    generator installed by ``rng.manual_seed``) -- and element e of the draw is a function of (seed, stream, sample_offset * chw + e) only.
      * ``torch.manual_seed(s)`` therefore makes a run reproducible exactly as it does for the reference's ``torch.randn_like`` calls
        (the VALUES differ from torch's device generator, which no script can rely on across devices or torch versions anyway), and
        ranks that seed alike draw alike;
      * the values do not depend on launch geometry, device count or how the batch is split: a process that holds samples
        [sample_offset, sample_offset + N_local) of the global batch gets the noise a single process would have drawn for them.
    """

    def __init__(self):
        self.generator = None          # None: torch's default CPU generator
        self.sample_offset = 0

    def manual_seed(self, seed: int) -> "DeviceRng":
        self.generator = torch.Generator().manual_seed(int(seed))
        return self

    def next_key(self):
        k = torch.randint(-(1 << 63), (1 << 63) - 1, (2,), dtype=torch.int64, generator=self.generator)
        return int(k[0]), int(k[1])

    def randn_like(self, t: torch.Tensor) -> torch.Tensor:
        if not t.is_cuda:
            raise RuntimeError("perceptor_amd sampler updates run on a HIP device only (no CPU fallback)")
        seed, stream = self.next_key()
        out = torch.empty(t.shape, dtype=torch.float32, device=t.device)
        chw = out[0].numel() if out.ndim > 1 else 1
        call("pmi_randn", ptr(out), out.numel(), self.sample_offset * chw, seed, stream)
        return out


def _i64(v: int) -> int:
    v &= (1 << 64) - 1
    return v - (1 << 64) if v >= (1 << 63) else v


rng = DeviceRng()


def randn_like(t: torch.Tensor) -> torch.Tensor:
    return rng.randn_like(t)


# ------------------------------------------------------------------------------------------- quantile / sort-based statistics
def quantile_abs(x: torch.Tensor, q: float) -> torch.Tensor:
    """torch.quantile(x.flatten(1).abs(), q, dim=1) by radix select (no sort, no |x| temporary)."""
    x = _prep(x)
    n = x.shape[0]
    out = torch.empty(n, dtype=torch.float32, device=x.device)
    call("pmi_quantile_abs", ptr(x), ptr(out), n, x[0].numel(), float(q))
    return out


def sort_rows(x: torch.Tensor) -> torch.Tensor:
    """Each sample's elements sorted ascending: [N, n] view of a padded work buffer."""
    from .._hip import lib
    x = _prep(x)
    rows, n = x.shape[0], x[0].numel()
    npad = lib().pmi_sort_rows_padded(n)
    if npad < 0:
        raise RuntimeError("pmi_sort_rows: row length out of range")
    work = torch.empty(rows, npad, dtype=torch.float32, device=x.device)
    call("pmi_sort_rows", ptr(x), ptr(work), rows, n)
    return work[:, :n]


def wasserstein(x: torch.Tensor, power: int) -> torch.Tensor:
    """mean |sort(x_n) - Normal(0,1).icdf(linspace(0.5/n, 1-0.5/n, n))|^power over all samples (0-dim tensor)."""
    from .._hip import lib
    x = _prep(x)
    rows, n = x.shape[0], x[0].numel()
    npad = lib().pmi_sort_rows_padded(n)
    if npad < 0:
        raise RuntimeError("pmi_sort_rows: row length out of range")
    work = torch.empty(rows, npad, dtype=torch.float32, device=x.device)
    partial = torch.empty(1024, dtype=torch.float32, device=x.device)
    out = torch.empty(1, dtype=torch.float32, device=x.device)
    call("pmi_sort_rows", ptr(x), ptr(work), rows, n)
    call("pmi_wasserstein", ptr(work), rows, n, int(power), ptr(partial), ptr(out))
    return out[0]


def clamp_grad(x, grad, lo, hi):
    """Backward of clamp_with_grad: grad * (grad * (x - clamp(x, lo, hi)) >= 0)."""
    x, grad = _prep(x), _prep(grad)
    n, chw = x.shape[0], x[0].numel()
    out = torch.empty_like(x)
    vlo, vhi = _vec(lo, n, x.device), _vec(hi, n, x.device)
    call("pmi_clamp_grad", ptr(x), ptr(grad), ptr(vlo), ptr(vhi), ptr(out), n, chw)
    return out
