"""TEST INFRASTRUCTURE (oracle) — schedules and Predictions algebra, CPU fp32.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this file; the product path never does.

Mirrors
  perceptor/models/guided_diffusion/gaussian_diffusion.py:14-30,131-132 (linear betas, cumprod)
  perceptor/models/guided_diffusion/guided_diffusion.py:44-52,58-96     (alpha/sigma tables, schedule_indices)
  perceptor/models/guided_diffusion/predictions.py:51-59,61-98,147-154,174-179
  perceptor/models/velocity_diffusion/velocity_diffusion.py:48-66       (schedule_ts)
  perceptor/models/velocity_diffusion/utils.py:24-49
  perceptor/models/velocity_diffusion/predictions.py:50-62,68-105,148-155,177-200
Pinned against tests/golden/sampling.npz.
"""
from __future__ import annotations

import math

import numpy as np
import torch


def gd_tables(n=1000):
    betas = np.linspace(1000 / n * 1e-4, 1000 / n * 0.02, n, dtype=np.float64)
    ac = np.cumprod(1.0 - betas, axis=0)
    return torch.from_numpy(ac).sqrt().float(), (1 - torch.from_numpy(ac)).sqrt().float()


def schedule_indices(alphas, sigmas, n_steps=500, from_index=999, to_index=0, rho=7.0):
    if from_index < to_index:
        raise ValueError("from_index must be greater than to_index")
    fa, fs = alphas[from_index], sigmas[from_index]
    ta, ts = alphas[to_index], sigmas[to_index]
    f_sig = (1 / torch.log(fa**2 / fs**2).exp()).sqrt().clamp(max=150)
    t_sig = (1 / torch.log(ta**2 / ts**2).exp()).sqrt().clamp(min=1e-3)
    ramp = torch.linspace(0, 1, n_steps + 1)
    s = (f_sig ** (1 / rho) + ramp * (t_sig ** (1 / rho) - f_sig ** (1 / rho))) ** rho
    target = torch.log(torch.ones_like(s) ** 2 / s**2)
    sched = torch.log(alphas**2 / sigmas**2)
    idx = (target[:, None] - sched[None, :]).abs().argmin(dim=1).unique().sort(descending=True)[0]
    assert len(idx) >= n_steps * 0.9
    return torch.stack([idx[:-1], idx[1:]], dim=1)


def t_to_alpha_sigma(t):
    return torch.cos(t * math.pi / 2), torch.sin(t * math.pi / 2)


def schedule_ts(n_steps=500, from_ts=1.0, to_ts=1e-2, rho=7.0):
    fa, fs = t_to_alpha_sigma(torch.as_tensor(from_ts))
    ta, ts = t_to_alpha_sigma(torch.as_tensor(to_ts))
    f_sig = (1 / torch.log(fa**2 / fs**2).exp()).sqrt().clamp(max=150)
    t_sig = (1 / torch.log(ta**2 / ts**2).exp()).sqrt().clamp(min=1e-3)
    ramp = torch.linspace(0, 1, n_steps + 1)
    s = (f_sig ** (1 / rho) + ramp * (t_sig ** (1 / rho) - f_sig ** (1 / rho))) ** rho
    log_snr = torch.log(torch.ones_like(s) ** 2 / s**2)
    alpha, sigma = log_snr.sigmoid().sqrt(), log_snr.neg().sigmoid().sqrt()
    t = torch.atan2(sigma, alpha) / math.pi * 2
    return torch.stack([t[:-1], t[1:]], dim=1)


def _b(v):
    return v[:, None, None, None]


# ---- epsilon form ---------------------------------------------------------
def eps_denoised_xs(images, eps, a_f, s_f):
    return (images * 2 - 1 - _b(s_f) * eps) / _b(a_f).clamp(min=1e-7)


def eps_step(images, eps, a_f, s_f, a_t, s_t):
    x0 = eps_denoised_xs(images, eps, a_f, s_f)
    return (x0 * _b(a_t) + eps * _b(s_t) + 1) / 2


def guided(pred, grad, s_f, guidance_scale=0.5, clamp_value=1e-6):
    return pred + guidance_scale * _b(s_f) * grad.clamp(-clamp_value, clamp_value) / clamp_value


def eps_forced_denoised_images(images, denoised_images, a_f, s_f):
    return (images * 2 - 1 - (denoised_images * 2 - 1) * _b(a_f)) / _b(s_f).clamp(min=1e-7)


# ---- v form ---------------------------------------------------------------
def v_denoised_xs(images, v, t_f):
    a, s = t_to_alpha_sigma(t_f)
    return (images * 2 - 1) * _b(a) - v * _b(s)


def v_predicted_noise(images, v, t_f):
    a, s = t_to_alpha_sigma(t_f)
    return (images * 2 - 1) * _b(s) + v * _b(a)


def v_step(images, v, t_f, t_t):
    a, s = t_to_alpha_sigma(t_t)
    return (v_denoised_xs(images, v, t_f) * _b(a) + v_predicted_noise(images, v, t_f) * _b(s) + 1) / 2


# ---- stochastic / sort / quantile variants (row f3) -----------------------
# Mirrors guided_diffusion/predictions.py:61-98 (eta > 0), :116-145 (resample, noisy_reverse_step), :156-172 (dynamic_threshold),
# :184-198 (wasserstein_*); velocity_diffusion/predictions.py:68-105,119-147,148-164,187-200.  The noise is an argument here (the
# reference draws it with torch.randn_like).  Pinned against tests/golden/sampling2.npz.
def step_eta(x0, eps, a_f, s_f, a_t, s_t, eta, noise, decode=True):
    ddim_sigma = eta * (_b(s_t) ** 2 / _b(s_f) ** 2).sqrt() * (1 - _b(a_f) ** 2 / _b(a_t) ** 2).sqrt()
    adjusted = (_b(s_t) ** 2 - ddim_sigma**2).sqrt()
    xs = x0 * _b(a_t) + eps * adjusted + noise * ddim_sigma
    return (xs + 1) / 2 if decode else xs


def resample_noise(eps, s_f, s_r, noise):
    return (_b(s_r) * eps + (_b(s_f) ** 2 - _b(s_r) ** 2).sqrt() * noise) / _b(s_f)


def resample(x0, eps, a_f, s_f, s_r, noise):
    return (x0 * _b(a_f) + resample_noise(eps, s_f, s_r, noise) * _b(s_f) + 1) / 2


def noisy_reverse_step(x0, eps, s_f, a_t, s_t, noise):
    return (x0 * _b(a_t) + _b(s_f) * eps + (_b(s_t) ** 2 - _b(s_f) ** 2).sqrt() * noise + 1) / 2


def quantile_abs(x, q):
    return torch.quantile(x.flatten(start_dim=1).abs(), q, dim=1)


def wasserstein(pred, power):
    s = pred.flatten(start_dim=1).sort(dim=1)[0]
    n = s.shape[1]
    expected = torch.distributions.Normal(0, 1).icdf(torch.linspace(0.5 / n, 1 - 0.5 / n, n))
    d = (s - expected[None]).abs()
    return (d if power == 1 else d.square()).mean()


def clamp_with_grad_backward(x, grad, lo, hi):
    """transforms/clamp_with_grad.py:16-23"""
    return grad * (grad * (x - x.clamp(lo, hi)) >= 0)


# ---- the product's counter-based generator, restated (csrc/sampling.hip: pmi_randn) ----------------------------------------
# Philox4x32-10 as published (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11; Random123 v1.x),
# pinned by that library's known-answer vectors (tests/test_oracle_golden.py).  There is no reference counterpart: the reference calls
# torch.randn_like, whose values are not reproducible across devices.
_M0, _M1, _W0, _W1 = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85


def philox4x32_10(counter, key):
    """counter: uint32 array [..., 4]; key: (k0, k1).  Returns uint32 [..., 4]."""
    c = [np.asarray(counter[..., i], dtype=np.uint64) for i in range(4)]
    k0, k1 = np.uint64(key[0]), np.uint64(key[1])
    mask = np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0, p1 = np.uint64(_M0) * c[0], np.uint64(_M1) * c[2]
        c = [(p1 >> np.uint64(32)) ^ c[1] ^ k0, p1 & mask, (p0 >> np.uint64(32)) ^ c[3] ^ k1, p0 & mask]
        k0, k1 = (k0 + np.uint64(_W0)) & mask, (k1 + np.uint64(_W1)) & mask
    return np.stack(c, axis=-1).astype(np.uint32)


def device_randn(shape, seed, stream, first_element=0):
    """Element e of the draw (seed, stream): lane (g & 3) of Philox block (g >> 2), g = first_element + e; Box-Muller on 23-bit uniforms."""
    n = int(np.prod(shape))
    g = np.arange(first_element, first_element + n, dtype=np.uint64)
    blk = g >> np.uint64(2)
    ctr = np.stack([blk & np.uint64(0xFFFFFFFF), blk >> np.uint64(32),
                    np.full_like(blk, stream & 0xFFFFFFFF), np.full_like(blk, (stream >> 32) & 0xFFFFFFFF)], axis=-1)
    r = philox4x32_10(ctr, (seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF))
    lane = (g & np.uint64(3)).astype(np.int64)
    pair = lane >> 1
    r0 = np.take_along_axis(r, (pair * 2)[:, None], axis=1)[:, 0]
    r1 = np.take_along_axis(r, (pair * 2 + 1)[:, None], axis=1)[:, 0]
    scale = np.float32(2.0**-23)
    u1 = ((r0 >> np.uint32(9)).astype(np.float32) + np.float32(0.5)) * scale
    u2 = ((r1 >> np.uint32(9)).astype(np.float32) + np.float32(0.5)) * scale
    rad = np.sqrt(np.float32(-2.0) * np.log(u1))
    th = np.float32(6.283185307179586) * u2
    z = np.where((lane & 1) == 0, rad * np.cos(th), rad * np.sin(th)).astype(np.float32)
    return torch.from_numpy(z.reshape(tuple(shape)))
