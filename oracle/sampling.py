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
