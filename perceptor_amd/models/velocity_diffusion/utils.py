"""t <-> (alpha, sigma) <-> log-SNR conversions (perceptor/models/velocity_diffusion/utils.py:24-49)."""
import math

import torch


def log_snr_to_alpha_sigma(log_snr):
    return log_snr.sigmoid().sqrt(), log_snr.neg().sigmoid().sqrt()


def alpha_sigma_to_log_snr(alpha, sigma):
    return torch.log(alpha**2 / sigma**2)


def t_to_alpha_sigma(t):
    return torch.cos(t * math.pi / 2), torch.sin(t * math.pi / 2)


def alpha_sigma_to_t(alpha, sigma):
    return torch.atan2(sigma, alpha) / math.pi * 2


def sigma_to_t(sigma):
    return torch.asin(sigma) / math.pi * 2
