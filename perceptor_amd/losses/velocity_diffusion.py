"""losses.VelocityDiffusion — drop-in for perceptor.losses.VelocityDiffusion (reference losses/velocity_diffusion.py:11-82).

``guided_resample_`` is the reference's "guidance through the denoiser": diffuse the current image with a persistent noise tensor,
denoise it with the UNet, let the caller put a loss on the denoised image, and move the NOISE along the loss gradient.  Upstream that
gradient comes from autograd through diffuse -> UNet -> Predictions; here the UNet leg is the HIP engine's input-gradient pass
(engine/vdiff.py: forward_train / backward) and the two elementwise legs around it are applied in closed form:

    x_d   = x0 * alpha + noise * sigma                 (diffuse, velocity_diffusion.py:121-131)         images = (x + 1) / 2
    pred  = x_d * alpha - v(x_d, t) * sigma            (Predictions.denoised_xs, predictions.py:50-55)
    dL/dv = -sigma/2 * g,   dL/dx_d = alpha/2 * g + J_v^T dL/dv,   dL/dnoise = sigma * dL/dx_d          g = dL/d denoised_images
"""
from __future__ import annotations

from contextlib import contextmanager

import torch
import torch.nn.functional as F

from .. import transforms
from ..engine import sampler
from ..models.velocity_diffusion import utils
from ..models.velocity_diffusion.predictions import Predictions
from .open_clip import LossInterface


class VelocityDiffusion(LossInterface):
    def __init__(self, model, noise, from_ts=0.5, resample_ts=0.3):
        super().__init__()
        self.from_ts = from_ts
        self.resample_ts = resample_ts
        self.model = model
        self.noise = torch.nn.Parameter(noise, requires_grad=True)

    def _ts(self, n):
        return torch.full((n,), float(self.from_ts), device=self.model.device)

    def diffuse_denoise(self, denoised, **extra_kwargs):
        predictions = self.model.predictions(self.model.diffuse(denoised, self.from_ts, noise=self.noise.data), self.from_ts, **extra_kwargs)
        return predictions.denoised_images

    def forward(self, images, frozen_diffused_denoised):
        return F.mse_loss(frozen_diffused_denoised.detach().clamp(0, 1), transforms.clamp_with_grad(images))

    @contextmanager
    def guided_resample_(self, denoised, guidance_scale=0.5, clamp_value=1e-6, **extra_kwargs):
        """
        Resamples noise in direction of the gradient

        Usage:

            with diffusion.guided_resample_(images) as diffused_denoised:
                clip(diffused_denoised).backward()
        """
        conditioning = extra_kwargs.pop("conditioning", None)
        if extra_kwargs:
            raise TypeError(f"unexpected arguments {sorted(extra_kwargs)}")
        if self.noise.grad is not None:
            self.noise.grad.zero_()
        model = self.model
        n = denoised.shape[0]
        ts = self._ts(n)
        a, s = utils.t_to_alpha_sigma(ts)
        noise = self.noise.data.to(model.device)
        from_diffused = model.diffuse(denoised, ts, noise=noise)
        ce = None
        if model.spec["cond"]:
            if conditioning is None:
                raise ValueError("this model is CLIP-conditioned: pass conditioning=")
            ce = conditioning.squeeze(dim=1).to(model.device)
            if ce.shape[0] == 1 and n > 1:
                ce = ce.expand(n, -1)
        v, tape = model.engine.forward_train(from_diffused, ts, ce)
        predictions = Predictions(from_diffused_images=from_diffused, from_ts=ts, velocities=v)
        diffuse_denoise = predictions.denoised_images.detach().requires_grad_(True)
        with torch.enable_grad():
            yield diffuse_denoise
        if diffuse_denoise.grad is None:
            raise RuntimeError("guided_resample_: nothing was backpropagated to the yielded images")
        g = diffuse_denoise.grad.float()
        d_v = sampler.lincomb2(g, -s / 2)                                                  # dL/dv
        d_img = model.engine.backward(tape, d_v, model.model.state_dict())                  # J_v^T dL/dv wrt the diffused IMAGES (x = 2 img - 1)
        d_x = sampler.lincomb2(d_img, 0.5, g, a / 2)                                        # dL/dx_d
        noise_grad = sampler.lincomb2(d_x, s)
        self.noise.grad = noise_grad.to(self.noise.device)
        guided = predictions.guided(sampler.lincomb2(noise_grad, -1.0), guidance_scale=guidance_scale, clamp_value=clamp_value)
        self.noise.data = guided.resample_noise(self.resample_ts).to(self.noise.device)
        self.noise.grad.zero_()

    def compensate_noise_(self, from_denoised, to_denoised):
        # noise -= encode(to) - encode(from) = 2 * (to - from)                              velocity_diffusion.py:63-67
        self.noise.data = sampler.lincomb2(self.noise.data.to(self.model.device), 1.0,
                                           sampler.lincomb2(to_denoised, 2.0, from_denoised, -2.0), -1.0).to(self.noise.device)

    def noise_step_(self, from_denoised, from_t, to_t, to_denoised):
        # the reference calls model.forced_eps / .step / .reverse_step / .eps, none of which exist on its VelocityDiffusion
        # (velocity_diffusion.py:69-82 raises AttributeError upstream)
        raise AttributeError("noise_step_ relies on VelocityDiffusion.forced_eps/.step/.reverse_step/.eps, which the reference does not define")
