"""Latent eps-form Predictions — drop-in for perceptor.models.stable_diffusion.predictions.Predictions.

Same fields, properties, method names, argument meaning and error behaviour as the reference
(perceptor/models/stable_diffusion/predictions.py:10-250); the arithmetic runs in the fused HIP kernels (pmi_lincomb2, pmi_guided_update,
pmi_clamp, pmi_quantile_abs, pmi_randn, pmi_sort_rows, pmi_wasserstein) with per-sample coefficients gathered in torch.
"""
from __future__ import annotations

import torch

from ...engine import sampler
from ...utils.record import FrozenRecord


class Predictions(FrozenRecord):
    _fields = ("from_diffused_latents", "from_indices", "predicted_noise", "schedule_alphas", "schedule_sigmas", "encode", "decode")

    def __repr__(self):
        return f"Predictions(from_diffused_latents=<{tuple(self.from_diffused_latents.shape)}>, from_indices={self.from_indices.tolist()})"

    @property
    def device(self):
        return self.predicted_noise.device

    def indices(self, indices):
        if isinstance(indices, (float, int)):
            indices = torch.as_tensor(indices)
        if indices.ndim == 0:
            indices = indices[None]
        if indices.ndim != 1:
            raise ValueError("indices must be a scalar or a 1D tensor")
        return indices.long().to(self.device)

    def _a(self, indices):
        return self.schedule_alphas.to(self.device)[self.indices(indices)]

    def _s(self, indices):
        return self.schedule_sigmas.to(self.device)[self.indices(indices)]

    def alphas(self, indices):
        return self._a(indices)[:, None, None, None]

    def sigmas(self, indices):
        return self._s(indices)[:, None, None, None]

    @property
    def from_alphas(self):
        return self.alphas(self.from_indices)

    @property
    def from_sigmas(self):
        return self.sigmas(self.from_indices)

    @property
    def denoised_latents(self):
        # (x - sigma*eps) / max(alpha, 1e-7)                                predictions.py:51-54
        a = self._a(self.from_indices).clamp(min=1e-7)
        return sampler.lincomb2(self.from_diffused_latents, 1.0 / a, self.predicted_noise, -self._s(self.from_indices) / a)

    @property
    def denoised_images(self):
        return self.decode(self.denoised_latents)

    def _to(self, at, s_eps):
        """denoised_latents * at + predicted_noise * s_eps in one pass."""
        a = self._a(self.from_indices).clamp(min=1e-7)
        return sampler.lincomb2(self.from_diffused_latents, at / a, self.predicted_noise, s_eps - at * self._s(self.from_indices) / a)

    def step(self, to_indices, eta=0.0):
        """Reduce noise level to ``to_indices`` (DDIM; eta > 0 adds fresh noise).   predictions.py:60-98"""
        af, sf = self._a(self.from_indices), self._s(self.from_indices)
        at, st = self._a(to_indices), self._s(to_indices)
        if eta > 0.0:
            ddim_sigma = eta * (st**2 / sf**2).sqrt() * (1 - af**2 / at**2).sqrt()
            adjusted = (st**2 - ddim_sigma**2).sqrt()
            nxt = self._to(at, adjusted)
            return sampler.lincomb2(nxt, 1.0, sampler.randn_like(nxt), ddim_sigma)
        return self._to(at, st)

    def correction(self, previous: "Predictions") -> "Predictions":
        # the reference calls a non-existent ``forced_denoised`` (predictions.py:118-120); the evident intent:
        return previous.forced_denoised_latents(sampler.lincomb2(self.denoised_latents, 0.5, previous.denoised_latents, 0.5))

    def reverse_step(self, to_indices):
        if (torch.as_tensor(self.from_indices).cpu() > torch.as_tensor(to_indices).cpu()).any():
            raise ValueError("from_indices must be less than to_indices")
        return self._to(self._a(to_indices), self._s(to_indices))

    def resample_noise(self, resample_indices):
        if (torch.as_tensor(self.from_indices).cpu() < torch.as_tensor(resample_indices).cpu()).any():
            raise ValueError("from_indices must be greater than resample_indices")
        sf, sr = self._s(self.from_indices), self._s(resample_indices)
        return sampler.lincomb2(self.predicted_noise, sr / sf, sampler.randn_like(self.predicted_noise), (sf**2 - sr**2).sqrt() / sf)

    def resample(self, resample_indices):
        """Harmonizing resampling (RePaint).   predictions.py:141-148"""
        return sampler.lincomb2(self.denoised_latents, self._a(self.from_indices), self.resample_noise(resample_indices), self._s(self.from_indices))

    def noisy_reverse_step(self, to_indices):
        at, st = self._a(to_indices), self._s(to_indices)
        sf = self._s(self.from_indices)
        noise_sigma = sampler.lincomb2(self.predicted_noise, sf, sampler.randn_like(self.predicted_noise), (st**2 - sf**2).sqrt())
        return sampler.lincomb2(self.denoised_latents, at, noise_sigma, 1.0)

    def guided(self, guiding, guidance_scale=0.5, clamp_value=1e-6) -> "Predictions":
        return self.replace(predicted_noise=sampler.guided_update(
            self.predicted_noise, guiding, self._s(self.from_indices), guidance_scale, clamp_value))

    def latent_dynamic_threshold(self, quantile=0.95) -> "Predictions":
        if quantile is None:
            return self
        thr = sampler.quantile_abs(self.predicted_noise, quantile).clamp(min=2.5)
        return self.forced_predicted_noise(sampler.clamp(self.predicted_noise, -thr, thr))

    def dynamic_threshold(self, quantile=0.95) -> "Predictions":
        """Imagen thresholding in image space (predictions.py:195-216): decode, clamp at the per-sample quantile, encode back."""
        if quantile is None:
            return self
        xs = sampler.lincomb2(self.decode(self.denoised_latents), 2.0, cc=-1.0)
        thr = sampler.quantile_abs(xs, quantile).clamp(min=1.0)
        xs = sampler.lincomb2(sampler.clamp(xs, -thr, thr), 0.5 / thr, cc=0.5)          # clamp / threshold, then decode: (x + 1) / 2
        return self.forced_denoised_latents(self.encode(xs))

    def forced_denoised_latents(self, denoised_latents) -> "Predictions":
        # eps = (x - x0*alpha) / max(sigma, 1e-7)                          predictions.py:218-222
        s = self._s(self.from_indices).clamp(min=1e-7)
        return self.replace(predicted_noise=sampler.lincomb2(self.from_diffused_latents, 1.0 / s, denoised_latents, -self._a(self.from_indices) / s))

    def forced_predicted_noise(self, predicted_noise) -> "Predictions":
        return self.replace(predicted_noise=predicted_noise)

    def wasserstein_distance(self):
        return sampler.wasserstein(self.predicted_noise, 1)

    def wasserstein_square_distance(self):
        return sampler.wasserstein(self.predicted_noise, 2)

    def classifier_free_guidance(self, positive_predictions: "Predictions", guidance_scale=7.0) -> "Predictions":
        # eps + (eps_pos - eps) * scale                                   predictions.py:243-250
        return self.replace(predicted_noise=sampler.lincomb2(self.predicted_noise, 1.0 - guidance_scale,
                                                             positive_predictions.predicted_noise, guidance_scale))
