"""v-form Predictions — drop-in for perceptor.models.velocity_diffusion.predictions.Predictions
(perceptor/models/velocity_diffusion/predictions.py:9-216); arithmetic in the fused HIP kernels
pmi_ddim_v_step / pmi_guided_update / pmi_lincomb2 / pmi_clamp."""
from __future__ import annotations

import torch

from ...engine import sampler
from ...utils.record import FrozenRecord
from . import utils


class Predictions(FrozenRecord):
    _fields = ("from_diffused_images", "from_ts", "velocities")

    @property
    def device(self):
        return self.velocities.device

    def _ts(self, ts):
        if isinstance(ts, float):
            ts = torch.tensor(ts)
        if ts.ndim == 0:
            ts = ts[None]
        if ts.ndim != 1:
            raise ValueError("ts must be a scalar or a 1D tensor")
        return ts.to(self.device)

    def _a(self, ts):
        return utils.t_to_alpha_sigma(self._ts(ts))[0]

    def _s(self, ts):
        return utils.t_to_alpha_sigma(self._ts(ts))[1]

    def alphas(self, ts):
        return self._a(ts)[:, None, None, None]

    def sigmas(self, ts):
        return self._s(ts)[:, None, None, None]

    @property
    def from_alphas(self):
        return self.alphas(self.from_ts)

    @property
    def from_sigmas(self):
        return self.sigmas(self.from_ts)

    @property
    def from_diffused_xs(self):
        return sampler.lincomb2(self.from_diffused_images, 2.0, cc=-1.0)

    @property
    def denoised_xs(self):
        a, s = self._a(self.from_ts), self._s(self.from_ts)
        return sampler.lincomb2(self.from_diffused_images, 2.0 * a, self.velocities, -s, -a)     # x*alpha - v*sigma

    @property
    def predicted_noise(self):
        a, s = self._a(self.from_ts), self._s(self.from_ts)
        return sampler.lincomb2(self.from_diffused_images, 2.0 * s, self.velocities, a, -s)      # x*sigma + v*alpha

    @property
    def denoised_images(self):
        _, den = sampler.ddim_step("v", self.from_diffused_images, self.velocities, self._a(self.from_ts), self._s(self.from_ts),
                                   want_next=False, want_denoised=True)
        return den

    def step(self, to_ts, eta=0.0):
        af, sf, at, st = self._a(self.from_ts), self._s(self.from_ts), self._a(to_ts), self._s(to_ts)
        if eta > 0.0:
            ddim_sigma = eta * (st**2 / sf**2).sqrt() * (1 - af**2 / at**2).sqrt()
            adjusted = (st**2 - ddim_sigma**2).sqrt()
            nxt, _ = sampler.ddim_step("v", self.from_diffused_images, self.velocities, af, sf, at, adjusted)
            return sampler.lincomb2(nxt, 1.0, sampler.randn_like(nxt), ddim_sigma / 2)
        nxt, _ = sampler.ddim_step("v", self.from_diffused_images, self.velocities, af, sf, at, st)
        return nxt

    def correction(self, previous: "Predictions") -> "Predictions":
        return previous.forced_denoised_images(sampler.lincomb2(self.denoised_images, 0.5, previous.denoised_images, 0.5))

    def reverse_step(self, to_ts):
        if (torch.as_tensor(self.from_ts).cpu() > torch.as_tensor(to_ts).cpu()).any():
            raise ValueError("from_ts must be less than to_ts")
        # the reference returns xs here (no decode): predictions.py:112-117
        return sampler.lincomb2(self.denoised_xs, self._a(to_ts), self.predicted_noise, self._s(to_ts))

    def resample_noise(self, resample_ts):
        if (torch.as_tensor(self.from_ts).cpu() < torch.as_tensor(resample_ts).cpu()).any():
            raise ValueError("from_ts must be greater than resample_ts")
        sf, sr = self._s(self.from_ts), self._s(resample_ts)
        return sampler.lincomb2(self.predicted_noise, sr / sf, sampler.randn_like(self.velocities), (sf**2 - sr**2).sqrt() / sf)

    def resample(self, resample_ts):
        xs = sampler.lincomb2(self.denoised_xs, self._a(self.from_ts), self.resample_noise(resample_ts), self._s(self.from_ts))
        return sampler.lincomb2(xs, 0.5, cc=0.5)

    def noisy_reverse_step(self, to_ts):
        at, st, sf = self._a(to_ts), self._s(to_ts), self._s(self.from_ts)
        ns = sampler.lincomb2(self.predicted_noise, sf, sampler.randn_like(self.velocities), (st**2 - sf**2).sqrt())
        return sampler.lincomb2(self.denoised_xs, at / 2, ns, 0.5, 0.5)

    def guided(self, guiding, guidance_scale=0.5, clamp_value=1e-6) -> "Predictions":
        return self.replace(velocities=sampler.guided_update(self.velocities, guiding, self._s(self.from_ts), guidance_scale, clamp_value))

    def dynamic_threshold(self, quantile=0.95) -> "Predictions":
        xs = self.denoised_xs
        thr = sampler.quantile_abs(xs, quantile).clamp(min=1.0)      # radix select, csrc/sampling.hip
        clamped = sampler.clamp(xs, -thr, thr)
        return self.forced_denoised_images(sampler.lincomb2(clamped, 0.5 / thr, cc=0.5))          # decode(xs / thr)

    def static_threshold(self):
        den = self.denoised_images
        return self.forced_denoised_images(sampler.clamp(den, torch.zeros(1), torch.ones(1)))

    def forced_denoised_images(self, denoised_images) -> "Predictions":
        a, s = self._a(self.from_ts), self._s(self.from_ts)
        x0 = sampler.lincomb2(denoised_images, 2.0, cc=-1.0)
        if (s >= 1e-3).all():
            eps = sampler.lincomb2(self.from_diffused_images, 2.0 / s, x0, -a / s, -1.0 / s)
        else:
            eps = self.predicted_noise
        return self.replace(velocities=sampler.lincomb2(eps, a, x0, -s))

    def forced_predicted_noise(self, predicted_noise) -> "Predictions":
        a, s = self._a(self.from_ts), self._s(self.from_ts)
        if (a >= 1e-3).all():
            x0 = sampler.lincomb2(self.from_diffused_images, 2.0 / a, predicted_noise, -s / a, -1.0 / a)
        else:
            x0 = self.denoised_xs
        return self.replace(velocities=sampler.lincomb2(predicted_noise, a, x0, -s))

    def _wasserstein(self, power):
        return sampler.wasserstein(self.predicted_noise, power)      # bitonic sort + fused statistic, csrc/sampling.hip

    def wasserstein_distance(self):
        return self._wasserstein(1)

    def wasserstein_square_distance(self):
        return self._wasserstein(2)
