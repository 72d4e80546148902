"""VelocityDiffusion — drop-in for perceptor.models.VelocityDiffusion
(perceptor/models/velocity_diffusion/velocity_diffusion.py:15-164); UNet forward in VDiffEngine (HIP)."""
from __future__ import annotations

from typing import Optional

import torch

from ...engine import sampler, vdiff
from ...utils.synth import synth_state_dict
from ..guided_diffusion.guided_diffusion import WeightStore
from ...utils.image_space import images_from_x
from . import utils
from .predictions import Predictions

_SPECS = {"yfcc_2": vdiff.yfcc2_spec, "yfcc_1": vdiff.yfcc1_spec, "cc12m_1": vdiff.cc12m1_spec, "cc12m_1_cfg": vdiff.cc12m1_spec,
          "wikiart": vdiff.wikiart_spec}
_LATER = ()


class _Velocities(torch.autograd.Function):
    """velocities(x, t) with an input gradient, as autograd provides upstream (the UNet is an nn.Module there): forward and backward
    are the HIP engine's forward_train / backward (engine/vdiff.py).  Weights are frozen: no parameter gradients."""

    @staticmethod
    def forward(ctx, diffused, t, model, clip_embed=None):
        v, tape = model.engine.forward_train(diffused, t, clip_embed)
        ctx.model, ctx.tape = model, tape
        ctx.cond = (t, clip_embed.detach()) if (clip_embed is not None and clip_embed.requires_grad) else None
        return v

    @staticmethod
    def backward(ctx, grad_v):
        m = ctx.model
        if ctx.cond is not None:            # gradient to the conditioning too (upstream: velocity_diffusion.py:96-109 keeps it in the graph)
            gx, gce = m.engine.backward(ctx.tape, grad_v.contiguous(), m.model.state_dict(), cond_grad=ctx.cond)
            return gx, None, None, gce
        return m.engine.backward(ctx.tape, grad_v.contiguous(), m.model.state_dict()), None, None, None


class VelocityDiffusion(torch.nn.Module):
    def __init__(self, name="yfcc_2", *, weights="synthetic", checkpoint: Optional[str] = None, dtype="bf16", seed=0, spec=None,
                 weight_gain: float = 1.0):
        """
        Args:
            name: The name of the model. Available models are: yfcc_2, yfcc_1, cc12m_1_cfg (conditioned), wikiart
        """
        super().__init__()
        self.name = name
        if spec is not None:
            self.spec = spec
        elif name in _SPECS:
            self.spec = _SPECS[name]()
        elif name in _LATER:
            raise NotImplementedError(f"{name}: this v-diffusion net is not on the HIP path yet")
        else:
            raise KeyError(name)       # the reference indexes MODELS[name] (velocity_diffusion/models.py:13-14)
        self.compute_dtype = dtype
        shapes = vdiff.state_dict_shapes(self.spec)
        if checkpoint is not None:
            sd = {k: v.float() for k, v in torch.load(checkpoint, map_location="cpu", weights_only=True).items()}
        elif weights == "synthetic":
            sd = synth_state_dict(shapes, seed, gain=weight_gain)
        else:
            raise ValueError("weights must be 'synthetic' or a checkpoint path must be given (no network access)")
        if set(sd) != set(shapes):
            raise RuntimeError("checkpoint keys do not match the model")
        self.model = WeightStore(sd)
        if self.spec["cond"]:
            self.model.clip_model = "ViT-B-16"      # cc12m_1.py:116
        self._engine: Optional[vdiff.VDiffEngine] = None
        self.register_load_state_dict_post_hook(lambda module, incompatible: setattr(module, "_engine", None))

    @property
    def engine(self) -> Optional[vdiff.VDiffEngine]:
        """Built on first use on a HIP device; dropped when the module moves or a state dict is loaded (see GuidedDiffusion.engine)."""
        if self._engine is None and self.device.type == "cuda":
            self._engine = vdiff.VDiffEngine(self.spec, self.model.state_dict(), self.device, self.compute_dtype)
        return self._engine

    def _apply(self, fn, *a, **k):
        self._engine = None
        return super()._apply(fn, *a, **k)

    def to(self, *args, **kwargs):
        # the reference halves the net on cuda (velocity_diffusion.py:34-38); here the fp32 masters stay fp32 and the engine packs
        # its own 16-bit (or precise) copies
        super().to(*args, **kwargs)
        self.engine                           # noqa: B018
        return self

    @property
    def device(self):
        return next(self.model.parameters()).device

    @property
    def shape(self):
        return self.spec["shape"]

    @staticmethod
    def schedule_ts(n_steps=500, from_ts=1.0, to_ts=1e-2, rho=7.0):
        """Karras-rho ramp in sigma space mapped to continuous t (velocity_diffusion.py:48-66)."""
        fa, fs = utils.t_to_alpha_sigma(torch.as_tensor(from_ts))
        ta, ts_ = utils.t_to_alpha_sigma(torch.as_tensor(to_ts))
        sigma_max = (1 / utils.alpha_sigma_to_log_snr(fa, fs).exp()).sqrt().clamp(max=150)
        sigma_min = (1 / utils.alpha_sigma_to_log_snr(ta, ts_).exp()).sqrt().clamp(min=1e-3)
        ramp = torch.linspace(0, 1, n_steps + 1)
        sig = (sigma_max ** (1 / rho) + ramp * (sigma_min ** (1 / rho) - sigma_max ** (1 / rho))) ** rho
        alpha, sigma = utils.log_snr_to_alpha_sigma(utils.alpha_sigma_to_log_snr(torch.ones_like(sig), sig))
        t = utils.alpha_sigma_to_t(alpha, sigma)
        return torch.stack([t[:-1], t[1:]], dim=1)

    def random_diffused(self, shape):
        return images_from_x(torch.randn(shape)).to(self.device)

    @staticmethod
    def sigmas_to_ts(sigmas):
        return utils.sigma_to_t(torch.as_tensor(sigmas))

    def _ts(self, ts):
        if isinstance(ts, float):
            ts = torch.tensor(ts)
        if ts.ndim == 0:
            ts = ts[None]
        if ts.ndim != 1:
            raise ValueError("t must be a scalar or a 1D tensor")
        return ts

    def alphas(self, ts):
        return utils.t_to_alpha_sigma(self._ts(ts))[0][:, None, None, None].to(self.device)

    def sigmas(self, ts):
        return utils.t_to_alpha_sigma(self._ts(ts))[1][:, None, None, None].to(self.device)

    def _need_engine(self):
        if self.engine is None:
            raise RuntimeError("VelocityDiffusion needs a HIP device: call .to('cuda') first (perceptor_amd has no CPU fallback)")
        return self.engine

    def velocities(self, diffused, t, conditioning=None):
        eng = self._need_engine()
        diffused = diffused.to(self.device)
        if isinstance(t, float):
            t = torch.full((diffused.shape[0],), float(t))
        elif t.ndim == 0:          # same value as the reference's float(t), without a device->host sync (HIP-graph capturable)
            t = t.reshape(1).expand(diffused.shape[0])
        ce = conditioning.squeeze(dim=1) if (self.spec["cond"] and conditioning is not None) else None
        if self.spec["cond"] and ce is not None and ce.shape[0] == 1 and diffused.shape[0] > 1:
            ce = ce.expand(diffused.shape[0], -1)
        if torch.is_grad_enabled() and (diffused.requires_grad or (ce is not None and ce.requires_grad)):   # autograd through the UNet (guided_resample_-style scripts)
            return _Velocities.apply(diffused, t.to(self.device), self, ce.to(self.device) if ce is not None else None)
        return eng.forward(diffused, t, ce)

    def forward(self, diffused_images, ts, conditioning=None) -> Predictions:
        if isinstance(ts, float):
            ts = torch.full((diffused_images.shape[0],), float(ts)).to(diffused_images)
        elif ts.ndim == 0:
            ts = ts.reshape(1).expand(diffused_images.shape[0]).to(diffused_images)
        return Predictions(from_diffused_images=diffused_images, from_ts=ts,
                           velocities=self.velocities(diffused_images, ts, conditioning))

    def predictions(self, diffused_images, ts, conditioning=None) -> Predictions:
        return self.forward(diffused_images, ts, conditioning)

    def conditioning(self, texts=None, images=None, encodings=None):
        all_encodings = []
        if texts is not None or images is not None:
            from .. import CLIP
            clip_model = CLIP(self.model.clip_model).to(self.device)
            if texts is not None:
                all_encodings.append(clip_model.encode_texts(texts))
            if images is not None:
                all_encodings.append(clip_model.encode_images(images))
        if encodings is not None:
            all_encodings.append(encodings)
        if len(all_encodings) == 0:
            raise ValueError("Must provide at least one of texts, images, or encodings")
        return torch.stack(all_encodings, dim=0).mean(dim=0)[None]

    def diffuse(self, denoised_images, ts, noise=None):
        if isinstance(ts, float) or ts.ndim == 0:
            ts = torch.full((denoised_images.shape[0],), float(ts))
        if noise is None:
            noise = sampler.randn_like(denoised_images)
        a, s = utils.t_to_alpha_sigma(ts.to(self.device))
        return sampler.lincomb2(denoised_images, a, noise, s / 2, (1 - a) / 2)

    def inject_noise(self, diffused_images, ts, reversed_ts, extra_noise_multiplier=1.003):
        dev = self.device
        af, sf = utils.t_to_alpha_sigma(self._ts(ts).to(dev))
        ar, sr = utils.t_to_alpha_sigma(self._ts(reversed_ts).to(dev))
        mult = ar / af
        add_std = (sr.square() - sf.square() * mult.square()).sqrt()
        diffused_images = diffused_images.to(dev)
        # decode(x*mult + std*noise*k) with x = 2*img-1
        return sampler.lincomb2(diffused_images, mult, sampler.randn_like(diffused_images), add_std * extra_noise_multiplier / 2, (1 - mult) / 2)
