"""GuidedDiffusion — drop-in for perceptor.models.GuidedDiffusion on MI355X.

Same call surface as perceptor/models/guided_diffusion/guided_diffusion.py:14-155; the UNet forward
(unet.py:626-654) runs in perceptor_amd.engine.adm.AdmEngine (hand-written HIP kernels).

Differences that are deliberate and visible:
  * checkpoints cannot be downloaded offline (reference :25-36): pass ``checkpoint=<path to the
    reference .pt>`` (loaded with weights_only=True) or get deterministic synthetic weights
    (perceptor_amd.utils.synth) — ``weights="synthetic"`` is the default;
  * compute needs a HIP device: ``.to("cuda")`` builds the engine, CPU calls raise RuntimeError.
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch

from ...engine import adm
from ...utils.synth import synth_state_dict
from ...utils.image_space import images_from_x
from ...utils.param_tree import ParamTree
from .predictions import Predictions


def linear_alphas_cumprod(n: int = 1000) -> np.ndarray:
    """get_named_beta_schedule('linear', n) + cumprod (gaussian_diffusion.py:14-30,131-132), float64."""
    scale = 1000 / n
    betas = np.linspace(scale * 0.0001, scale * 0.02, n, dtype=np.float64)
    return np.cumprod(1.0 - betas, axis=0)


WeightStore = ParamTree      # frozen reference-named parameters as a module tree (utils/param_tree.py)


_CONFIGS = {"standard": (adm.openimages_config, (3, 512, 512)), "pixelart": (adm.pixelart_config, (3, 256, 256))}


class _PredictedNoise(torch.autograd.Function):
    """predicted_noise(x, t) with an input gradient, as autograd provides upstream (the UNet is an nn.Module there): forward and backward are
    the HIP engine's forward_train / backward (engine/adm.py).  Weights are frozen: no parameter gradients."""

    @staticmethod
    def forward(ctx, diffused, idx, model):
        eng = model.grad_engine
        eps, tape = eng.forward_train(diffused, idx, model.model.state_dict(), out_channels=3)
        if eng is not model.engine:          # mixed / precise model: the VALUE is that mode's, the tape (and so the gradient) the f16 engine's
            eps = model.engine.forward(diffused, idx, out_channels=3)
        ctx.model, ctx.tape = model, tape
        return eps

    @staticmethod
    def backward(ctx, grad_eps):
        m = ctx.model
        return m.grad_engine.backward(ctx.tape, grad_eps.contiguous(), m.model.state_dict()), None, None


class GuidedDiffusion(torch.nn.Module):
    def __init__(self, name="standard", *, weights="synthetic", checkpoint: Optional[str] = None, dtype="bf16", seed=0,
                 config: Optional[adm.AdmConfig] = None):
        """
        Args:
            name: The name of the model. Available models are "standard" and "pixelart"
        """
        super().__init__()
        self.name = name
        if config is not None:
            self.config, self.shape = config, (3, config.image_size, config.image_size)
        elif name in _CONFIGS:
            self.config, self.shape = _CONFIGS[name][0](), _CONFIGS[name][1]
        else:
            raise ValueError(f"Unknown model name {self.name}")
        self.compute_dtype = dtype
        shapes = adm.state_dict_shapes(self.config)
        if checkpoint is not None:
            sd = torch.load(checkpoint, map_location="cpu", weights_only=True)
            sd = {k: v.float() for k, v in sd.items()}
        elif weights == "synthetic":
            sd = synth_state_dict(shapes, seed)
        else:
            raise ValueError("weights must be 'synthetic' or a checkpoint path must be given (no network access)")
        if set(sd) != set(shapes):
            raise RuntimeError("checkpoint keys do not match the UNet configuration")
        self.model = WeightStore(sd)
        ac = linear_alphas_cumprod(1000)
        self.schedule_alphas = torch.nn.Parameter(torch.from_numpy(ac).sqrt().float(), requires_grad=False)
        self.schedule_sigmas = torch.nn.Parameter((1 - torch.from_numpy(ac)).sqrt().float(), requires_grad=False)
        self._engine: Optional[adm.AdmEngine] = None
        self.register_load_state_dict_post_hook(lambda module, incompatible: module._drop_engines())

    @property
    def engine(self) -> Optional[adm.AdmEngine]:
        """The HIP engine holds packed 16-bit copies of the weights: built on first use on a HIP device, dropped (and rebuilt from the
        current parameters) whenever the module moves or a state dict is loaded."""
        if self._engine is None and self.device.type == "cuda":
            if self.compute_dtype == "mixed":       # eps max-abs error < 1e-3 at ~1.4x the f16 MFMA work (engine/adm_mixed.py)
                from ...engine.adm_mixed import AdmMixedEngine
                self._engine = AdmMixedEngine(self.config, self.model.state_dict(), self.device)
            else:
                self._engine = adm.AdmEngine(self.config, self.model.state_dict(), self.device, self.compute_dtype)
        return self._engine

    @property
    def grad_engine(self):
        """The engine whose training-mode forward + backward give the input gradient: the model's own in the 16-bit modes; a lazily built f16
        engine for the mixed / precise modes (their split tensors have no backward kernels: the gradient is then the f16 path's, 1.4e-3
        relative to fp32 autograd on the shipped net, tests/test_gpu_backward.py -- the forward value stays the mode's own)."""
        eng = self.engine
        if eng is None or not getattr(eng, "precise", False):
            return eng
        if self.__dict__.get("_grad_engine") is None:
            self.__dict__["_grad_engine"] = adm.AdmEngine(self.config, self.model.state_dict(), self.device, "f16")
        return self.__dict__["_grad_engine"]

    def _drop_engines(self) -> None:
        self._engine = None
        self.__dict__["_grad_engine"] = None

    def _apply(self, fn, *a, **k):          # .to() / .cuda() / .cpu() / .float() ... all come through here
        self._drop_engines()
        return super()._apply(fn, *a, **k)

    def to(self, *args, **kwargs):
        super().to(*args, **kwargs)
        self.engine                           # noqa: B018  build the engine now (as before): packing 558 M weights takes seconds
        return self

    @property
    def device(self):
        return self.schedule_alphas.device

    def schedule_indices(self, n_steps=500, from_index=999, to_index=0, rho=7.0):
        """Karras-rho ramp in sigma space snapped to the 1000 discrete log-SNRs (guided_diffusion.py:58-96)."""
        if from_index < to_index:
            raise ValueError("from_index must be greater than to_index")
        alphas, sigmas = self.schedule_alphas.detach().cpu(), self.schedule_sigmas.detach().cpu()
        from_log_snr = torch.log(alphas[from_index] ** 2 / sigmas[from_index] ** 2)
        to_log_snr = torch.log(alphas[to_index] ** 2 / sigmas[to_index] ** 2)
        sigma_max = (1 / from_log_snr.exp()).sqrt().clamp(max=150)
        sigma_min = (1 / to_log_snr.exp()).sqrt().clamp(min=1e-3)
        ramp = torch.linspace(0, 1, n_steps + 1)
        karras = (sigma_max ** (1 / rho) + ramp * (sigma_min ** (1 / rho) - sigma_max ** (1 / rho))) ** rho
        target = torch.log(torch.ones_like(karras) ** 2 / karras**2)
        table = torch.log(alphas**2 / sigmas**2)
        idx = (target[:, None] - table[None, :]).abs().argmin(dim=1).unique().sort(descending=True)[0]
        assert len(idx) >= n_steps * 0.9
        assert (idx[:-1] != idx[1:]).all()
        return torch.stack([idx[:-1], idx[1:]], dim=1).to(self.device)

    def random_diffused(self, shape):
        n, c, h, w = shape
        if h % 8 != 0:
            raise ValueError("Height must be divisible by 32")
        if w % 8 != 0:
            raise ValueError("Width must be divisible by 32")
        return images_from_x(torch.randn(shape).to(self.device))

    def indices(self, indices):
        if isinstance(indices, (float, int)):
            indices = torch.as_tensor(indices)
        if indices.ndim == 0:
            indices = indices[None]
        if indices.ndim != 1:
            raise ValueError("indices must be a scalar or a 1-dimensional tensor")
        return indices.long().to(self.device)

    def alphas(self, indices):
        return self.schedule_alphas[self.indices(indices)][:, None, None, None].to(self.device)

    def sigmas(self, indices):
        return self.schedule_sigmas[self.indices(indices)][:, None, None, None].to(self.device)

    def _need_engine(self):
        if self.engine is None:
            raise RuntimeError("GuidedDiffusion needs a HIP device: call .to('cuda') first (perceptor_amd has no CPU fallback)")
        return self.engine

    def predicted_noise(self, diffused_images, from_indices):
        idx = self.indices(from_indices)
        n = diffused_images.shape[0]
        if idx.numel() == 1 and n > 1:
            idx = idx.expand(n)
        if torch.is_grad_enabled() and diffused_images.requires_grad:
            # upstream this call is differentiable (guided_diffusion.py:125-133: autocast, no no_grad; the blocks run through
            # CheckpointFunction, nn.py:138-189): the engine's training-mode forward + tape backward give the same input gradient
            self._need_engine()
            return _PredictedNoise.apply(diffused_images.to(self.device), idx, self)
        return self._need_engine().forward(diffused_images.to(self.device), idx, out_channels=3)

    def predictions(self, diffused_images, indices) -> Predictions:
        indices = self.indices(indices)
        return Predictions(
            from_diffused_images=diffused_images,
            from_indices=indices,
            predicted_noise=self.predicted_noise(diffused_images, indices),
            schedule_alphas=self.schedule_alphas,
            schedule_sigmas=self.schedule_sigmas,
        )

    def forward(self, diffused_images, indices) -> Predictions:
        return self.predictions(diffused_images, indices)

    def diffuse_images(self, denoised_images, indices, noise=None):
        from ...engine import sampler
        indices = self.indices(indices)
        if noise is None:
            noise = sampler.randn_like(denoised_images)
        a, s = self.schedule_alphas[indices], self.schedule_sigmas[indices]
        # decode(encode(img)*alpha + noise*sigma) = img*alpha + noise*sigma/2 + (1-alpha)/2
        return sampler.lincomb2(denoised_images, a, noise, s / 2, (1 - a) / 2)
