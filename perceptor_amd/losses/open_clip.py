"""losses.OpenCLIP / losses.CLIP — drop-ins for perceptor/losses/open_clip.py:7-97 and losses/clip/clip.py:10-99.

loss = mean_{n,k}( w_k * 2*asin(|e_n - t_k| / 2)^2 ) [* multiplier]; its gradient w.r.t. the image is the
guidance gradient.  ``forward(images)`` keeps the reference contract (a scalar you can ``.backward()``
when ``images.requires_grad``); ``loss_and_grad(images)`` is the fused path the sampler loop and
bench.py use: ViT forward -> pmi_spherical_loss (loss + dL/d emb through F.normalize) -> ViT input-gradient.
"""
from __future__ import annotations

import json
from pathlib import Path

import torch
import torch.nn.functional as F

from .. import models
from .._hip import call, ptr


class LossInterface(torch.nn.Module):
    """A differentiable score on images: ``forward(images) -> scalar`` (role of perceptor/losses/interface.py)."""

    def forward(self, images):
        raise NotImplementedError(type(self).__name__ + ".forward")


class _SphericalBase(LossInterface):
    multiplier = 1.0
    _normalize_targets = False

    @property
    def device(self):
        return self.model.device

    def to(self, device):
        super().to(device)
        self.model.to(device)
        return self

    def add_texts_(self, texts, weights=None):
        return self.add_encodings_(self.model.encode_texts(texts), weights)

    def add_images_(self, images, weights=None):
        return self.add_encodings_(self.model.encode_images(images), weights)

    def add_encodings_(self, encodings, weights=None):
        if isinstance(weights, (list, tuple)):
            weights = torch.tensor(weights)
        elif weights is None:
            weights = torch.ones_like(encodings[:, 0])
        enc = F.normalize(encodings) if self._normalize_targets else encodings
        enc, weights = enc.detach().float().to(self.device), weights.detach().float().to(self.device)
        if self.encodings is None:
            self.encodings = torch.nn.Parameter(enc, requires_grad=False)
            self.weights = torch.nn.Parameter(weights, requires_grad=False)
        else:
            self.encodings = torch.nn.Parameter(torch.cat([self.encodings, enc]), requires_grad=False)
            self.weights = torch.nn.Parameter(torch.cat([self.weights, weights]), requires_grad=False)
        return self

    def forward(self, images):
        image_encodings = self.model.encode_images(images)
        d = (image_encodings[:, None] - self.encodings[None, :]).norm(dim=2).div(2).arcsin().square().mul(2)
        loss = (d * self.weights).mean()
        return loss.mul(self.multiplier) if self.multiplier != 1.0 else loss

    @torch.no_grad()
    def loss_and_grad(self, images, n_total=None):
        """(loss, dloss/dimages).  ``n_total``: global batch when this rank holds a shard, so the
        mean (and therefore the gradient) equals the single-process value."""
        eng = self.model._need_engine()
        images = images.to(self.device)
        n = images.shape[0]
        emb = eng.forward(images, save=True).contiguous()
        k, dim = self.encodings.shape
        loss = torch.empty(1, dtype=torch.float32, device=self.device)
        demb = torch.empty_like(emb)
        enc, wts = self.encodings.data.contiguous(), self.weights.data.contiguous()
        call("pmi_spherical_loss", ptr(emb), ptr(enc), ptr(wts), ptr(loss), ptr(demb),
             n, k, dim, int(n_total or n), float(self.multiplier), float(eng.gscale))
        return loss[0], eng.backward(demb)


class OpenCLIP(_SphericalBase):
    def __init__(self, architecture="ViT-H-14", weights="laion2b_s32b_b79k", **kw):
        super().__init__()
        self.architecture = architecture
        self.model = models.OpenCLIP(architecture, weights, kw.pop("dtype", None), **kw)
        self.encodings = None
        self.weights = None


class CLIP(_SphericalBase):
    _normalize_targets = True   # losses/clip/clip.py:70-82 normalises stored targets, losses/open_clip.py:68-85 does not

    def __init__(self, name="ViT-B-32", precision="fp32", jit=False, **kw):
        """The reference passes (name, precision, jit) to models.CLIP(architecture, precision) and raises TypeError
        (losses/clip/clip.py:28 vs models/clip.py:6); ``jit`` is accepted and ignored here."""
        super().__init__()
        self.name = name
        self.model = models.CLIP(name, precision, **kw)
        self.encodings = None
        self.weights = None
        self.multiplier = 0.01 if name in ("ViT-L-14", "ViT-L-14-336") else 1.0

    def mul_(self, multiplier):
        self.multiplier *= multiplier
        return self

    def add_text_off_(self, weight=None):
        path = Path("perceptor/losses/clip/vectors/textoff.json")
        if not path.exists():
            raise ValueError(f"There is no textoff for this model: {self.name} ({path} not found)")
        textoff_json = json.loads(path.read_text())
        if self.name in textoff_json:
            return self.add_encodings_(torch.tensor(textoff_json[self.name]), weight)
        raise ValueError(f"There is no textoff for this model: {self.name}")
