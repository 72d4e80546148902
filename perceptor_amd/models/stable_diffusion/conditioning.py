"""Conditioning — drop-in for perceptor.models.stable_diffusion.conditioning.Conditioning (conditioning.py:6-42)."""
from __future__ import annotations

from typing import Optional

import torch


class Conditioning(torch.nn.Module):
    def __init__(self, model_name: str, encodings: torch.Tensor, inpainting_latent_masks: Optional[torch.Tensor] = None,
                 inpainting_latents: Optional[torch.Tensor] = None):
        super().__init__()
        self.model_name = model_name
        self.encodings = torch.nn.Parameter(encodings, requires_grad=False)
        self.inpainting_latent_masks = inpainting_latent_masks
        self.inpainting_latents = inpainting_latents

    @property
    def device(self):
        return self.encodings.device

    def __neg__(self):
        # the reference passes -encodings as model_name here (conditioning.py:24-29, raises TypeError upstream); the evident intent:
        return Conditioning(self.model_name, -self.encodings, inpainting_latent_masks=self.inpainting_latent_masks,
                            inpainting_latents=self.inpainting_latents)

    def input(self, diffused_latents):
        if self.model_name == "runwayml/stable-diffusion-inpainting":
            return torch.cat([diffused_latents, self.inpainting_latent_masks.ge(0.5).float(), self.inpainting_latents], dim=1)
        return diffused_latents
