"""Value ranges on the sampling path: public tensors are images in [0, 1]; the UNets see x = 2*image - 1 in [-1, 1].

(The reference keeps one two-function module per model family for this: models/*/diffusion_space.py.)
On the HIP path the conversion is fused into pmi_prep_input / the sampler-update kernels; these helpers only serve
host-side code such as ``random_diffused``.
"""
import torch


def images_from_x(x: torch.Tensor) -> torch.Tensor:
    return (x + 1.0) * 0.5


def x_from_images(images: torch.Tensor) -> torch.Tensor:
    return images * 2.0 - 1.0
