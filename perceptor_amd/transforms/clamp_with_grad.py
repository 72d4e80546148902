"""clamp_with_grad — drop-in for perceptor.transforms.clamp_with_grad (reference transforms/clamp_with_grad.py:8-40).

Forward clamps; backward lets the gradient through wherever the clamp was inactive AND wherever following the gradient would move the
value back towards [min, max] (``grad * (grad * (x - clamp(x)) >= 0)``), so an optimiser is never stuck outside the interval.  Both
directions are HIP kernels (csrc/elementwise.hip pmi_clamp, csrc/sampling.hip pmi_clamp_grad); min / max may be numbers or one value per
sample (the form Predictions.dynamic_threshold needs).
"""
from __future__ import annotations

import torch

from ..engine import sampler


class ClampWithGradFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, input, min=0, max=1):
        ctx.min, ctx.max = min, max
        ctx.save_for_backward(input)
        return sampler.clamp(input, min, max).to(input.dtype).view_as(input)

    @staticmethod
    def backward(ctx, grad_in):
        (input,) = ctx.saved_tensors
        return sampler.clamp_grad(input, grad_in, ctx.min, ctx.max).to(grad_in.dtype).view_as(grad_in), None, None


def clamp_with_grad(tensor, min=0.0, max=1.0):
    return ClampWithGradFunction.apply(tensor, min, max)


class ClampWithGrad(torch.nn.Module):
    """TransformInterface-shaped wrapper (reference clamp_with_grad.py:30-40): encode clamps, decode is the identity."""

    def __init__(self, min=0, max=1):
        super().__init__()
        self.min = min
        self.max = max

    def encode(self, tensor):
        return clamp_with_grad(tensor, self.min, self.max)

    def decode(self, tensor):
        return tensor

    def forward(self, tensor):
        return self.encode(tensor)
