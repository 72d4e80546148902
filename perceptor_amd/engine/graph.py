"""HIP-graph capture of a denoising step.

A guided step is ~1000-2000 kernel launches issued from Python.  At the benchmark's headline size (512x512, batch 8) the GPU is
the bottleneck and eager launching keeps it fed; at small sizes (batch 1, 256x256) the host is, and replaying a captured graph
removes the launch cost (cc12m_1 256x256 batch 1: 9.5 -> 7.6 ms per step).  Nothing on the step path synchronises with the host
(timesteps travel as device tensors, scratch comes from the stream-ordered caching allocator), so the whole step -- UNet,
CLIP forward + input gradient, guidance, DDIM update -- is capturable as is.

    step = GraphedStep(lambda images, t_from, t_to: one_step(images, t_from, t_to), images, t_from, t_to)
    for t_from, t_to in schedule:
        images = step(images, t_from, t_to)        # copies the three inputs into static buffers and replays

The returned tensor is the graph's static output buffer: clone it if it must survive the next call.
"""
from __future__ import annotations

from typing import Callable

import torch


class GraphedStep:
    def __init__(self, fn: Callable[..., torch.Tensor], *example_inputs: torch.Tensor, warmup: int = 2):
        if not all(isinstance(t, torch.Tensor) and t.is_cuda for t in example_inputs):
            raise RuntimeError("GraphedStep needs HIP tensors as example inputs (perceptor_amd has no CPU fallback)")
        self._static_in = [t.clone() for t in example_inputs]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):          # warm-up off the capture: library handles, autotuned GEMM plans, allocator pools
            for _ in range(warmup):
                fn(*self._static_in)
        torch.cuda.current_stream().wait_stream(side)
        self._graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self._graph):
            self._static_out = fn(*self._static_in)

    def __call__(self, *inputs: torch.Tensor) -> torch.Tensor:
        if len(inputs) != len(self._static_in):
            raise ValueError(f"expected {len(self._static_in)} inputs, got {len(inputs)}")
        for dst, src in zip(self._static_in, inputs):
            if dst.shape != src.shape:
                raise ValueError(f"input shape {tuple(src.shape)} differs from the captured {tuple(dst.shape)}")
            dst.copy_(src)
        self._graph.replay()
        return self._static_out
