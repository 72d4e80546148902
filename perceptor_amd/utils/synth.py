"""Deterministic, name-keyed synthetic weights.

Real checkpoints are unreachable offline (reference downloads them:
perceptor/models/guided_diffusion/guided_diffusion.py:25-36,
perceptor/models/velocity_diffusion/velocity_diffusion.py:28,
perceptor/models/open_clip.py:65-72), so parity and benchmarks run on weights
generated from the state-dict *names* alone.  The same function is applied to
the reference modules (when generating tests/golden/*), to the oracle and to
the HIP engine, so all three see bit-identical parameters.

Every matrix-shaped parameter is rounded to a bf16-representable value.  Such
values are exact in bf16, in fp16 (normal range) and in fp32, which removes
weight quantisation as an error source: the only difference between the fp32
oracle and the 16-bit MFMA path is activation rounding.  Layers the reference
zero-initialises (unet.py:207-209,289,607) get ordinary random values, since
a zero layer would make the parity check vacuous.
"""
from __future__ import annotations

import zlib
from typing import Dict, Mapping, Sequence

import numpy as np
import torch


def _round_bf16(x: np.ndarray) -> np.ndarray:
    u = x.astype(np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32)


def _rng(name: str, seed: int) -> np.random.Generator:
    key = zlib.crc32(name.encode()) ^ (seed * 0x9E3779B1 & 0xFFFFFFFF)
    return np.random.Generator(np.random.Philox(key=key))


def synth_tensor(name: str, shape: Sequence[int], seed: int = 0, gain: float = 1.0, rounding: str = "bf16") -> torch.Tensor:
    """rounding="bf16" (default): matrix-shaped parameters are bf16-representable; "none": full fp32 values (used with the
    reference's own fp16 cast of the torso convolutions for the weight-packing fixtures)."""
    shape = tuple(int(s) for s in shape)
    g = _rng(name, seed)
    leaf = name.rsplit(".", 1)[-1]
    n = int(np.prod(shape)) if len(shape) else 1
    z = g.standard_normal(n, dtype=np.float32).reshape(shape)
    if leaf in ("positional_embedding", "class_embedding"):
        w = z * float(shape[-1]) ** -0.5
    elif leaf == "proj" and len(shape) == 2:  # ViT output projection, used as x @ proj
        w = z * float(shape[0]) ** -0.5
    elif len(shape) >= 2:
        fan_in = int(np.prod(shape[1:]))
        w = z * (gain * float(fan_in) ** -0.5)   # gain < 1 keeps deep norm-free nets (wikiart) inside the fp16 range
    elif leaf in ("weight",):  # norm gains
        w = 1.0 + 0.1 * z
    else:  # biases and other vectors
        w = 0.05 * z
    if len(shape) >= 2 and rounding == "bf16":
        w = _round_bf16(w)
    return torch.from_numpy(np.ascontiguousarray(w, dtype=np.float32))


def synth_state_dict(shapes: Mapping[str, Sequence[int]], seed: int = 0, workers: int = 8, gain: float = 1.0,
                     rounding: str = "bf16") -> Dict[str, torch.Tensor]:
    """Each tensor has its own name-keyed stream, so generation order / threading cannot change values."""
    from concurrent.futures import ThreadPoolExecutor
    keys = list(shapes)
    with ThreadPoolExecutor(max_workers=workers) as ex:
        vals = list(ex.map(lambda k: synth_tensor(k, shapes[k], seed, gain, rounding), keys))
    return dict(zip(keys, vals))


def synth_like(state_dict: Mapping[str, torch.Tensor], seed: int = 0, gain: float = 1.0, rounding: str = "bf16") -> Dict[str, torch.Tensor]:
    """Synthetic replacement for every floating-point entry of ``state_dict``."""
    out = {}
    for k, v in state_dict.items():
        if torch.is_floating_point(v):
            out[k] = synth_tensor(k, v.shape, seed, gain, rounding).to(v.dtype)
        else:
            out[k] = v.clone()
    return out


def seeded_noise(shape: Sequence[int], seed: int) -> torch.Tensor:
    """CPU noise as the reference draws it (guided_diffusion.py:104), platform-stable."""
    g = np.random.Generator(np.random.Philox(key=seed))
    return torch.from_numpy(g.standard_normal(int(np.prod(shape)), dtype=np.float32).reshape(tuple(shape)))
