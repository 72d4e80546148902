"""Reference-named frozen weights as a real nn.Module tree.

The reference wrappers hold their network as `self.model` (an nn.Module whose state-dict keys are
`input_blocks.0.0.weight`, `net.3.main.0.weight`, `visual.conv1.weight`, ...: SURVEY.md §8b); `.to()`, `.parameters()`,
`.state_dict()` and `.load_state_dict()` of the wrapper cover those tensors.  ParamTree rebuilds that module nesting from the
key names alone (containers are empty nn.Modules, leaves are `requires_grad=False` Parameters), so the standard nn.Module
machinery gives the same keys -- `torch.save(model.state_dict())` round-trips and a reference checkpoint loads as is.
The HIP engines pack their own 16-bit copies from this tree; the wrappers rebuild them after `.to()` / `load_state_dict()`.
"""
from __future__ import annotations

from typing import Dict

import torch


class ParamTree(torch.nn.Module):
    def __init__(self, sd: Dict[str, torch.Tensor]):
        super().__init__()
        for key, v in sd.items():
            node = self
            *path, leaf = key.split(".")
            for part in path:
                if part not in node._modules:
                    node.add_module(part, torch.nn.Module())
                node = node._modules[part]
            node.register_parameter(leaf, torch.nn.Parameter(v.detach().clone(), requires_grad=False))

    def forward(self, *a, **k):
        raise RuntimeError("the network runs in the HIP engine of the wrapper (perceptor_amd.engine), not in this parameter holder")
