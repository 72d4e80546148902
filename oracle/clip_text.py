"""TEST INFRASTRUCTURE (oracle) — CPU fp32 restatement of the CLIP text tower.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file; the product path never does.

The arithmetic lives in third-party packages absent here (open-clip-torch 2.0.2, poetry.lock:1321-1322, reached through
perceptor/models/open_clip.py:99-107; transformers' CLIPTextModel reached through
perceptor/models/stable_diffusion/stable_diffusion.py:295-323).  Both follow the OpenAI-CLIP text transformer, of which the
reference holds an in-tree copy; this file restates
  ruclip/model.py:204-228  (CLIP.encode_text: token + positional embedding, transformer, ln_final, EOT row @ text_projection)
  ruclip/model.py:181-185  (causal mask: -inf above the diagonal)
  ruclip/model.py:27-58    (ResidualAttentionBlock), :20-23 (QuickGELU); laion weights use exact GELU (models/clip.py:21-27)
Pinned against tests/golden/clip_text_*.npz: the reference's ruclip CLIP.encode_text (QuickGELU) and, independently,
transformers 5.15 CLIPTextModelWithProjection(config) on the same name-keyed weights (exact GELU and QuickGELU).
"""
from __future__ import annotations

from typing import Dict, Tuple

import torch
import torch.nn.functional as F

TEXT_CONFIGS = {
    # name: (context, vocab, width, layers, heads, out_dim)
    "ViT-B-32": (77, 49408, 512, 12, 8, 512),
    "ViT-L-14": (77, 49408, 768, 12, 12, 768),
    "ViT-H-14": (77, 49408, 1024, 24, 16, 1024),
    "tiny": (16, 96, 64, 2, 1, 32),
    "tiny-wide": (24, 200, 128, 3, 2, 48),
}


def text_state_dict_shapes(cfg) -> Dict[str, Tuple[int, ...]]:
    ctx, vocab, width, layers, heads, out = cfg
    S = {"token_embedding.weight": (vocab, width), "positional_embedding": (ctx, width),
         "ln_final.weight": (width,), "ln_final.bias": (width,), "text_projection": (width, out)}
    for i in range(layers):
        p = f"transformer.resblocks.{i}."
        S[p + "attn.in_proj_weight"] = (3 * width, width); S[p + "attn.in_proj_bias"] = (3 * width,)
        S[p + "attn.out_proj.weight"] = (width, width); S[p + "attn.out_proj.bias"] = (width,)
        S[p + "ln_1.weight"] = (width,); S[p + "ln_1.bias"] = (width,)
        S[p + "mlp.c_fc.weight"] = (4 * width, width); S[p + "mlp.c_fc.bias"] = (4 * width,)
        S[p + "mlp.c_proj.weight"] = (width, 4 * width); S[p + "mlp.c_proj.bias"] = (width,)
        S[p + "ln_2.weight"] = (width,); S[p + "ln_2.bias"] = (width,)
    return S


def text_forward(sd, cfg, ids: torch.Tensor, quick_gelu: bool):
    """ids [N, T] int64 -> (hidden [N, T, width] after ln_final, pooled [N, out] = hidden[n, argmax ids[n]] @ text_projection)."""
    ctx, vocab, width, layers, heads, out = cfg
    n, t = ids.shape
    d = width // heads
    x = sd["token_embedding.weight"][ids] + sd["positional_embedding"][:t]
    mask = torch.full((t, t), float("-inf")).triu_(1)
    for i in range(layers):
        p = f"transformer.resblocks.{i}."
        h = F.layer_norm(x, (width,), sd[p + "ln_1.weight"], sd[p + "ln_1.bias"], 1e-5)
        qkv = F.linear(h, sd[p + "attn.in_proj_weight"], sd[p + "attn.in_proj_bias"])
        q, k, v = (z.reshape(n, t, heads, d).transpose(1, 2) for z in qkv.chunk(3, dim=-1))
        a = torch.softmax((q * d ** -0.5) @ k.transpose(-1, -2) + mask, dim=-1) @ v
        x = x + F.linear(a.transpose(1, 2).reshape(n, t, width), sd[p + "attn.out_proj.weight"], sd[p + "attn.out_proj.bias"])
        h = F.layer_norm(x, (width,), sd[p + "ln_2.weight"], sd[p + "ln_2.bias"], 1e-5)
        h = F.linear(h, sd[p + "mlp.c_fc.weight"], sd[p + "mlp.c_fc.bias"])
        h = h * torch.sigmoid(1.702 * h) if quick_gelu else F.gelu(h)
        x = x + F.linear(h, sd[p + "mlp.c_proj.weight"], sd[p + "mlp.c_proj.bias"])
    hidden = F.layer_norm(x, (width,), sd["ln_final.weight"], sd["ln_final.bias"], 1e-5)
    pooled = hidden[torch.arange(n), ids.argmax(dim=-1)] @ sd["text_projection"]
    return hidden, pooled
