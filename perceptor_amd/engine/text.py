"""HIP engine for the CLIP text tower (forward only: prompts are encoded once per run, no gradient flows to them).

Replaces ``open_clip``'s ``model.encode_text`` behind ``models.OpenCLIP.encode_texts`` (perceptor/models/open_clip.py:99-107)
and ``transformers.CLIPTextModel`` behind ``StableDiffusion.text_encodings``
(perceptor/models/stable_diffusion/stable_diffusion.py:295-323).  The arithmetic is the OpenAI-CLIP text transformer, of which
the reference holds an in-tree copy: perceptor/models/ruclip/model.py:164-228 (token + positional embedding, causal
ResidualAttentionBlocks, ln_final, the EOT token's row @ text_projection).

State-dict keys follow open_clip / OpenAI-CLIP: token_embedding.weight, positional_embedding,
transformer.resblocks.{i}.{ln_1,attn.in_proj_*,attn.out_proj,ln_2,mlp.c_fc,mlp.c_proj}, ln_final, text_projection.
The residual stream is fp32 in HBM; GEMM operands are 16-bit on the MFMA (fp32 accumulation); LayerNorm and softmax in fp32.
"""
from __future__ import annotations

from typing import Dict, Tuple

import torch

from .. import _hip
from .._hip import ACT_GELU, ACT_QUICKGELU, call, ptr
from . import ops
from .ops import PackedLinear

TEXT_CONFIGS = {
    # name: (context, vocab, width, layers, heads, out_dim)   (open_clip model configs / OpenAI CLIP)
    "ViT-B-32": (77, 49408, 512, 12, 8, 512),
    "ViT-B-16": (77, 49408, 512, 12, 8, 512),
    "ViT-L-14": (77, 49408, 768, 12, 12, 768),
    "ViT-H-14": (77, 49408, 1024, 24, 16, 1024),
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


def from_hf_text_state_dict(hf: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """transformers' CLIPTextModel(WithProjection) state dict (``text_model.encoder.layers.{i}.self_attn.q_proj...``, the form the
    reference loads at stable_diffusion.py:299-301 and transformers_openai_clip.py:90) -> the OpenAI-CLIP names this engine packs."""
    g = lambda k: hf[k].detach().float()
    out = {"token_embedding.weight": g("text_model.embeddings.token_embedding.weight"),
           "positional_embedding": g("text_model.embeddings.position_embedding.weight"),
           "ln_final.weight": g("text_model.final_layer_norm.weight"), "ln_final.bias": g("text_model.final_layer_norm.bias")}
    if "text_projection.weight" in hf:
        out["text_projection"] = g("text_projection.weight").t().contiguous()
    i = 0
    while f"text_model.encoder.layers.{i}.layer_norm1.weight" in hf:
        a, b = f"text_model.encoder.layers.{i}.", f"transformer.resblocks.{i}."
        out[b + "attn.in_proj_weight"] = torch.cat([g(a + f"self_attn.{n}_proj.weight") for n in "qkv"], dim=0)
        out[b + "attn.in_proj_bias"] = torch.cat([g(a + f"self_attn.{n}_proj.bias") for n in "qkv"], dim=0)
        for src, dst in (("self_attn.out_proj", "attn.out_proj"), ("layer_norm1", "ln_1"), ("layer_norm2", "ln_2"),
                         ("mlp.fc1", "mlp.c_fc"), ("mlp.fc2", "mlp.c_proj")):
            out[b + dst + ".weight"], out[b + dst + ".bias"] = g(a + src + ".weight"), g(a + src + ".bias")
        i += 1
    return out


def hf_text_state_dict_shapes(cfg, projection: bool = True) -> Dict[str, Tuple[int, ...]]:
    """The text tower under transformers' key names (CLIPTextModel[WithProjection])."""
    ctx, vocab, width, layers, heads, out = cfg
    S = {"text_model.embeddings.token_embedding.weight": (vocab, width), "text_model.embeddings.position_embedding.weight": (ctx, width),
         "text_model.final_layer_norm.weight": (width,), "text_model.final_layer_norm.bias": (width,)}
    if projection:
        S["text_projection.weight"] = (out, width)
    for i in range(layers):
        a = f"text_model.encoder.layers.{i}."
        for n in "qkvo":
            nm = "out_proj" if n == "o" else f"{n}_proj"
            S[a + f"self_attn.{nm}.weight"] = (width, width); S[a + f"self_attn.{nm}.bias"] = (width,)
        for nm in ("layer_norm1", "layer_norm2"):
            S[a + nm + ".weight"] = (width,); S[a + nm + ".bias"] = (width,)
        S[a + "mlp.fc1.weight"] = (4 * width, width); S[a + "mlp.fc1.bias"] = (4 * width,)
        S[a + "mlp.fc2.weight"] = (width, 4 * width); S[a + "mlp.fc2.bias"] = (width,)
    return S


class TextEngine:
    def __init__(self, cfg, state_dict, device, dtype="bf16", quick_gelu=True):
        self.cfg, self.device = cfg, torch.device(device)
        self.dt = _hip.dtype_code(dtype)
        if self.dt not in (_hip.DT_BF16, _hip.DT_F16):
            raise ValueError("the text tower runs in 'bf16' or 'f16'")
        self.act = ACT_QUICKGELU if quick_gelu else ACT_GELU
        _hip.lib()
        ctx, vocab, width, layers, heads, out = cfg
        sd, dev, dt = state_dict, self.device, self.dt
        f32 = lambda k: sd[k].detach().float().to(dev).contiguous()
        self.tok, self.pos = f32("token_embedding.weight"), f32("positional_embedding")
        self.ln_final = (f32("ln_final.weight"), f32("ln_final.bias"))
        self.proj = PackedLinear(sd["text_projection"].detach().float().t().contiguous(), None, dt, dev) if "text_projection" in sd else None
        self.blocks = []
        for i in range(layers):
            p = f"transformer.resblocks.{i}."
            lin = lambda k: PackedLinear(sd[p + k + ("_weight" if k.endswith("in_proj") else ".weight")].float(),
                                         sd[p + k + ("_bias" if k.endswith("in_proj") else ".bias")], dt, dev)
            self.blocks.append(dict(ln1=(f32(p + "ln_1.weight"), f32(p + "ln_1.bias")), ln2=(f32(p + "ln_2.weight"), f32(p + "ln_2.bias")),
                                    qkv=lin("attn.in_proj"), out=lin("attn.out_proj"), fc=lin("mlp.c_fc"), pr=lin("mlp.c_proj")))

    def _ln(self, x, gb, m, d, want16=True, want32=False):
        y16 = torch.empty((m, d), dtype=_hip.TORCH_DTYPE[self.dt], device=x.device) if want16 else None
        y32 = torch.empty((m, d), dtype=torch.float32, device=x.device) if want32 else None
        call("pmi_layernorm_fwd", ptr(x), d, ptr(gb[0]), ptr(gb[1]), ptr(y16), ptr(y32), None, m, d, 1e-5, self.dt)
        return y16, y32

    @torch.no_grad()
    def forward(self, ids: torch.Tensor):
        """ids [N, T] int64 (T <= context) -> (hidden [N, T, width] fp32 after ln_final, pooled [N, out_dim] fp32 or None).

        pooled = hidden[n, argmax(ids[n])] @ text_projection: the EOT token has the largest id of the CLIP vocabulary
        (OpenAI-CLIP ``x[arange, text.argmax(-1)]``; ruclip/model.py:224-227 selects the same row by eos_id)."""
        ctx, vocab, width, layers, heads, out = self.cfg
        if ids.ndim != 2 or ids.dtype != torch.int64:
            raise ValueError("ids must be an int64 tensor of shape [N, T]")
        n, t = ids.shape
        if t > ctx:
            raise ValueError(f"sequence length {t} exceeds the context length {ctx}")
        if int(ids.min()) < 0 or int(ids.max()) >= vocab:
            raise ValueError("token id out of range")
        dev, dt = self.device, self.dt
        ids = ids.to(dev).contiguous()
        m = n * t
        x = torch.empty((n, t, width), dtype=torch.float32, device=dev)
        call("pmi_embed_tokens", ptr(ids), ptr(self.tok), ptr(self.pos), ptr(x), n, t, width, vocab)
        x = x.view(m, width)
        for blk in self.blocks:
            h, _ = self._ln(x, blk["ln1"], m, width)
            qkv = ops.igemm(h, blk["qkv"])                                          # [m, 3*width], (q|k|v) x (head, d)
            a = ops.attention(qkv.view(n, t, 3 * width), heads, 1, dt, causal=True).view(m, width)
            x_mid = ops.igemm(a, blk["out"], residual=x, out_f32=True)
            h2, _ = self._ln(x_mid, blk["ln2"], m, width)
            hact = ops.igemm(h2, blk["fc"], act=self.act)
            x = ops.igemm(hact, blk["pr"], residual=x_mid, out_f32=True)
        _, hidden = self._ln(x, self.ln_final, m, width, want16=False, want32=True)
        pooled = None
        if self.proj is not None:
            rows = ids.argmax(dim=1) + torch.arange(n, device=dev) * t
            eot = torch.empty((n, width), dtype=torch.float32, device=dev)
            call("pmi_gather_rows", ptr(hidden), ptr(rows), ptr(eot), n, width, width, m)
            e16 = torch.empty((n, width), dtype=_hip.TORCH_DTYPE[dt], device=dev)
            call("pmi_cast_f32_to_16", ptr(eot), ptr(e16), n * width, _hip.ACT_NONE, dt)
            pooled = ops.igemm(e16, self.proj, out_f32=True)
            pooled = pooled[:, :out] if pooled.shape[1] != out else pooled
        return hidden.view(n, t, width), pooled
