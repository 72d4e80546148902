"""TEST INFRASTRUCTURE (oracle) — CPU fp32 restatement of the StableDiffusion path (BASELINE config 4).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file; the product path never does.

PINNING.  The UNet and VAE arithmetic of the reference's StableDiffusion path lives in diffusers 0.6.0 (poetry.lock:365-366;
`UNet2DConditionModel`, `AutoencoderKL`, `DDPMScheduler` used at perceptor/models/stable_diffusion/stable_diffusion.py:82-100,259-271), which is
not vendored and not installed here, and the reference's only value test for it (`test_stable_diffusion_step`, :574-658) needs downloaded
weights: parity with diffusers' OWN code is unpinned.  The network itself is pinned with reference code: the reference vendors the CompVis
latent-diffusion modules the diffusers classes are ports of (perceptor/models/latent_diffusion/ldm/modules/diffusionmodules/openaimodel.py:
UNetModel with SpatialTransformer; .../model.py: Encoder / Decoder of AutoencoderKL), and tests/golden/sd_ldm_*.npz hold THEIR outputs on the
name-keyed weights under the published diffusers <-> CompVis key correspondence (oracle/gen_golden.py: gen_sd_ldm, strict loads): this file
matches them to 3e-5 on the full 860 M-parameter SD-v1 UNet and on both VAE halves (tests/test_oracle_golden.py).  Also restated from the tree:
  stable_diffusion/attention.py:120-188   SpatialTransformer (GroupNorm eps 1e-6, 1x1 in/out, residual)
  stable_diffusion/attention.py:191-247   BasicTransformerBlock (attn1 self, attn2 cross, GEGLU feed-forward, pre-LayerNorm, residuals)
  stable_diffusion/attention.py:250-298   CrossAttention (to_q/to_k/to_v without bias, softmax(q k^T d^-1/2) v = the xformers call at :285, to_out)
  stable_diffusion/attention.py:301-348   FeedForward / GEGLU
  stable_diffusion/attention.py:23-117    AttentionBlock (VAE mid block: q/k/v linears, both scaled by d^-1/4, fp32 softmax, proj_attn, residual)
  stable_diffusion/predictions.py:51-98,243-250   denoised_latents, DDIM step, classifier_free_guidance
  stable_diffusion/stable_diffusion.py:98-114     scaled-linear beta schedule 0.00085..0.012 -> sqrt(alpha-bar), sqrt(1 - alpha-bar)
The layer wiring restates the published diffusers 0.6.0 algorithm (UNet2DConditionModel.forward: sinusoidal timesteps [cos|sin] with
flip_sin_to_cos, two-layer time MLP, ResnetBlock2D with additive time projection, CrossAttnDown/Up blocks, stride-2 conv
down-sampling with padding 1, nearest x2 + conv up-sampling, skip concat [h, skip]; AutoencoderKL.decode: post_quant_conv, mid block,
four up blocks of three ResnetBlock2D (eps 1e-6), GroupNorm+SiLU+conv out) over the diffusers state-dict key names.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, Tuple

import torch
import torch.nn.functional as F


@dataclass(frozen=True)
class SdConfig:
    in_channels: int = 4
    out_channels: int = 4
    block_out: Tuple[int, ...] = (320, 640, 1280, 1280)
    cross_attn: Tuple[bool, ...] = (True, True, True, False)      # CrossAttnDownBlock2D x3, DownBlock2D (mirrored on the way up)
    layers_per_block: int = 2
    heads: int = 8                                                  # `attention_head_dim` of the v1 configs is the head COUNT
    context_dim: int = 768
    groups: int = 32


@dataclass(frozen=True)
class VaeConfig:
    latent_channels: int = 4
    out_channels: int = 3
    block_out: Tuple[int, ...] = (128, 256, 512, 512)
    layers_per_block: int = 2
    groups: int = 32


SD_V1 = SdConfig()
SD_TINY = SdConfig(block_out=(32, 64, 64), cross_attn=(True, True, False), heads=2, context_dim=32)
SD_MID = SdConfig(block_out=(128, 256), cross_attn=(True, False), heads=4, context_dim=64, layers_per_block=1)
VAE_V1 = VaeConfig()
VAE_TINY = VaeConfig(block_out=(32, 64), layers_per_block=1)


def unet_state_dict_shapes(cfg: SdConfig) -> Dict[str, Tuple[int, ...]]:
    S: Dict[str, Tuple[int, ...]] = {}
    bo, ted = cfg.block_out, 4 * cfg.block_out[0]

    def lin(k, o, i, bias=True):
        S[k + ".weight"] = (o, i)
        if bias:
            S[k + ".bias"] = (o,)

    def conv(k, o, i, ks):
        S[k + ".weight"] = (o, i, ks, ks); S[k + ".bias"] = (o,)

    def norm(k, c):
        S[k + ".weight"] = (c,); S[k + ".bias"] = (c,)

    def resnet(k, i, o):
        norm(k + ".norm1", i); conv(k + ".conv1", o, i, 3); lin(k + ".time_emb_proj", o, ted)
        norm(k + ".norm2", o); conv(k + ".conv2", o, o, 3)
        if i != o:
            conv(k + ".conv_shortcut", o, i, 1)

    def transformer(k, c):
        norm(k + ".norm", c); conv(k + ".proj_in", c, c, 1); conv(k + ".proj_out", c, c, 1)
        b = k + ".transformer_blocks.0"
        for a, ctx in ((".attn1", c), (".attn2", cfg.context_dim)):
            lin(b + a + ".to_q", c, c, False); lin(b + a + ".to_k", c, ctx, False); lin(b + a + ".to_v", c, ctx, False)
            lin(b + a + ".to_out.0", c, c)
        for nm in (".norm1", ".norm2", ".norm3"):
            norm(b + nm, c)
        lin(b + ".ff.net.0.proj", 8 * c, c); lin(b + ".ff.net.2", c, 4 * c)

    lin("time_embedding.linear_1", ted, bo[0]); lin("time_embedding.linear_2", ted, ted)
    conv("conv_in", bo[0], cfg.in_channels, 3)
    ch = bo[0]
    for i, o in enumerate(bo):
        for j in range(cfg.layers_per_block):
            resnet(f"down_blocks.{i}.resnets.{j}", ch, o)
            ch = o
            if cfg.cross_attn[i]:
                transformer(f"down_blocks.{i}.attentions.{j}", o)
        if i != len(bo) - 1:
            conv(f"down_blocks.{i}.downsamplers.0.conv", o, o, 3)
    resnet("mid_block.resnets.0", ch, ch); transformer("mid_block.attentions.0", ch); resnet("mid_block.resnets.1", ch, ch)
    rev = list(reversed(bo))
    for i, o in enumerate(rev):
        prev = rev[max(i - 1, 0)] if i else bo[-1]
        inn = rev[min(i + 1, len(bo) - 1)]
        for j in range(cfg.layers_per_block + 1):
            skip = inn if j == cfg.layers_per_block else o
            rin = prev if j == 0 else o
            resnet(f"up_blocks.{i}.resnets.{j}", rin + skip, o)
            if list(reversed(cfg.cross_attn))[i]:
                transformer(f"up_blocks.{i}.attentions.{j}", o)
        if i != len(bo) - 1:
            conv(f"up_blocks.{i}.upsamplers.0.conv", o, o, 3)
    norm("conv_norm_out", bo[0]); conv("conv_out", cfg.out_channels, bo[0], 3)
    return S


def vae_decoder_state_dict_shapes(cfg: VaeConfig) -> Dict[str, Tuple[int, ...]]:
    S: Dict[str, Tuple[int, ...]] = {}

    def conv(k, o, i, ks):
        S[k + ".weight"] = (o, i, ks, ks); S[k + ".bias"] = (o,)

    def norm(k, c):
        S[k + ".weight"] = (c,); S[k + ".bias"] = (c,)

    def resnet(k, i, o):
        norm(k + ".norm1", i); conv(k + ".conv1", o, i, 3); norm(k + ".norm2", o); conv(k + ".conv2", o, o, 3)
        if i != o:
            conv(k + ".conv_shortcut", o, i, 1)

    top = cfg.block_out[-1]
    conv("post_quant_conv", cfg.latent_channels, cfg.latent_channels, 1)
    conv("decoder.conv_in", top, cfg.latent_channels, 3)
    resnet("decoder.mid_block.resnets.0", top, top); resnet("decoder.mid_block.resnets.1", top, top)
    a = "decoder.mid_block.attentions.0"
    norm(a + ".group_norm", top)
    for nm in ("query", "key", "value", "proj_attn"):
        S[f"{a}.{nm}.weight"] = (top, top); S[f"{a}.{nm}.bias"] = (top,)
    ch = top
    for i, o in enumerate(reversed(cfg.block_out)):
        for j in range(cfg.layers_per_block + 1):
            resnet(f"decoder.up_blocks.{i}.resnets.{j}", ch, o)
            ch = o
        if i != len(cfg.block_out) - 1:
            conv(f"decoder.up_blocks.{i}.upsamplers.0.conv", o, o, 3)
    norm("decoder.conv_norm_out", ch); conv("decoder.conv_out", cfg.out_channels, ch, 3)
    return S


def vae_encoder_state_dict_shapes(cfg: VaeConfig) -> Dict[str, Tuple[int, ...]]:
    S: Dict[str, Tuple[int, ...]] = {}

    def conv(k, o, i, ks):
        S[k + ".weight"] = (o, i, ks, ks); S[k + ".bias"] = (o,)

    def norm(k, c):
        S[k + ".weight"] = (c,); S[k + ".bias"] = (c,)

    def resnet(k, i, o):
        norm(k + ".norm1", i); conv(k + ".conv1", o, i, 3); norm(k + ".norm2", o); conv(k + ".conv2", o, o, 3)
        if i != o:
            conv(k + ".conv_shortcut", o, i, 1)

    conv("encoder.conv_in", cfg.block_out[0], cfg.out_channels, 3)
    ch = cfg.block_out[0]
    for i, o in enumerate(cfg.block_out):
        for j in range(cfg.layers_per_block):
            resnet(f"encoder.down_blocks.{i}.resnets.{j}", ch, o)
            ch = o
        if i != len(cfg.block_out) - 1:
            conv(f"encoder.down_blocks.{i}.downsamplers.0.conv", o, o, 3)
    resnet("encoder.mid_block.resnets.0", ch, ch); resnet("encoder.mid_block.resnets.1", ch, ch)
    a = "encoder.mid_block.attentions.0"
    norm(a + ".group_norm", ch)
    for nm in ("query", "key", "value", "proj_attn"):
        S[f"{a}.{nm}.weight"] = (ch, ch); S[f"{a}.{nm}.bias"] = (ch,)
    norm("encoder.conv_norm_out", ch); conv("encoder.conv_out", 2 * cfg.latent_channels, ch, 3)
    conv("quant_conv", 2 * cfg.latent_channels, 2 * cfg.latent_channels, 1)
    return S


def timestep_embedding(t, dim):
    half = dim // 2
    freqs = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half)
    a = t.float()[:, None] * freqs[None]
    return torch.cat([a.cos(), a.sin()], dim=-1)           # flip_sin_to_cos=True, freq_shift=0


def _gn(x, sd, k, groups, eps):
    return F.group_norm(x, groups, sd[k + ".weight"], sd[k + ".bias"], eps)


def _conv(x, sd, k, stride=1, pad=1):
    return F.conv2d(x, sd[k + ".weight"], sd[k + ".bias"], stride=stride, padding=pad)


def _resnet(sd, k, x, emb_silu, groups, eps):
    h = _conv(F.silu(_gn(x, sd, k + ".norm1", groups, eps)), sd, k + ".conv1")
    if emb_silu is not None:
        h = h + F.linear(emb_silu, sd[k + ".time_emb_proj.weight"], sd[k + ".time_emb_proj.bias"])[:, :, None, None]
    h = _conv(F.silu(_gn(h, sd, k + ".norm2", groups, eps)), sd, k + ".conv2")
    if k + ".conv_shortcut.weight" in sd:
        x = _conv(x, sd, k + ".conv_shortcut", pad=0)
    return x + h


def _attention(sd, k, x, ctx, heads):
    n, t, c = x.shape
    d = c // heads
    q = F.linear(x, sd[k + ".to_q.weight"]); kk = F.linear(ctx, sd[k + ".to_k.weight"]); v = F.linear(ctx, sd[k + ".to_v.weight"])
    sp = lambda z: z.reshape(n, z.shape[1], heads, d).transpose(1, 2)
    a = torch.softmax(sp(q) @ sp(kk).transpose(-1, -2) * d ** -0.5, dim=-1) @ sp(v)
    return F.linear(a.transpose(1, 2).reshape(n, t, c), sd[k + ".to_out.0.weight"], sd[k + ".to_out.0.bias"])


def _transformer(sd, k, x, context, heads, groups):
    n, c, hh, ww = x.shape
    h = _conv(_gn(x, sd, k + ".norm", groups, 1e-6), sd, k + ".proj_in", pad=0)
    h = h.permute(0, 2, 3, 1).reshape(n, hh * ww, c)
    b = k + ".transformer_blocks.0"
    ln = lambda z, nm: F.layer_norm(z, (c,), sd[b + nm + ".weight"], sd[b + nm + ".bias"], 1e-5)
    z = ln(h, ".norm1")
    h = _attention(sd, b + ".attn1", z, z, heads) + h
    h = _attention(sd, b + ".attn2", ln(h, ".norm2"), context, heads) + h
    f = F.linear(ln(h, ".norm3"), sd[b + ".ff.net.0.proj.weight"], sd[b + ".ff.net.0.proj.bias"])
    val, gate = f.chunk(2, dim=-1)
    h = F.linear(val * F.gelu(gate), sd[b + ".ff.net.2.weight"], sd[b + ".ff.net.2.bias"]) + h
    h = h.reshape(n, hh, ww, c).permute(0, 3, 1, 2)
    return _conv(h, sd, k + ".proj_out", pad=0) + x


def unet_forward(sd, cfg: SdConfig, sample, timesteps, context):
    """sample [N, in, H, W], timesteps [N], context [N, T, context_dim] -> predicted noise [N, out, H, W]."""
    bo, g = cfg.block_out, cfg.groups
    emb = timestep_embedding(timesteps, bo[0])
    emb = F.linear(F.silu(F.linear(emb, sd["time_embedding.linear_1.weight"], sd["time_embedding.linear_1.bias"])),
                   sd["time_embedding.linear_2.weight"], sd["time_embedding.linear_2.bias"])
    es = F.silu(emb)
    h = _conv(sample, sd, "conv_in")
    skips = [h]
    for i in range(len(bo)):
        for j in range(cfg.layers_per_block):
            h = _resnet(sd, f"down_blocks.{i}.resnets.{j}", h, es, g, 1e-5)
            if cfg.cross_attn[i]:
                h = _transformer(sd, f"down_blocks.{i}.attentions.{j}", h, context, cfg.heads, g)
            skips.append(h)
        if i != len(bo) - 1:
            h = _conv(h, sd, f"down_blocks.{i}.downsamplers.0.conv", stride=2)
            skips.append(h)
    h = _resnet(sd, "mid_block.resnets.0", h, es, g, 1e-5)
    h = _transformer(sd, "mid_block.attentions.0", h, context, cfg.heads, g)
    h = _resnet(sd, "mid_block.resnets.1", h, es, g, 1e-5)
    ca = list(reversed(cfg.cross_attn))
    for i in range(len(bo)):
        for j in range(cfg.layers_per_block + 1):
            h = _resnet(sd, f"up_blocks.{i}.resnets.{j}", torch.cat([h, skips.pop()], dim=1), es, g, 1e-5)
            if ca[i]:
                h = _transformer(sd, f"up_blocks.{i}.attentions.{j}", h, context, cfg.heads, g)
        if i != len(bo) - 1:
            h = _conv(F.interpolate(h, scale_factor=2.0, mode="nearest"), sd, f"up_blocks.{i}.upsamplers.0.conv")
    return _conv(F.silu(_gn(h, sd, "conv_norm_out", g, 1e-5)), sd, "conv_out")


def _vae_attention(sd, a, h, g):
    n, c, hh, ww = h.shape
    z = _gn(h, sd, a + ".group_norm", g, 1e-6).reshape(n, c, hh * ww).transpose(1, 2)
    q, k, v = (F.linear(z, sd[f"{a}.{nm}.weight"], sd[f"{a}.{nm}.bias"]) for nm in ("query", "key", "value"))
    p = torch.softmax((q * c ** -0.25) @ (k * c ** -0.25).transpose(-1, -2), dim=-1)
    z = F.linear(p @ v, sd[a + ".proj_attn.weight"], sd[a + ".proj_attn.bias"])
    return z.transpose(1, 2).reshape(n, c, hh, ww) + h


def vae_encode_moments(sd, cfg: VaeConfig, x):
    """AutoencoderKL.encode(x).latent_dist parameters: x in [-1, 1] [N, 3, H, W] -> (mean, logvar) [N, 4, H/8, W/8] each
    (mode() = mean; stable_diffusion.py:185-190 multiplies by 0.18215).  Down-sampling pads right/bottom only, then a stride-2 conv."""
    g = cfg.groups
    h = _conv(x, sd, "encoder.conv_in")
    for i in range(len(cfg.block_out)):
        for j in range(cfg.layers_per_block):
            h = _resnet(sd, f"encoder.down_blocks.{i}.resnets.{j}", h, None, g, 1e-6)
        if i != len(cfg.block_out) - 1:
            h = _conv(F.pad(h, (0, 1, 0, 1)), sd, f"encoder.down_blocks.{i}.downsamplers.0.conv", stride=2, pad=0)
    h = _resnet(sd, "encoder.mid_block.resnets.0", h, None, g, 1e-6)
    h = _vae_attention(sd, "encoder.mid_block.attentions.0", h, g)
    h = _resnet(sd, "encoder.mid_block.resnets.1", h, None, g, 1e-6)
    h = _conv(F.silu(_gn(h, sd, "encoder.conv_norm_out", g, 1e-6)), sd, "encoder.conv_out")
    return _conv(h, sd, "quant_conv", pad=0).chunk(2, dim=1)


def vae_decode(sd, cfg: VaeConfig, latents):
    """AutoencoderKL.decode(latents).sample: latents [N, 4, h, w] (already divided by 0.18215) -> x in [-1, 1], [N, 3, 8h, 8w]."""
    g = cfg.groups
    h = _conv(_conv(latents, sd, "post_quant_conv", pad=0), sd, "decoder.conv_in")
    h = _resnet(sd, "decoder.mid_block.resnets.0", h, None, g, 1e-6)
    a = "decoder.mid_block.attentions.0"
    n, c, hh, ww = h.shape
    z = _gn(h, sd, a + ".group_norm", g, 1e-6).reshape(n, c, hh * ww).transpose(1, 2)
    q, k, v = (F.linear(z, sd[f"{a}.{nm}.weight"], sd[f"{a}.{nm}.bias"]) for nm in ("query", "key", "value"))
    p = torch.softmax((q * c ** -0.25) @ (k * c ** -0.25).transpose(-1, -2), dim=-1)
    z = F.linear(p @ v, sd[a + ".proj_attn.weight"], sd[a + ".proj_attn.bias"])
    h = z.transpose(1, 2).reshape(n, c, hh, ww) + h
    h = _resnet(sd, "decoder.mid_block.resnets.1", h, None, g, 1e-6)
    for i in range(len(cfg.block_out)):
        for j in range(cfg.layers_per_block + 1):
            h = _resnet(sd, f"decoder.up_blocks.{i}.resnets.{j}", h, None, g, 1e-6)
        if i != len(cfg.block_out) - 1:
            h = _conv(F.interpolate(h, scale_factor=2.0, mode="nearest"), sd, f"decoder.up_blocks.{i}.upsamplers.0.conv")
    return _conv(F.silu(_gn(h, sd, "decoder.conv_norm_out", g, 1e-6)), sd, "decoder.conv_out")


def schedule():
    """stable_diffusion.py:98-114: DDPMScheduler(beta_start=0.00085, beta_end=0.012, 'scaled_linear') -> (alphas, sigmas) [1000]."""
    betas = torch.linspace(0.00085 ** 0.5, 0.012 ** 0.5, 1000, dtype=torch.float32) ** 2
    ac = torch.cumprod(1.0 - betas, dim=0)
    return ac.sqrt(), (1 - ac).sqrt()


def denoised_latents(x, eps, alpha, sigma):                   # predictions.py:51-54
    return (x - sigma * eps) / alpha.clamp(min=1e-7)


def ddim_step(x, eps, alpha, sigma, to_alpha, to_sigma):     # predictions.py:94-96 (eta = 0)
    return denoised_latents(x, eps, alpha, sigma) * to_alpha + eps * to_sigma


def classifier_free_guidance(eps_uncond, eps_pos, scale=7.0):   # predictions.py:243-250
    return eps_uncond + (eps_pos - eps_uncond) * scale
