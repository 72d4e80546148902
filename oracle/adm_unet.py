"""TEST INFRASTRUCTURE (oracle) — CPU fp32 restatement of the ADM UNet forward.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this file; the product path (perceptor_amd/) never does.

Functional form: ``adm_unet_forward(sd, cfg, x, timesteps)`` consumes a state
dict with the reference's key names and mirrors
  perceptor/models/guided_diffusion/unet.py:626-654  (UNetModel.forward)
  perceptor/models/guided_diffusion/unet.py:232-252  (ResBlock._forward)
  perceptor/models/guided_diffusion/unet.py:294-300  (AttentionBlock._forward)
  perceptor/models/guided_diffusion/unet.py:332-348, 364-382 (QKV attention, both orders)
  perceptor/models/guided_diffusion/nn.py:17-19, 101-118 (GroupNorm32, timestep_embedding)
Pinned against tests/golden/adm_*.npz (outputs of the reference modules).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Tuple

import torch
import torch.nn.functional as F


@dataclass(frozen=True)
class AdmConfig:
    """Hyper-parameters as resolved by script_util.create_model (script_util.py:130-184)."""
    image_size: int
    model_channels: int
    num_res_blocks: int
    channel_mult: Tuple[float, ...]
    attention_ds: Tuple[int, ...]
    num_heads: int = 1
    num_head_channels: int = -1
    num_heads_upsample: int = -1
    use_scale_shift_norm: bool = False
    resblock_updown: bool = False
    use_new_attention_order: bool = False
    in_channels: int = 3
    out_channels: int = 6
    conv_resample: bool = True


def default_channel_mult(image_size: int) -> Tuple[float, ...]:
    # script_util.py:148-160
    return {512: (0.5, 1, 1, 2, 2, 4, 4), 256: (1, 1, 2, 2, 4, 4), 128: (1, 1, 2, 3, 4), 64: (1, 2, 3, 4)}[image_size]


def openimages_config() -> AdmConfig:
    # create_models.py:8-33
    return AdmConfig(512, 256, 2, default_channel_mult(512), tuple(512 // r for r in (32, 16, 8)),
                     num_head_channels=64, use_scale_shift_norm=True, resblock_updown=True)


def pixelart_config() -> AdmConfig:
    # create_models.py:36-62
    return AdmConfig(256, 128, 2, default_channel_mult(256), (256 // 16,), num_heads=1)


def timestep_embedding(t: torch.Tensor, dim: int, max_period: float = 10000.0) -> torch.Tensor:
    # nn.py:101-118 — cos first, then sin
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(half, dtype=torch.float32) / half)
    args = t[:, None].float() * freqs[None]
    return torch.cat([torch.cos(args), torch.sin(args)], dim=-1)


def _gn(sd, p, x, groups=32):
    return F.group_norm(x.float(), groups, sd[p + ".weight"], sd[p + ".bias"], eps=1e-5)


def _conv(sd, p, x, **kw):
    return F.conv2d(x, sd[p + ".weight"], sd.get(p + ".bias"), **kw)


def _resblock(sd, p, x, emb, cfg: AdmConfig, up=False, down=False):
    h = F.silu(_gn(sd, p + ".in_layers.0", x))
    if up:
        h = F.interpolate(h, scale_factor=2, mode="nearest")
        x = F.interpolate(x, scale_factor=2, mode="nearest")
    elif down:
        h = F.avg_pool2d(h, 2)
        x = F.avg_pool2d(x, 2)
    h = _conv(sd, p + ".in_layers.2", h, padding=1)
    e = F.linear(F.silu(emb), sd[p + ".emb_layers.1.weight"], sd[p + ".emb_layers.1.bias"])[:, :, None, None]
    if cfg.use_scale_shift_norm:
        scale, shift = e.chunk(2, dim=1)
        h = _gn(sd, p + ".out_layers.0", h) * (1 + scale) + shift
        h = F.silu(h)
    else:
        h = F.silu(_gn(sd, p + ".out_layers.0", h + e))
    h = _conv(sd, p + ".out_layers.3", h, padding=1)
    if (p + ".skip_connection.weight") in sd:
        w = sd[p + ".skip_connection.weight"]
        x = F.conv2d(x, w, sd[p + ".skip_connection.bias"], padding=w.shape[-1] // 2)
    return x + h


def _attention(sd, p, x, heads: int, new_order: bool):
    b, c, hh, ww = x.shape
    xf = x.reshape(b, c, -1)
    qkv = F.conv1d(_gn(sd, p + ".norm", xf), sd[p + ".qkv.weight"], sd[p + ".qkv.bias"])
    t = xf.shape[-1]
    ch = c // heads
    if new_order:   # unet.py:364-382 : split q,k,v then heads
        q, k, v = qkv.chunk(3, dim=1)
        q, k, v = (z.reshape(b * heads, ch, t) for z in (q, k, v))
    else:           # unet.py:332-348 : split heads then q,k,v
        q, k, v = qkv.reshape(b * heads, 3 * ch, t).split(ch, dim=1)
    s = ch ** -0.25
    w = torch.softmax(torch.einsum("bct,bcs->bts", q * s, k * s).float(), dim=-1)
    a = torch.einsum("bts,bcs->bct", w, v).reshape(b, c, t)
    a = F.conv1d(a, sd[p + ".proj_out.weight"], sd[p + ".proj_out.bias"])
    return (xf + a).reshape(b, c, hh, ww)


def block_plan(cfg: AdmConfig):
    """Enumerate (prefix, kind, params) for every layer, in execution order.

    Restates the constructor loops of unet.py:471-601 so the oracle knows which
    sub-modules exist under which state-dict prefix.
    """
    mc = cfg.model_channels

    def heads(c, upsample=False):
        if cfg.num_head_channels != -1:
            return c // cfg.num_head_channels
        if upsample and cfg.num_heads_upsample != -1:
            return cfg.num_heads_upsample
        return cfg.num_heads

    inp: List[List[tuple]] = [[("input_blocks.0.0", "conv", {})]]
    ch = int(cfg.channel_mult[0] * mc)
    chans = [ch]
    ds = 1
    for level, mult in enumerate(cfg.channel_mult):
        for _ in range(cfg.num_res_blocks):
            i = len(inp)
            layers = [(f"input_blocks.{i}.0", "res", {})]
            ch = int(mult * mc)
            if ds in cfg.attention_ds:
                layers.append((f"input_blocks.{i}.1", "attn", {"heads": heads(ch)}))
            inp.append(layers)
            chans.append(ch)
        if level != len(cfg.channel_mult) - 1:
            i = len(inp)
            if cfg.resblock_updown:
                inp.append([(f"input_blocks.{i}.0", "res", {"down": True})])
            else:
                inp.append([(f"input_blocks.{i}.0", "downsample", {})])
            chans.append(ch)
            ds *= 2
    mid = [("middle_block.0", "res", {}), ("middle_block.1", "attn", {"heads": heads(ch)}), ("middle_block.2", "res", {})]
    out: List[List[tuple]] = []
    for level, mult in list(enumerate(cfg.channel_mult))[::-1]:
        for i in range(cfg.num_res_blocks + 1):
            j = len(out)
            chans.pop()
            layers = [(f"output_blocks.{j}.0", "res", {})]
            ch = int(mc * mult)
            k = 1
            if ds in cfg.attention_ds:
                layers.append((f"output_blocks.{j}.{k}", "attn", {"heads": heads(ch, True)}))
                k += 1
            if level and i == cfg.num_res_blocks:
                if cfg.resblock_updown:
                    layers.append((f"output_blocks.{j}.{k}", "res", {"up": True}))
                else:
                    layers.append((f"output_blocks.{j}.{k}", "upsample", {}))
                ds //= 2
            out.append(layers)
    return inp, mid, out


def _run(sd, cfg, layers, h, emb):
    for p, kind, kw in layers:
        if kind == "conv":
            h = _conv(sd, p, h, padding=1)
        elif kind == "res":
            h = _resblock(sd, p, h, emb, cfg, **kw)
        elif kind == "attn":
            h = _attention(sd, p, h, kw["heads"], cfg.use_new_attention_order)
        elif kind == "downsample":   # unet.py:112-138 with conv_resample=True
            h = _conv(sd, p + ".op", h, stride=2, padding=1) if cfg.conv_resample else F.avg_pool2d(h, 2)
        elif kind == "upsample":     # unet.py:81-109
            h = F.interpolate(h, scale_factor=2, mode="nearest")
            if cfg.conv_resample:
                h = _conv(sd, p + ".conv", h, padding=1)
    return h


@torch.no_grad()
def adm_unet_forward(sd: Dict[str, torch.Tensor], cfg: AdmConfig, x: torch.Tensor, timesteps: torch.Tensor) -> torch.Tensor:
    sd = {k: v.float() for k, v in sd.items()}
    inp, mid, out = block_plan(cfg)
    emb = timestep_embedding(timesteps, cfg.model_channels)
    emb = F.linear(emb, sd["time_embed.0.weight"], sd["time_embed.0.bias"])
    emb = F.linear(F.silu(emb), sd["time_embed.2.weight"], sd["time_embed.2.bias"])
    h = x.float()
    hs = []
    for layers in inp:
        h = _run(sd, cfg, layers, h, emb)
        hs.append(h)
    h = _run(sd, cfg, mid, h, emb)
    for layers in out:
        h = torch.cat([h, hs.pop()], dim=1)
        h = _run(sd, cfg, layers, h, emb)
    h = F.silu(_gn(sd, "out.0", h))
    return _conv(sd, "out.2", h, padding=1)


def state_dict_shapes(cfg: AdmConfig) -> Dict[str, Tuple[int, ...]]:
    """Parameter names/shapes of the reference UNetModel for ``cfg``."""
    mc = cfg.model_channels
    ted = 4 * mc
    S: Dict[str, Tuple[int, ...]] = {}

    def lin(p, i, o):
        S[p + ".weight"] = (o, i); S[p + ".bias"] = (o,)

    def conv(p, i, o, k):
        S[p + ".weight"] = (o, i, k, k); S[p + ".bias"] = (o,)

    def norm(p, c):
        S[p + ".weight"] = (c,); S[p + ".bias"] = (c,)

    def res(p, cin, cout):
        norm(p + ".in_layers.0", cin); conv(p + ".in_layers.2", cin, cout, 3)
        lin(p + ".emb_layers.1", ted, 2 * cout if cfg.use_scale_shift_norm else cout)
        norm(p + ".out_layers.0", cout); conv(p + ".out_layers.3", cout, cout, 3)
        if cin != cout:
            conv(p + ".skip_connection", cin, cout, 1)

    def attn(p, c):
        norm(p + ".norm", c)
        S[p + ".qkv.weight"] = (3 * c, c, 1); S[p + ".qkv.bias"] = (3 * c,)
        S[p + ".proj_out.weight"] = (c, c, 1); S[p + ".proj_out.bias"] = (c,)

    lin("time_embed.0", mc, ted); lin("time_embed.2", ted, ted)
    inp, mid, out = block_plan(cfg)
    ch = int(cfg.channel_mult[0] * mc)
    conv("input_blocks.0.0", cfg.in_channels, ch, 3)
    chans = [ch]
    level_of = []
    for level, mult in enumerate(cfg.channel_mult):
        level_of += [mult] * cfg.num_res_blocks
        if level != len(cfg.channel_mult) - 1:
            level_of.append(None)
    for layers, mult in zip(inp[1:], level_of):
        for p, kind, kw in layers:
            if kind == "res":
                cout = ch if mult is None else int(mult * mc)
                res(p, ch, cout); ch = cout
            elif kind == "attn":
                attn(p, ch)
            elif kind == "downsample" and cfg.conv_resample:
                conv(p + ".op", ch, ch, 3)
        chans.append(ch)
    res("middle_block.0", ch, ch); attn("middle_block.1", ch); res("middle_block.2", ch, ch)
    mults = [m for m in cfg.channel_mult[::-1] for _ in range(cfg.num_res_blocks + 1)]
    for layers, mult in zip(out, mults):
        first = True
        for p, kind, kw in layers:
            if kind == "res":
                if first:
                    cin = ch + chans.pop(); cout = int(mc * mult); first = False
                else:
                    cin = cout = ch
                res(p, cin, cout); ch = cout
            elif kind == "attn":
                attn(p, ch)
            elif kind == "upsample" and cfg.conv_resample:
                conv(p + ".conv", ch, ch, 3)
    norm("out.0", ch)
    conv("out.2", int(cfg.channel_mult[0] * mc), cfg.out_channels, 3)
    return S
