"""HIP execution engines for the StableDiffusion path: the latent UNet (eps-prediction with text cross-attention) and the VAE decoder.

Replaces ``diffusers.UNet2DConditionModel`` / ``AutoencoderKL.decode`` as called at
perceptor/models/stable_diffusion/stable_diffusion.py:195-198,259-271 (diffusers 0.6.0, poetry.lock:365-366), with the transformer
blocks of perceptor/models/stable_diffusion/attention.py:120-348 (SpatialTransformer, BasicTransformerBlock, CrossAttention incl. the
fused-attention call at :285, FeedForward/GEGLU) and the VAE's AttentionBlock (:23-117).  NHWC 16-bit activations through
libperceptor_hip.so:

  * ResnetBlock2D = conv -> conv with GroupNorm-apply+SiLU fused into each conv's patch staging, statistics out of the producer's
    epilogue, the additive time projection as a per-sample bias, shortcut / residual add in the second conv's epilogue; skip
    concatenations are two source pointers; nearest-x2 up-sampling is fused into the following conv's gather;
  * all ``time_emb_proj`` linears run as ONE GEMM per step;
  * transformer blocks keep an fp32 residual stream ([tokens][C]); q|k|v (self) and k|v (cross) projections are one GEMM each;
    GEGLU is one pass over the 8C-wide projection;
  * the context's k|v projections of all cross-attention layers depend only on the prompt: computed once per conditioning and cached.

State-dict keys are diffusers' (time_embedding.linear_1, down_blocks.{i}.resnets.{j}.conv1, ...attentions.{j}.transformer_blocks.0.attn2.to_k, ...),
so a real checkpoint's tensors load as they are.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import torch

from .. import _hip
from .._hip import ACT_NONE, ACT_SILU, call, ptr
from . import ops
from .ops import PackedLinear


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
SD_INPAINTING = SdConfig(in_channels=9)          # runwayml/stable-diffusion-inpainting: latents | mask | masked-image latents
VAE_V1 = VaeConfig()


class _Shapes:
    def __init__(self):
        self.S: Dict[str, Tuple[int, ...]] = {}

    def lin(self, k, o, i, bias=True):
        self.S[k + ".weight"] = (o, i)
        if bias:
            self.S[k + ".bias"] = (o,)

    def conv(self, k, o, i, ks):
        self.S[k + ".weight"] = (o, i, ks, ks); self.S[k + ".bias"] = (o,)

    def norm(self, k, c):
        self.S[k + ".weight"] = (c,); self.S[k + ".bias"] = (c,)

    def resnet(self, k, i, o, ted=None):
        self.norm(k + ".norm1", i); self.conv(k + ".conv1", o, i, 3)
        if ted:
            self.lin(k + ".time_emb_proj", o, ted)
        self.norm(k + ".norm2", o); self.conv(k + ".conv2", o, o, 3)
        if i != o:
            self.conv(k + ".conv_shortcut", o, i, 1)


def unet_plan(cfg: SdConfig):
    """Blocks in execution order: ("res", key, cin, cout, srcs) / ("attn", key, c) / ("down"|"up", key, c)."""
    bo = cfg.block_out
    down: List[list] = []
    ch = bo[0]
    skip_ch = [ch]
    for i, o in enumerate(bo):
        for j in range(cfg.layers_per_block):
            blk = [("res", f"down_blocks.{i}.resnets.{j}", ch, o, None)]
            ch = o
            if cfg.cross_attn[i]:
                blk.append(("attn", f"down_blocks.{i}.attentions.{j}", o))
            down.append(blk)
            skip_ch.append(ch)
        if i != len(bo) - 1:
            down.append([("down", f"down_blocks.{i}.downsamplers.0.conv", o)])
            skip_ch.append(ch)
    mid = [("res", "mid_block.resnets.0", ch, ch, None), ("attn", "mid_block.attentions.0", ch), ("res", "mid_block.resnets.1", ch, ch, None)]
    up: List[list] = []
    ca = list(reversed(cfg.cross_attn))
    for i, o in enumerate(reversed(bo)):
        for j in range(cfg.layers_per_block + 1):
            s = skip_ch.pop()
            blk = [("res", f"up_blocks.{i}.resnets.{j}", ch + s, o, (ch, s))]
            ch = o
            if ca[i]:
                blk.append(("attn", f"up_blocks.{i}.attentions.{j}", o))
            if j == cfg.layers_per_block and i != len(bo) - 1:
                blk.append(("up", f"up_blocks.{i}.upsamplers.0.conv", o))
            up.append(blk)
    return down, mid, up


def unet_gflop(cfg: SdConfig, h: int, w: int, tc: int = 77) -> float:
    """Algorithmic GFLOP of ONE UNet evaluation of ONE sample at h x w latents (2 per multiply-add; the prompt's k|v projections are
    per-conditioning work and not counted): convolutions, linears and the attention products."""
    fl = 0.0
    down, mid, up = unet_plan(cfg)
    hh, ww = h, w
    fl += 2.0 * hh * ww * cfg.in_channels * cfg.block_out[0] * 9 + 2.0 * hh * ww * cfg.block_out[0] * cfg.out_channels * 9
    ted = 4 * cfg.block_out[0]
    fl += 2.0 * (cfg.block_out[0] * ted + ted * ted)
    for blk in down + [mid] + up:
        for l in blk:
            if l[0] == "res":
                _, _, ci, co, _ = l
                fl += 2.0 * hh * ww * (ci * co * 9 + co * co * 9 + (ci * co if ci != co else 0)) + 2.0 * ted * co
            elif l[0] == "attn":
                c, t = l[2], hh * ww
                fl += 2.0 * t * c * c * (2 + 3 + 1 + 1 + 1 + 8 + 4) + 4.0 * t * t * c + 4.0 * t * tc * c
            elif l[0] == "down":
                hh, ww = hh // 2, ww // 2
                fl += 2.0 * hh * ww * l[2] * l[2] * 9
            else:
                hh, ww = hh * 2, ww * 2
                fl += 2.0 * hh * ww * l[2] * l[2] * 9
    return fl / 1e9


def unet_state_dict_shapes(cfg: SdConfig) -> Dict[str, Tuple[int, ...]]:
    sh, ted = _Shapes(), 4 * cfg.block_out[0]
    sh.lin("time_embedding.linear_1", ted, cfg.block_out[0]); sh.lin("time_embedding.linear_2", ted, ted)
    sh.conv("conv_in", cfg.block_out[0], cfg.in_channels, 3)
    down, mid, up = unet_plan(cfg)
    for blk in down + [mid] + up:
        for l in blk:
            if l[0] == "res":
                sh.resnet(l[1], l[2], l[3], ted)
            elif l[0] == "attn":
                k, c = l[1], l[2]
                sh.norm(k + ".norm", c); sh.conv(k + ".proj_in", c, c, 1); sh.conv(k + ".proj_out", c, c, 1)
                b = k + ".transformer_blocks.0"
                for a, ctx in ((".attn1", c), (".attn2", cfg.context_dim)):
                    sh.lin(b + a + ".to_q", c, c, False); sh.lin(b + a + ".to_k", c, ctx, False); sh.lin(b + a + ".to_v", c, ctx, False)
                    sh.lin(b + a + ".to_out.0", c, c)
                for nm in (".norm1", ".norm2", ".norm3"):
                    sh.norm(b + nm, c)
                sh.lin(b + ".ff.net.0.proj", 8 * c, c); sh.lin(b + ".ff.net.2", c, 4 * c)
            else:
                sh.conv(l[1], l[2], l[2], 3)
    sh.norm("conv_norm_out", cfg.block_out[0]); sh.conv("conv_out", cfg.out_channels, cfg.block_out[0], 3)
    return sh.S


def vae_decoder_state_dict_shapes(cfg: VaeConfig) -> Dict[str, Tuple[int, ...]]:
    sh, top = _Shapes(), cfg.block_out[-1]
    sh.conv("post_quant_conv", cfg.latent_channels, cfg.latent_channels, 1)
    sh.conv("decoder.conv_in", top, cfg.latent_channels, 3)
    sh.resnet("decoder.mid_block.resnets.0", top, top); sh.resnet("decoder.mid_block.resnets.1", top, top)
    a = "decoder.mid_block.attentions.0"
    sh.norm(a + ".group_norm", top)
    for nm in ("query", "key", "value", "proj_attn"):
        sh.lin(f"{a}.{nm}", top, top)
    ch = top
    for i, o in enumerate(reversed(cfg.block_out)):
        for j in range(cfg.layers_per_block + 1):
            sh.resnet(f"decoder.up_blocks.{i}.resnets.{j}", ch, o)
            ch = o
        if i != len(cfg.block_out) - 1:
            sh.conv(f"decoder.up_blocks.{i}.upsamplers.0.conv", o, o, 3)
    sh.norm("decoder.conv_norm_out", ch); sh.conv("decoder.conv_out", cfg.out_channels, ch, 3)
    return sh.S


def vae_encoder_state_dict_shapes(cfg: VaeConfig) -> Dict[str, Tuple[int, ...]]:
    sh = _Shapes()
    sh.conv("encoder.conv_in", cfg.block_out[0], cfg.out_channels, 3)
    ch = cfg.block_out[0]
    for i, o in enumerate(cfg.block_out):
        for j in range(cfg.layers_per_block):
            sh.resnet(f"encoder.down_blocks.{i}.resnets.{j}", ch, o)
            ch = o
        if i != len(cfg.block_out) - 1:
            sh.conv(f"encoder.down_blocks.{i}.downsamplers.0.conv", o, o, 3)
    sh.resnet("encoder.mid_block.resnets.0", ch, ch); sh.resnet("encoder.mid_block.resnets.1", ch, ch)
    a = "encoder.mid_block.attentions.0"
    sh.norm(a + ".group_norm", ch)
    for nm in ("query", "key", "value", "proj_attn"):
        sh.lin(f"{a}.{nm}", ch, ch)
    sh.norm("encoder.conv_norm_out", ch); sh.conv("encoder.conv_out", 2 * cfg.latent_channels, ch, 3)
    sh.conv("quant_conv", 2 * cfg.latent_channels, 2 * cfg.latent_channels, 1)
    return sh.S


class _Blocks:
    """What the UNet and the VAE decoder share: packed weights by key and the ResnetBlock2D launch sequence."""

    def _init_common(self, state_dict, device, dtype):
        self.device = torch.device(device)
        self.dt = _hip.dtype_code(dtype)
        if self.dt not in (_hip.DT_BF16, _hip.DT_F16):
            raise ValueError("the StableDiffusion engines run in 'bf16' or 'f16'")
        _hip.lib()
        self.sd = state_dict
        self.w: Dict[str, object] = {}

    def _f32(self, k):
        return self.sd[k].detach().float().to(self.device).contiguous()

    def _lin(self, k, **kw):
        return PackedLinear(self.sd[k + ".weight"], self.sd.get(k + ".bias"), self.dt, self.device, **kw)

    def _pack_resnet(self, k, srcs=None):
        w = self.w
        w[k + ".gn1"] = (self._f32(k + ".norm1.weight"), self._f32(k + ".norm1.bias"))
        w[k + ".conv1"] = self._lin(k + ".conv1", sources=srcs)
        w[k + ".gn2"] = (self._f32(k + ".norm2.weight"), self._f32(k + ".norm2.bias"))
        w[k + ".conv2"] = self._lin(k + ".conv2")
        if k + ".conv_shortcut.weight" in self.sd:
            w[k + ".skip"] = self._lin(k + ".conv_shortcut", sources=srcs)

    def _resnet(self, k, x, x1, nbias, groups, eps):
        dt, w = self.dt, self.w
        ca, cb = ops.group_norm_coeffs(x, *w[k + ".gn1"], groups, dt, x1=x1, eps=eps)
        h = ops.igemm(x, w[k + ".conv1"], a1=x1, nbias=nbias, prologue=(ca, cb, ACT_SILU), want_stats=True)
        ca, cb = ops.group_norm_coeffs(h, *w[k + ".gn2"], groups, dt, eps=eps)
        skip = ops.igemm(x, w[k + ".skip"], a1=x1) if (k + ".skip") in w else x
        assert (k + ".skip") in w or x1 is None
        return ops.igemm(h, w[k + ".conv2"], residual=skip, prologue=(ca, cb, ACT_SILU), want_stats=True)


    def _pack_vae_attention(self, a):
        sd = self.sd
        self.w[a + ".gn"] = (self._f32(a + ".group_norm.weight"), self._f32(a + ".group_norm.bias"))
        self.w[a + ".qkv"] = PackedLinear(torch.cat([sd[f"{a}.{nm}.weight"].float() for nm in ("query", "key", "value")], 0),
                                          torch.cat([sd[f"{a}.{nm}.bias"].float() for nm in ("query", "key", "value")], 0), self.dt, self.device)
        self.w[a + ".proj"] = self._lin(a + ".proj_attn")

    def _vae_attention(self, a, h, groups):
        """AttentionBlock (stable_diffusion/attention.py:71-117): GroupNorm -> q|k|v -> one head over all pixels -> proj + residual."""
        dt, w = self.dt, self.w
        n, h2, w2, cc = h.shape
        m = n * h2 * w2
        hn = ops.group_norm(h, *w[a + ".gn"], groups, dt, eps=1e-6)
        qkv = ops.igemm(hn.view(m, cc), w[a + ".qkv"])
        at = ops.attention(qkv.view(n, h2 * w2, 3 * cc), 1, 1, dt)
        o = ops.igemm(at.view(m, cc), w[a + ".proj"], residual=h.view(m, cc), want_stats=True, hw=h2 * w2)
        h4 = o.view(n, h2, w2, cc)
        if hasattr(o, "_pmi_stats"):
            h4._pmi_stats = o._pmi_stats
        return h4


class SdUnetEngine(_Blocks):
    def __init__(self, cfg: SdConfig, state_dict: Dict[str, torch.Tensor], device, dtype="f16"):
        self._init_common(state_dict, device, dtype)
        self.cfg = cfg
        sd = state_dict
        self.down, self.mid, self.up = unet_plan(cfg)
        self.te1, self.te2 = self._lin("time_embedding.linear_1"), self._lin("time_embedding.linear_2")
        self.conv_in = self._lin("conv_in", cin_pad=(cfg.in_channels + 7) // 8 * 8)
        emb_w, emb_b, off = [], [], 0
        self.emb_off: Dict[str, Tuple[int, int]] = {}
        cat = lambda keys: torch.cat([sd[k].detach().float() for k in keys], dim=0)
        for blk in self.down + [self.mid] + self.up:
            for l in blk:
                if l[0] == "res":
                    k = l[1]
                    self._pack_resnet(k, l[4])
                    emb_w.append(sd[k + ".time_emb_proj.weight"].float()); emb_b.append(sd[k + ".time_emb_proj.bias"].float())
                    self.emb_off[k] = (off, l[3])
                    off += l[3]
                elif l[0] == "attn":
                    k, b, w = l[1], l[1] + ".transformer_blocks.0", self.w
                    w[k + ".gn"] = (self._f32(k + ".norm.weight"), self._f32(k + ".norm.bias"))
                    w[k + ".proj_in"], w[k + ".proj_out"] = self._lin(k + ".proj_in"), self._lin(k + ".proj_out")
                    for nm in ("norm1", "norm2", "norm3"):
                        w[f"{b}.{nm}"] = (self._f32(f"{b}.{nm}.weight"), self._f32(f"{b}.{nm}.bias"))
                    w[b + ".qkv1"] = PackedLinear(cat([b + ".attn1.to_q.weight", b + ".attn1.to_k.weight", b + ".attn1.to_v.weight"]), None, self.dt, self.device)
                    w[b + ".out1"] = self._lin(b + ".attn1.to_out.0")
                    w[b + ".q2"] = self._lin(b + ".attn2.to_q")
                    w[b + ".kv2"] = PackedLinear(cat([b + ".attn2.to_k.weight", b + ".attn2.to_v.weight"]), None, self.dt, self.device)
                    w[b + ".out2"] = self._lin(b + ".attn2.to_out.0")
                    wf, bf = ops.interleave_geglu(sd[b + ".ff.net.0.proj.weight"].detach().float(), sd[b + ".ff.net.0.proj.bias"].detach().float())
                    w[b + ".ff1"] = PackedLinear(wf, bf, self.dt, self.device)      # (16 value | 16 gate) column groups: GEGLU in the GEMM's epilogue
                    w[b + ".ff2"] = self._lin(b + ".ff.net.2")
                else:
                    self.w[l[1]] = self._lin(l[1])
        self.emb_all = PackedLinear(torch.cat(emb_w, 0), torch.cat(emb_b, 0), self.dt, self.device)
        self.gn_out = (self._f32("conv_norm_out.weight"), self._f32("conv_norm_out.bias"))
        self.conv_out = self._lin("conv_out")
        self.sd = None                                      # packed: drop the reference to the caller's tensors
        self._kv_cache: Tuple[Optional[tuple], Dict[str, torch.Tensor]] = (None, {})

    # ---- transformer block (attention.py:173-188, 236-247) -------------------------------------------
    def _ln(self, x32, gb, m, c):
        y = torch.empty((m, c), dtype=_hip.TORCH_DTYPE[self.dt], device=x32.device)
        call("pmi_layernorm_fwd", ptr(x32), c, ptr(gb[0]), ptr(gb[1]), ptr(y), None, None, m, c, 1e-5, self.dt)
        return y

    def _context_kv(self, context: torch.Tensor) -> Dict[str, torch.Tensor]:
        """k|v projections of the prompt encodings for every cross-attention layer; they depend on the conditioning only, so they
        are computed when a new context tensor is seen and reused over the sampling steps."""
        key = (context.data_ptr(), context._version, tuple(context.shape))
        if self._kv_cache[0] != key:
            n, tc, cd = context.shape
            c16 = torch.empty((n * tc, cd), dtype=_hip.TORCH_DTYPE[self.dt], device=self.device)
            ctx = context.float().contiguous()
            call("pmi_cast_f32_to_16", ptr(ctx), ptr(c16), ctx.numel(), ACT_NONE, self.dt)
            kv = {k: ops.igemm(c16, w) for k, w in self.w.items() if k.endswith(".kv2")}
            self._kv_cache = (key, kv, context)             # holding the tensor keeps its data_ptr from being recycled
        return self._kv_cache[1]

    def _attn(self, k, x, kv_all, tc):
        dt, w, heads = self.dt, self.w, self.cfg.heads
        n, hh, ww, c = x.shape
        t, m = hh * ww, n * hh * ww
        b = k + ".transformer_blocks.0"
        hn = ops.group_norm(x, *w[k + ".gn"], self.cfg.groups, dt, eps=1e-6)
        h = ops.igemm(hn.view(m, c), w[k + ".proj_in"], out_f32=True)                             # fp32 token stream [m, c]
        qkv = ops.igemm(self._ln(h, w[b + ".norm1"], m, c), w[b + ".qkv1"])
        a = ops.attention(qkv.view(n, t, 3 * c), heads, 1, dt)
        h = ops.igemm(a.view(m, c), w[b + ".out1"], residual=h, out_f32=True)
        q = ops.igemm(self._ln(h, w[b + ".norm2"], m, c), w[b + ".q2"])
        a = ops.cross_attention(q.view(n, t, c), kv_all[b + ".kv2"].view(n, tc, 2 * c), heads, dt)
        h = ops.igemm(a.view(m, c), w[b + ".out2"], residual=h, out_f32=True)
        gg = ops.geglu_linear(self._ln(h, w[b + ".norm3"], m, c), w[b + ".ff1"])                  # value * gelu(gate), [m, 4c]
        h16 = ops.igemm(gg, w[b + ".ff2"], residual=h)                                            # fp32 residual in, 16-bit tokens out
        out = ops.igemm(h16, w[k + ".proj_out"], residual=x.view(m, c), want_stats=True, hw=t)
        o4 = out.view(n, hh, ww, c)
        if hasattr(out, "_pmi_stats"):
            o4._pmi_stats = out._pmi_stats
        return o4

    def _run(self, blk, h, h1, emb, kv_all, tc):
        for l in blk:
            if l[0] == "res":
                off, co = self.emb_off[l[1]]
                h = self._resnet(l[1], h, h1, emb[:, off:off + co], self.cfg.groups, 1e-5)
            elif l[0] == "attn":
                h = self._attn(l[1], h, kv_all, tc)
            elif l[0] == "down":
                h = ops.igemm(h, self.w[l[1]], stride=2, want_stats=True)
            else:
                h = ops.igemm(h, self.w[l[1]], up=True, want_stats=True)
            h1 = None
        return h

    @torch.no_grad()
    def forward(self, latents: torch.Tensor, timesteps: torch.Tensor, context: torch.Tensor) -> torch.Tensor:
        """latents NCHW fp32 [N, in, h, w], timesteps [N], context [N, T, context_dim] fp32 -> predicted noise NCHW fp32."""
        cfg, dt, dev = self.cfg, self.dt, self.device
        if not latents.is_cuda or not context.is_cuda:
            raise RuntimeError("SdUnetEngine runs on a HIP device only (no CPU fallback)")
        latents = latents.float().contiguous()
        n, cin, hh, ww = latents.shape
        levels = len(cfg.block_out) - 1
        if cin != cfg.in_channels or hh % (1 << levels) or ww % (1 << levels):
            raise ValueError(f"latents must be [N, {cfg.in_channels}, h, w] with h, w divisible by {1 << levels}")
        if context.ndim != 3 or context.shape[0] != n or context.shape[2] != cfg.context_dim:
            raise ValueError(f"context must be [N, T, {cfg.context_dim}] with N = {n}")
        tdt = _hip.TORCH_DTYPE[dt]
        t = timesteps.to(device=dev, dtype=torch.float32).contiguous()
        temb = torch.empty((n, cfg.block_out[0]), dtype=tdt, device=dev)
        call("pmi_timestep_embedding", ptr(t), ptr(temb), n, cfg.block_out[0], 10000.0, dt)
        e = ops.igemm(temb, self.te1, act=ACT_SILU)
        e = ops.igemm(e, self.te2, act=ACT_SILU)              # SiLU(emb): the only form the ResnetBlocks consume
        emb = ops.igemm(e, self.emb_all, out_f32=True)        # all time_emb_proj outputs, [N, sum Cout]
        kv_all, tc = self._context_kv(context), context.shape[1]
        cp = self.conv_in.cin_p
        x = torch.empty((n, hh, ww, cp), dtype=tdt, device=dev)
        call("pmi_nchw_to_nhwc", ptr(latents), ptr(x), n, cin, hh, ww, cp, 1.0, 0.0, dt)
        h = ops.igemm(x, self.conv_in, want_stats=True)
        hs = [h]
        for blk in self.down:
            h = self._run(blk, h, None, emb, kv_all, tc)
            hs.append(h)
        h = self._run(self.mid, h, None, emb, kv_all, tc)
        for blk in self.up:
            h = self._run(blk, h, hs.pop(), emb, kv_all, tc)
        ca, cb = ops.group_norm_coeffs(h, *self.gn_out, cfg.groups, dt)
        y = ops.igemm(h, self.conv_out, out_f32=True, prologue=(ca, cb, ACT_SILU))
        out = torch.empty((n, cfg.out_channels, hh, ww), dtype=torch.float32, device=dev)
        call("pmi_nhwc_to_nchw", ptr(y), y.shape[-1], ptr(out), n, hh, ww, cfg.out_channels, 1.0, 0.0)
        return out


class VaeDecoderEngine(_Blocks):
    """AutoencoderKL.decode (stable_diffusion.py:195-198): latents / 0.18215 -> post_quant_conv -> decoder -> (x + 1) / 2."""

    def __init__(self, cfg: VaeConfig, state_dict: Dict[str, torch.Tensor], device, dtype="bf16"):
        self._init_common(state_dict, device, dtype)
        self.cfg = cfg
        sd = state_dict
        # post_quant_conv (1x1, 4 -> 4) with its output channels zero-padded to the 8 the next convolution reads
        lc, lp = cfg.latent_channels, (cfg.latent_channels + 7) // 8 * 8
        wq = torch.zeros((lp, lc, 1, 1)); wq[:lc] = sd["post_quant_conv.weight"].detach().float()
        bq = torch.zeros(lp); bq[:lc] = sd["post_quant_conv.bias"].detach().float()
        self.pq = PackedLinear(wq, bq, self.dt, self.device, cin_pad=lp)
        self.conv_in = self._lin("decoder.conv_in", cin_pad=lp)
        self._pack_resnet("decoder.mid_block.resnets.0"); self._pack_resnet("decoder.mid_block.resnets.1")
        self._pack_vae_attention("decoder.mid_block.attentions.0")
        self.plan = []
        for i, o in enumerate(reversed(cfg.block_out)):
            for j in range(cfg.layers_per_block + 1):
                k = f"decoder.up_blocks.{i}.resnets.{j}"
                self._pack_resnet(k)
                self.plan.append(("res", k))
            if i != len(cfg.block_out) - 1:
                k = f"decoder.up_blocks.{i}.upsamplers.0.conv"
                self.w[k] = self._lin(k)
                self.plan.append(("up", k))
        self.gn_out = (self._f32("decoder.conv_norm_out.weight"), self._f32("decoder.conv_norm_out.bias"))
        self.conv_out = self._lin("decoder.conv_out")
        self.sd = None

    @torch.no_grad()
    def forward(self, latents: torch.Tensor, scale: float = 1.0 / 0.18215, to_images: bool = True) -> torch.Tensor:
        """latents NCHW fp32 (the UNet's space) -> images NCHW fp32 in [0, 1] (to_images) or the decoder's x in [-1, 1]."""
        cfg, dt, dev, w = self.cfg, self.dt, self.device, self.w
        if not latents.is_cuda:
            raise RuntimeError("VaeDecoderEngine runs on a HIP device only (no CPU fallback)")
        latents = latents.float().contiguous()
        n, c, hh, ww = latents.shape
        if c != cfg.latent_channels:
            raise ValueError(f"latents must have {cfg.latent_channels} channels")
        tdt = _hip.TORCH_DTYPE[dt]
        x = torch.empty((n, hh, ww, self.pq.cin_p), dtype=tdt, device=dev)
        call("pmi_nchw_to_nhwc", ptr(latents), ptr(x), n, c, hh, ww, self.pq.cin_p, float(scale), 0.0, dt)
        z = ops.igemm(x, self.pq)
        h = ops.igemm(z, self.conv_in, want_stats=True)
        g, eps = cfg.groups, 1e-6
        h = self._resnet("decoder.mid_block.resnets.0", h, None, None, g, eps)
        h4 = self._vae_attention("decoder.mid_block.attentions.0", h, g)
        h = self._resnet("decoder.mid_block.resnets.1", h4, None, None, g, eps)
        for kind, k in self.plan:
            h = self._resnet(k, h, None, None, g, eps) if kind == "res" else ops.igemm(h, w[k], up=True, want_stats=True)
        ca, cb = ops.group_norm_coeffs(h, *self.gn_out, g, dt, eps=eps)
        y = ops.igemm(h, self.conv_out, out_f32=True, prologue=(ca, cb, ACT_SILU))
        _, ho, wo, _ = y.shape
        out = torch.empty((n, cfg.out_channels, ho, wo), dtype=torch.float32, device=dev)
        mul, add = (0.5, 0.5) if to_images else (1.0, 0.0)          # diffusion_space.decode: (x + 1) / 2
        call("pmi_nhwc_to_nchw", ptr(y), y.shape[-1], ptr(out), n, ho, wo, cfg.out_channels, mul, add)
        return out


class VaeEncoderEngine(_Blocks):
    """AutoencoderKL.encode (stable_diffusion.py:176-192): images -> 2*img-1 -> encoder -> quant_conv -> (mean, logvar)."""

    def __init__(self, cfg: VaeConfig, state_dict: Dict[str, torch.Tensor], device, dtype="bf16"):
        self._init_common(state_dict, device, dtype)
        self.cfg = cfg
        self.conv_in = self._lin("encoder.conv_in", cin_pad=8)
        self.plan = []
        for i, o in enumerate(cfg.block_out):
            for j in range(cfg.layers_per_block):
                k = f"encoder.down_blocks.{i}.resnets.{j}"
                self._pack_resnet(k)
                self.plan.append(("res", k))
            if i != len(cfg.block_out) - 1:
                k = f"encoder.down_blocks.{i}.downsamplers.0.conv"
                self.w[k] = self._lin(k)
                self.plan.append(("down", k))
        self._pack_resnet("encoder.mid_block.resnets.0"); self._pack_resnet("encoder.mid_block.resnets.1")
        self._pack_vae_attention("encoder.mid_block.attentions.0")
        self.gn_out = (self._f32("encoder.conv_norm_out.weight"), self._f32("encoder.conv_norm_out.bias"))
        self.conv_out = self._lin("encoder.conv_out")
        self.quant = (self._f32("quant_conv.weight").flatten(1), self._f32("quant_conv.bias"))
        self.sd = None

    def _down(self, h, lin):
        """Downsample2D(padding=0): pad right/bottom by one, 3x3 stride-2 conv without padding.  Run as the symmetric pad-1 stride-2
        convolution on a copy with a one-pixel zero frame: its output (oy+1, ox+1) reads exactly x[2oy..2oy+2, 2ox..2ox+2]."""
        n, hh, ww, c = h.shape
        xp = torch.zeros((n, hh + 2, ww + 2, c), dtype=h.dtype, device=h.device)
        xp[:, 1:hh + 1, 1:ww + 1] = h
        return ops.igemm(xp, lin, stride=2)[:, 1:, 1:].contiguous()

    @torch.no_grad()
    def forward(self, images: torch.Tensor):
        """images NCHW fp32 in [0, 1] -> (mean, logvar) NCHW fp32 [N, latent, H/8, W/8] of the latent distribution."""
        cfg, dt, dev, w = self.cfg, self.dt, self.device, self.w
        if not images.is_cuda:
            raise RuntimeError("VaeEncoderEngine runs on a HIP device only (no CPU fallback)")
        images = images.float().contiguous()
        n, c, hh, ww = images.shape
        down = 1 << (len(cfg.block_out) - 1)
        if c != cfg.out_channels or hh % down or ww % down:
            raise ValueError(f"images must be [N, {cfg.out_channels}, H, W] with H, W divisible by {down}")
        x = torch.empty((n, hh, ww, 8), dtype=_hip.TORCH_DTYPE[dt], device=dev)
        call("pmi_nchw_to_nhwc", ptr(images), ptr(x), n, c, hh, ww, 8, 2.0, -1.0, dt)      # diffusion_space.encode: 2*img - 1
        h = ops.igemm(x, self.conv_in, want_stats=True)
        g, eps = cfg.groups, 1e-6
        for kind, k in self.plan:
            h = self._resnet(k, h, None, None, g, eps) if kind == "res" else self._down(h, w[k])
        h = self._resnet("encoder.mid_block.resnets.0", h, None, None, g, eps)
        h = self._vae_attention("encoder.mid_block.attentions.0", h, g)
        h = self._resnet("encoder.mid_block.resnets.1", h, None, None, g, eps)
        ca, cb = ops.group_norm_coeffs(h, *self.gn_out, g, dt, eps=eps)
        y = ops.igemm(h, self.conv_out, out_f32=True, prologue=(ca, cb, ACT_SILU))        # [n, h/8, w/8, 8] fp32
        _, ho, wo, c2 = y.shape
        m = n * ho * wo
        mom = ops.linear_f32(y.view(m, c2), self.quant[0], self.quant[1])                  # quant_conv (1x1, 8 -> 8) in exact fp32
        out = torch.empty((n, c2, ho, wo), dtype=torch.float32, device=dev)
        call("pmi_nhwc_to_nchw", ptr(mom), c2, ptr(out), n, ho, wo, c2, 1.0, 0.0)
        lc = cfg.latent_channels
        return out[:, :lc].contiguous(), out[:, lc:2 * lc].contiguous()
