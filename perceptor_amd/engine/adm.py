"""HIP execution engine for the ADM UNet (ε-prediction) — replaces UNetModel.forward.

Mirrors perceptor/models/guided_diffusion/unet.py:626-654 (forward), :232-252
(ResBlock), :294-300 (AttentionBlock) and the constructor loops :471-601, but runs
NHWC 16-bit activations through libperceptor_hip.so:

  * every conv3x3 / conv1x1 / linear is one pmi_igemm launch (MFMA implicit GEMM) with
    bias, per-sample bias, residual add, nearest-up gather and skip-concat (two source
    pointers) fused, so th.cat / F.interpolate / "+ skip" never touch HBM on their own;
  * GroupNorm32+SiLU(+FiLM)(+AvgPool) is stats -> finalize -> apply, fp32 statistics;
  * all 49 emb_layers linears run as ONE GEMM per step (their weights are concatenated);
  * attention with 64-channel heads runs the fused flash kernel.

State-dict keys are the reference's, so a real checkpoint loads as is.
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
class AdmConfig:
    """Resolved hyper-parameters (script_util.py:130-184)."""
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


_DEFAULT_MULT = {512: (0.5, 1, 1, 2, 2, 4, 4), 256: (1, 1, 2, 2, 4, 4), 128: (1, 1, 2, 3, 4), 64: (1, 2, 3, 4)}


def openimages_config() -> AdmConfig:   # create_models.py:8-33
    return AdmConfig(512, 256, 2, _DEFAULT_MULT[512], (16, 32, 64), num_head_channels=64,
                     use_scale_shift_norm=True, resblock_updown=True)


def pixelart_config() -> AdmConfig:     # create_models.py:36-62
    return AdmConfig(256, 128, 2, _DEFAULT_MULT[256], (16,), num_heads=1)


class _Res:
    def __init__(self, prefix, cin, cout, up=False, down=False, srcs=None):
        self.p, self.cin, self.cout, self.up, self.down = prefix, cin, cout, up, down
        self.srcs = srcs          # channel counts of a concat input (h, skip), None for a single source
        self.emb_off = 0


class _Attn:
    def __init__(self, prefix, c, heads):
        self.p, self.c, self.heads = prefix, c, heads


class _Resample:
    def __init__(self, prefix, c, up):
        self.p, self.c, self.up = prefix, c, up


def build_plan(cfg: AdmConfig):
    """Layer descriptors in execution order + parameter shapes (names as in the reference)."""
    mc = cfg.model_channels

    def heads(c, upsample=False):
        if cfg.num_head_channels != -1:
            return c // cfg.num_head_channels
        if upsample and cfg.num_heads_upsample != -1:
            return cfg.num_heads_upsample
        return cfg.num_heads

    ch = int(cfg.channel_mult[0] * mc)
    inp: List[list] = [[("conv0", "input_blocks.0.0", cfg.in_channels, ch)]]
    skip_ch = [ch]
    ds = 1
    for level, mult in enumerate(cfg.channel_mult):
        for _ in range(cfg.num_res_blocks):
            i = len(inp)
            cout = int(mult * mc)
            layers = [_Res(f"input_blocks.{i}.0", ch, cout)]
            ch = cout
            if ds in cfg.attention_ds:
                layers.append(_Attn(f"input_blocks.{i}.1", ch, heads(ch)))
            inp.append(layers)
            skip_ch.append(ch)
        if level != len(cfg.channel_mult) - 1:
            i = len(inp)
            if cfg.resblock_updown:
                inp.append([_Res(f"input_blocks.{i}.0", ch, ch, down=True)])
            else:
                inp.append([_Resample(f"input_blocks.{i}.0", ch, up=False)])
            skip_ch.append(ch)
            ds *= 2
    mid = [_Res("middle_block.0", ch, ch), _Attn("middle_block.1", ch, heads(ch)), _Res("middle_block.2", ch, ch)]
    out: List[list] = []
    for level, mult in list(enumerate(cfg.channel_mult))[::-1]:
        for i in range(cfg.num_res_blocks + 1):
            j = len(out)
            ich = skip_ch.pop()
            cout = int(mc * mult)
            layers = [_Res(f"output_blocks.{j}.0", ch + ich, cout, srcs=(ch, ich))]
            ch = cout
            k = 1
            if ds in cfg.attention_ds:
                layers.append(_Attn(f"output_blocks.{j}.{k}", ch, heads(ch, True)))
                k += 1
            if level and i == cfg.num_res_blocks:
                if cfg.resblock_updown:
                    layers.append(_Res(f"output_blocks.{j}.{k}", ch, ch, up=True))
                else:
                    layers.append(_Resample(f"output_blocks.{j}.{k}", ch, up=True))
                ds //= 2
            out.append(layers)
    return inp, mid, out, ch


def state_dict_shapes(cfg: AdmConfig) -> Dict[str, Tuple[int, ...]]:
    mc, ted = cfg.model_channels, 4 * cfg.model_channels
    inp, mid, out, ch_last = build_plan(cfg)
    S: Dict[str, Tuple[int, ...]] = {
        "time_embed.0.weight": (ted, mc), "time_embed.0.bias": (ted,),
        "time_embed.2.weight": (ted, ted), "time_embed.2.bias": (ted,)}
    for layers in inp + [mid] + out:
        for l in layers:
            if isinstance(l, tuple):
                S[l[1] + ".weight"] = (l[3], l[2], 3, 3); S[l[1] + ".bias"] = (l[3],)
            elif isinstance(l, _Res):
                p = l.p
                S[p + ".in_layers.0.weight"] = (l.cin,); S[p + ".in_layers.0.bias"] = (l.cin,)
                S[p + ".in_layers.2.weight"] = (l.cout, l.cin, 3, 3); S[p + ".in_layers.2.bias"] = (l.cout,)
                e = 2 * l.cout if cfg.use_scale_shift_norm else l.cout
                S[p + ".emb_layers.1.weight"] = (e, ted); S[p + ".emb_layers.1.bias"] = (e,)
                S[p + ".out_layers.0.weight"] = (l.cout,); S[p + ".out_layers.0.bias"] = (l.cout,)
                S[p + ".out_layers.3.weight"] = (l.cout, l.cout, 3, 3); S[p + ".out_layers.3.bias"] = (l.cout,)
                if l.cin != l.cout:
                    S[p + ".skip_connection.weight"] = (l.cout, l.cin, 1, 1); S[p + ".skip_connection.bias"] = (l.cout,)
            elif isinstance(l, _Attn):
                p = l.p
                S[p + ".norm.weight"] = (l.c,); S[p + ".norm.bias"] = (l.c,)
                S[p + ".qkv.weight"] = (3 * l.c, l.c, 1); S[p + ".qkv.bias"] = (3 * l.c,)
                S[p + ".proj_out.weight"] = (l.c, l.c, 1); S[p + ".proj_out.bias"] = (l.c,)
            elif isinstance(l, _Resample) and cfg.conv_resample:
                n = l.p + (".conv" if l.up else ".op")
                S[n + ".weight"] = (l.c, l.c, 3, 3); S[n + ".bias"] = (l.c,)
    S["out.0.weight"] = (ch_last,); S["out.0.bias"] = (ch_last,)
    S["out.2.weight"] = (cfg.out_channels, int(cfg.channel_mult[0] * mc), 3, 3); S["out.2.bias"] = (cfg.out_channels,)
    return S


class AdmEngine:
    def __init__(self, cfg: AdmConfig, state_dict: Dict[str, torch.Tensor], device, dtype="bf16"):
        self.cfg, self.device = cfg, torch.device(device)
        self.dt = _hip.dtype_code(dtype)
        _hip.lib()
        sd, dev, dt = state_dict, self.device, self.dt
        self.inp, self.mid, self.out, self.ch_last = build_plan(cfg)
        f32 = lambda k: sd[k].detach().float().to(dev).contiguous()
        lin = lambda k, **kw: PackedLinear(sd[k + ".weight"], sd.get(k + ".bias"), dt, dev, **kw)
        self.w: Dict[str, object] = {}
        self.precise = dt == _hip.DT_F16X2
        if self.precise:      # fp32 time MLPs (exact-fp32 MFMA GEMM): their weights stay fp32 in the reference too (unet.py:610-616)
            self.te0, self.te2 = (f32("time_embed.0.weight"), f32("time_embed.0.bias")), (f32("time_embed.2.weight"), f32("time_embed.2.bias"))
        else:
            self.te0, self.te2 = lin("time_embed.0"), lin("time_embed.2")
        emb_w, emb_b, off = [], [], 0
        for layers in self.inp + [self.mid] + self.out:
            for l in layers:
                if isinstance(l, tuple):
                    self.w[l[1]] = lin(l[1], cin_pad=8)
                elif isinstance(l, _Res):
                    p = l.p
                    self.w[p + ".gn1"] = (f32(p + ".in_layers.0.weight"), f32(p + ".in_layers.0.bias"))
                    self.w[p + ".conv1"] = lin(p + ".in_layers.2", sources=l.srcs)
                    self.w[p + ".gn2"] = (f32(p + ".out_layers.0.weight"), f32(p + ".out_layers.0.bias"))
                    self.w[p + ".conv2"] = lin(p + ".out_layers.3")
                    if l.cin != l.cout:
                        self.w[p + ".skip"] = lin(p + ".skip_connection", sources=l.srcs)
                    l.emb_off = off
                    emb_w.append(sd[p + ".emb_layers.1.weight"].float()); emb_b.append(sd[p + ".emb_layers.1.bias"].float())
                    off += emb_w[-1].shape[0]
                elif isinstance(l, _Attn):
                    p = l.p
                    self.w[p + ".gn"] = (f32(p + ".norm.weight"), f32(p + ".norm.bias"))
                    self.w[p + ".qkv"] = lin(p + ".qkv")
                    self.w[p + ".proj"] = lin(p + ".proj_out")
                elif isinstance(l, _Resample) and cfg.conv_resample:
                    self.w[l.p] = lin(l.p + (".conv" if l.up else ".op"))
        if self.precise:
            self.emb_all = (torch.cat(emb_w, 0).to(dev).contiguous(), torch.cat(emb_b, 0).to(dev).contiguous())
        else:
            self.emb_all = PackedLinear(torch.cat(emb_w, 0), torch.cat(emb_b, 0), dt, dev)
        self.gn_out = (f32("out.0.weight"), f32("out.0.bias"))
        self.conv_out = lin("out.2")

    # ---- blocks ----------------------------------------------------------------------------------
    def _res(self, l: _Res, x, x1, emb):
        cfg, dt, w = self.cfg, self.dt, self.w
        g1, b1 = w[l.p + ".gn1"]
        ecols = 2 * l.cout if cfg.use_scale_shift_norm else l.cout
        e = emb[:, l.emb_off:l.emb_off + ecols]
        nb = None if cfg.use_scale_shift_norm else e
        skip, skip1 = x, x1
        if l.down:
            # SiLU(GN(x)) must be pooled AFTER the activation: streaming apply+pool kernel, then the conv
            if x1 is None:      # both pooled tensors from one pass over x
                h, skip = ops.group_norm_pool_skip(x, g1, b1, 32, dt, act=ACT_SILU)
            else:
                h = ops.group_norm(x, g1, b1, 32, dt, x1=x1, act=ACT_SILU, pool=True)
                skip = ops.avgpool2(x, dt)
            h = ops.igemm(h, w[l.p + ".conv1"], nbias=nb, want_stats=True)
        else:
            # GroupNorm-apply + SiLU fused into the conv's patch staging (no activated copy in HBM)
            ca, cb = ops.group_norm_coeffs(x, g1, b1, 32, dt, x1=x1)
            h = ops.igemm(x, w[l.p + ".conv1"], a1=x1, up=l.up, nbias=nb, prologue=(ca, cb, ACT_SILU), want_stats=True)
        g2, b2 = w[l.p + ".gn2"]
        if cfg.use_scale_shift_norm:
            ca, cb = ops.group_norm_coeffs(h, g2, b2, 32, dt, film=e, film_ld=emb.stride(0))
        else:
            ca, cb = ops.group_norm_coeffs(h, g2, b2, 32, dt)
        if l.cin != l.cout:
            skip = ops.igemm(skip, w[l.p + ".skip"], a1=skip1)
        elif skip1 is not None:
            raise NotImplementedError("identity skip over a concatenated input does not occur in the shipped configs")
        return ops.igemm(h, w[l.p + ".conv2"], residual=skip, res_up=l.up, prologue=(ca, cb, ACT_SILU), want_stats=True)

    def _attn(self, l: _Attn, x):
        dt, w = self.dt, self.w
        n, hh, ww, c = x.shape
        g, b = w[l.p + ".gn"]
        hn = ops.group_norm(x, g, b, 32, dt)
        qkv = ops.igemm(hn.view(n * hh * ww, c), w[l.p + ".qkv"])
        a = ops.attention(qkv.view(n, hh * ww, 3 * c), l.heads, 1 if self.cfg.use_new_attention_order else 0, dt)
        out = ops.igemm(a.view(n * hh * ww, c), w[l.p + ".proj"], residual=x.view(n * hh * ww, c), want_stats=True, hw=hh * ww)
        o4 = out.view(n, hh, ww, c)
        if hasattr(out, "_pmi_stats"):
            o4._pmi_stats = out._pmi_stats
        return o4

    def _run(self, layers, h, h1, emb):
        for l in layers:
            if isinstance(l, tuple):
                h = ops.igemm(h, self.w[l[1]], want_stats=True)
            elif isinstance(l, _Res):
                h = self._res(l, h, h1, emb)
            elif isinstance(l, _Attn):
                h = self._attn(l, h)
            elif isinstance(l, _Resample):
                if self.cfg.conv_resample:
                    h = ops.igemm(h, self.w[l.p], up=l.up, stride=1 if l.up else 2, want_stats=True)
                else:
                    raise NotImplementedError("conv_resample=False is not used by the shipped configs")
            h1 = None
        return h

    @torch.no_grad()
    def forward(self, images: torch.Tensor, timesteps: torch.Tensor, out_channels: Optional[int] = None) -> torch.Tensor:
        """images: NCHW fp32 in [0,1] (encoded to x = 2*img-1 on the fly); returns NCHW fp32 model output."""
        cfg, dt, dev = self.cfg, self.dt, self.device
        if not images.is_cuda:
            raise RuntimeError("AdmEngine runs on a HIP device only (no CPU fallback)")
        images = images.float().contiguous()
        n, _, hh, ww = images.shape
        t = timesteps.to(device=dev, dtype=torch.float32).contiguous()
        tdt = _hip.TORCH_DTYPE[dt]
        temb = torch.empty((n, cfg.model_channels), dtype=torch.float32 if self.precise else tdt, device=dev)
        call("pmi_timestep_embedding", ptr(t), ptr(temb), n, cfg.model_channels, 10000.0, dt)
        if self.precise:
            e = ops.linear_f32(temb, *self.te0, act=ACT_SILU)
            e = ops.linear_f32(e, *self.te2, act=ACT_SILU)
            emb = ops.linear_f32(e, *self.emb_all)
        else:
            e = ops.igemm(temb, self.te0, act=ACT_SILU)
            e = ops.igemm(e, self.te2, act=ACT_SILU)            # = SiLU(emb): the only form the ResBlocks consume
            emb = ops.igemm(e, self.emb_all, out_f32=True)       # [N, sum of all emb_layers outputs]
        x = torch.empty((n, hh, ww, 16 if self.precise else 8), dtype=tdt, device=dev)
        call("pmi_prep_input", ptr(images), None, 0, ptr(x), n, hh, ww, 8, dt)
        h, hs = x, []
        for layers in self.inp:
            h = self._run(layers, h, None, emb)
            hs.append(h)
        h = self._run(self.mid, h, None, emb)
        for layers in self.out:
            h = self._run(layers, h, hs.pop(), emb)
        g, b = self.gn_out
        ca, cb = ops.group_norm_coeffs(h, g, b, 32, dt)
        y = ops.igemm(h, self.conv_out, out_f32=True, prologue=(ca, cb, ACT_SILU))
        co = out_channels or cfg.out_channels
        out = torch.empty((n, co, hh, ww), dtype=torch.float32, device=dev)
        call("pmi_finish_output", ptr(y), y.shape[-1], ptr(out), n, hh, ww, co)
        return out

    # ---- input gradient -----------------------------------------------------------------------------------
    # Upstream GuidedDiffusion.predicted_noise is differentiable (guided_diffusion.py:125-133) and runs its blocks through CheckpointFunction
    # (nn.py:138-189, unet.py:228-229, 292): activations are dropped and recomputed in the backward to fit 16-40 GB cards.  Here the
    # training-mode forward keeps them instead -- every tensor it keeps is one the inference forward writes to HBM anyway (block inputs,
    # conv1 outputs, attention operands), about 26 GB for GD "standard" at 512x512 x 8 of 288 GB -- and backward() walks the tape once: no
    # recomputation.  dX of a convolution is the forward kernel on transposed + flipped weights (packed lazily); GroupNorm32 (+FiLM) + SiLU
    # backward is pmi_gn_bwd_*; attention backward is the ViT's flash backward (64-channel heads) or batched GEMMs with the kept softmax.
    def _check_train(self):
        if self.precise:
            raise NotImplementedError("the ADM input gradient runs in the 16-bit modes (bf16 / f16)")

    def _wt(self, key, weight, cin_pad=None):
        """Packed weights of the input-gradient convolution of `weight` [Cout, Cin, k(, k)]: [Cin, Cout, k, k] with both taps flipped."""
        if key not in self.w:
            w = weight.detach().float()
            if w.ndim == 3:
                w = w[..., None]
            w = w.permute(1, 0, 2, 3)
            if w.shape[-1] == 3:
                w = w.flip(2, 3)
            self.w[key] = PackedLinear(w.contiguous(), None, self.dt, self.device, cin_pad=cin_pad)
        return self.w[key]

    def _qkv_order1(self, l: _Attn, sd):
        """qkv projection producing channels (q|k|v, head, d) -- the layout of pmi_vit_attn_fwd / ops.attention_train -- whatever the
        checkpoint's order (unet.py:332-348 legacy: (head, q|k|v, d))."""
        key = l.p + ".qkv_o1"
        if key not in self.w:
            wq, bq = sd[l.p + ".qkv.weight"].detach().float(), sd[l.p + ".qkv.bias"].detach().float()
            if not self.cfg.use_new_attention_order:
                d = l.c // l.heads
                idx = torch.arange(3 * l.c).view(l.heads, 3, d).permute(1, 0, 2).reshape(-1)
                wq, bq = wq[idx], bq[idx]
            self.w[key] = PackedLinear(wq, bq, self.dt, self.device)
            self.w[key + ".w"] = wq
        return self.w[key], self.w[key + ".w"]

    def _res_train(self, l: _Res, x, x1, emb, tape):
        cfg, dt, w = self.cfg, self.dt, self.w
        g1, b1 = w[l.p + ".gn1"]
        ecols = 2 * l.cout if cfg.use_scale_shift_norm else l.cout
        e = emb[:, l.emb_off:l.emb_off + ecols]
        nb = None if cfg.use_scale_shift_norm else e
        ca, cb, parts = ops.group_norm_coeffs_train(x, g1, b1, 32, dt, x1=x1)
        skip, skip1 = x, x1
        if l.down:
            if x1 is not None:
                raise NotImplementedError("down-sampling ResBlock over a concatenated input does not occur in the shipped configs")
            n, hh, ww, c = x.shape
            hp = torch.empty((n, hh // 2, ww // 2, c), dtype=x.dtype, device=x.device)
            skip = torch.empty_like(hp)
            call("pmi_gn_apply_pool_skip", ptr(x), ptr(ca), ptr(cb), ptr(hp), ptr(skip), n, hh, ww, c, ACT_SILU, dt)
            h = ops.igemm(hp, w[l.p + ".conv1"], nbias=nb, want_stats=True)
        else:
            h = ops.igemm(x, w[l.p + ".conv1"], a1=x1, up=l.up, nbias=nb, prologue=(ca, cb, ACT_SILU), want_stats=True)
        g2, b2 = w[l.p + ".gn2"]
        film = e if cfg.use_scale_shift_norm else None
        ca2, cb2, parts2 = ops.group_norm_coeffs_train(h, g2, b2, 32, dt, film=film, film_ld=emb.stride(0) if film is not None else 0)
        if l.cin != l.cout:
            skip = ops.igemm(skip, w[l.p + ".skip"], a1=skip1)
        elif skip1 is not None:
            raise NotImplementedError("identity skip over a concatenated input does not occur in the shipped configs")
        out = ops.igemm(h, w[l.p + ".conv2"], residual=skip, res_up=l.up, prologue=(ca2, cb2, ACT_SILU), want_stats=True)
        tape.append(("res", l, x, x1, (ca, cb, parts), h, (ca2, cb2, parts2), film))
        return out

    def _attn_train(self, l: _Attn, x, tape, sd):
        dt, w = self.dt, self.w
        n, hh, ww, c = x.shape
        t, d = hh * ww, c // l.heads
        g, b = w[l.p + ".gn"]
        ca, cb, parts = ops.group_norm_coeffs_train(x, g, b, 32, dt)
        hn = torch.empty_like(x)
        call("pmi_gn_apply", ptr(x), None, c, ptr(ca), ptr(cb), None, ptr(hn), n, hh, ww, c, ACT_NONE, 0, dt)
        lin, _ = self._qkv_order1(l, sd)
        qkv = ops.igemm(hn.view(n * t, c), lin)
        if d == 64:                                  # flash-style forward keeping the log-sum-exp (csrc/attn.hip)
            tp32 = (t + 31) // 32 * 32
            aws = torch.empty((6, n * l.heads, tp32, 64), dtype=x.dtype, device=x.device)
            lse = torch.empty((n * l.heads, tp32), dtype=torch.float32, device=x.device)
            a = torch.empty((n * t, c), dtype=x.dtype, device=x.device)
            call("pmi_vit_attn_fwd", ptr(qkv), ptr(aws), ptr(lse), ptr(a), n, t, l.heads, 64.0 ** -0.5, dt)
            saved = (aws, lse, a)
        else:                                        # other head dims (the tiny test configs): batched GEMMs, the softmax is kept
            a, pm = ops.attention_train(qkv.view(n, t, 3 * c), l.heads, dt)
            a = a.view(n * t, c)
            saved = (qkv, pm)
        out = ops.igemm(a, w[l.p + ".proj"], residual=x.view(n * t, c), want_stats=True, hw=t)
        o4 = out.view(n, hh, ww, c)
        if hasattr(out, "_pmi_stats"):
            o4._pmi_stats = out._pmi_stats
        tape.append(("attn", l, x, (ca, cb, parts), saved))
        return o4

    def _run_train(self, layers, h, h1, emb, tape, sd):
        for l in layers:
            if isinstance(l, tuple):
                x_in = h
                h = ops.igemm(h, self.w[l[1]], want_stats=True)
                tape.append(("conv", l, x_in))
            elif isinstance(l, _Res):
                h = self._res_train(l, h, h1, emb, tape)
            elif isinstance(l, _Attn):
                h = self._attn_train(l, h, tape, sd)
            elif isinstance(l, _Resample) and self.cfg.conv_resample:      # unet.py:81-138: nearest x2 + conv / stride-2 conv
                h = ops.igemm(h, self.w[l.p], up=l.up, stride=1 if l.up else 2, want_stats=True)
                tape.append(("resample", l))
            else:
                raise NotImplementedError("conv_resample=False is not used by the shipped configs")
            h1 = None
        return h

    @torch.no_grad()
    def forward_train(self, images: torch.Tensor, timesteps: torch.Tensor, state_dict, out_channels: Optional[int] = None):
        """As forward(), keeping what backward() needs.  Returns (model output NCHW fp32, tape)."""
        self._check_train()
        cfg, dt, dev = self.cfg, self.dt, self.device
        if not images.is_cuda:
            raise RuntimeError("AdmEngine runs on a HIP device only (no CPU fallback)")
        sd = state_dict
        images = images.float().contiguous()
        n, _, hh, ww = images.shape
        t = timesteps.to(device=dev, dtype=torch.float32).contiguous()
        tdt = _hip.TORCH_DTYPE[dt]
        temb = torch.empty((n, cfg.model_channels), dtype=tdt, device=dev)
        call("pmi_timestep_embedding", ptr(t), ptr(temb), n, cfg.model_channels, 10000.0, dt)
        e = ops.igemm(temb, self.te0, act=ACT_SILU)
        e = ops.igemm(e, self.te2, act=ACT_SILU)
        emb = ops.igemm(e, self.emb_all, out_f32=True)
        x = torch.empty((n, hh, ww, 8), dtype=tdt, device=dev)
        call("pmi_prep_input", ptr(images), None, 0, ptr(x), n, hh, ww, 8, dt)
        tape = {"inp": [], "mid": [], "out": []}
        h, hs = x, []
        for layers in self.inp:
            tp = []
            h = self._run_train(layers, h, None, emb, tp, sd)
            tape["inp"].append(tp)
            hs.append(h)
        h = self._run_train(self.mid, h, None, emb, tape["mid"], sd)
        for layers in self.out:
            tp = []
            h = self._run_train(layers, h, hs.pop(), emb, tp, sd)
            tape["out"].append(tp)
        g, b = self.gn_out
        ca, cb, parts = ops.group_norm_coeffs_train(h, g, b, 32, dt)
        y = ops.igemm(h, self.conv_out, out_f32=True, prologue=(ca, cb, ACT_SILU))
        tape["last"] = (h, (ca, cb, parts))
        tape["emb_ld"] = emb.stride(0)
        co = out_channels or cfg.out_channels
        out = torch.empty((n, co, hh, ww), dtype=torch.float32, device=dev)
        call("pmi_finish_output", ptr(y), y.shape[-1], ptr(out), n, hh, ww, co)
        return out, tape

    def _resample_bwd(self, l: _Res, g):
        """Gradient through the block's resampling of a path: up -> sum of the 2x2 block, down -> a quarter to each of the 4 pixels."""
        n, h, w_, c = g.shape
        if l.up:
            out = torch.empty((n, h // 2, w_ // 2, c), dtype=g.dtype, device=g.device)
            call("pmi_upsample_nearest2_bwd", ptr(g), ptr(out), n, h // 2, w_ // 2, c, self.dt)
            return out
        if l.down:
            out = torch.empty((n, 2 * h, 2 * w_, c), dtype=g.dtype, device=g.device)
            call("pmi_avgpool2_bwd", ptr(g), ptr(out), n, 2 * h, 2 * w_, c, self.dt)
            return out
        return g

    def _res_back(self, rec, g, sd, ld):
        _, l, x, x1, gn1, h, gn2, film = rec
        dt, w = self.dt, self.w
        p = l.p
        d_a2 = ops.igemm(g, self._wt(p + ".conv2T", sd[p + ".out_layers.3.weight"]))            # wrt SiLU(GN2(h))
        dh, _ = ops.group_norm_backward(h, d_a2, gn2[0], gn2[1], gn2[2], w[p + ".gn2"][0], 32, dt, film=film, film_ld=ld if film is not None else 0,
                                        act=ACT_SILU)
        d_a1 = self._resample_bwd(l, ops.igemm(dh, self._wt(p + ".conv1T", sd[p + ".in_layers.2.weight"])))     # wrt SiLU(GN1(cat(x, x1)))
        if l.cin != l.cout:
            skw = sd[p + ".skip_connection.weight"]
            if x1 is None:
                gs0, gs1 = ops.igemm(g, self._wt(p + ".skipT", skw)), None
            else:
                c0 = x.shape[-1]
                gs0 = ops.igemm(g, self._wt(p + ".skipT0", skw[:, :c0]))
                gs1 = ops.igemm(g, self._wt(p + ".skipT1", skw[:, c0:]))
        else:
            gs0, gs1 = self._resample_bwd(l, g), None
        return ops.group_norm_backward(x, d_a1, gn1[0], gn1[1], gn1[2], w[p + ".gn1"][0], 32, dt, x1=x1, act=ACT_SILU, gadd0=gs0, gadd1=gs1)

    def _attn_back(self, rec, g, sd):
        _, l, x, gn, saved = rec
        dt, w = self.dt, self.w
        n, hh, ww, c = x.shape
        t, d = hh * ww, c // l.heads
        g2 = g.reshape(n * t, c)
        da = ops.igemm(g2, self._wt(l.p + ".projT", sd[l.p + ".proj_out.weight"]))
        if d == 64:
            aws, lse, a = saved
            tp32 = (t + 31) // 32 * 32
            bws = torch.empty((2, n * l.heads, tp32, 64), dtype=x.dtype, device=x.device)
            delta = torch.empty((n * l.heads, tp32), dtype=torch.float32, device=x.device)
            dqkv = torch.empty((n * t, 3 * c), dtype=x.dtype, device=x.device)
            call("pmi_vit_attn_bwd", ptr(aws), ptr(lse), ptr(a), ptr(da), ptr(bws), ptr(delta), ptr(dqkv), n, t, l.heads, 64.0 ** -0.5, dt)
        else:
            qkv, pm = saved
            dqkv = ops.attention_backward(qkv.view(n, t, 3 * c), pm, da.view(n, t, c), l.heads, dt).view(n * t, 3 * c)
        _, wq = self._qkv_order1(l, sd)
        dhn = ops.igemm(dqkv, self._wt(l.p + ".qkvT", wq)).view(n, hh, ww, c)
        gx, _ = ops.group_norm_backward(x, dhn, gn[0], gn[1], gn[2], w[l.p + ".gn"][0], 32, dt, act=ACT_NONE, gadd0=g.contiguous())
        return gx

    def _back(self, tp, g, sd, ld):
        """Gradient wrt the input(s) of the layer list recorded in `tp` given g = gradient wrt its output: (g_in, g_skip or None)."""
        g1 = None
        for rec in reversed(tp):
            assert g1 is None
            if rec[0] == "res":
                g, g1 = self._res_back(rec, g, sd, ld)
            elif rec[0] == "attn":
                g = self._attn_back(rec, g, sd)
            elif rec[0] == "resample":
                l = rec[1]
                wt = self._wt(l.p + "T", sd[l.p + (".conv" if l.up else ".op") + ".weight"])
                if l.up:                             # conv over nearest-up(x): dX at the high resolution, then the sum of each 2x2 block
                    d = ops.igemm(g, wt)
                    n, h_, w_, c = d.shape
                    g = torch.empty((n, h_ // 2, w_ // 2, c), dtype=d.dtype, device=d.device)
                    call("pmi_upsample_nearest2_bwd", ptr(d), ptr(g), n, h_ // 2, w_ // 2, c, self.dt)
                else:                                # stride-2 conv: dX = stride-1 conv of the zero-inserted gradient with the flipped weights
                    n, h_, w_, c = g.shape           # (two torch ops for the layout step: only the conv_resample configs -- pixelart -- come here)
                    z = torch.zeros((n, 2 * h_, 2 * w_, c), dtype=g.dtype, device=g.device)
                    z[:, ::2, ::2] = g
                    g = ops.igemm(z, wt)
            else:                                    # the first convolution: fp32 gradient wrt the padded input
                g = ops.igemm(g, self._wt(rec[1][1] + "T", sd[rec[1][1] + ".weight"]), out_f32=True)
        return g, g1

    @torch.no_grad()
    def backward(self, tape, d_out: torch.Tensor, state_dict) -> torch.Tensor:
        """d loss / d images (NCHW fp32, images in [0, 1]) from d loss / d output (NCHW fp32, the first d_out.shape[1] output channels) and the
        tape of forward_train().  f16 engines scale the gradient by a power of two on the way in and back on the way out (image gradients
        of a CLIP loss are ~1e-6 and would flush to zero in f16); bf16 needs no scaling."""
        self._check_train()
        dev, dt = self.device, self.dt
        n, co, hh, ww = d_out.shape
        sd = {k: v.detach() for k, v in state_dict.items()}
        scale = 1.0
        if dt == _hip.DT_F16:
            amax = float(d_out.abs().max())
            if amax > 0.0 and amax == amax:
                scale = 2.0 ** max(-24, min(24, -int(torch.tensor(amax).log2().ceil())))
        g = torch.zeros((n, hh, ww, 8), dtype=_hip.TORCH_DTYPE[dt], device=dev)                # output channels + padding (layout only)
        g[..., :co] = (d_out.to(dev).float() * scale).permute(0, 2, 3, 1)
        ld = tape["emb_ld"]
        h, gn = tape["last"]
        d_act = ops.igemm(g, self._wt("out.2T", sd["out.2.weight"], cin_pad=8))
        g, _ = ops.group_norm_backward(h, d_act, gn[0], gn[1], gn[2], self.gn_out[0], 32, dt, act=ACT_SILU)
        g_hs = []                                                                               # gradients of the skip tensors, in pop order
        for tp in reversed(tape["out"]):
            g, gk = self._back(tp, g, sd, ld)
            g_hs.append(gk)
        g, _ = self._back(tape["mid"], g, sd, ld)
        # the out blocks popped hs from the end: the LAST out list read hs[0]; walking them reversed gives g_hs = [for hs[0], hs[1], ...]
        for i in range(len(tape["inp"]) - 1, -1, -1):
            gk = g_hs[i]
            tot = torch.empty_like(g)
            call("pmi_add16", ptr(g), ptr(gk), ptr(tot), g.numel(), dt)                         # hs[i] feeds the next block AND an out block
            g, _ = self._back(tape["inp"][i], tot, sd, ld)
        out = torch.empty((n, 3, hh, ww), dtype=torch.float32, device=dev)
        call("pmi_finish_output", ptr(g), g.shape[-1], ptr(out), n, hh, ww, 3)
        return out * (2.0 / scale)                                                              # x = 2 * images - 1
