"""HIP engine for the v-diffusion UNets (yfcc_2, cc12m_1 and same-family nets).

Replaces YFCC2Model.forward (perceptor/models/velocity_diffusion/yfcc_2.py:247-249, blocks :17-70)
and CC12M1Model.forward (cc12m_1.py:293-302, blocks :19-61).  The reference's nested nn.Sequential
is described by a list program (levels -> [blocks, Skip[...], blocks]); state-dict keys follow from
list positions exactly as nn.Sequential numbers them.

Kernel mapping: conv3x3 (+bias +ReLU +residual) and the bias-free 1x1 skips are pmi_igemm launches,
SkipBlock's torch.cat is never materialised (two source pointers in the K loop), AvgPool2d /
bilinear x2 are streaming kernels, SelfAttention2d = GroupNorm(1,C) -> 1x1 qkv -> flash attention
(d = 64) -> 1x1 out + residual.  cc12m_1: GroupNorm(1,C,affine=False) + Modulation2d + ReLU is one
stats/finalize/apply chain whose FiLM coefficients for ALL blocks come from a single GEMM on the
mapping network's output.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import torch

from .. import _hip
from .._hip import ACT_NONE, ACT_RELU, call, ptr
from . import ops
from .ops import PackedLinear


class Res:
    def __init__(self, cin, cmid, cout, last=False):
        self.cin, self.cmid, self.cout, self.last = cin, cmid, cout, last
        self.mod1 = self.mod2 = 0
        self.srcs = None          # (c, c) when the block reads the concat a SkipBlock leaves behind


class Attn:
    def __init__(self, c):
        self.c = c


class Down:
    pass


class Up:
    pass


class Skip:
    def __init__(self, main):
        self.main = main


def _level(cs: List[int], i: int, attn_from: int, side: int, inner: int) -> list:
    """Contents of the SkipBlock running at resolution level i (i = 1 is the first down-sampled level)."""
    prog: list = [Down()]

    def block(a, b, c):
        prog.append(Res(a, b, c))
        if i >= attn_from:
            prog.append(Attn(c))

    if i < len(cs) - 1:
        block(cs[i - 1], cs[i], cs[i])
        for _ in range(side - 1):
            block(cs[i], cs[i], cs[i])
        prog.append(Skip(_level(cs, i + 1, attn_from, side, inner)))
        block(2 * cs[i], cs[i], cs[i])
        for _ in range(side - 2):
            block(cs[i], cs[i], cs[i])
        block(cs[i], cs[i], cs[i - 1])
    else:
        block(cs[i - 1], cs[i], cs[i])
        for _ in range(inner - 2):
            block(cs[i], cs[i], cs[i])
        block(cs[i], cs[i], cs[i - 1])
    prog.append(Up())
    return prog


def make_spec(name: str, shape, cs: List[int], top: int, side: int, inner: int, attn_from: int, cond: bool, feats: int = 1024,
              head_dim: int = 64, attn_norm: bool = True, up_mode: str = "bilinear", t_input: str = "t", skip_first: bool = False):
    prog = [Res(3 + 16, cs[0], cs[0])] + [Res(cs[0], cs[0], cs[0]) for _ in range(top - 1)]
    prog.append(Skip(_level(cs, 1, attn_from, side, inner)))
    prog.append(Res(2 * cs[0], cs[0], cs[0]))
    prog += [Res(cs[0], cs[0], cs[0]) for _ in range(top - 2)]
    prog.append(Res(cs[0], cs[0], 3, last=True))
    return dict(name=name, shape=tuple(shape), cond=cond, feats=feats, net=prog, head_dim=head_dim, attn_norm=attn_norm,
                up_mode=up_mode, t_input=t_input, skip_first=skip_first)


def yfcc2_spec():   # yfcc_2.py:77-245
    c = 256
    return make_spec("yfcc_2", (3, 512, 512), [c // 2, c, c * 2, c * 2, c * 4, c * 4, c * 8, c * 8], 2, 2, 4, 5, False)


def cc12m1_spec():  # cc12m_1.py:112-291
    c = 128
    return make_spec("cc12m_1", (3, 256, 256), [c, c * 2, c * 2, c * 4, c * 4, c * 8, c * 8], 4, 4, 8, 4, True)


def yfcc1_spec():   # yfcc_1.py:78-336
    c = 128
    return make_spec("yfcc_1", (3, 512, 512), [c, c, c * 2, c * 2, c * 4, c * 4, c * 8, c * 8], 4, 4, 8, 5, False)


def wikiart_spec():  # wikiart_256.py:105-291: no norm in attention, 128-channel heads, nearest upsampling, Fourier features of log-SNR(t)
    c = 128
    return make_spec("wikiart", (3, 256, 256), [c // 2, c, c * 2, c * 2, c * 4, c * 4, c * 8], 4, 4, 8, 4, False,
                     head_dim=128, attn_norm=False, up_mode="nearest", t_input="log_snr", skip_first=True)


def _walk(prog, prefix, fn):
    for i, l in enumerate(prog):
        p = f"{prefix}.{i}"
        if isinstance(l, Skip):
            _walk(l.main, p + ".main", fn)
            if i + 1 < len(prog) and isinstance(prog[i + 1], Res):
                prog[i + 1].srcs = (prog[i + 1].cin // 2, prog[i + 1].cin // 2)
        else:
            fn(l, p)


def state_dict_shapes(spec) -> Dict[str, Tuple[int, ...]]:
    S: Dict[str, Tuple[int, ...]] = {}
    cond, f = spec["cond"], spec["feats"]
    if cond:
        S["mapping_timestep_embed.weight"] = (64, 1)
        S["mapping.0.main.0.weight"] = (f, 640); S["mapping.0.main.0.bias"] = (f,)
        S["mapping.0.main.2.weight"] = (f, f); S["mapping.0.main.2.bias"] = (f,)
        S["mapping.0.skip.weight"] = (f, 640)
        S["mapping.1.main.0.weight"] = (f, f); S["mapping.1.main.0.bias"] = (f,)
        S["mapping.1.main.2.weight"] = (f, f); S["mapping.1.main.2.bias"] = (f,)
    S["timestep_embed.weight"] = (8, 1)

    def add(l, p):
        if isinstance(l, Res):
            j = 4 if cond else 2
            S[p + ".main.0.weight"] = (l.cmid, l.cin, 3, 3); S[p + ".main.0.bias"] = (l.cmid,)
            S[p + f".main.{j}.weight"] = (l.cout, l.cmid, 3, 3); S[p + f".main.{j}.bias"] = (l.cout,)
            if cond:
                S[p + ".main.2.layer.weight"] = (2 * l.cmid, f)
                if not l.last:
                    S[p + ".main.6.layer.weight"] = (2 * l.cout, f)
            if l.cin != l.cout:
                S[p + ".skip.weight"] = (l.cout, l.cin, 1, 1)
        elif isinstance(l, Attn):
            if spec.get("attn_norm", True):
                S[p + ".norm.weight"] = (l.c,); S[p + ".norm.bias"] = (l.c,)
            S[p + ".qkv_proj.weight"] = (3 * l.c, l.c, 1, 1); S[p + ".qkv_proj.bias"] = (3 * l.c,)
            S[p + ".out_proj.weight"] = (l.c, l.c, 1, 1); S[p + ".out_proj.bias"] = (l.c,)

    _walk(spec["net"], "net", add)
    return S


class VDiffEngine:
    def __init__(self, spec, state_dict, device, dtype="bf16"):
        self.spec, self.device = spec, torch.device(device)
        self.dt = _hip.dtype_code(dtype)
        _hip.lib()
        sd, dev, dt = state_dict, self.device, self.dt
        self.cond = spec["cond"]
        self.precise = dt == _hip.DT_F16X2
        f32 = lambda k: sd[k].detach().float().to(dev).contiguous()
        self.w: Dict[str, object] = {}
        mods, off, mod_keys = [], [0], []
        self._mod_keys, self._dmod = mod_keys, None
        first = [True]

        def pack(l, p):
            if isinstance(l, Res):
                j = 4 if self.cond else 2
                self.w[p + ".c1"] = PackedLinear(sd[p + ".main.0.weight"], sd[p + ".main.0.bias"], dt, dev,
                                                 cin_pad=24 if first[0] else None, sources=l.srcs)
                first[0] = False
                self.w[p + ".c2"] = PackedLinear(sd[p + f".main.{j}.weight"], sd[p + f".main.{j}.bias"], dt, dev)
                if l.cin != l.cout:
                    self.w[p + ".skip"] = PackedLinear(sd[p + ".skip.weight"], None, dt, dev, cin_pad=24 if l.cin == 19 else None, sources=l.srcs)
                if self.cond:
                    l.mod1 = off[0]; mods.append(sd[p + ".main.2.layer.weight"].float()); off[0] += 2 * l.cmid
                    mod_keys.append(p + ".main.2.layer.weight")
                    if not l.last:
                        l.mod2 = off[0]; mods.append(sd[p + ".main.6.layer.weight"].float()); off[0] += 2 * l.cout
                        mod_keys.append(p + ".main.6.layer.weight")
            elif isinstance(l, Attn):
                if spec.get("attn_norm", True):
                    self.w[p + ".gn"] = (f32(p + ".norm.weight"), f32(p + ".norm.bias"))
                self.w[p + ".qkv"] = PackedLinear(sd[p + ".qkv_proj.weight"], sd[p + ".qkv_proj.bias"], dt, dev)
                self.w[p + ".out"] = PackedLinear(sd[p + ".out_proj.weight"], sd[p + ".out_proj.bias"], dt, dev)

        _walk(spec["net"], "net", pack)
        self.tw = f32("timestep_embed.weight").reshape(-1)
        if self.cond:
            self.mtw = f32("mapping_timestep_embed.weight").reshape(-1)
            if self.precise:      # the mapping network runs in exact fp32 (tiny GEMMs)
                self.mod_all = torch.cat(mods, 0).to(dev).contiguous()
                L = lambda k, bias=True: (f32(k + ".weight"), f32(k + ".bias") if bias else None)
            else:
                self.mod_all = PackedLinear(torch.cat(mods, 0), None, dt, dev)
                L = lambda k, bias=True: PackedLinear(sd[k + ".weight"], sd.get(k + ".bias") if bias else None, dt, dev)
            self.m = dict(a0=L("mapping.0.main.0"), a2=L("mapping.0.main.2"), askip=L("mapping.0.skip", False),
                          b0=L("mapping.1.main.0"), b2=L("mapping.1.main.2"))

    # ---- blocks --------------------------------------------------------------------------------------
    def _res(self, l: Res, p, x, x1, mod):
        dt, w = self.dt, self.w
        skip = x
        if l.cin != l.cout:
            # precise mode: the last block's 3-channel skip stays fp32 (a 4-channel row cannot hold hi + lo groups)
            skip = ops.igemm(x, w[p + ".skip"], a1=x1, out_f32=self.precise and l.last)
        elif x1 is not None:
            raise NotImplementedError("identity skip over a concatenated input does not occur in this model family")
        if not self.cond:
            h = ops.igemm(x, w[p + ".c1"], a1=x1, act=ACT_RELU)
            if l.last:
                return ops.igemm(h, w[p + ".c2"], residual=skip, out_f32=True)
            return ops.igemm(h, w[p + ".c2"], act=ACT_RELU, residual=skip)
        ld = mod.stride(0)
        # conv1 leaves the statistics of its output behind; GroupNorm(1, C) + Modulation2d + ReLU are applied by conv2 while it
        # stages its input (no normalised copy in HBM); where a shape is not eligible ops.igemm falls back to the streaming kernels
        h = ops.igemm(x, w[p + ".c1"], a1=x1, want_stats=True)
        ca, cb = ops.group_norm_coeffs(h, None, None, 1, dt, film=mod[:, l.mod1:], film_ld=ld)
        if l.last:
            return ops.igemm(h, w[p + ".c2"], residual=skip, out_f32=True, prologue=(ca, cb, ACT_RELU))
        h = ops.igemm(h, w[p + ".c2"], prologue=(ca, cb, ACT_RELU), want_stats=True)
        return ops.group_norm(h, None, None, 1, dt, film=mod[:, l.mod2:], film_ld=ld, act=ACT_RELU, residual=skip)

    def _attn(self, l: Attn, p, x):
        dt, w = self.dt, self.w
        n, hh, ww, c = x.shape
        hn = x
        if self.spec.get("attn_norm", True):
            g, b = w[p + ".gn"]
            hn = ops.group_norm(x, g, b, 1, dt)
        qkv = ops.igemm(hn.view(n * hh * ww, c), w[p + ".qkv"])
        a = ops.attention(qkv.view(n, hh * ww, 3 * c), ops.logical_c(x, dt) // self.spec.get("head_dim", 64), 1, dt)
        return ops.igemm(a.view(n * hh * ww, c), w[p + ".out"], residual=x.view(n * hh * ww, c)).view(n, hh, ww, c)

    def _run(self, prog, prefix, x, mod):
        x1 = None
        for i, l in enumerate(prog):
            p = f"{prefix}.{i}"
            if isinstance(l, Res):
                x, x1 = self._res(l, p, x, x1, mod), None
            elif isinstance(l, Attn):
                x = self._attn(l, p, x)
            elif isinstance(l, Down):
                x = ops.avgpool2(x, self.dt)
            elif isinstance(l, Up):
                x = ops.upsample_nearest2(x) if self.spec.get("up_mode") == "nearest" else ops.upsample_bilinear2(x, self.dt)
            elif isinstance(l, Skip):
                # torch.cat([main(x), x], dim=1): main first, then the skip path (yfcc_2.py:31-38);
                # wikiart concatenates the other way round (wikiart_256.py:86-87)
                inner = self._run(l.main, p + ".main", x, mod)
                x, x1 = (x, inner) if self.spec.get("skip_first") else (inner, x)
        assert x1 is None
        return x

    def _mapping(self, t, clip_embed):
        dt, dev = self.dt, self.device
        n = t.shape[0]
        tdt = _hip.TORCH_DTYPE[dt]
        ce = clip_embed.to(device=dev, dtype=torch.float32).contiguous()
        cen = torch.empty_like(ce)
        call("pmi_l2norm_rows", ptr(ce), ptr(cen), n, ce.shape[1], float(ce.shape[1]) ** 0.5)      # cc12m_1.py:294
        ff = torch.empty((n, 2 * self.mtw.numel()), dtype=torch.float32, device=dev)
        call("pmi_fourier_features", ptr(t), ptr(self.mtw), ptr(ff), n, self.mtw.numel())
        if self.precise:
            m = self.m
            xin = torch.cat([cen, ff], dim=1)                                   # [N, 640] fp32 (layout only)
            h = ops.linear_f32(xin, *m["a0"], act=ACT_RELU)
            s = ops.linear_f32(xin, *m["askip"])
            z1 = ops.linear_f32(h, *m["a2"], act=ACT_RELU, residual=s)
            h = ops.linear_f32(z1, *m["b0"], act=ACT_RELU)
            cond = ops.linear_f32(h, *m["b2"], residual=z1)
            return ops.linear_f32(cond, self.mod_all, None)
        ce16, ff16 = torch.empty(cen.shape, dtype=tdt, device=dev), torch.empty(ff.shape, dtype=tdt, device=dev)
        call("pmi_cast_f32_to_16", ptr(cen), ptr(ce16), cen.numel(), ACT_NONE, dt)
        call("pmi_cast_f32_to_16", ptr(ff), ptr(ff16), ff.numel(), ACT_NONE, dt)
        m = self.m
        h = ops.igemm(ce16, m["a0"], a1=ff16, act=ACT_RELU)       # Linear on cat([clip_embed, fourier]) without the cat
        s = ops.igemm(ce16, m["askip"], a1=ff16)
        z1 = ops.igemm(h, m["a2"], act=ACT_RELU, residual=s)
        h = ops.igemm(z1, m["b0"], act=ACT_RELU)
        cond = ops.igemm(h, m["b2"], residual=z1)
        return ops.igemm(cond, self.mod_all, out_f32=True)           # every Modulation2d's (scale|shift) at once

    @torch.no_grad()
    def forward(self, images: torch.Tensor, t: torch.Tensor, clip_embed: Optional[torch.Tensor] = None) -> torch.Tensor:
        """images NCHW fp32 in [0,1]; t [N] float in (0,1]; returns v, NCHW fp32 [N,3,H,W]."""
        if not images.is_cuda:
            raise RuntimeError("VDiffEngine runs on a HIP device only (no CPU fallback)")
        dt, dev = self.dt, self.device
        images = images.float().contiguous()
        n, _, hh, ww = images.shape
        t = t.to(device=dev, dtype=torch.float32).contiguous()
        mod = None
        if self.cond:
            if clip_embed is None:
                raise ValueError("this model is CLIP-conditioned: clip_embed is required")
            mod = self._mapping(t, clip_embed)
        planes = torch.empty((n, 16), dtype=torch.float32, device=dev)
        tf = t
        if self.spec.get("t_input") == "log_snr":       # wikiart_256.py:288-292: features of log(alpha^2 / sigma^2), [N] scalars
            tf = torch.log(torch.cos(t * (torch.pi / 2)) ** 2 / torch.sin(t * (torch.pi / 2)) ** 2).contiguous()
        call("pmi_fourier_features", ptr(tf), ptr(self.tw), ptr(planes), n, 8)
        x = torch.empty((n, hh, ww, 48 if self.precise else 24), dtype=_hip.TORCH_DTYPE[dt], device=dev)
        call("pmi_prep_input", ptr(images), ptr(planes), 16, ptr(x), n, hh, ww, 24, dt)
        y = self._run(self.spec["net"], "net", x, mod)
        out = torch.empty((n, 3, hh, ww), dtype=torch.float32, device=dev)
        call("pmi_finish_output", ptr(y), y.shape[-1], ptr(out), n, hh, ww, 3)
        return out

    # ---- input gradient -----------------------------------------------------------------------------------
    # What autograd gives the reference when losses/velocity_diffusion.py:33-61 (guided_resample_) backpropagates a loss on the
    # denoised image to the noise: d loss / d images through the UNet, weights frozen.  forward_train() runs the same kernels as
    # forward() but keeps, per block, the tensors a backward needs (the post-ReLU outputs double as ReLU masks); backward() walks the
    # tape in reverse.  dX of a convolution is the forward convolution on transposed + flipped weights (packed lazily, once), so the
    # MFMA kernels, their fused residual add and the fragment-ordered weight paths are reused unchanged.
    def _check_backward_support(self):
        if self.precise:
            raise NotImplementedError("input gradient runs in the 16-bit modes (bf16 / f16)")

    def _wt(self, key, weight, cin_pad=None):
        """Packed weights of the input-gradient convolution of `weight` [Cout, Cin, k, k]: [Cin, Cout, k, k] with both taps flipped."""
        if key not in self.w:
            w = weight.detach().float().permute(1, 0, 2, 3)
            if w.shape[-1] == 3:
                w = w.flip(2, 3)
            self.w[key] = PackedLinear(w.contiguous(), None, self.dt, self.device, cin_pad=cin_pad)
        return self.w[key]

    def _res_train_cond(self, l: Res, p, x, x1, mod, tape):
        """cc12m_1.py:46-61: conv -> GroupNorm(1, C, affine=False) -> Modulation2d -> ReLU -> conv [-> the same again] + skip, with the
        pre-norm tensors and the post-ReLU tensors kept (norm inputs and ReLU masks of the backward)."""
        dt, w = self.dt, self.w
        ld = mod.stride(0)
        h = ops.igemm(x, w[p + ".c1"], a1=x1)
        hn = ops.group_norm(h, None, None, 1, dt, film=mod[:, l.mod1:], film_ld=ld, act=ACT_RELU)
        if l.last:
            skip = ops.igemm(x, w[p + ".skip"], a1=x1)
            y = ops.igemm(hn, w[p + ".c2"], residual=skip, out_f32=True)
            h2 = r2 = None
        else:
            h2 = ops.igemm(hn, w[p + ".c2"])
            r2 = ops.group_norm(h2, None, None, 1, dt, film=mod[:, l.mod2:], film_ld=ld, act=ACT_RELU)
            if l.cin != l.cout:
                y = ops.igemm(x, w[p + ".skip"], a1=x1, residual=r2)
            else:
                y = torch.empty_like(r2)
                call("pmi_add16", ptr(r2), ptr(x), ptr(y), r2.numel(), dt)
        tape.append(("res", l, p, hn, r2, x1 is not None, h, h2, mod))
        return y

    def _res_train(self, l: Res, p, x, x1, tape, mod=None):
        if self.cond:
            return self._res_train_cond(l, p, x, x1, mod, tape)
        dt, w = self.dt, self.w
        h1 = ops.igemm(x, w[p + ".c1"], a1=x1, act=ACT_RELU)
        if l.last:
            skip = ops.igemm(x, w[p + ".skip"], a1=x1)
            y = ops.igemm(h1, w[p + ".c2"], residual=skip, out_f32=True)
            r2 = None
        else:
            r2 = ops.igemm(h1, w[p + ".c2"], act=ACT_RELU)                      # kept apart from the skip path: its sign is the ReLU mask
            if l.cin != l.cout:
                y = ops.igemm(x, w[p + ".skip"], a1=x1, residual=r2)
            else:
                y = torch.empty_like(r2)
                call("pmi_add16", ptr(r2), ptr(x), ptr(y), r2.numel(), dt)
        tape.append(("res", l, p, h1, r2, x1 is not None, None, None, None))
        return y

    def _attn_train(self, l: Attn, p, x, tape):
        dt, w = self.dt, self.w
        n, hh, ww, c = x.shape
        d = self.spec.get("head_dim", 64)
        t, heads = hh * ww, c // d
        hn = x
        if self.spec.get("attn_norm", True):
            g, b = w[p + ".gn"]
            hn = ops.group_norm(x, g, b, 1, dt)
        qkv = ops.igemm(hn.view(n * t, c), w[p + ".qkv"])
        if d == 64:                                  # flash-style forward keeping the log-sum-exp (csrc/attn.hip)
            tp32 = (t + 31) // 32 * 32
            aws = torch.empty((6, n * heads, tp32, 64), dtype=x.dtype, device=x.device)
            lse = torch.empty((n * heads, tp32), dtype=torch.float32, device=x.device)
            a = torch.empty((n * t, c), dtype=x.dtype, device=x.device)
            call("pmi_vit_attn_fwd", ptr(qkv), ptr(aws), ptr(lse), ptr(a), n, t, heads, 64.0 ** -0.5, dt)
            saved = (aws, lse, a)
        else:                                        # other head dims (wikiart: 128): batched GEMMs, the softmax is kept
            a, pm = ops.attention_train(qkv.view(n, t, 3 * c), heads, dt)
            a = a.view(n * t, c)
            saved = (qkv, pm)
        y = ops.igemm(a, w[p + ".out"], residual=x.view(n * t, c)).view(n, hh, ww, c)
        tape.append(("attn", l, p, x, saved))
        return y

    def _run_train(self, prog, prefix, x, tape, mod=None):
        x1 = None
        for i, l in enumerate(prog):
            p = f"{prefix}.{i}"
            if isinstance(l, Res):
                x, x1 = self._res_train(l, p, x, x1, tape, mod), None
            elif isinstance(l, Attn):
                x = self._attn_train(l, p, x, tape)
            elif isinstance(l, Down):
                x = ops.avgpool2(x, self.dt)
                tape.append(("down",))
            elif isinstance(l, Up):
                x = ops.upsample_nearest2(x) if self.spec.get("up_mode") == "nearest" else ops.upsample_bilinear2(x, self.dt)
                tape.append(("up",))
            elif isinstance(l, Skip):
                inner_tape = []
                inner = self._run_train(l.main, p + ".main", x, inner_tape, mod)
                tape.append(("skip", inner_tape))
                # torch.cat([main(x), x], dim=1) (yfcc_2.py:31-38); wikiart concatenates the other way round (wikiart_256.py:86-87)
                x, x1 = (x, inner) if self.spec.get("skip_first") else (inner, x)
        assert x1 is None
        return x

    @torch.no_grad()
    def forward_train(self, images: torch.Tensor, t: torch.Tensor, clip_embed: Optional[torch.Tensor] = None):
        """As forward(), keeping what backward() needs.  Returns (v NCHW fp32, tape)."""
        self._check_backward_support()
        if not images.is_cuda:
            raise RuntimeError("VDiffEngine runs on a HIP device only (no CPU fallback)")
        dt, dev = self.dt, self.device
        images = images.float().contiguous()
        n, _, hh, ww = images.shape
        t = t.to(device=dev, dtype=torch.float32).contiguous()
        mod = None
        if self.cond:
            if clip_embed is None:
                raise ValueError("this model is CLIP-conditioned: clip_embed is required")
            mod = self._mapping(t, clip_embed)                 # no gradient flows to the conditioning: it only scales / shifts
        planes = torch.empty((n, 16), dtype=torch.float32, device=dev)
        tf = t
        if self.spec.get("t_input") == "log_snr":       # wikiart_256.py:288-292
            tf = torch.log(torch.cos(t * (torch.pi / 2)) ** 2 / torch.sin(t * (torch.pi / 2)) ** 2).contiguous()
        call("pmi_fourier_features", ptr(tf), ptr(self.tw), ptr(planes), n, 8)
        x = torch.empty((n, hh, ww, 24), dtype=_hip.TORCH_DTYPE[dt], device=dev)
        call("pmi_prep_input", ptr(images), ptr(planes), 16, ptr(x), n, hh, ww, 24, dt)
        tape = []
        y = self._run_train(self.spec["net"], "net", x, tape, mod)
        out = torch.empty((n, 3, hh, ww), dtype=torch.float32, device=dev)
        call("pmi_finish_output", ptr(y), y.shape[-1], ptr(out), n, hh, ww, 3)
        return out, tape

    def _mask(self, g, y):
        """g * (y > 0) in place: y is a post-ReLU tensor, its sign is the mask."""
        call("pmi_act_bwd", ptr(g), ptr(y), ptr(g), g.numel(), ACT_RELU, self.dt)
        return g

    def _film_norm_back(self, xpre, d, mod, off):
        """d (gradient wrt Modulation2d's output, ReLU mask already applied) -> gradient wrt the GroupNorm(1, C) input xpre."""
        n, hh, ww, c = xpre.shape
        if self._dmod is not None:
            self._cond_grad_layer(xpre, d, off)
        out = torch.empty_like(xpre)
        scale = mod[:, off:]                                                     # [N, >= C] view: (scale | shift) of this layer
        part = torch.empty((n, _hip.lib().pmi_gn1_bwd_partials(hh * ww, c), 4), dtype=torch.float64, device=xpre.device)
        call("pmi_gn1_bwd", ptr(xpre), ptr(d), scale.data_ptr(), mod.stride(0), 1.0, None, ptr(out), ptr(part), n, hh * ww, c, 1e-5, self.dt)
        return out

    # ---- gradient to the conditioning (cc12m_1: upstream velocity_diffusion.py:96-109 lets autograd reach `conditioning`) -----------------
    # Modulation2d (cc12m_1.py:33-43) is out = xhat * (1 + scale[n][c]) + shift[n][c] on xhat = GroupNorm(1, C)(x): with d = the (masked)
    # gradient wrt out, d shift = sum_p d and d scale = sum_p d xhat = r (sum_p d x - mu sum_p d).  The per-channel sums are one streaming pass
    # (pmi_gn_bwd_stats), the per-sample moments come from the forward statistics pass; all layers accumulate into one [N, sum 2C] row that
    # goes back through the mapping network (four tiny fp32 GEMMs on the exact-fp32 MFMA, recomputed forward for the ReLU masks).
    def _cond_grad_layer(self, xpre, d, off):
        n, hh, ww, c = xpre.shape
        hw, dev, dt = hh * ww, xpre.device, self.dt
        nchunk = max(1, min(hw // 8, (1024 + n - 1) // n))
        ws = torch.empty((n, nchunk, c, 2), dtype=torch.float32, device=dev)
        call("pmi_gn_stats", ptr(xpre), None, c, ptr(ws), n, hw, c, 1, nchunk, dt)
        fs = ws.double().sum(dim=(1, 2))                                   # [n, 2]: sum, sumsq over the sample (per-sample scalars: torch)
        cnt = float(hw * c)
        mu = fs[:, 0] / cnt
        r = 1.0 / torch.sqrt((fs[:, 1] / cnt - mu * mu).clamp_min(0.0) + 1e-5)
        one = torch.ones((n, c), dtype=torch.float32, device=dev)
        zero = torch.zeros((n, c), dtype=torch.float32, device=dev)
        wb = torch.empty((n, nchunk, c, 2), dtype=torch.float32, device=dev)
        call("pmi_gn_bwd_stats", ptr(xpre), None, c, ptr(d), ptr(one), ptr(zero), ACT_NONE, ptr(wb), n, hw, c, nchunk, dt)
        ab = wb.double().sum(dim=1)                                        # [n, c, 2]: sum_p d, sum_p d x
        self._dmod[:, off:off + c] += (r[:, None] * (ab[..., 1] - mu[:, None] * ab[..., 0])).float()
        self._dmod[:, off + c:off + 2 * c] += ab[..., 0].float()

    def _mapping_back(self, t, clip_embed, d_mod, sd):
        """d loss / d clip_embed from d loss / d (every layer's scale | shift): back through Modulation2d.layer, the two ResLinearBlocks
        (cc12m_1.py:19-31, 121-124) and F.normalize(clip_embed) * sqrt(D) (cc12m_1.py:294).  fp32 throughout."""
        dev = self.device
        key = "_map_f32"
        if key not in self.w:
            f = lambda k: sd[k].detach().float().to(dev).contiguous()
            self.w[key] = dict(w0=f("mapping.0.main.0.weight"), b0=f("mapping.0.main.0.bias"), w2=f("mapping.0.main.2.weight"), b2=f("mapping.0.main.2.bias"),
                               ws=f("mapping.0.skip.weight"), v0=f("mapping.1.main.0.weight"), c0=f("mapping.1.main.0.bias"),
                               v2=f("mapping.1.main.2.weight"), c2=f("mapping.1.main.2.bias"),
                               mod=torch.cat([sd[k].detach().float() for k in self._mod_keys], 0).to(dev).contiguous())
        m = self.w[key]
        n = t.shape[0]
        ce = clip_embed.to(device=dev, dtype=torch.float32).contiguous()
        dim = ce.shape[1]
        nrm = ce.norm(dim=1, keepdim=True).clamp_min(1e-12)
        chat = ce / nrm
        ff = torch.empty((n, 2 * self.mtw.numel()), dtype=torch.float32, device=dev)
        call("pmi_fourier_features", ptr(t), ptr(self.mtw), ptr(ff), n, self.mtw.numel())
        xin = torch.cat([chat * dim ** 0.5, ff], dim=1).contiguous()
        ha = ops.linear_f32(xin, m["w0"], m["b0"], act=ACT_RELU)
        u = ops.linear_f32(ha, m["w2"], m["b2"])                           # pre-ReLU of block 0's main path
        z1 = torch.relu(u) + ops.linear_f32(xin, m["ws"], None)
        hb = ops.linear_f32(z1.contiguous(), m["v0"], m["c0"], act=ACT_RELU)

        def back(g, w):                                                     # g [n, out] @ w [out, in] -> [n, in]
            out = torch.empty((n, w.shape[1]), dtype=torch.float32, device=dev)
            return ops.gemm_f32(g.contiguous(), w, out, M=n, N=w.shape[1], K=w.shape[0], lda=g.shape[1], ldb=w.shape[1], ldd=w.shape[1], trans_b=True)

        d_cond = back(d_mod, m["mod"])
        d_z1 = back(back(d_cond, m["v2"]) * (hb > 0), m["v0"]) + d_cond
        d_u = d_z1 * (u > 0)
        d_xin = back(back(d_u, m["w2"]) * (ha > 0), m["w0"]) + back(d_z1, m["ws"])
        d_cen = d_xin[:, :dim]
        return (dim ** 0.5 / nrm) * (d_cen - chat * (chat * d_cen).sum(dim=1, keepdim=True))

    def _res_back(self, rec, g, sd, first):
        _, l, p, h1, r2, two, hpre, h2pre, mod = rec
        j = 4 if self.cond else 2
        w2 = sd[p + f".main.{j}.weight"]
        w1 = sd[p + ".main.0.weight"]
        if l.last:
            d2 = g                                                               # [N,H,W,8]: 3 channels + padding
            c2t = self._wt(p + ".c2T", w2, cin_pad=8)
        else:
            d2 = self._mask(g.clone(), r2)
            if self.cond:
                d2 = self._film_norm_back(h2pre, d2, mod, l.mod2)
            c2t = self._wt(p + ".c2T", w2)
        dh1 = self._mask(ops.igemm(d2, c2t), h1)
        if self.cond:
            dh1 = self._film_norm_back(hpre, dh1, mod, l.mod1)
        has_skip = l.cin != l.cout
        skw = sd[p + ".skip.weight"] if has_skip else None
        spad = 8 if l.last else None
        if not two:
            gs = ops.igemm(g, self._wt(p + ".skipT", skw, cin_pad=spad)) if has_skip else g
            return ops.igemm(dh1, self._wt(p + ".c1T", w1), residual=gs, out_f32=first), None
        half = l.cin // 2                                                        # conv1 / skip read cat([main(x), x]): one dX per source
        outs = []
        for k in range(2):
            sl = slice(k * half, (k + 1) * half)
            gs = ops.igemm(g, self._wt(p + f".skipT{k}", skw[:, sl], cin_pad=spad))
            outs.append(ops.igemm(dh1, self._wt(p + f".c1T{k}", w1[:, sl]), residual=gs))
        return outs[0], outs[1]

    def _attn_back(self, rec, g, sd):
        _, l, p, x, saved = rec
        dt = self.dt
        n, hh, ww, c = x.shape
        d = self.spec.get("head_dim", 64)
        t, heads = hh * ww, c // d
        g2 = g.reshape(n * t, c)
        da = ops.igemm(g2, self._wt(p + ".outT", sd[p + ".out_proj.weight"]))
        if d == 64:
            aws, lse, a = saved
            tp32 = (t + 31) // 32 * 32
            bws = torch.empty((2, n * heads, tp32, 64), dtype=x.dtype, device=x.device)
            delta = torch.empty((n * heads, tp32), dtype=torch.float32, device=x.device)
            dqkv = torch.empty((n * t, 3 * c), dtype=x.dtype, device=x.device)
            call("pmi_vit_attn_bwd", ptr(aws), ptr(lse), ptr(a), ptr(da), ptr(bws), ptr(delta), ptr(dqkv), n, t, heads, 64.0 ** -0.5, dt)
        else:
            qkv, pm = saved
            dqkv = ops.attention_backward(qkv.view(n, t, 3 * c), pm, da.view(n, t, c), heads, dt).view(n * t, 3 * c)
        dhn = ops.igemm(dqkv, self._wt(p + ".qkvT", sd[p + ".qkv_proj.weight"]))
        gx = torch.empty_like(x)
        if self.spec.get("attn_norm", True):
            part = torch.empty((n, _hip.lib().pmi_gn1_bwd_partials(t, c), 4), dtype=torch.float64, device=x.device)
            call("pmi_gn1_bwd", ptr(x), ptr(dhn), ptr(self.w[p + ".gn"][0]), 0, 0.0, ptr(g2), ptr(gx), ptr(part), n, t, c, 1e-5, dt)
        else:                                        # no norm in front of the projections: the two paths just add
            call("pmi_add16", ptr(dhn), ptr(g2), ptr(gx), gx.numel(), dt)
        return gx

    def _back(self, tape, g, sd, outermost=False):
        """Gradient wrt the input of the block sequence recorded in `tape`, given g = gradient wrt its output."""
        g1 = None
        for idx in range(len(tape) - 1, -1, -1):
            rec = tape[idx]
            kind = rec[0]
            if kind == "res":
                assert g1 is None
                g, g1 = self._res_back(rec, g, sd, first=outermost and idx == 0)
            elif kind == "skip":
                assert g1 is not None, "a SkipBlock is always followed by the block that reads its concat"
                g_inner, g_x = (g1, g) if self.spec.get("skip_first") else (g, g1)
                gm = self._back(rec[1], g_inner, sd)                             # through main(x)
                out = torch.empty_like(gm)
                call("pmi_add16", ptr(gm), ptr(g_x), ptr(out), gm.numel(), self.dt)
                g, g1 = out, None
            else:
                assert g1 is None
                if kind == "attn":
                    g = self._attn_back(rec, g, sd)
                elif kind == "down":
                    n, h, w_, c = g.shape
                    out = torch.empty((n, 2 * h, 2 * w_, c), dtype=g.dtype, device=g.device)
                    call("pmi_avgpool2_bwd", ptr(g), ptr(out), n, 2 * h, 2 * w_, c, self.dt)
                    g = out
                elif kind == "up":
                    n, h, w_, c = g.shape
                    out = torch.empty((n, h // 2, w_ // 2, c), dtype=g.dtype, device=g.device)
                    call("pmi_upsample_nearest2_bwd" if self.spec.get("up_mode") == "nearest" else "pmi_upsample_bilinear2_bwd",
                         ptr(g), ptr(out), n, h // 2, w_ // 2, c, self.dt)
                    g = out
        assert g1 is None
        return g

    @torch.no_grad()
    def backward(self, tape, d_v: torch.Tensor, state_dict, cond_grad=None):
        """d loss / d images (NCHW fp32, images in [0, 1]) from d loss / d v (NCHW fp32 [N, 3, H, W]) and the tape of forward_train().
        `state_dict`: the model's parameters (reference key names) -- the transposed weight packings are built from it on first use.
        f16 engines scale the gradient by a power of two (largest incoming value -> 1) on the way in and back on the way out: image
        gradients of a CLIP loss are ~1e-6 and would flush to zero in f16; bf16 needs no scaling.
        cond_grad = (t [N], clip_embed [N, D]) of a conditioned net: also returns d loss / d clip_embed -> (d_images, d_clip_embed)."""
        self._check_backward_support()
        dev, dt = self.device, self.dt
        n, _, hh, ww = d_v.shape
        if cond_grad is not None:
            if not self.cond:
                raise ValueError("cond_grad: this net takes no conditioning")
            self._dmod = torch.zeros((n, sum(state_dict[k].shape[0] for k in self._mod_keys)), dtype=torch.float32, device=dev)
        scale = 1.0
        if dt == _hip.DT_F16:                         # keep the f16 gradient tensors in range: largest incoming value -> 1 (power of two: exact)
            amax = float(d_v.abs().max())
            if amax > 0.0 and amax == amax:
                scale = 2.0 ** max(-24, min(24, -int(torch.tensor(amax).log2().ceil())))
        g = torch.zeros((n, hh, ww, 8), dtype=_hip.TORCH_DTYPE[dt], device=dev)                # 3 channels + padding (layout only)
        g[..., :3] = (d_v.to(dev).float() * scale).permute(0, 2, 3, 1)
        sd = {k: v.detach() for k, v in state_dict.items()}
        gx = self._back(tape, g, sd, outermost=True)                                            # fp32 [N,H,W,20]: d / d (x, Fourier planes)
        out = torch.empty((n, 3, hh, ww), dtype=torch.float32, device=dev)
        call("pmi_finish_output", ptr(gx), gx.shape[-1], ptr(out), n, hh, ww, 3)
        if cond_grad is not None:
            d_mod, self._dmod = self._dmod / scale, None
            t_, ce_ = cond_grad
            d_ce = self._mapping_back(t_.to(device=dev, dtype=torch.float32).contiguous(), ce_, d_mod.contiguous(), sd)
            return out * (2.0 / scale), d_ce
        return out * (2.0 / scale)                                                              # x = 2 * images - 1

