"""TEST INFRASTRUCTURE (oracle side) -- error-budget sweep for the 16-bit / mixed precision modes of the ADM UNet engine.

The HIP engine differs from the fp32 reference only by where a tensor is rounded to the MFMA operand type (weights are exact).  This script
restates the oracle's forward (oracle/adm_unet.py, same reference lines) with an explicit rounding point for every tensor CLASS the engine
rounds, so that one class at a time can be promoted to full precision ON THE CPU and its share of the max-abs error read off -- the sweep
VERDICT r1/r2 asked for -- without spending GPU minutes.  Classes (engine/adm.py, csrc/conv_wd.hip):

  x0    the network input (pmi_prep_input writes 2*img-1 as 16 bit)
  xin   conv3x3 operands: SiLU(GroupNorm(.)) rounded as it is staged into LDS (conv_wd.hip store_piece), pooled operands of down blocks
  c1    conv1 output, stored 16 bit in front of GroupNorm 2
  h     the residual stream: every ResBlock / attention / plain conv output stored 16 bit
  sk    1x1 skip_connection output (residual operand of conv2), and its (un-normalised) input when pooled
  at    attention internals: GroupNorm output, qkv, softmax P, attention output
  emb   timestep embedding MLP operands

Usage:  python -m oracle.error_budget [--size 128] [--fmt f16|bf16] [--levels]
Prints max-abs / rms error against the all-fp32 run for: everything rounded, each class alone rounded, each class alone promoted.
"""
from __future__ import annotations

import argparse
import sys
import time

import torch
import torch.nn.functional as F

from . import adm_unet as O


class Policy:
    """Which tensor classes are rounded, optionally restricted to resolution levels (ds = 1, 2, 4, ...)."""

    def __init__(self, fmt="f16", classes=(), split=(), levels=None, else_split=False):
        self.dt = {"f16": torch.float16, "bf16": torch.bfloat16}[fmt]
        self.classes = set(classes)      # rounded once to the 16-bit type
        self.split = set(split)          # kept as hi + lo 16-bit pairs (what the precise kernels hold): rounded to ~22 bits
        self.levels = levels             # None = all; else set of ds values where `classes` applies (elsewhere: full precision)
        self.else_split = else_split     # ... or hi + lo pairs elsewhere
        self.ds = 1
        self.single = None               # set of conv names whose operand is a single 16-bit value (all other conv operands: hi + lo)
        self.seen = []
        self.hi_only = False             # single-operand convs read only the high halves of their (hi + lo) input
        self.plain_from = None           # ds >= plain_from: the plain 16-bit engine (every class rounded once), whatever the other fields say

    def r(self, x, cls, name=None):
        if self.plain_from is not None and self.ds >= self.plain_from and cls != "emb":
            if name is not None:
                self.seen.append((name, self.ds, tuple(x.shape)))
            return x.to(self.dt).float()
        if name is not None:
            self.seen.append((name, self.ds, tuple(x.shape)))
            if self.single is not None:          # explicit per-layer choice for the conv operands: single 16-bit if listed, hi + lo otherwise
                cls = "__single" if name in self.single else "__split"
        if cls == "__single":
            return x.to(self.dt).float()
        if cls == "__split" or cls in self.split or (cls in self.classes and self.levels is not None and self.ds not in self.levels and self.else_split):
            hi = x.to(self.dt).float()
            return hi + (x - hi).to(self.dt).float()
        if cls in self.classes and (self.levels is None or self.ds in self.levels):
            return x.to(self.dt).float()
        return x


def _gn(sd, p, x):
    return F.group_norm(x, 32, sd[p + ".weight"], sd[p + ".bias"], eps=1e-5)


def _res(sd, p, x, emb, cfg, P: Policy, up=False, down=False):
    xg = x
    if P.hi_only and P.single is not None and (p + ".conv1") in P.single:
        xg = x.to(P.dt).float()
    h = F.silu(_gn(sd, p + ".in_layers.0", xg))
    if up:
        h = F.interpolate(h, scale_factor=2, mode="nearest")
        x = F.interpolate(x, scale_factor=2, mode="nearest")
        P.ds //= 2
    elif down:
        h = F.avg_pool2d(h, 2)
        x = P.r(F.avg_pool2d(x, 2), "sk")
        P.ds *= 2
    h = P.r(h, "xin", p + ".conv1")
    h = F.conv2d(h, sd[p + ".in_layers.2.weight"], sd[p + ".in_layers.2.bias"], padding=1)
    e = F.linear(emb, sd[p + ".emb_layers.1.weight"], sd[p + ".emb_layers.1.bias"])[:, :, None, None]
    if cfg.use_scale_shift_norm:
        h = P.r(h, "c1")
        if P.hi_only and P.single is not None and (p + ".conv2") in P.single:
            h = h.to(P.dt).float()
        scale, shift = e.chunk(2, dim=1)
        h = F.silu(_gn(sd, p + ".out_layers.0", h) * (1 + scale) + shift)
    else:
        h = P.r(h + e, "c1")
        h = F.silu(_gn(sd, p + ".out_layers.0", h))
    h = P.r(h, "xin", p + ".conv2")
    h = F.conv2d(h, sd[p + ".out_layers.3.weight"], sd[p + ".out_layers.3.bias"], padding=1)
    if (p + ".skip_connection.weight") in sd:
        w = sd[p + ".skip_connection.weight"]
        x = P.r(F.conv2d(P.r(x, "skin"), w, sd[p + ".skip_connection.bias"], padding=w.shape[-1] // 2), "sk")
    return P.r(x + h, "h")


def _attn(sd, p, x, heads, new_order, P: Policy):
    b, c, hh, ww = x.shape
    xf = x.reshape(b, c, -1)
    hn = P.r(_gn(sd, p + ".norm", xf), "at")
    qkv = P.r(F.conv1d(hn, sd[p + ".qkv.weight"], sd[p + ".qkv.bias"]), "at")
    t = xf.shape[-1]
    ch = c // heads
    if new_order:
        q, k, v = qkv.chunk(3, dim=1)
        q, k, v = (z.reshape(b * heads, ch, t) for z in (q, k, v))
    else:
        q, k, v = qkv.reshape(b * heads, 3 * ch, t).split(ch, dim=1)
    s = ch ** -0.25
    w = P.r(torch.softmax(torch.einsum("bct,bcs->bts", q * s, k * s), dim=-1), "at")
    a = P.r(torch.einsum("bts,bcs->bct", w, v).reshape(b, c, t), "at")
    a = F.conv1d(a, sd[p + ".proj_out.weight"], sd[p + ".proj_out.bias"])
    return P.r((xf + a).reshape(b, c, hh, ww), "h")


def _run(sd, cfg, layers, h, emb, P):
    for p, kind, kw in layers:
        if kind == "conv":
            h = P.r(F.conv2d(P.r(h, "x0"), sd[p + ".weight"], sd[p + ".bias"], padding=1), "h")
        elif kind == "res":
            h = _res(sd, p, h, emb, cfg, P, **kw)
        elif kind == "attn":
            h = _attn(sd, p, h, kw["heads"], cfg.use_new_attention_order, P)
        else:
            raise NotImplementedError(kind)
    return h


@torch.no_grad()
def forward(sd, cfg, x, t, P: Policy):
    inp, mid, out = O.block_plan(cfg)
    P.ds = 1
    emb = P.r(O.timestep_embedding(t, cfg.model_channels), "emb")
    emb = P.r(F.silu(F.linear(emb, sd["time_embed.0.weight"], sd["time_embed.0.bias"])), "emb")
    emb = P.r(F.silu(F.linear(emb, sd["time_embed.2.weight"], sd["time_embed.2.bias"])), "emb")
    h, hs = x.float(), []
    for layers in inp:
        h = _run(sd, cfg, layers, h, emb, P)
        hs.append(h)
    h = _run(sd, cfg, mid, h, emb, P)
    for layers in out:
        h = torch.cat([h, hs.pop()], dim=1)
        h = _run(sd, cfg, layers, h, emb, P)
    h = P.r(F.silu(_gn(sd, "out.0", h)), "xin", "out")
    return F.conv2d(h, sd["out.2.weight"], sd["out.2.bias"], padding=1)


CLASSES = ("x0", "xin", "c1", "h", "sk", "at", "emb")


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=128)
    ap.add_argument("--fmt", default="f16")
    ap.add_argument("--levels", action="store_true", help="also sweep the xin class per resolution level")
    ap.add_argument("--t", type=int, default=500)
    ap.add_argument("--per-conv", action="store_true", help="variance share and FLOPs of every conv3x3 operand; greedy choice of the layers to keep single")
    ap.add_argument("--budget", type=float, default=1.0e-4, help="rms error budget of the mixed mode (--per-conv)")
    a = ap.parse_args(argv)
    sys.path.insert(0, ".")
    from perceptor_amd.utils.synth import seeded_noise, synth_state_dict
    cfg = O.openimages_config()
    sd = {k: v.float() for k, v in synth_state_dict(O.state_dict_shapes(cfg), 0).items()}
    x = seeded_noise((1, 3, a.size, a.size), 1234)
    t = torch.tensor([a.t])
    t0 = time.time()
    ref = forward(sd, cfg, x, t, Policy(a.fmt))
    print(f"# GD standard @{a.size}, {a.fmt}; fp32 run {time.time() - t0:.1f} s; max|eps| {float(ref.abs().max()):.3f}", flush=True)

    def rep(tag, P):
        y = forward(sd, cfg, x, t, P)
        d = y - ref
        print(f"{tag:34s} max-abs {float(d.abs().max()):.3e}   rms {float(d.pow(2).mean().sqrt()):.3e}", flush=True)

    rep("all rounded", Policy(a.fmt, CLASSES))
    for c in CLASSES:
        rep(f"only {c} rounded", Policy(a.fmt, (c,)))
    for c in CLASSES:
        rep(f"all but {c} (promoted to fp32)", Policy(a.fmt, [k for k in CLASSES if k != c]))
    rep("storage (c1, h, sk) as hi+lo pairs", Policy(a.fmt, ("x0", "xin", "at", "emb"), split=("c1", "h", "sk")))
    rep("storage hi+lo, xin only rounded", Policy(a.fmt, ("xin",), split=("c1", "h", "sk", "x0", "at", "emb")))
    allds = [2 ** i for i in range(len(cfg.channel_mult))]
    for lo in (2, 4, 8):     # the candidate "mixed" modes: everything stored as hi + lo pairs, conv operands hi + lo on the levels below `lo`, single 16-bit above
        rep(f"mixed: xin single 16-bit at ds>={lo}", Policy(a.fmt, ("xin",), split=("c1", "h", "sk", "x0", "at", "emb"),
                                                           levels={d for d in allds if d >= lo}, else_split=True))
    if a.per_conv:
        P0 = Policy(a.fmt)
        forward(sd, cfg, x, t, P0)
        convs = P0.seen
        STO = ("c1", "h", "sk", "x0", "at", "emb")
        rows = []
        for name, ds, shp in convs:
            P = Policy(a.fmt, split=STO)
            P.single = {name}
            d = forward(sd, cfg, x, t, P) - ref
            wkey = {"out": "out.2.weight"}.get(name) or (name[:-6] + (".in_layers.2.weight" if name.endswith("conv1") else ".out_layers.3.weight"))
            w = sd[wkey]
            hw = shp[2] * shp[3]
            fl = 2.0 * hw * w.shape[0] * w.shape[1] * 9
            rows.append((name, ds, shp[1], w.shape[0], fl, float(d.pow(2).mean())))
        tot = sum(r[4] for r in rows)
        rows.sort(key=lambda r: r[5] / r[4])          # least variance per FLOP first: these stay single
        var, keep, fkeep = 0.0, [], 0.0
        print(f"{'conv':34s} ds  cin->cout   GFLOP   var      cum-rms   kept-single-FLOP-share")
        for name, ds, ci, co, fl, v in rows:
            ok = (var + v) ** 0.5 <= a.budget
            if ok:
                var += v; keep.append(name); fkeep += fl
            print(f"{name:34s} {ds:2d} {ci:5d}->{co:4d} {fl / 1e9:7.2f} {v:.2e} {(var ** 0.5):.2e} {fkeep / tot:6.3f} {'single' if ok else 'hi+lo'}")
        P = Policy(a.fmt, split=STO)
        P.single = set(keep)
        rep(f"greedy mixed ({len(keep)}/{len(rows)} single, {fkeep / tot:.3f} of FLOPs)", P)
        print("SINGLE =", sorted(keep))
        P = Policy(a.fmt, ("skin",), split=STO)
        P.single = set(keep)
        rep("  + skip_connection 1x1 operand single", P)
        for pf in (64, 32, 16, 8, 4):
            P = Policy(a.fmt, split=STO)
            P.single = set(keep)
            P.plain_from = pf
            rep(f"  + plain 16-bit engine at ds >= {pf}", P)
        P = Policy(a.fmt, split=STO)
        P.single = set(keep)
        P.hi_only = True
        rep("  + single convs read hi only", P)
    if a.levels:
        ds = 1
        while ds <= 2 ** (len(cfg.channel_mult) - 1):
            rep(f"only xin @ds={ds}", Policy(a.fmt, ("xin",), levels={ds}))
            ds *= 2


if __name__ == "__main__":
    main()
