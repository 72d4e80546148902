"""TEST INFRASTRUCTURE (oracle) — CPU fp32 restatement of the v-diffusion UNets.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this file; the product path never does.

Mirrors
  perceptor/models/velocity_diffusion/yfcc_2.py:17-28 (ResConvBlock), :41-49
      (FourierFeatures), :52-70 (SelfAttention2d), :77-249 (YFCC2Model)
  perceptor/models/velocity_diffusion/cc12m_1.py:19-30 (ResLinearBlock), :33-43
      (Modulation2d), :46-61 (ResModConvBlock), :112-302 (CC12M1Model)
The nested nn.Sequential of the reference is described by a recursive spec
(lists of tuples); state-dict key names follow from list positions.
Pinned against tests/golden/vdiff_*.npz.
"""
from __future__ import annotations

import math
from typing import Dict, List, Tuple

import torch
import torch.nn.functional as F


def _level_spec(cs, i, n_inner, attn_from, blocks_per_side, innermost_blocks):
    """Layers inside the SkipBlock that runs at resolution level ``i`` (1-based)."""
    L: List[tuple] = [("down",)]
    att = i >= attn_from
    last = len(cs) - 1

    def add(cin, cmid, cout):
        L.append(("res", cin, cmid, cout, False))
        if att:
            L.append(("attn", cout))

    if i < last:
        add(cs[i - 1], cs[i], cs[i])
        for _ in range(blocks_per_side - 1):
            add(cs[i], cs[i], cs[i])
        L.append(("skip", _level_spec(cs, i + 1, n_inner, attn_from, blocks_per_side, innermost_blocks)))
        add(cs[i] * 2, cs[i], cs[i])
        for _ in range(blocks_per_side - 2):
            add(cs[i], cs[i], cs[i])
        add(cs[i], cs[i], cs[i - 1])
    else:
        add(cs[i - 1], cs[i], cs[i])
        for _ in range(innermost_blocks - 2):
            add(cs[i], cs[i], cs[i])
        add(cs[i], cs[i], cs[i - 1])
    L.append(("up",))
    return L


def yfcc2_spec():
    c = 256
    cs = [c // 2, c, c * 2, c * 2, c * 4, c * 4, c * 8, c * 8]
    inner = _level_spec(cs, 1, None, 5, 2, 4)
    return dict(name="yfcc_2", shape=(3, 512, 512), cond=False, cs=cs, net=[
        ("res", 3 + 16, cs[0], cs[0], False), ("res", cs[0], cs[0], cs[0], False),
        ("skip", inner),
        ("res", cs[0] * 2, cs[0], cs[0], False), ("res", cs[0], cs[0], 3, True)])


def cc12m1_spec():
    c = 128
    cs = [c, c * 2, c * 2, c * 4, c * 4, c * 8, c * 8]
    inner = _level_spec(cs, 1, None, 4, 4, 8)
    return dict(name="cc12m_1", shape=(3, 256, 256), cond=True, cs=cs, feats=1024, net=[
        ("res", 3 + 16, cs[0], cs[0], False), ("res", cs[0], cs[0], cs[0], False),
        ("res", cs[0], cs[0], cs[0], False), ("res", cs[0], cs[0], cs[0], False),
        ("skip", inner),
        ("res", cs[0] * 2, cs[0], cs[0], False), ("res", cs[0], cs[0], cs[0], False),
        ("res", cs[0], cs[0], cs[0], False), ("res", cs[0], cs[0], 3, True)])


def yfcc1_spec():   # yfcc_1.py:78-336
    c = 128
    cs = [c, c, c * 2, c * 2, c * 4, c * 4, c * 8, c * 8]
    inner = _level_spec(cs, 1, None, 5, 4, 8)
    return dict(name="yfcc_1", shape=(3, 512, 512), cond=False, cs=cs, net=[
        ("res", 3 + 16, cs[0], cs[0], False)] + [("res", cs[0], cs[0], cs[0], False)] * 3 + [
        ("skip", inner),
        ("res", cs[0] * 2, cs[0], cs[0], False), ("res", cs[0], cs[0], cs[0], False), ("res", cs[0], cs[0], cs[0], False),
        ("res", cs[0], cs[0], 3, True)])


def wikiart_spec():  # wikiart_256.py:105-291
    c = 128
    cs = [c // 2, c, c * 2, c * 2, c * 4, c * 4, c * 8]
    inner = _level_spec(cs, 1, None, 4, 4, 8)
    return dict(name="wikiart", shape=(3, 256, 256), cond=False, cs=cs, head_dim=128, attn_norm=False, up_mode="nearest",
                t_input="log_snr", skip_first=True, net=[
        ("res", 3 + 16, cs[0], cs[0], False)] + [("res", cs[0], cs[0], cs[0], False)] * 3 + [
        ("skip", inner),
        ("res", cs[0] * 2, cs[0], cs[0], False), ("res", cs[0], cs[0], cs[0], False), ("res", cs[0], cs[0], cs[0], False),
        ("res", cs[0], cs[0], 3, True)])


def tiny_spec(cond: bool, c: int = 64):
    """Small net of the same family (2 levels + attention) for fast full-tensor parity."""
    cs = [c, c * 2, c * 2]
    inner = _level_spec(cs, 1, None, 1, 2, 4)
    d = dict(name="tiny_cond" if cond else "tiny", shape=(3, 32, 32), cond=cond, cs=cs, net=[
        ("res", 3 + 16, cs[0], cs[0], False), ("res", cs[0], cs[0], cs[0], False),
        ("skip", inner),
        ("res", cs[0] * 2, cs[0], cs[0], False), ("res", cs[0], cs[0], 3, True)])
    if cond:
        d["feats"] = 1024
    return d


def state_dict_shapes(spec) -> Dict[str, Tuple[int, ...]]:
    S: Dict[str, Tuple[int, ...]] = {}
    cond = spec["cond"]

    def walk(layers, prefix):
        for idx, l in enumerate(layers):
            p = f"{prefix}.{idx}"
            if l[0] == "res":
                _, cin, cmid, cout, last = l
                S[p + ".main.0.weight"] = (cmid, cin, 3, 3); S[p + ".main.0.bias"] = (cmid,)
                j = 4 if cond else 2
                S[p + f".main.{j}.weight"] = (cout, cmid, 3, 3); S[p + f".main.{j}.bias"] = (cout,)
                if cond:
                    S[p + ".main.2.layer.weight"] = (2 * cmid, spec["feats"])
                    if not last:
                        S[p + ".main.6.layer.weight"] = (2 * cout, spec["feats"])
                if cin != cout:
                    S[p + ".skip.weight"] = (cout, cin, 1, 1)
            elif l[0] == "attn":
                c = l[1]
                if spec.get("attn_norm", True):
                    S[p + ".norm.weight"] = (c,); S[p + ".norm.bias"] = (c,)
                S[p + ".qkv_proj.weight"] = (3 * c, c, 1, 1); S[p + ".qkv_proj.bias"] = (3 * c,)
                S[p + ".out_proj.weight"] = (c, c, 1, 1); S[p + ".out_proj.bias"] = (c,)
            elif l[0] == "skip":
                walk(l[1], p + ".main")

    if cond:
        S["mapping_timestep_embed.weight"] = (64, 1)
        f = spec["feats"]
        S["mapping.0.main.0.weight"] = (f, 512 + 128); S["mapping.0.main.0.bias"] = (f,)
        S["mapping.0.main.2.weight"] = (f, f); S["mapping.0.main.2.bias"] = (f,)
        S["mapping.0.skip.weight"] = (f, 512 + 128)
        S["mapping.1.main.0.weight"] = (f, f); S["mapping.1.main.0.bias"] = (f,)
        S["mapping.1.main.2.weight"] = (f, f); S["mapping.1.main.2.bias"] = (f,)
    S["timestep_embed.weight"] = (8, 1)
    walk(spec["net"], "net")
    return S


def fourier_features(t, weight):
    # yfcc_2.py:41-49 / cc12m_1.py:74-84 : [cos f | sin f], f = 2*pi*t*W^T
    f = 2 * math.pi * t[:, None].float() @ weight.T
    return torch.cat([f.cos(), f.sin()], dim=-1)


def _attn(sd, p, x, heads):
    # yfcc_2.py:62-70 ; wikiart_256.py:61-77 has no norm
    n, c, h, w = x.shape
    xn = F.group_norm(x, 1, sd[p + ".norm.weight"], sd[p + ".norm.bias"], eps=1e-5) if (p + ".norm.weight") in sd else x
    qkv = F.conv2d(xn, sd[p + ".qkv_proj.weight"], sd[p + ".qkv_proj.bias"])
    qkv = qkv.view(n, heads * 3, c // heads, h * w).transpose(2, 3)
    q, k, v = qkv.chunk(3, dim=1)
    s = k.shape[3] ** -0.25
    att = ((q * s) @ (k.transpose(2, 3) * s)).softmax(3)
    y = (att @ v).transpose(2, 3).contiguous().view(n, c, h, w)
    return x + F.conv2d(y, sd[p + ".out_proj.weight"], sd[p + ".out_proj.bias"])


def _mod(sd, key, x, cond):
    # cc12m_1.py:39-43 : scales first, then shifts
    scales, shifts = F.linear(cond, sd[key]).chunk(2, dim=-1)
    return torch.addcmul(shifts[..., None, None], x, scales[..., None, None] + 1)


def _res(sd, p, x, l, cond_vec):
    _, cin, cmid, cout, last = l
    if cond_vec is None:
        h = F.relu(F.conv2d(x, sd[p + ".main.0.weight"], sd[p + ".main.0.bias"], padding=1))
        h = F.conv2d(h, sd[p + ".main.2.weight"], sd[p + ".main.2.bias"], padding=1)
        if not last:
            h = F.relu(h)
    else:
        h = F.conv2d(x, sd[p + ".main.0.weight"], sd[p + ".main.0.bias"], padding=1)
        h = F.relu(_mod(sd, p + ".main.2.layer.weight", F.group_norm(h, 1, eps=1e-5), cond_vec))
        h = F.conv2d(h, sd[p + ".main.4.weight"], sd[p + ".main.4.bias"], padding=1)
        if not last:
            h = F.relu(_mod(sd, p + ".main.6.layer.weight", F.group_norm(h, 1, eps=1e-5), cond_vec))
    s = x if cin == cout else F.conv2d(x, sd[p + ".skip.weight"])
    return h + s


def _walk(sd, layers, prefix, x, cond_vec, head_dim=64, up_mode="bilinear", skip_first=False):
    for idx, l in enumerate(layers):
        p = f"{prefix}.{idx}"
        if l[0] == "res":
            x = _res(sd, p, x, l, cond_vec)
        elif l[0] == "attn":
            x = _attn(sd, p, x, l[1] // head_dim)
        elif l[0] == "down":
            x = F.avg_pool2d(x, 2)
        elif l[0] == "up":
            x = F.interpolate(x, scale_factor=2, mode="nearest") if up_mode == "nearest" else \
                F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=False)
        elif l[0] == "skip":
            inner = _walk(sd, l[1], p + ".main", x, cond_vec, head_dim, up_mode, skip_first)
            # yfcc_2.py:37-38 cat([main, skip]); wikiart_256.py:86-87 cat([skip, main])
            x = torch.cat([x, inner], dim=1) if skip_first else torch.cat([inner, x], dim=1)
    return x


def mapping_cond(sd, t, clip_embed):
    # cc12m_1.py:293-298 and :19-30
    ce = F.normalize(clip_embed.float(), dim=-1) * clip_embed.shape[-1] ** 0.5
    z = torch.cat([ce, fourier_features(t, sd["mapping_timestep_embed.weight"])], dim=1)
    h = F.relu(F.linear(z, sd["mapping.0.main.0.weight"], sd["mapping.0.main.0.bias"]))
    h = F.relu(F.linear(h, sd["mapping.0.main.2.weight"], sd["mapping.0.main.2.bias"]))
    z = h + F.linear(z, sd["mapping.0.skip.weight"])
    h = F.relu(F.linear(z, sd["mapping.1.main.0.weight"], sd["mapping.1.main.0.bias"]))
    h = F.linear(h, sd["mapping.1.main.2.weight"], sd["mapping.1.main.2.bias"])
    return h + z


@torch.no_grad()
def vdiff_forward(sd, spec, x, t, clip_embed=None):
    sd = {k: v.float() for k, v in sd.items()}
    cond_vec = mapping_cond(sd, t, clip_embed) if spec["cond"] else None
    tf = t.float()
    if spec.get("t_input") == "log_snr":   # wikiart_256.py:288-292
        tf = torch.log(torch.cos(tf * math.pi / 2) ** 2 / torch.sin(tf * math.pi / 2) ** 2)
    te = fourier_features(tf, sd["timestep_embed.weight"])
    planes = te[..., None, None].repeat(1, 1, x.shape[2], x.shape[3])
    return _walk(sd, spec["net"], "net", torch.cat([x.float(), planes], dim=1), cond_vec,
                 spec.get("head_dim", 64), spec.get("up_mode", "bilinear"), spec.get("skip_first", False))
