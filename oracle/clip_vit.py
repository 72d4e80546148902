"""TEST INFRASTRUCTURE (oracle) — CPU fp32 restatement of the CLIP image path.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this file; the product path never does.

The ViT arithmetic lives in the third-party open-clip-torch 2.0.2
(poetry.lock:1321-1322), which is not vendored and not installed.  Its visual
tower follows the OpenAI-CLIP VisionTransformer, of which the reference holds an
in-tree copy: perceptor/models/ruclip/model.py:11-131.  This file restates
  ruclip/model.py:72-131  (VisionTransformer.forward: patch conv, cls+pos, ln_pre,
                           blocks, ln_post on cls token, @ proj)
  ruclip/model.py:27-58   (ResidualAttentionBlock: x+attn(ln_1 x); x+mlp(ln_2 x))
  ruclip/model.py:20-23   (QuickGELU); laion weights use exact GELU (models/clip.py:21-27)
  perceptor/models/open_clip.py:109-123  (resize -> Normalize -> encode_image -> F.normalize)
  perceptor/transforms/resize/resize_right.py:34-189 (+interpolation_methods.py:38-76)
  perceptor/losses/clip/clip.py:89-99 and losses/open_clip.py:87-97 (spherical loss)
Differentiable (plain torch ops), so the guidance gradient is autograd of this file.
Pinned against tests/golden/clip_*.npz (reference ruclip ViT + reference resize);
parity of open_clip's own code is unpinned (package absent) — see DESIGN.md.
"""
from __future__ import annotations

from math import ceil, pi
from typing import Dict, Tuple

import torch
import torch.nn.functional as F

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)   # ruclip/processor.py:23-24
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)

VIT_CONFIGS = {
    # name: (image, patch, width, layers, heads, out_dim)
    "ViT-B-32": (224, 32, 768, 12, 12, 512),
    "ViT-B-16": (224, 16, 768, 12, 12, 512),
    "ViT-L-14": (224, 14, 1024, 24, 16, 768),
    "ViT-H-14": (224, 14, 1280, 32, 16, 1024),
    "tiny": (32, 8, 64, 2, 1, 32),
    "tiny-odd": (28, 14, 128, 2, 2, 48),
}


def vit_state_dict_shapes(cfg) -> Dict[str, Tuple[int, ...]]:
    res, patch, width, layers, heads, out = cfg
    S = {"conv1.weight": (width, 3, patch, patch), "class_embedding": (width,),
         "positional_embedding": ((res // patch) ** 2 + 1, width),
         "ln_pre.weight": (width,), "ln_pre.bias": (width,),
         "ln_post.weight": (width,), "ln_post.bias": (width,), "proj": (width, out)}
    for i in range(layers):
        p = f"transformer.resblocks.{i}."
        S[p + "attn.in_proj_weight"] = (3 * width, width); S[p + "attn.in_proj_bias"] = (3 * width,)
        S[p + "attn.out_proj.weight"] = (width, width); S[p + "attn.out_proj.bias"] = (width,)
        S[p + "ln_1.weight"] = (width,); S[p + "ln_1.bias"] = (width,)
        S[p + "mlp.c_fc.weight"] = (4 * width, width); S[p + "mlp.c_fc.bias"] = (4 * width,)
        S[p + "mlp.c_proj.weight"] = (width, 4 * width); S[p + "mlp.c_proj.bias"] = (width,)
        S[p + "ln_2.weight"] = (width,); S[p + "ln_2.bias"] = (width,)
    return S


def vit_forward(sd, cfg, x, quick_gelu: bool):
    res, patch, width, layers, heads, out = cfg
    n = x.shape[0]
    x = F.conv2d(x, sd["conv1.weight"], stride=patch)
    x = x.reshape(n, width, -1).permute(0, 2, 1)
    cls = sd["class_embedding"][None, None, :].expand(n, 1, width)
    x = torch.cat([cls, x], dim=1) + sd["positional_embedding"]
    x = F.layer_norm(x, (width,), sd["ln_pre.weight"], sd["ln_pre.bias"], 1e-5)
    t = x.shape[1]
    d = width // heads
    for i in range(layers):
        p = f"transformer.resblocks.{i}."
        h = F.layer_norm(x, (width,), sd[p + "ln_1.weight"], sd[p + "ln_1.bias"], 1e-5)
        qkv = F.linear(h, sd[p + "attn.in_proj_weight"], sd[p + "attn.in_proj_bias"])
        q, k, v = (z.reshape(n, t, heads, d).transpose(1, 2) for z in qkv.chunk(3, dim=-1))
        a = torch.softmax((q * d ** -0.5) @ k.transpose(-1, -2), dim=-1) @ v
        a = a.transpose(1, 2).reshape(n, t, width)
        x = x + F.linear(a, sd[p + "attn.out_proj.weight"], sd[p + "attn.out_proj.bias"])
        h = F.layer_norm(x, (width,), sd[p + "ln_2.weight"], sd[p + "ln_2.bias"], 1e-5)
        h = F.linear(h, sd[p + "mlp.c_fc.weight"], sd[p + "mlp.c_fc.bias"])
        h = h * torch.sigmoid(1.702 * h) if quick_gelu else F.gelu(h)
        x = x + F.linear(h, sd[p + "mlp.c_proj.weight"], sd[p + "mlp.c_proj.bias"])
    x = F.layer_norm(x[:, 0, :], (width,), sd["ln_post.weight"], sd["ln_post.bias"], 1e-5)
    return x @ sd["proj"]


# ---- ResizeRight (antialiased, zero pad): dense 1-D operator ----------------
def _lanczos3(x):
    eps = torch.finfo(torch.float32).eps
    return ((torch.sin(pi * x) * torch.sin(pi * x / 3) + eps) / ((pi**2 * x**2 / 3) + eps)) * (x.abs() < 3).to(x.dtype)


def _cubic(x):
    a = x.abs(); a2 = a**2; a3 = a**3
    return (1.5 * a3 - 2.5 * a2 + 1.0) * (a <= 1.0).to(x.dtype) + \
        (-0.5 * a3 + 2.5 * a2 - 4.0 * a + 2.0) * ((1.0 < a) & (a <= 2.0)).to(x.dtype)


def resize_matrix(in_sz: int, out_sz: int, method: str) -> torch.Tensor:
    """[out_sz, in_sz] matrix equal to one 1-D pass of resize_right.resize.

    resize_right.py:198-207 (projected grid), :210-219 (field of view),
    :222-234 (generalised zero padding == dropping out-of-range taps),
    :275-285 (weights normalised to sum 1 *including* out-of-range taps),
    :463-472 (antialiasing stretches the kernel by the scale when shrinking).
    """
    fn, support = {"lanczos3": (_lanczos3, 6), "cubic": (_cubic, 4)}[method]
    scale = out_sz / in_sz
    eps = torch.finfo(torch.float32).eps
    grid = torch.arange(out_sz) / float(scale) + (in_sz - 1) / 2 - (out_sz - 1) / (2 * float(scale))
    if scale < 1.0:
        cur_support = support / scale
        f = lambda a: scale * fn(scale * a)
    else:
        cur_support, f = support, fn
    left = (grid - cur_support / 2 - eps).ceil().long()
    fov = left[:, None] + torch.arange(ceil(cur_support - eps))
    pad0 = -int(fov[0, 0])           # calc_pad_sz shifts both by the left pad before the kernel is evaluated
    w = f((grid + pad0)[:, None] - (fov + pad0))
    s = w.sum(1, keepdim=True)
    s[s == 0] = 1
    w = w / s
    m = torch.zeros(out_sz, in_sz)
    valid = (fov >= 0) & (fov < in_sz)
    rows = torch.arange(out_sz)[:, None].expand_as(fov)
    m.index_put_((rows[valid], fov[valid]), w[valid], accumulate=True)
    return m


def resize(images, out_hw):
    """Default-argument resize_right.resize: lanczos3 when shrinking both dims, else bicubic
    (resize_right.py:102-108); dims processed in order of increasing scale factor (:113-117)."""
    h, w = images.shape[-2:]
    oh, ow = out_hw
    method = "lanczos3" if (h >= oh and w >= ow) else "cubic"
    dims = sorted([(oh / h, -2, h, oh), (ow / w, -1, w, ow)], key=lambda z: (z[0],))
    out = images
    for sf, dim, i, o in dims:
        if sf == 1.0:
            continue
        m = resize_matrix(i, o, method).to(images.dtype)
        out = torch.einsum("oi,nciw->ncow", m, out) if dim == -2 else torch.einsum("oi,nchi->ncho", m, out)
    return out


def encode_images(sd, cfg, images, quick_gelu, normalize=True):
    res = cfg[0]
    x = resize(images, (res, res))
    mean = torch.tensor(CLIP_MEAN, dtype=x.dtype)[None, :, None, None]
    std = torch.tensor(CLIP_STD, dtype=x.dtype)[None, :, None, None]
    e = vit_forward(sd, cfg, (x - mean) / std, quick_gelu)
    return F.normalize(e) if normalize else e


def spherical_loss(image_enc, target_enc, weights, multiplier=1.0):
    d = (image_enc[:, None] - target_enc[None, :]).norm(dim=2).div(2).arcsin().square().mul(2)
    return (d * weights).mean() * multiplier


def clip_loss_and_grad(sd, cfg, images, target_enc, weights, quick_gelu, multiplier=1.0):
    images = images.detach().clone().requires_grad_(True)
    with torch.enable_grad():
        loss = spherical_loss(encode_images(sd, cfg, images, quick_gelu), target_enc, weights, multiplier)
        (g,) = torch.autograd.grad(loss, images)
    return loss.detach(), g
