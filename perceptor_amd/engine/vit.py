"""HIP engine for the CLIP ViT image tower: forward and gradient w.r.t. the input image.

Replaces open_clip's ``model.encode_image`` + autograd on the guidance path
(perceptor/models/open_clip.py:109-123; tower = OpenAI-CLIP VisionTransformer, in-tree copy
perceptor/models/ruclip/model.py:72-131).  Only the *input* gradient is needed
(weights are frozen, models/open_clip.py:75-76), so backward is one dX = dY.W GEMM per linear,
LayerNorm / softmax / activation input-grads and the attention score products — all on pmi_igemm
and the kernels of csrc/clip.hip.  The residual stream and its gradient stay fp32 in HBM; GEMM
operands are 16-bit (bf16 by default: image gradients are ~1e-6 and would flush in fp16).

State-dict keys follow open_clip / OpenAI-CLIP ``visual.*``: conv1.weight, class_embedding,
positional_embedding, ln_pre, transformer.resblocks.{i}.{ln_1,attn.in_proj_*,attn.out_proj,ln_2,mlp.c_fc,mlp.c_proj},
ln_post, proj.
"""
from __future__ import annotations

from typing import Dict, Tuple

import torch

from .. import _hip
from .._hip import ACT_GELU, ACT_NONE, ACT_QUICKGELU, call, ptr
from ..transforms.resize import resize as _resize, resize_backward as _resize_backward
from . import ops
from .ops import PackedLinear

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)   # OpenAI-CLIP constants (reference: ruclip/processor.py:23-24)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)

VIT_CONFIGS = {
    # name: (image, patch, width, layers, heads, out_dim)
    "ViT-B-32": (224, 32, 768, 12, 12, 512),
    "ViT-B-16": (224, 16, 768, 12, 12, 512),
    "ViT-L-14": (224, 14, 1024, 24, 16, 768),
    "ViT-H-14": (224, 14, 1280, 32, 16, 1024),
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


def from_hf_vision_state_dict(hf: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """transformers' CLIPModel / CLIPVisionModelWithProjection keys (``vision_model.encoder.layers.{i}.self_attn.q_proj...``,
    ``visual_projection.weight``; the form models/transformers_openai_clip.py:58-65 holds) -> the OpenAI-CLIP names this engine packs."""
    g = lambda k: hf[k].detach().float()
    out = {"class_embedding": g("vision_model.embeddings.class_embedding"), "conv1.weight": g("vision_model.embeddings.patch_embedding.weight"),
           "positional_embedding": g("vision_model.embeddings.position_embedding.weight"),
           "ln_pre.weight": g("vision_model.pre_layrnorm.weight"), "ln_pre.bias": g("vision_model.pre_layrnorm.bias"),
           "ln_post.weight": g("vision_model.post_layernorm.weight"), "ln_post.bias": g("vision_model.post_layernorm.bias"),
           "proj": g("visual_projection.weight").t().contiguous()}
    i = 0
    while f"vision_model.encoder.layers.{i}.layer_norm1.weight" in hf:
        a, b = f"vision_model.encoder.layers.{i}.", f"transformer.resblocks.{i}."
        out[b + "attn.in_proj_weight"] = torch.cat([g(a + f"self_attn.{n}_proj.weight") for n in "qkv"], dim=0)
        out[b + "attn.in_proj_bias"] = torch.cat([g(a + f"self_attn.{n}_proj.bias") for n in "qkv"], dim=0)
        for src, dst in (("self_attn.out_proj", "attn.out_proj"), ("layer_norm1", "ln_1"), ("layer_norm2", "ln_2"),
                         ("mlp.fc1", "mlp.c_fc"), ("mlp.fc2", "mlp.c_proj")):
            out[b + dst + ".weight"], out[b + dst + ".bias"] = g(a + src + ".weight"), g(a + src + ".bias")
        i += 1
    return out


def hf_vision_state_dict_shapes(cfg) -> Dict[str, Tuple[int, ...]]:
    """The same tower under transformers' key names (what ``TransformersOpenAICLIP.state_dict()`` holds)."""
    res, patch, width, layers, heads, out = cfg
    S = {"vision_model.embeddings.class_embedding": (width,), "vision_model.embeddings.patch_embedding.weight": (width, 3, patch, patch),
         "vision_model.embeddings.position_embedding.weight": ((res // patch) ** 2 + 1, width),
         "vision_model.pre_layrnorm.weight": (width,), "vision_model.pre_layrnorm.bias": (width,),
         "vision_model.post_layernorm.weight": (width,), "vision_model.post_layernorm.bias": (width,), "visual_projection.weight": (out, width)}
    for i in range(layers):
        a = f"vision_model.encoder.layers.{i}."
        for n in "qkvo":
            nm = "out_proj" if n == "o" else f"{n}_proj"
            S[a + f"self_attn.{nm}.weight"] = (width, width); S[a + f"self_attn.{nm}.bias"] = (width,)
        for nm in ("layer_norm1", "layer_norm2"):
            S[a + nm + ".weight"] = (width,); S[a + nm + ".bias"] = (width,)
        S[a + "mlp.fc1.weight"] = (4 * width, width); S[a + "mlp.fc1.bias"] = (4 * width,)
        S[a + "mlp.fc2.weight"] = (width, 4 * width); S[a + "mlp.fc2.bias"] = (width,)
    return S


class _Lin:
    """Forward weights + the transposed copy used by the input-gradient GEMM."""

    def __init__(self, w, b, dt, dev):
        self.fwd = PackedLinear(w, b, dt, dev)
        self.bwd = PackedLinear(w.t().contiguous(), None, dt, dev)


class VitEngine:
    def __init__(self, cfg, state_dict, device, dtype="bf16", quick_gelu=True):
        self.cfg, self.device = cfg, torch.device(device)
        self.dt = _hip.dtype_code(dtype)
        # fp16 cannot hold ~1e-6 gradients: scale the loss gradient up and the image gradient back down
        self.gscale = 1.0 if self.dt == _hip.DT_BF16 else 65536.0
        self.act = ACT_QUICKGELU if quick_gelu else ACT_GELU
        _hip.lib()
        res, patch, width, layers, heads, out = cfg
        sd, dev, dt = state_dict, self.device, self.dt
        f32 = lambda k: sd[k].detach().float().to(dev).contiguous()
        k = 3 * patch * patch
        self.kp = (k + 7) // 8 * 8
        wc = sd["conv1.weight"].detach().float().reshape(width, k)
        wc_p = torch.zeros(width, self.kp)
        wc_p[:, :k] = wc
        self.conv1 = _Lin(wc_p, None, dt, dev)
        self.cls, self.pos = f32("class_embedding"), f32("positional_embedding")
        self.ln_pre = (f32("ln_pre.weight"), f32("ln_pre.bias"))
        self.ln_post = (f32("ln_post.weight"), f32("ln_post.bias"))
        self.proj = _Lin(sd["proj"].detach().float().t().contiguous(), None, dt, dev)
        self.blocks = []
        for i in range(layers):
            p = f"transformer.resblocks.{i}."
            self.blocks.append(dict(
                ln1=(f32(p + "ln_1.weight"), f32(p + "ln_1.bias")), ln2=(f32(p + "ln_2.weight"), f32(p + "ln_2.bias")),
                qkv=_Lin(sd[p + "attn.in_proj_weight"].float(), sd[p + "attn.in_proj_bias"], dt, dev),
                out=_Lin(sd[p + "attn.out_proj.weight"].float(), sd[p + "attn.out_proj.bias"], dt, dev),
                fc=_Lin(sd[p + "mlp.c_fc.weight"].float(), sd[p + "mlp.c_fc.bias"], dt, dev),
                pr=_Lin(sd[p + "mlp.c_proj.weight"].float(), sd[p + "mlp.c_proj.bias"], dt, dev)))
        self.mean = torch.tensor(CLIP_MEAN, device=dev)
        self.std = torch.tensor(CLIP_STD, device=dev)
        self.output_dim = out
        self.saved = None

    # ---- helpers ------------------------------------------------------------------------------------
    def _ln(self, x, ld, gb, m, d, want16=True, want32=False):
        dev = x.device
        y16 = torch.empty((m, d), dtype=_hip.TORCH_DTYPE[self.dt], device=dev) if want16 else None
        y32 = torch.empty((m, d), dtype=torch.float32, device=dev) if want32 else None
        mr = torch.empty((2, m), dtype=torch.float32, device=dev)
        call("pmi_layernorm_fwd", ptr(x), ld, ptr(gb[0]), ptr(gb[1]), ptr(y16), ptr(y32), ptr(mr), m, d, 1e-5, self.dt)
        return y16, y32, mr

    def _ln_bwd(self, dy, dy_ld, x, gb, mr, gres, m, d, row_stride=1, g32=None, g16=None, want32=True, want16=True):
        dev = x.device
        if g32 is None and want32:
            g32 = torch.empty((m, d), dtype=torch.float32, device=dev)
        if g16 is None and want16:
            g16 = torch.empty((m, d), dtype=_hip.TORCH_DTYPE[self.dt], device=dev)
        call("pmi_layernorm_bwd", ptr(dy), ptr(x), ptr(gb[0]), ptr(mr), ptr(gres), ptr(g32), ptr(g16), m, d, dy_ld, row_stride, self.dt)
        return g32, g16

    def _ln_slabs(self, slabs, bias, residual, gb, m, d):
        """x = sum of the GEMM's split-K slabs + bias + residual (the residual stream, fp32) and LayerNorm(x) as the next GEMM's operand,
        one pass (pmi_layernorm_fwd_slabs)."""
        _, ws, sk = slabs
        dev = ws.device
        x = torch.empty((m, d), dtype=torch.float32, device=dev)
        y16 = torch.empty((m, d), dtype=_hip.TORCH_DTYPE[self.dt], device=dev)
        mr = torch.empty((2, m), dtype=torch.float32, device=dev)
        call("pmi_layernorm_fwd_slabs", ptr(ws), sk, m * d, ptr(bias), ptr(residual), ptr(x), ptr(gb[0]), ptr(gb[1]), ptr(y16), ptr(mr), m, d, 1e-5, self.dt)
        return x, y16, mr

    def _ln_bwd_any(self, dy, x, gb, mr, gres, m, d):
        """LayerNorm input gradient from the input-gradient GEMM's output -- or straight from its unreduced split-K slabs."""
        if isinstance(dy, tuple):
            _, ws, sk = dy
            g32 = torch.empty((m, d), dtype=torch.float32, device=x.device)
            g16 = torch.empty((m, d), dtype=_hip.TORCH_DTYPE[self.dt], device=x.device)
            call("pmi_layernorm_bwd_slabs", ptr(ws), sk, m * d, ptr(x), ptr(gb[0]), ptr(mr), ptr(gres), ptr(g32), ptr(g16), m, d, self.dt)
            return g32, g16
        return self._ln_bwd(dy, d, x, gb, mr, gres, m, d)

    def _transpose(self, src: torch.Tensor, off: int, rows: int, cols: int, ld: int, s_o: int, s_i: int, inner: int, batch: int):
        rp = (rows + 7) // 8 * 8
        out = torch.empty((batch, cols, rp), dtype=src.dtype, device=src.device)
        call("pmi_transpose_16", src.data_ptr() + off * src.element_size(), ptr(out), rows, cols, ld, s_o, s_i, inner, batch)
        return out

    # ---- forward --------------------------------------------------------------------------------------
    @torch.no_grad()
    def forward(self, images: torch.Tensor, save: bool = False, features: bool = False):
        """images NCHW fp32 in [0,1] (any size) -> un-normalised embeddings [N, out_dim] fp32
        (features=True: also the last hidden state [N, T, width] and the post-LayerNorm class token [N, width], both fp32)."""
        if not images.is_cuda:
            raise RuntimeError("VitEngine runs on a HIP device only (no CPU fallback)")
        res, patch, width, layers, heads, out = self.cfg
        dt, dev = self.dt, self.device
        tdt = _hip.TORCH_DTYPE[dt]
        n = images.shape[0]
        g = res // patch
        t = g * g + 1
        m = n * t
        tp = (t + 7) // 8 * 8
        d = width // heads
        resized = _resize(images, (res, res))
        col = torch.empty((n * g * g, self.kp), dtype=tdt, device=dev)
        call("pmi_patchify", ptr(resized), ptr(self.mean), ptr(self.std), ptr(col), n, res, patch, self.kp, 0, dt)
        pe = ops.igemm(col, self.conv1.fwd, out_f32=True)
        x0 = torch.empty((n, t, width), dtype=torch.float32, device=dev)
        call("pmi_vit_assemble", ptr(pe), ptr(self.cls), ptr(self.pos), ptr(x0), n, t, width, 0)
        _, x, mr_pre = self._ln(x0, width, self.ln_pre, m, width, want16=False, want32=True)
        sv = dict(in_hw=tuple(images.shape[2:]), n=n, x0=x0, mr_pre=mr_pre, layers=[]) if save else None
        scale = float(d) ** -0.5
        fuse = width % 256 == 0 and width <= 2048            # split-K reduce of the MLP's last GEMM + the next block's LayerNorm in one pass
        nxt = None                                           # (h, mr1) of this block when the previous block's tail already produced them
        for bi, blk in enumerate(self.blocks):
            if nxt is None:
                h, _, mr1 = self._ln(x, width, blk["ln1"], m, width)
            else:
                h, mr1 = nxt
                nxt = None
            qkv = ops.igemm(h, blk["qkv"].fwd)                                    # [m, 3*width], (q|k|v) x (head, d)
            a = torch.empty((m, width), dtype=tdt, device=dev)
            fused = d == 64
            if fused:   # flash-style kernels: no T x T matrix, no transposes (csrc/attn.hip)
                tp32 = (t + 31) // 32 * 32
                aws = torch.empty((6, n * heads, tp32, 64), dtype=tdt, device=dev)
                lse = torch.empty((n * heads, tp32), dtype=torch.float32, device=dev)
                call("pmi_vit_attn_fwd", ptr(qkv), ptr(aws), ptr(lse), ptr(a), n, t, heads, scale, dt)
                p = None
            else:
                s = torch.empty((n * heads, t, tp), dtype=torch.float32, device=dev)
                ops.bgemm(qkv, qkv, s, M=t, N=t, K=d, lda=3 * width, ldb=3 * width, ldd=tp, batch=n * heads, batch_inner=heads,
                          sA=(t * 3 * width, d), sB=(t * 3 * width, d), sD=(heads * t * tp, t * tp), dt=dt, b_off=width)
                p = torch.empty((n * heads, t, tp), dtype=tdt, device=dev)
                call("pmi_softmax_fwd", ptr(s), ptr(p), n * heads * t, t, tp, tp, scale, dt)
                vt = self._transpose(qkv, 2 * width, t, d, 3 * width, t * 3 * width, d, heads, n * heads)
                ops.bgemm(p, vt, a, M=t, N=d, K=tp, lda=tp, ldb=tp, ldd=width, batch=n * heads, batch_inner=heads,
                          sA=(heads * t * tp, t * tp), sB=(heads * d * tp, d * tp), sD=(t * width, d), dt=dt)
                aws = lse = None
            x_mid = ops.igemm(a, blk["out"].fwd, residual=x, out_f32=True)
            h2, _, mr2 = self._ln(x_mid, width, blk["ln2"], m, width)
            if ops.fused_mlp_epilogues(blk["fc"].fwd, m):    # activation in the GEMM epilogue; the pre-activation value is kept only for a backward pass
                hpre = torch.empty((m, blk["fc"].fwd.n_p), dtype=tdt, device=dev) if save else None
                hact = ops.igemm(h2, blk["fc"].fwd, act=self.act, pre_out=hpre)
            else:
                hpre = ops.igemm(h2, blk["fc"].fwd)
                hact = torch.empty_like(hpre)
                call("pmi_act_fwd", ptr(hpre), ptr(hact), hpre.numel(), self.act, dt)
            last = bi + 1 == len(self.blocks)
            x_out = ops.igemm(hact, blk["pr"].fwd, residual=x_mid, out_f32=True, defer_reduce=fuse and not last)
            if isinstance(x_out, tuple):
                x_out, hn, mrn = self._ln_slabs(x_out, blk["pr"].fwd.b, x_mid, self.blocks[bi + 1]["ln1"], m, width)
                nxt = (hn, mrn)
            if save:
                sv["layers"].append(dict(x_in=x, mr1=mr1, qkv=qkv if not fused else None, p=p, aws=aws, lse=lse, a=a if fused else None,
                                         x_mid=x_mid, mr2=mr2, hpre=hpre))
            x = x_out
        y16, y32, mr_post = self._ln(x, t * width, self.ln_post, n, width, want32=features)         # cls token rows only
        emb = ops.igemm(y16, self.proj.fwd, out_f32=True)
        if save:
            sv.update(x_final=x, mr_post=mr_post)
            self.saved = sv
        emb = emb[:, :out] if emb.shape[1] != out else emb
        return (emb, x.view(n, t, width), y32) if features else emb

    # ---- input gradient -----------------------------------------------------------------------------------
    @torch.no_grad()
    def backward(self, d_emb: torch.Tensor) -> torch.Tensor:
        """d_emb: dL/d(embedding) * self.gscale, fp32 [N, out_dim]  ->  dL/d(images), fp32 NCHW."""
        sv = self.saved
        if sv is None:
            raise RuntimeError("call forward(images, save=True) before backward()")
        res, patch, width, layers, heads, out = self.cfg
        dt, dev = self.dt, self.device
        tdt = _hip.TORCH_DTYPE[dt]
        n = sv["n"]
        g = res // patch
        t = g * g + 1
        m = n * t
        tp = (t + 7) // 8 * 8
        d = width // heads
        scale = float(d) ** -0.5
        d16 = torch.empty((n, out), dtype=tdt, device=dev)
        d_emb = d_emb.contiguous()          # named: a temporary would be released before the launch is queued
        call("pmi_cast_f32_to_16", ptr(d_emb), ptr(d16), n * out, ACT_NONE, dt)
        dy_post = ops.igemm(d16, self.proj.bwd, out_f32=True)                      # [n, width]
        g32 = torch.zeros((m, width), dtype=torch.float32, device=dev)
        g16 = torch.zeros((m, width), dtype=tdt, device=dev)
        self._ln_bwd(dy_post, dy_post.shape[1], sv["x_final"], self.ln_post, sv["mr_post"], None, n, width, row_stride=t, g32=g32, g16=g16)
        for blk, L in zip(reversed(self.blocks), reversed(sv["layers"])):
            # ---- MLP branch
            if ops.fused_mlp_epilogues(blk["pr"].bwd, m):    # dh * act'(h_pre) in the GEMM epilogue
                dh = ops.igemm(g16, blk["pr"].bwd, act_grad_of=L["hpre"], act_grad=self.act)    # [m, 4w]
            else:
                dh = ops.igemm(g16, blk["pr"].bwd)
                call("pmi_act_bwd", ptr(dh), ptr(L["hpre"]), ptr(dh), dh.numel(), self.act, dt)
            fuse = width % 256 == 0 and width <= 2048
            dln2 = ops.igemm(dh, blk["fc"].bwd, out_f32=True, defer_reduce=fuse)
            gm32, gm16 = self._ln_bwd_any(dln2, L["x_mid"], blk["ln2"], L["mr2"], g32, m, width)
            # ---- attention branch
            da = ops.igemm(gm16, blk["out"].bwd)                                    # dO, [m, w] (head, d)
            w3 = 3 * width
            dqkv = torch.empty((m, w3), dtype=tdt, device=dev)
            if L["aws"] is not None:
                tp32 = (t + 31) // 32 * 32
                bws = torch.empty((2, n * heads, tp32, 64), dtype=tdt, device=dev)
                delta = torch.empty((n * heads, tp32), dtype=torch.float32, device=dev)
                call("pmi_vit_attn_bwd", ptr(L["aws"]), ptr(L["lse"]), ptr(L["a"]), ptr(da), ptr(bws), ptr(delta), ptr(dqkv),
                     n, t, heads, scale, dt)
            else:
                qkv, p = L["qkv"], L["p"]
                dp = torch.empty((n * heads, t, tp), dtype=torch.float32, device=dev)
                ops.bgemm(da, qkv, dp, M=t, N=t, K=d, lda=width, ldb=w3, ldd=tp, batch=n * heads, batch_inner=heads,
                          sA=(t * width, d), sB=(t * w3, d), sD=(heads * t * tp, t * tp), dt=dt, b_off=2 * width)
                ds = torch.empty((n * heads, t, tp), dtype=tdt, device=dev)
                call("pmi_softmax_bwd", ptr(dp), ptr(p), ptr(ds), n * heads * t, t, tp, tp, scale, dt)
                pt = self._transpose(p, 0, t, t, tp, heads * t * tp, t * tp, heads, n * heads)          # [nh, t(s), tp(t)]
                dot = self._transpose(da, 0, t, d, width, t * width, d, heads, n * heads)               # [nh, d, tp]
                ops.bgemm(pt, dot, dqkv, M=t, N=d, K=tp, lda=tp, ldb=tp, ldd=w3, batch=n * heads, batch_inner=heads,
                          sA=(heads * t * tp, t * tp), sB=(heads * d * tp, d * tp), sD=(t * w3, d), dt=dt, d_off=2 * width)   # dV
                kt = self._transpose(qkv, width, t, d, w3, t * w3, d, heads, n * heads)
                ops.bgemm(ds, kt, dqkv, M=t, N=d, K=tp, lda=tp, ldb=tp, ldd=w3, batch=n * heads, batch_inner=heads,
                          sA=(heads * t * tp, t * tp), sB=(heads * d * tp, d * tp), sD=(t * w3, d), dt=dt)                    # dQ
                dst = self._transpose(ds, 0, t, t, tp, heads * t * tp, t * tp, heads, n * heads)
                qt = self._transpose(qkv, 0, t, d, w3, t * w3, d, heads, n * heads)
                ops.bgemm(dst, qt, dqkv, M=t, N=d, K=tp, lda=tp, ldb=tp, ldd=w3, batch=n * heads, batch_inner=heads,
                          sA=(heads * t * tp, t * tp), sB=(heads * d * tp, d * tp), sD=(t * w3, d), dt=dt, d_off=width)       # dK
            dln1 = ops.igemm(dqkv, blk["qkv"].bwd, out_f32=True, defer_reduce=fuse)
            g32, g16 = self._ln_bwd_any(dln1, L["x_in"], blk["ln1"], L["mr1"], gm32, m, width)
        _, g0 = self._ln_bwd(g32, width, sv["x0"], self.ln_pre, sv["mr_pre"], None, m, width, want32=False)
        dcol = torch.empty((n * (t - 1), self.kp), dtype=torch.float32, device=dev)
        ops.bgemm(g0, self.conv1.bwd.w, dcol, M=t - 1, N=self.kp, K=width, lda=width, ldb=width, ldd=self.kp, batch=n, batch_inner=1,
                  sA=(t * width, 0), sB=(0, 0), sD=((t - 1) * self.kp, 0), dt=dt, a_off=width)
        dres = torch.empty((n, 3, res, res), dtype=torch.float32, device=dev)
        call("pmi_unpatchify", ptr(dcol), ptr(self.std), ptr(dres), n, res, patch, self.kp, 1.0 / self.gscale)
        self.saved = None
        return _resize_backward(dres, sv["in_hw"])
