"""OpenCLIP / CLIP image encoders — drop-in for perceptor.models.OpenCLIP and perceptor.models.CLIP.

Call surface of perceptor/models/open_clip.py:12-140 and perceptor/models/clip.py:6-27.  The ViT image
tower runs in perceptor_amd.engine.vit.VitEngine (HIP); pre-processing is the reference's:
resize (ResizeRight lanczos3/bicubic) -> Normalize(mean, std) -> tower -> F.normalize.

The text tower (``encode_texts``, models/open_clip.py:99-107) runs in perceptor_amd.engine.text.TextEngine behind
perceptor_amd.utils.tokenizer.ClipTokenizer; the tokenizer's merge list is data that ships with CLIP checkpoints
(``bpe_path=`` / ``PERCEPTOR_AMD_BPE``), token ids can also be passed directly (``encode_tokens``).

Not available here (stated loudly instead of faked): pretrained weights (no network;
``weights="synthetic"`` gives name-keyed deterministic weights, or pass ``checkpoint=`` with an
open_clip state dict: ``visual.*`` and, for the text side, the root-level text-tower tensors) and ResNet towers.
"""
from __future__ import annotations

from typing import Optional

import torch

from .._hip import call, ptr
from ..engine import text as text_engine
from ..engine import vit
from ..utils.param_tree import ParamTree
from ..utils.synth import synth_state_dict

# (architecture, weights) pairs the reference accepts (docstring of models/open_clip.py:24-44)
PRETRAINED = {
    ("ViT-H-14", "laion2b_s32b_b79k"), ("ViT-g-14", "laion2b_s12b_b42k"), ("ViT-L-14", "laion2b_s32b_b82k"),
    ("ViT-B-32", "laion2b_s34b_b79k"), ("ViT-B-16-plus-240", "laion400m_e32"), ("ViT-B-32", "laion2b_e16"),
    ("ViT-B-16", "laion400m_e32"), ("ViT-B-32", "laion400m_e32"), ("ViT-L-14", "laion400m_e32"),
    ("RN101", "yfcc15m"), ("RN50", "yfcc15m"), ("RN50", "cc12m"), ("RN50-quickgelu", "openai"),
    ("RN101-quickgelu", "openai"), ("RN50x4", "openai"), ("RN50x16", "openai"), ("RN50x64", "openai"),
    ("ViT-B-32-quickgelu", "openai"), ("ViT-B-16", "openai"), ("ViT-L-14", "openai"), ("ViT-L-14-336", "openai"),
}


class _EncodeImages(torch.autograd.Function):
    """Lets ``loss(images).backward()`` work as in the reference: forward/backward are the HIP engine's."""

    @staticmethod
    def forward(ctx, images, model, normalize):
        if getattr(model, "want_features", False):      # HF-flavoured surface: the (detached) hidden state and pooled class token of this same pass
            emb, hidden, pooled = model.engine.forward(images, save=True, features=True)
            model.features = (hidden.detach(), pooled.detach())
        else:
            emb = model.engine.forward(images, save=True)
        ctx.model, ctx.normalize = model, normalize
        ctx.saved_state = model.engine.saved
        ctx.save_for_backward(emb)
        if normalize:
            out = torch.empty_like(emb)
            emb = emb.contiguous()
            call("pmi_l2norm_rows", ptr(emb), ptr(out), emb.shape[0], emb.shape[1], 1.0)
            return out
        return emb.clone()

    @staticmethod
    def backward(ctx, grad_out):
        (emb,) = ctx.saved_tensors
        eng = ctx.model.engine
        g = grad_out.float()
        if ctx.normalize:   # d/d emb of emb/|emb|  (tiny [N, D] tensors)
            nrm = emb.norm(dim=1, keepdim=True).clamp(min=1e-12)
            e = emb / nrm
            g = (g - e * (e * g).sum(dim=1, keepdim=True)) / nrm
        eng.saved = ctx.saved_state
        return eng.backward(g * eng.gscale), None, None


class OpenCLIP(torch.nn.Module):
    def __init__(self, architecture="ViT-H-14", weights="laion2b_s32b_b79k", precision=None, *, checkpoint: Optional[str] = None,
                 seed: int = 0, quick_gelu: Optional[bool] = None, config: Optional[tuple] = None, text_config: Optional[tuple] = None,
                 bpe_path: Optional[str] = None):
        """
        Args:
            architecture (str): name of the clip model
            weights (str): name of the weights ("synthetic" for deterministic offline weights)
            precision (str): "bf16" (default on HIP) or "fp16"
        """
        super().__init__()
        self.architecture, self.weights = architecture, weights
        if weights != "synthetic" and (architecture, weights) not in PRETRAINED:
            raise ValueError(f"Invalid architecture/weights: {architecture}/{weights}")
        base = architecture.replace("-quickgelu", "")
        if config is not None:
            vit.VIT_CONFIGS.setdefault(base, tuple(config))
        if base not in vit.VIT_CONFIGS:
            if weights == "synthetic":
                raise ValueError(f"Invalid architecture/weights: {architecture}/{weights}")
            raise NotImplementedError(f"{architecture}: only the ViT image towers {sorted(vit.VIT_CONFIGS)} run on the HIP path")
        self.cfg = vit.VIT_CONFIGS[base]
        if weights != "synthetic" and checkpoint is None:
            raise RuntimeError(f"pretrained weights {architecture}/{weights} cannot be downloaded (no network): "
                               "pass checkpoint=<visual state dict> or weights='synthetic'")
        self.quick_gelu = quick_gelu if quick_gelu is not None else ("-quickgelu" in architecture or weights == "openai")
        if precision == "fp32":
            # the reference default (losses/clip/clip.py:11); this path has no fp32 tower: say so instead of silently lowering it
            import warnings
            warnings.warn("precision='fp32' requested: the HIP CLIP tower multiplies in bf16 on the MFMA (fp32 accumulation, fp32 residual "
                          "stream, LayerNorm and softmax); embedding rel-L2 error vs fp32 ~4e-3 (tests/test_gpu_clip.py)", RuntimeWarning, stacklevel=2)
        self.precision = {None: "bf16", "fp16": "f16", "f16": "f16", "bf16": "bf16", "fp32": "bf16"}[precision]
        shapes = vit.vit_state_dict_shapes(self.cfg)
        # text tower (context, vocab, width, layers, heads, out_dim): open_clip's config of the architecture, or text_config=
        self.text_cfg = tuple(text_config) if text_config is not None else (text_engine.TEXT_CONFIGS.get(base) if config is None else None)
        tshapes = text_engine.text_state_dict_shapes(self.text_cfg) if self.text_cfg is not None else {}
        if checkpoint is not None:
            raw = torch.load(checkpoint, map_location="cpu", weights_only=True)
            sd = {k[len("visual."):]: v.float() for k, v in raw.items() if k.startswith("visual.")} or {k: v.float() for k, v in raw.items()}
            sd = {k: v for k, v in sd.items() if k in shapes}
            if set(sd) != set(shapes):
                raise RuntimeError("checkpoint does not contain the visual tower's tensors")
            tsd = {k: v.float() for k, v in raw.items() if k in tshapes}
            if set(tsd) != set(tshapes):          # a visual-only file: the model has no text side
                tsd, self.text_cfg = {}, None
        else:
            sd = synth_state_dict(shapes, seed)
            tsd = synth_state_dict(tshapes, seed) if tshapes else {}
        # the reference holds the open_clip model as self.model: image tower under "visual.", text tower at its root (models/open_clip.py:65-76)
        self.model = ParamTree({**{"visual." + k: v for k, v in sd.items()}, **tsd})
        self._engine: Optional[vit.VitEngine] = None
        self._text_engine: Optional[text_engine.TextEngine] = None
        self._tokenizer, self._bpe_path = None, bpe_path
        self.register_load_state_dict_post_hook(lambda module, incompatible: (setattr(module, "_engine", None), setattr(module, "_text_engine", None)))
        self.output_dim = self.cfg[5]

    def visual_state_dict(self):
        return {k[len("visual."):]: v for k, v in self.model.state_dict().items() if k.startswith("visual.")}

    @property
    def text(self) -> text_engine.TextEngine:
        """The text tower's engine, packed on first use (prompts are encoded once per run)."""
        if self.text_cfg is None:
            raise RuntimeError("this OpenCLIP holds no text tower (visual-only checkpoint or a custom config without text_config=)")
        if self.device.type != "cuda":
            raise RuntimeError("OpenCLIP needs a HIP device: call .to('cuda') first (perceptor_amd has no CPU fallback)")
        if self._text_engine is None:
            sd = {k: v for k, v in self.model.state_dict().items() if not k.startswith("visual.")}
            self._text_engine = text_engine.TextEngine(self.text_cfg, sd, self.device, self.precision, self.quick_gelu)
        return self._text_engine

    @property
    def engine(self) -> Optional[vit.VitEngine]:
        if self._engine is None and self.device.type == "cuda":
            self._engine = vit.VitEngine(self.cfg, self.visual_state_dict(), self.device, self.precision, self.quick_gelu)
        return self._engine

    def _apply(self, fn, *a, **k):
        self._engine = None
        self._text_engine = None
        return super()._apply(fn, *a, **k)

    def to(self, *args, **kwargs):
        super().to(*args, **kwargs)
        self.engine                           # noqa: B018
        return self

    @property
    def device(self):
        return next(self.model.parameters()).device

    @property
    def image_size(self):
        return (self.cfg[0], self.cfg[0])

    def _need_engine(self):
        if self.engine is None:
            raise RuntimeError("OpenCLIP needs a HIP device: call .to('cuda') first (perceptor_amd has no CPU fallback)")
        return self.engine

    def tokenize(self, text_prompts) -> torch.Tensor:
        """open_clip.tokenize (models/open_clip.py:101-103): int64 [N, context], zero padded."""
        if self._tokenizer is None:
            from ..utils.tokenizer import ClipTokenizer
            self._tokenizer = ClipTokenizer(self._bpe_path)
        ctx = self.text_cfg[0] if self.text_cfg is not None else 77
        return self._tokenizer(text_prompts, context_length=ctx)

    def encode_tokens(self, token_ids: torch.Tensor, normalize=True):
        """The text tower on token ids [N, T] (int64): what encode_texts runs behind the tokenizer."""
        _, pooled = self.text.forward(token_ids)
        if normalize:
            out = torch.empty_like(pooled)
            pooled = pooled.contiguous()
            call("pmi_l2norm_rows", ptr(pooled), ptr(out), pooled.shape[0], pooled.shape[1], 1.0)
            return out
        return pooled

    def encode_texts(self, text_prompts, normalize=True):
        return self.encode_tokens(self.tokenize(text_prompts), normalize)

    def encode_images(self, images, normalize=True):
        self._need_engine()
        images = images.to(self.device)
        if images.requires_grad and torch.is_grad_enabled():
            return _EncodeImages.apply(images, self, normalize)
        emb = self.engine.forward(images)
        if normalize:
            out = torch.empty_like(emb)
            emb = emb.contiguous()
            call("pmi_l2norm_rows", ptr(emb), ptr(out), emb.shape[0], emb.shape[1], 1.0)
            return out
        return emb

    @staticmethod
    def spherical_distance(encodings_a, encodings_b):
        return (encodings_a[:, None] - encodings_b[None, :]).norm(dim=2).div(2).arcsin().square().mul(2)

    def forward(self, _):
        raise NotImplementedError


def CLIP(architecture: str, precision: Optional[str] = None, **kw):
    """perceptor/models/clip.py:6-27 — OpenAI weights; RN50/RN101/ViT-B-32 get the -quickgelu config."""
    if "-quickgelu" not in architecture and architecture in ["RN50", "RN101", "ViT-B-32"]:
        architecture = architecture + "-quickgelu"
    return OpenCLIP(architecture, kw.pop("weights", "openai"), precision, **kw)
