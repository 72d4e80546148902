"""TransformersOpenAICLIP — drop-in for perceptor.models.TransformersOpenAICLIP (the Hugging Face flavoured CLIP surface).

Call surface of perceptor/models/transformers_openai_clip.py:17-137: ``encode_images`` / ``encode_texts`` return an ``Encodings`` record
(``features`` with ``last_hidden_state`` and ``pooler_output``, ``unnormalized_encodings``, ``encodings``), ``spherical_distance`` works
on two such records, gradients flow from the encodings to the images.  The towers are the same OpenAI-CLIP transformers as behind
``models.OpenCLIP`` and run in the same HIP engines (engine/vit.py, engine/text.py); the module's state dict uses transformers' key names
(``vision_model.*``, ``visual_projection.weight``, ``text_projection.weight``, ``logit_scale``), so ``CLIPModel`` checkpoints load as they are.
The text model is loaded on demand upstream (:90) and is not part of the state dict: ``text_checkpoint=`` (a CLIPTextModel state dict) or
synthetic weights.  Nothing is downloaded: ``weights="synthetic"`` (default) or ``checkpoint=``.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from types import SimpleNamespace
from typing import Optional

import torch

from ..engine import text as text_engine
from ..engine import vit
from ..utils.param_tree import ParamTree
from ..utils.synth import synth_state_dict
from .open_clip import _EncodeImages

# name -> (open_clip architecture, QuickGELU)
_NAMES = {
    "openai/clip-vit-base-patch32": ("ViT-B-32", True), "openai/clip-vit-base-patch16": ("ViT-B-16", True),
    "openai/clip-vit-large-patch14": ("ViT-L-14", True),
    "laion/CLIP-ViT-H-14-laion2B-s32B-b79K": ("ViT-H-14", False), "laion/CLIP-ViT-L-14-laion2B-s32B-b82K": ("ViT-L-14", False),
    "laion/CLIP-ViT-B-32-laion2B-s34B-b79K": ("ViT-B-32", False),
}


@dataclass
class Encodings:
    features: object                      # .last_hidden_state, .pooler_output (transformers' BaseModelOutputWithPooling fields)
    unnormalized_encodings: torch.Tensor
    encodings: torch.Tensor


class _Shim:
    """What _EncodeImages needs of a model: the engine."""

    def __init__(self, engine, want_features=False):
        self.engine, self.want_features, self.features = engine, want_features, None


class TransformersOpenAICLIP(torch.nn.Module):
    def __init__(self, name="openai/clip-vit-large-patch14", bfloat16=True, *, weights: str = "synthetic", checkpoint: Optional[str] = None,
                 text_checkpoint: Optional[str] = None, bpe_path: Optional[str] = None, seed: int = 0, config: Optional[tuple] = None,
                 text_config: Optional[tuple] = None, quick_gelu: Optional[bool] = None):
        """
        CLIP text-image similarity with easy feature extraction.

        Args:
            name (str): huggingface model id (one of the OpenAI / LAION ViT CLIPs) -- selects the architecture
            bfloat16 (bool): bf16 MFMA operands (default) or f16
        """
        super().__init__()
        self.name = name
        if config is None and name not in _NAMES:
            raise NotImplementedError(f"{name}: only the ViT CLIP towers {sorted(_NAMES)} run on the HIP path")
        if weights != "synthetic" and checkpoint is None:
            raise RuntimeError(f"pretrained weights {name} cannot be downloaded (no network): pass checkpoint= or weights='synthetic'")
        arch, qg = _NAMES.get(name, (None, True))
        self.cfg = tuple(config) if config is not None else vit.VIT_CONFIGS[arch]
        self.text_cfg = tuple(text_config) if text_config is not None else (text_engine.TEXT_CONFIGS[arch] if config is None else None)
        self.quick_gelu = qg if quick_gelu is None else quick_gelu
        self.precision = "bf16" if bfloat16 else "f16"
        shapes = vit.hf_vision_state_dict_shapes(self.cfg)
        if self.text_cfg is not None:
            shapes["text_projection.weight"] = (self.text_cfg[5], self.text_cfg[2])
        if checkpoint is not None:
            raw = torch.load(checkpoint, map_location="cpu", weights_only=True)
            sd = {k: raw[k].float() for k in shapes if k in raw}
            if set(sd) != set(shapes):
                raise RuntimeError("checkpoint does not contain the CLIPModel vision tower / projections")
        else:
            sd = synth_state_dict(shapes, seed)
        sd["logit_scale"] = torch.tensor(math.log(1 / 0.07))
        self._tree = ParamTree(sd)
        for child_name, child in list(self._tree.named_children()):     # vision_model, visual_projection, text_projection at the root
            self.add_module(child_name, child)
        self.logit_scale = self._tree.logit_scale
        del self._tree
        tshapes = text_engine.hf_text_state_dict_shapes(self.text_cfg, projection=False) if self.text_cfg is not None else {}
        if text_checkpoint is not None:
            raw = torch.load(text_checkpoint, map_location="cpu", weights_only=True)
            self.__dict__["_text_hf"] = {k: raw[k].float() for k in tshapes}
        else:
            self.__dict__["_text_hf"] = None
        self.__dict__["_text_shapes"], self.__dict__["_seed"] = tshapes, seed
        self.image_size = [self.cfg[0], self.cfg[0]]
        self._bpe_path, self._tokenizer = bpe_path, None
        self._engines = {}
        self.register_load_state_dict_post_hook(lambda module, incompatible: module._engines.clear())

    def _apply(self, fn, *a, **k):
        self._engines.clear()
        return super()._apply(fn, *a, **k)

    @property
    def device(self):
        return next(iter(self.parameters())).device

    def _engine(self, which):
        if self.device.type != "cuda":
            raise RuntimeError("TransformersOpenAICLIP needs a HIP device: call .to('cuda') first (perceptor_amd has no CPU fallback)")
        if which not in self._engines:
            sd = self.state_dict()
            if which == "vision":
                e = vit.VitEngine(self.cfg, vit.from_hf_vision_state_dict(sd), self.device, self.precision, self.quick_gelu)
            else:
                if self.text_cfg is None:
                    raise RuntimeError("this model was built without a text tower (custom config without text_config=)")
                if self._text_hf is None:
                    self.__dict__["_text_hf"] = synth_state_dict(self._text_shapes, self._seed)
                hf = dict(self._text_hf)
                hf["text_projection.weight"] = sd["text_projection.weight"]
                e = text_engine.TextEngine(self.text_cfg, text_engine.from_hf_text_state_dict(hf), self.device, self.precision, self.quick_gelu)
            self._engines[which] = e
        return self._engines[which]

    def tokenize(self, texts):
        """CLIPTokenizer(padding=True) (transformers_openai_clip.py:84-86): ids padded with the end token to the longest prompt."""
        if self._tokenizer is None:
            from ..utils.tokenizer import ClipTokenizer
            self._tokenizer = ClipTokenizer(self._bpe_path)
        tk = self._tokenizer
        ids = tk(texts, context_length=self.text_cfg[0], pad="eot")
        longest = int((ids != tk.eot).sum(dim=1).max()) + 1
        return SimpleNamespace(input_ids=ids[:, :longest], attention_mask=(torch.arange(longest)[None] <= (ids[:, :longest] != tk.eot).sum(dim=1)[:, None]).long())

    def encode_token_ids(self, input_ids: torch.Tensor) -> Encodings:
        hidden, pooled = self._engine("text").forward(input_ids)
        eot = input_ids.to(hidden.device).argmax(dim=1)
        feats = SimpleNamespace(last_hidden_state=hidden, pooler_output=hidden[torch.arange(hidden.shape[0], device=hidden.device), eot])
        return Encodings(features=feats, unnormalized_encodings=pooled, encodings=pooled / pooled.norm(p=2, dim=-1, keepdim=True))

    def encode_texts(self, texts) -> Encodings:
        return self.encode_token_ids(self.tokenize(texts).input_ids)

    def encode_images(self, images) -> Encodings:
        eng = self._engine("vision")
        images = images.to(self.device)
        if images.requires_grad and torch.is_grad_enabled():
            # the reference always returns features (its gradient test reads them, transformers_openai_clip.py:88-116): here they are the
            # detached hidden state / pooled class token of the same pass; gradients flow through the encodings
            shim = _Shim(eng, want_features=True)
            un = _EncodeImages.apply(images, shim, False)
            feats = SimpleNamespace(last_hidden_state=shim.features[0], pooler_output=shim.features[1])
        else:
            un, hidden, pooled = eng.forward(images, features=True)
            feats = SimpleNamespace(last_hidden_state=hidden, pooler_output=pooled)
        return Encodings(features=feats, unnormalized_encodings=un, encodings=un / un.norm(p=2, dim=-1, keepdim=True))

    @staticmethod
    def spherical_distance(encodings_a: Encodings, encodings_b: Encodings):
        return (encodings_a.encodings[:, None] - encodings_b.encodings[None, :]).norm(dim=2).div(2).arcsin().square().mul(2)

    def forward(self, _):
        raise NotImplementedError
