"""StableDiffusion — drop-in for perceptor.models.StableDiffusion on MI355X (BASELINE config 4).

Same call surface as perceptor/models/stable_diffusion/stable_diffusion.py:32-491.  The latent UNet, the VAE and the CLIP ViT-L/14
text encoder run in perceptor_amd.engine.sd / engine.text (hand-written HIP kernels); the state dict keeps the reference's layout
(``unet.*`` / ``vae.*`` with diffusers' key names, ``schedule_alphas``, ``schedule_sigmas``), so real checkpoints load as they are.

Deliberate, visible differences:
  * nothing can be downloaded (reference :82-97,298-301): ``weights="synthetic"`` (default) gives name-keyed deterministic weights;
    ``unet_checkpoint= / vae_checkpoint= / text_checkpoint=`` take state-dict files (diffusers / transformers key names, loaded with
    weights_only=True); the tokenizer's merge list is data passed by ``bpe_path=`` / PERCEPTOR_AMD_BPE;
  * compute needs a HIP device; CPU calls raise RuntimeError;
  * ``sample`` evaluates the unconditioned and the conditioned prediction of a step in ONE batched UNet launch sequence (batch 2N)
    instead of two calls -- same values, half the launches;
  * ``latent_masks`` restates kornia's ``gaussian_blur2d`` (reflect border, normalised 1-D Gaussian of ``int(2 * blur) + 1`` taps, applied
    separably) -- kornia is absent here: that one helper is parity-unpinned; it runs once per conditioning on a 1-channel mask.
"""
from __future__ import annotations

from contextlib import contextmanager
from typing import List, Optional

import torch

from ...engine import sampler, sd as sd_engine, text as text_engine
from ...utils.param_tree import ParamTree
from ...utils.synth import synth_state_dict
from .conditioning import Conditioning
from .predictions import Predictions

TEXT_CFG = (77, 49408, 768, 12, 12, 768)      # openai/clip-vit-large-patch14 text model (stable_diffusion.py:298-301)


def scaled_linear_alphas_cumprod(n: int = 1000, beta_start: float = 0.00085, beta_end: float = 0.012) -> torch.Tensor:
    """DDPMScheduler(beta_schedule="scaled_linear") (stable_diffusion.py:98-100): betas = linspace(sqrt(b0), sqrt(b1), n)^2, fp32."""
    betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, n, dtype=torch.float32) ** 2
    return torch.cumprod(1.0 - betas, dim=0)


def _load(path):
    return {k: v.float() for k, v in torch.load(path, map_location="cpu", weights_only=True).items()}


class StableDiffusion(torch.nn.Module):
    def __init__(self, name: str = "runwayml/stable-diffusion-v1-5", decoder_name: Optional[str] = "stabilityai/sd-vae-ft-mse",
                 fp16: bool = True, auth_token=True, flash_attention: bool = True, attention_slicing=None, *,
                 weights: str = "synthetic", unet_checkpoint: Optional[str] = None, vae_checkpoint: Optional[str] = None,
                 text_checkpoint: Optional[str] = None, bpe_path: Optional[str] = None, seed: int = 0,
                 config: Optional[sd_engine.SdConfig] = None, vae_config: Optional[sd_engine.VaeConfig] = None,
                 text_config: Optional[tuple] = None, vae_dtype: str = "bf16"):
        """
        Stable Diffusion text2image model.

        Args:
            name (str): "runwayml/stable-diffusion-v1-5" / "CompVis/stable-diffusion-v1-4" (same architecture)
            decoder_name (str, optional): name of the decoder weights (informational offline)
            fp16 (bool): f16 MFMA operands (default, as the reference's fp16 + autocast); False -> bf16
            flash_attention / attention_slicing: accepted for compatibility; attention never materialises more than one layer's scores
        """
        super().__init__()
        self.name, self.decoder_name = name, decoder_name
        if weights != "synthetic" and unet_checkpoint is None:
            raise RuntimeError(f"pretrained weights {name} cannot be downloaded (no network): pass unet_checkpoint= / vae_checkpoint= "
                               "or weights='synthetic'")
        # the inpainting checkpoint takes latents | mask | masked-image latents: 9 input channels (conditioning.py:31-40)
        self.config = config or (sd_engine.SD_INPAINTING if name == "runwayml/stable-diffusion-inpainting" else sd_engine.SD_V1)
        self.vae_config = vae_config or sd_engine.VAE_V1
        self.text_config = tuple(text_config) if text_config is not None else TEXT_CFG
        self.compute_dtype, self.vae_dtype = ("f16" if fp16 else "bf16"), vae_dtype
        ushapes = sd_engine.unet_state_dict_shapes(self.config)
        vshapes = {**sd_engine.vae_encoder_state_dict_shapes(self.vae_config), **sd_engine.vae_decoder_state_dict_shapes(self.vae_config)}
        usd = _load(unet_checkpoint) if unet_checkpoint else synth_state_dict(ushapes, seed)
        vsd = _load(vae_checkpoint) if vae_checkpoint else synth_state_dict(vshapes, seed)
        if set(usd) != set(ushapes) or any(tuple(usd[k].shape) != tuple(ushapes[k]) for k in ushapes):
            raise RuntimeError("unet checkpoint keys / shapes do not match the UNet configuration")
        if not set(vshapes) <= set(vsd):
            raise RuntimeError("vae checkpoint does not contain the encoder / decoder tensors")
        self.unet, self.vae = ParamTree(usd), ParamTree(vsd)
        # the text encoder is loaded on demand upstream and is not part of the module's state dict (stable_diffusion.py:295-301)
        tshapes = {k: v for k, v in text_engine.text_state_dict_shapes(self.text_config).items() if k != "text_projection"}
        tsd = text_engine.from_hf_text_state_dict(_load(text_checkpoint)) if text_checkpoint else None
        self.__dict__["_text_sd"] = tsd
        self.__dict__["_text_shapes"], self.__dict__["_seed"] = tshapes, seed
        ac = scaled_linear_alphas_cumprod()
        self.schedule_alphas = torch.nn.Parameter(ac.sqrt(), requires_grad=False)
        self.schedule_sigmas = torch.nn.Parameter((1 - ac).sqrt(), requires_grad=False)
        self.vae_original_requires_grads = [False for _ in self.vae.parameters()]
        self._bpe_path, self._tokenizer = bpe_path, None
        self._engines = {}
        self.register_load_state_dict_post_hook(lambda module, incompatible: module._drop_caches())

    # ---- engines (packed 16-bit copies, rebuilt after .to() / load_state_dict) -------------------------------
    def _drop_caches(self):
        self._engines.clear()
        self.__dict__.pop("_pair_key", None)
        self.__dict__.pop("_pair_ctx", None)

    def _apply(self, fn, *a, **k):
        self._drop_caches()
        return super()._apply(fn, *a, **k)

    def _engine(self, which):
        if self.device.type != "cuda":
            raise RuntimeError("StableDiffusion needs a HIP device: call .to('cuda') first (perceptor_amd has no CPU fallback)")
        if which not in self._engines:
            if which == "unet":
                e = sd_engine.SdUnetEngine(self.config, self.unet.state_dict(), self.device, self.compute_dtype)
            elif which == "decoder":
                e = sd_engine.VaeDecoderEngine(self.vae_config, self.vae.state_dict(), self.device, self.vae_dtype)
            elif which == "encoder":
                e = sd_engine.VaeEncoderEngine(self.vae_config, self.vae.state_dict(), self.device, self.vae_dtype)
            else:
                if self._text_sd is None:
                    self.__dict__["_text_sd"] = synth_state_dict(self._text_shapes, self._seed)
                e = text_engine.TextEngine(self.text_config, self._text_sd, self.device, self.compute_dtype, quick_gelu=True)
            self._engines[which] = e
        return self._engines[which]

    def to(self, *args, **kwargs):
        super().to(*args, **kwargs)
        if self.device.type == "cuda":
            self._engine("unet")                     # pack the 860 M UNet weights now, not inside the first step
        return self

    @property
    def device(self):
        return self.schedule_alphas.device

    @property
    def shape(self):
        raise AttributeError("'StableDiffusion' object has no attribute 'model'")      # as upstream (stable_diffusion.py:128-130)

    def schedule_indices(self, n_steps=500, from_index=999, to_index=0, rho=3.0):
        """Karras-rho ramp in sigma space snapped to the 1000 discrete log-SNRs (stable_diffusion.py:132-173)."""
        if from_index < to_index:
            raise ValueError("from_index must be greater than to_index")
        alphas, sigmas = self.schedule_alphas.detach().cpu(), self.schedule_sigmas.detach().cpu()
        from_log_snr = torch.log(alphas[from_index] ** 2 / sigmas[from_index] ** 2)
        to_log_snr = torch.log(alphas[to_index] ** 2 / sigmas[to_index] ** 2)
        sigma_max = (1 / from_log_snr.exp()).sqrt().clamp(max=150)
        sigma_min = (1 / to_log_snr.exp()).sqrt().clamp(min=1e-3)
        ramp = torch.linspace(0, 1, n_steps + 1)
        karras = (sigma_max ** (1 / rho) + ramp * (sigma_min ** (1 / rho) - sigma_max ** (1 / rho))) ** rho
        target = torch.log(torch.ones_like(karras) ** 2 / karras**2)
        table = torch.log(alphas**2 / sigmas**2)
        idx = (target[:, None] - table[None, :]).abs().argmin(dim=1).unique().sort(descending=True)[0]
        if len(idx) <= n_steps * 0.9:
            raise ValueError(f"Scheduled steps {len(idx)} is too far from wanted number of steps {n_steps}")
        assert (idx[:-1] != idx[1:]).all()
        return torch.stack([idx[:-1], idx[1:]], dim=1).to(self.device)

    # ---- VAE -----------------------------------------------------------------------------------------------
    def encode(self, images, method="mode"):
        _, _, h, w = images.shape
        if h % 32 != 0:
            raise Exception(f"Height must be divisible by 32, got {h}")
        if w % 32 != 0:
            raise Exception(f"Width must be divisible by 32, got {w}")
        mean, logvar = self._engine("encoder").forward(images.to(self.device))
        if method == "sample":
            std = torch.exp(0.5 * logvar.clamp(-30.0, 20.0))
            return sampler.lincomb2(mean, 0.18215, sampler.randn_like(mean) * std, 0.18215)
        if method == "mode":
            return sampler.lincomb2(mean, 0.18215)
        raise ValueError(f"Unknown encoding method {method}")

    def decode(self, latents):
        return self._engine("decoder").forward(latents.to(self.device))

    @contextmanager
    def finetuneable_vae(self):
        raise NotImplementedError("the VAE runs forward-only in the HIP engine (no weight gradients)")
        yield self

    def latents(self, images):
        return self.encode(images).float()

    def images(self, latents):
        return self.decode(latents).float()

    def random_diffused_latents(self, shape):
        n, c, h, w = shape
        if h % 32 != 0:
            raise ValueError("Height must be divisible by 32")
        if w % 32 != 0:
            raise ValueError("Width must be divisible by 32")
        return torch.randn((n, self.config.in_channels, h // 8, w // 8)).to(self.device) * 1.0     # DDPMScheduler.init_noise_sigma = 1

    def indices(self, indices):
        if isinstance(indices, (float, int)):
            indices = torch.as_tensor(indices)
        if indices.ndim == 0:
            indices = indices[None]
        if indices.ndim != 1:
            raise ValueError("indices must be a scalar or a 1-dimensional tensor")
        return indices.long().to(self.device)

    def alphas(self, indices):
        return self.schedule_alphas[self.indices(indices)][:, None, None, None].to(self.device)

    def sigmas(self, indices):
        return self.schedule_sigmas[self.indices(indices)][:, None, None, None].to(self.device)

    # ---- UNet ----------------------------------------------------------------------------------------------
    def predicted_noise(self, diffused_latents, from_indices, conditioning: Conditioning):
        idx = self.indices(from_indices)
        x = conditioning.input(diffused_latents).to(self.device)
        n = x.shape[0]
        if idx.numel() == 1 and n > 1:
            idx = idx.expand(n)
        enc = conditioning.encodings
        if enc.shape[0] == 1 and n > 1:
            enc = enc.expand(n, -1, -1).contiguous()
        return self._engine("unet").forward(x, idx, enc)

    def forward(self, diffused_latents, indices, conditioning: Optional[Conditioning] = None) -> Predictions:
        indices = self.indices(indices)
        return Predictions(from_diffused_latents=diffused_latents, from_indices=indices,
                           predicted_noise=self.predicted_noise(diffused_latents, indices, conditioning),
                           schedule_alphas=self.schedule_alphas, schedule_sigmas=self.schedule_sigmas,
                           encode=self.encode, decode=self.decode)

    def predictions(self, diffused_latents, indices, conditioning) -> Predictions:
        return self.forward(diffused_latents, indices, conditioning)

    def predictions_pair(self, diffused_latents, indices, neutral: Conditioning, positive: Conditioning):
        """(unconditioned, conditioned) Predictions of the same latents from ONE batched UNet evaluation (batch 2N)."""
        idx = self.indices(indices)
        n = diffused_latents.shape[0]
        if idx.numel() == 1 and n > 1:
            idx = idx.expand(n)
        ex = lambda c: c.encodings.expand(n, -1, -1) if c.encodings.shape[0] == 1 and n > 1 else c.encodings
        # One context tensor per (prompt pair, batch) keeps the engine's k|v cache valid across the steps of a chain.  The cache entry HOLDS the two
        # encodings tensors (their addresses cannot be recycled for another prompt while it lives) and is compared by identity and
        # in-place version (an edited encodings tensor, e.g. prompt weighting, is a new context).
        ne, pe = neutral.encodings, positive.encodings
        key = self.__dict__.get("_pair_key")
        if key is None or key[0] is not ne or key[1] is not pe or key[2:] != (ne._version, pe._version, n):
            self.__dict__["_pair_key"] = (ne, pe, ne._version, pe._version, n)
            self.__dict__["_pair_ctx"] = torch.cat([ex(neutral), ex(positive)], dim=0).contiguous()
        x = diffused_latents.to(self.device)
        eps = self._engine("unet").forward(torch.cat([neutral.input(x), positive.input(x)], dim=0), torch.cat([idx, idx], dim=0), self._pair_ctx)
        mk = lambda e: Predictions(from_diffused_latents=diffused_latents, from_indices=idx, predicted_noise=e.contiguous(),
                                   schedule_alphas=self.schedule_alphas, schedule_sigmas=self.schedule_sigmas, encode=self.encode, decode=self.decode)
        return mk(eps[:n]), mk(eps[n:])

    # ---- text ----------------------------------------------------------------------------------------------
    def tokenize(self, texts) -> torch.Tensor:
        """CLIPTokenizer(padding="max_length", truncation=True) (stable_diffusion.py:304-311): int64 [N, 77], end-token padded."""
        if self._tokenizer is None:
            from ...utils.tokenizer import ClipTokenizer
            self._tokenizer = ClipTokenizer(self._bpe_path)
        return self._tokenizer(texts, context_length=self.text_config[0], pad="eot")

    def token_encodings(self, token_ids: torch.Tensor) -> torch.Tensor:
        hidden, _ = self._engine("text").forward(token_ids)
        return hidden

    def text_encodings(self, texts):
        return self.token_encodings(self.tokenize(texts))

    def latent_masks(self, masks, blur):
        """Masks [N, 1, H, W] in [0, 1] -> Gaussian-blurred, bilinearly down-sampled latent masks [N, 1, H/8, W/8] (stable_diffusion.py:325-341)."""
        n, c, h, w = masks.shape
        if h % 8 != 0:
            raise ValueError("Height must be divisible by 8")
        if w % 8 != 0:
            raise ValueError("Width must be divisible by 8")
        if c != 1:
            raise ValueError("Masks must be 1-channel")
        if masks.gt(1).any() or masks.lt(0).any():
            raise ValueError("Masks must be between 0 and 1")
        masks = masks.to(self.device).float()
        if blur is not None and blur > 0:
            ks = int(blur * 2) + 1
            x = torch.arange(ks, device=masks.device, dtype=torch.float32) - ks // 2
            if ks % 2 == 0:
                x = x + 0.5
            g = torch.exp(-x**2 / (2.0 * float(blur) ** 2))
            g = g / g.sum()
            p = ks // 2
            masks = torch.nn.functional.pad(masks, (p, p, p, p), mode="reflect")
            masks = torch.nn.functional.conv2d(torch.nn.functional.conv2d(masks, g.view(1, 1, 1, ks)), g.view(1, 1, ks, 1))
        return torch.nn.functional.interpolate(masks, size=(h // 8, w // 8), mode="bilinear")

    def conditioning(self, texts: List[str] = [""], inpainting_masks=None, inpainting_images=None, mask_blur=4.0, *,
                     token_ids: Optional[torch.Tensor] = None) -> Conditioning:
        """Conditioning from a list of texts (unconditional = the empty string), or from ``token_ids`` [N, T] directly; with the inpainting
        checkpoint also the latent masks and the latents of the masked images (stable_diffusion.py:343-375)."""
        enc = self.token_encodings(token_ids) if token_ids is not None else self.text_encodings(texts)
        if self.name == "runwayml/stable-diffusion-inpainting":
            inpainting_masks, inpainting_images = inpainting_masks.to(self.device), inpainting_images.to(self.device)
            latent_masks = self.latent_masks(inpainting_masks, mask_blur)
            latents = self.latents(inpainting_images * inpainting_masks.le(0.5) + 0.5 * inpainting_masks.gt(0.5).float())
            return Conditioning(model_name=self.name, encodings=enc, inpainting_latent_masks=latent_masks, inpainting_latents=latents)
        return Conditioning(model_name=self.name, encodings=enc)

    def diffuse_latents(self, denoised_latents, indices, noise=None):
        indices = self.indices(indices)
        if noise is None:
            noise = sampler.randn_like(denoised_latents)
        return sampler.lincomb2(denoised_latents, self.schedule_alphas[indices], noise, self.schedule_sigmas[indices])

    @torch.no_grad()
    def sample(self, text: str, from_index: int = 999, to_index: int = 0, n_steps: int = 50, guidance_scale: float = 7.0,
               n_resample: int = 0, init_image=None, inpainting_mask=None, mask_blur: float = 4.0, replace_diffused: bool = True):
        """Helper to sample a single image (stable_diffusion.py:384-491): yields the conditioned Predictions of every step."""
        neutral = self.conditioning(texts=[""], inpainting_masks=inpainting_mask, inpainting_images=init_image, mask_blur=mask_blur)
        positive = self.conditioning(texts=[text], inpainting_masks=inpainting_mask, inpainting_images=init_image, mask_blur=mask_blur)
        schedule_indices = self.schedule_indices(from_index=from_index, to_index=to_index, n_steps=n_steps)
        from_index = schedule_indices[0, 0]
        if init_image is None:
            if from_index != 999:
                raise ValueError("init_image must be provided if from_index < 999")
            diffused_latents = self.random_diffused_latents((1, 3, 512, 512))
        else:
            init_latents = self.latents(init_image)
            diffused_latents = self.diffuse_latents(init_latents, from_index)
        for from_index, to_index in schedule_indices:
            for _ in range(n_resample):
                un, pos = self.predictions_pair(diffused_latents, from_index, neutral, positive)
                diffused_latents = un.classifier_free_guidance(pos, guidance_scale=guidance_scale).resample(to_index)
            un, pos = self.predictions_pair(diffused_latents, from_index, neutral, positive)
            diffused_latents = un.classifier_free_guidance(pos, guidance_scale=guidance_scale).step(to_index)
            if replace_diffused and inpainting_mask is not None:     # peeks into the original masked image (stable_diffusion.py:477-483)
                lm = positive.inpainting_latent_masks
                diffused_latents = self.diffuse_latents(init_latents, to_index) * (1 - lm) + diffused_latents * lm
            yield pos
        yield self.predictions(diffused_latents, to_index, positive)
