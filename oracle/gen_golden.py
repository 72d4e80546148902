"""TEST INFRASTRUCTURE — generates tests/golden/*.npz by running the *reference* code.

Run in the build container only (needs /root/reference):  python -m oracle.gen_golden
Each fixture stores inputs + reference outputs (never reference source).  Weights
are not stored: they are re-derived from parameter names by
perceptor_amd.utils.synth (seed 0), identically here and on the GPU box.
"""
from __future__ import annotations

import os
import sys
import warnings

import numpy as np
import torch

warnings.filterwarnings("ignore")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import _refimport as R  # noqa: E402
from perceptor_amd.utils.synth import seeded_noise, synth_like  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def save(name, **arrs):
    os.makedirs(OUT, exist_ok=True)
    np.savez_compressed(os.path.join(OUT, name + ".npz"),
                        **{k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in arrs.items()})
    print("wrote", name, {k: tuple(np.asarray(v).shape) for k, v in arrs.items()})


def moments(y):
    f = y.flatten(1).double()
    return torch.stack([f.mean(1), f.std(1), f.norm(dim=1)], dim=1).float()


def gen_sampling():
    gd = R.ref("models.guided_diffusion.guided_diffusion")
    su = R.ref("models.guided_diffusion.script_util")
    P = R.ref("models.guided_diffusion.predictions").Predictions
    vd = R.ref("models.velocity_diffusion.velocity_diffusion")
    VP = R.ref("models.velocity_diffusion.predictions").Predictions
    # wrapper built by hand: the real __init__ downloads a checkpoint (guided_diffusion.py:25-36)
    g = gd.GuidedDiffusion.__new__(gd.GuidedDiffusion)
    torch.nn.Module.__init__(g)
    diffusion = su.create_gaussian_diffusion(steps=1000, learn_sigma=True, noise_schedule="linear")
    g.schedule_alphas = torch.nn.Parameter(torch.from_numpy(diffusion.alphas_cumprod).sqrt().float(), requires_grad=False)
    g.schedule_sigmas = torch.nn.Parameter((1 - torch.from_numpy(diffusion.alphas_cumprod)).sqrt().float(), requires_grad=False)
    out = dict(alphas=g.schedule_alphas.data, sigmas=g.schedule_sigmas.data,
               idx_50_r3=g.schedule_indices(n_steps=50, rho=3.0), idx_50_r7=g.schedule_indices(n_steps=50, rho=7.0),
               idx_250_r7=g.schedule_indices(n_steps=250, rho=7.0),
               idx_20_400=g.schedule_indices(n_steps=20, from_index=400, to_index=0),
               ts_50=vd.VelocityDiffusion.schedule_ts(n_steps=50), ts_500=vd.VelocityDiffusion.schedule_ts())
    img = seeded_noise((2, 3, 16, 16), 21) * 0.3 + 0.5
    eps = seeded_noise((2, 3, 16, 16), 22)
    grad = seeded_noise((2, 3, 16, 16), 23) * 2e-6
    fi, ti = torch.tensor([900, 37]), torch.tensor([850, 0])
    p = P(from_diffused_images=img, from_indices=fi, predicted_noise=eps,
          schedule_alphas=g.schedule_alphas.data, schedule_sigmas=g.schedule_sigmas.data)
    pg = p.guided(grad, guidance_scale=0.5, clamp_value=1e-6)
    out.update(img=img, eps=eps, grad=grad, fi=fi, ti=ti, eps_denoised=p.denoised_images, eps_step=p.step(ti),
               eps_guided=pg.predicted_noise, eps_guided_step=pg.step(ti),
               eps_forced=p.forced_denoised_images(p.denoised_images.clamp(0, 1)).predicted_noise,
               # reference defect: dynamic_threshold broadcasts an [N] threshold against NCHW, so it only
               # runs for N == 1 (predictions.py:156-172) -> fixture uses the first sample alone
               eps_dynthr=P(from_diffused_images=img[:1], from_indices=fi[:1], predicted_noise=eps[:1] * 3,
                            schedule_alphas=g.schedule_alphas.data, schedule_sigmas=g.schedule_sigmas.data
                            ).dynamic_threshold(0.95).predicted_noise,
               eps_reverse=P(from_diffused_images=img, from_indices=ti, predicted_noise=eps,
                             schedule_alphas=g.schedule_alphas.data, schedule_sigmas=g.schedule_sigmas.data).reverse_step(fi))
    ft, tt = torch.tensor([0.9, 0.05]), torch.tensor([0.8, 0.01])
    v = VP(from_diffused_images=img, from_ts=ft, velocities=eps)
    vg = v.guided(grad, guidance_scale=0.5, clamp_value=1e-6)
    out.update(ft=ft, tt=tt, v_denoised=v.denoised_images, v_eps=v.predicted_noise, v_step=v.step(tt),
               v_guided=vg.velocities, v_guided_step=vg.step(tt),
               v_forced=v.forced_denoised_images(v.denoised_images.clamp(0, 1)).velocities,
               v_forced_eps=v.forced_predicted_noise(eps * 0.5).velocities,
               v_static=v.static_threshold().velocities,
               v_dynthr=VP(from_diffused_images=img[:1], from_ts=ft[:1], velocities=eps[:1] * 3).dynamic_threshold(0.95).velocities)
    save("sampling", **out)


def gen_sampling2():
    """Stochastic and sort/quantile-based Predictions variants (row f3) and clamp_with_grad, from the reference classes.  The reference
    draws with torch.randn_like; here that name is bound to a function returning committed noise arrays (in call order), so the
    fixture pins the ARITHMETIC around the noise -- the product's own generator is pinned by known-answer vectors instead."""
    su = R.ref("models.guided_diffusion.script_util")
    P = R.ref("models.guided_diffusion.predictions").Predictions
    VP = R.ref("models.velocity_diffusion.predictions").Predictions
    cwg = R.ref("transforms.clamp_with_grad")
    diffusion = su.create_gaussian_diffusion(steps=1000, learn_sigma=True, noise_schedule="linear")
    alphas = torch.from_numpy(diffusion.alphas_cumprod).sqrt().float()
    sigmas = (1 - torch.from_numpy(diffusion.alphas_cumprod)).sqrt().float()
    img = seeded_noise((2, 3, 16, 16), 21) * 0.3 + 0.5
    eps = seeded_noise((2, 3, 16, 16), 22)
    noise = seeded_noise((2, 3, 16, 16), 24)
    fi, ti, hi = torch.tensor([900, 37]), torch.tensor([850, 0]), torch.tensor([950, 400])
    real = torch.randn_like
    torch.randn_like = lambda t, **kw: noise.to(t)
    try:
        p = P(from_diffused_images=img, from_indices=fi, predicted_noise=eps, schedule_alphas=alphas, schedule_sigmas=sigmas)
        out = dict(noise=noise, hi=hi,
                   eps_step_eta=p.step(ti, eta=0.7), eps_resample_noise=p.resample_noise(ti), eps_resample=p.resample(ti),
                   eps_noisy_reverse=p.noisy_reverse_step(hi),
                   eps_wasserstein=torch.stack([p.wasserstein_distance(), p.wasserstein_square_distance()]))
        ft, tt, ht = torch.tensor([0.9, 0.05]), torch.tensor([0.8, 0.01]), torch.tensor([0.95, 0.4])
        v = VP(from_diffused_images=img, from_ts=ft, velocities=eps)
        out.update(ht=ht, v_step_eta=v.step(tt, eta=0.7), v_resample_noise=v.resample_noise(tt), v_resample=v.resample(tt),
                   v_noisy_reverse=v.noisy_reverse_step(ht), v_reverse=VP(from_diffused_images=img, from_ts=tt, velocities=eps).reverse_step(ft),
                   v_wasserstein=torch.stack([v.wasserstein_distance(), v.wasserstein_square_distance()]))
    finally:
        torch.randn_like = real
    # a longer row for the sort / quantile kernels (crosses the 4096-element LDS block of the bitonic network), heavy ties included
    big = seeded_noise((3, 3, 40, 50), 25)
    big[1] = (big[1] * 4).round() / 4
    pb = P(from_diffused_images=big * 0.2 + 0.5, from_indices=torch.tensor([500, 20, 999]), predicted_noise=big, schedule_alphas=alphas,
           schedule_sigmas=sigmas)
    out.update(big=big, big_wasserstein=torch.stack([pb.wasserstein_distance(), pb.wasserstein_square_distance()]),
               big_quantiles=torch.stack([torch.quantile(big.flatten(1).abs(), q, dim=1) for q in (0.0, 0.5, 0.95, 0.999, 1.0)]))
    # clamp_with_grad forward and backward (transforms/clamp_with_grad.py:8-23)
    x = (seeded_noise((2, 3, 16, 16), 26) * 0.8 + 0.5).requires_grad_()
    g = seeded_noise((2, 3, 16, 16), 27)
    y = cwg.clamp_with_grad(x, 0.0, 1.0)
    y.backward(g)
    out.update(cwg_x=x.detach(), cwg_g=g, cwg_y=y.detach(), cwg_dx=x.grad)
    save("sampling2", **out)


ADM_TINY = {
    "a": dict(image_size=64, num_channels=32, num_res_blocks=1, channel_mult="1,2,2", learn_sigma=True,
              attention_resolutions="32,16", num_head_channels=16, use_scale_shift_norm=True, resblock_updown=True),
    "b": dict(image_size=64, num_channels=32, num_res_blocks=2, channel_mult="1,2", learn_sigma=True,
              attention_resolutions="32", num_heads=2, use_scale_shift_norm=False, resblock_updown=False,
              use_new_attention_order=True),
}


def gen_adm():
    su = R.ref("models.guided_diffusion.script_util")
    cm = R.ref("models.guided_diffusion.create_models")
    for tag, kw in ADM_TINY.items():
        m = su.create_model(**kw).eval()
        m.load_state_dict(synth_like(m.state_dict(), 0))
        x = seeded_noise((2, 3, 64, 64), 31)
        t = torch.tensor([10, 500])
        with torch.no_grad():
            y = m(x, t)
        save(f"adm_tiny_{tag}", x=x, t=t, y=y)
    for name, ctor, res in (("standard", cm.create_openimages_model, 128), ("pixelart", cm.create_pixelart_model, 64)):
        m, _ = ctor()
        m.convert_to_fp32()
        m.dtype = torch.float32          # fp32 oracle: the torso's fp16 cast is a GPU-side precision choice
        m.eval()
        m.load_state_dict(synth_like(m.state_dict(), 0))
        x = seeded_noise((1, 3, res, res), 32)
        t = torch.tensor([333])
        with torch.no_grad():
            y = m(x, t)
        save(f"adm_{name}_{res}", x=x, t=t, y_sub=y[:, :, ::4, ::4].contiguous(), y_mom=moments(y))


def gen_adm_grad():
    """Row f2, ADM UNet: the input gradient autograd computes through the reference's UNetModel -- upstream GuidedDiffusion.predicted_noise is
    differentiable (guided_diffusion.py:125-133) and its blocks run through CheckpointFunction (nn.py:138-189, unet.py:228-229, 292), whose
    recomputation gives autograd's gradient.  d sum(probe * eps) / d x with eps = the first 3 output channels (what predicted_noise returns):
    both tiny configs (both attention orders, both ResBlock flavours, up / down blocks) and the shipped 558 M-parameter net at 128x128."""
    su = R.ref("models.guided_diffusion.script_util")
    cm = R.ref("models.guided_diffusion.create_models")
    for tag, kw in ADM_TINY.items():
        m = su.create_model(**kw).eval()
        m.load_state_dict(synth_like(m.state_dict(), 0))
        for p_ in m.parameters():
            p_.requires_grad_(False)
        x = seeded_noise((2, 3, 64, 64), 31).requires_grad_(True)
        t = torch.tensor([10, 500])
        probe = seeded_noise((2, 3, 64, 64), 61)
        y = m(x, t)
        (g,) = torch.autograd.grad((y[:, :3] * probe).sum(), x)
        save(f"adm_tiny_{tag}_grad", t=t, g=g, y_mom=moments(y.detach()))
    m, _ = cm.create_openimages_model()
    m.convert_to_fp32()
    m.dtype = torch.float32
    m.eval()
    m.load_state_dict(synth_like(m.state_dict(), 0))
    for p_ in m.parameters():
        p_.requires_grad_(False)
    x = seeded_noise((1, 3, 128, 128), 32).requires_grad_(True)
    t = torch.tensor([333])
    probe = seeded_noise((1, 3, 128, 128), 62)
    y = m(x, t)
    (g,) = torch.autograd.grad((y[:, :3] * probe).sum(), x)
    save("adm_standard_128_grad", t=t, g_sub=g[:, :, ::2, ::2].contiguous(), g_mom=moments(g), y_mom=moments(y.detach()))


def gen_adm_fp16w():
    """Weights as a real checkpoint has them: full-precision fp32 values, of which the reference's own convert_to_fp16()
    (unet.py:610-616, fp16_util.py:16-23) casts the torso convolutions to fp16 -- NOT bf16-representable, so packing them to bf16
    loses 3 mantissa bits, and the fp32 time / output layers are not representable in either 16-bit type.  The names of the
    tensors the reference rounded are stored with the fixture (data, so the test can apply the same rounding)."""
    su = R.ref("models.guided_diffusion.script_util")
    kw = ADM_TINY["a"]
    m = su.create_model(**kw).eval()
    raw = synth_like(m.state_dict(), 0, rounding="none")
    m.load_state_dict(raw)
    m.convert_to_fp16()
    rounded = [k for k, v in m.state_dict().items() if v.dtype == torch.float16]
    m.convert_to_fp32()
    m.dtype = torch.float32          # run the fp16-VALUED weights in fp32 arithmetic (CPU oracle of the GPU path's weights)
    x = seeded_noise((2, 3, 64, 64), 31)
    t = torch.tensor([10, 500])
    with torch.no_grad():
        y = m(x, t)
    chk = {k: float(v.double().abs().sum()) for k, v in m.state_dict().items()}
    save("adm_tiny_a_fp16w", x=x, t=t, y=y, rounded_keys=np.array(rounded), weight_abs_sum=np.array([chk[k] for k in sorted(chk)]))


def gen_vdiff():
    y2 = R.ref("models.velocity_diffusion.yfcc_2")
    cc = R.ref("models.velocity_diffusion.cc12m_1")
    m = y2.YFCC2Model().eval()
    m.load_state_dict(synth_like(m.state_dict(), 0))
    x = seeded_noise((1, 3, 128, 128), 41)
    t = torch.tensor([0.3])
    with torch.no_grad():
        y = m(x, t)
    save("vdiff_yfcc_2_128", x=x, t=t, y_sub=y[:, :, ::4, ::4].contiguous(), y_mom=moments(y))
    del m
    m = cc.CC12M1Model().eval()
    m.load_state_dict(synth_like(m.state_dict(), 0))
    x = seeded_noise((1, 3, 64, 64), 42)
    t = torch.tensor([0.7])
    ce = seeded_noise((1, 512), 43)
    with torch.no_grad():
        y = m(x, t, ce)
    save("vdiff_cc12m_1_64", x=x, t=t, clip_embed=ce, y_sub=y[:, :, ::2, ::2].contiguous(), y_mom=moments(y))


def gen_vdiff_grad():
    """Row f2: the input gradient autograd computes through the reference's YFCC2Model (what guided_resample_ backpropagates,
    losses/velocity_diffusion.py:33-61): d sum(probe * v) / d x at 128x128, full 968 M-parameter net, name-keyed weights."""
    y2 = R.ref("models.velocity_diffusion.yfcc_2")
    m = y2.YFCC2Model().eval()
    m.load_state_dict(synth_like(m.state_dict(), 0))
    for p_ in m.parameters():
        p_.requires_grad_(False)
    x = seeded_noise((1, 3, 128, 128), 41).requires_grad_(True)
    t = torch.tensor([0.3])
    probe = seeded_noise((1, 3, 128, 128), 46)
    v = m(x, t)
    (g,) = torch.autograd.grad((v * probe).sum(), x)
    save("vdiff_yfcc_2_128_grad", t=t, g_sub=g[:, :, ::2, ::2].contiguous(), g_mom=moments(g), v_mom=moments(v.detach()))


def gen_vdiff_grad_cc12m():
    """As gen_vdiff_grad for the CLIP-conditioned CC12M1Model (GroupNorm(1, C) + Modulation2d blocks) at 64x64."""
    cc = R.ref("models.velocity_diffusion.cc12m_1")
    m = cc.CC12M1Model().eval()
    m.load_state_dict(synth_like(m.state_dict(), 0))
    for p_ in m.parameters():
        p_.requires_grad_(False)
    x = seeded_noise((1, 3, 64, 64), 42).requires_grad_(True)
    t = torch.tensor([0.7])
    ce = seeded_noise((1, 512), 43).requires_grad_(True)         # (round 3: also d / d clip_embed -- upstream keeps the conditioning in the graph)
    probe = seeded_noise((1, 3, 64, 64), 47)
    v = m(x, t, ce)
    g, g_ce = torch.autograd.grad((v * probe).sum(), (x, ce))
    save("vdiff_cc12m_1_64_grad", t=t, g=g, g_ce=g_ce, v_mom=moments(v.detach()))


def gen_vdiff_grad_wikiart():
    """As gen_vdiff_grad for WikiArt256Model (no normalisation, 128-channel heads, nearest upsampling, skip-first concat) at 64x64."""
    wa = R.ref("models.velocity_diffusion.wikiart_256")
    m = wa.WikiArt256Model().eval()
    m.load_state_dict(synth_like(m.state_dict(), 0, 0.6))
    for p_ in m.parameters():
        p_.requires_grad_(False)
    x = seeded_noise((1, 3, 64, 64), 45).requires_grad_(True)
    t = torch.tensor([0.6])
    probe = seeded_noise((1, 3, 64, 64), 48)
    v = m(x, t)
    (g,) = torch.autograd.grad((v * probe).sum(), x)
    save("vdiff_wikiart_64_grad", t=t, g=g, v_mom=moments(v.detach()))


def gen_vdiff2():
    y1 = R.ref("models.velocity_diffusion.yfcc_1")
    wa = R.ref("models.velocity_diffusion.wikiart_256")
    # wikiart has no normalisation anywhere (not even in attention): unit-gain random weights overflow fp16, so gain 0.6
    for name, cls, res, t, seed, gain in (("yfcc_1", y1.YFCC1Model, 128, 0.45, 44, 1.0), ("wikiart", wa.WikiArt256Model, 64, 0.6, 45, 0.6)):
        m = cls().eval()
        m.load_state_dict(synth_like(m.state_dict(), 0, gain))
        x = seeded_noise((1, 3, res, res), seed)
        tt = torch.tensor([t])
        with torch.no_grad():
            y = m(x, tt)
        save(f"vdiff_{name}_{res}", x=x, t=tt, y_sub=y[:, :, ::2, ::2].contiguous(), y_mom=moments(y))
        del m


def gen_clip():
    rz = R.ref("transforms.resize.resize_right")
    ru = R.ref("models.ruclip.model")
    out = {}
    for tag, shape, target in (("512_224", (1, 3, 512, 512), (224, 224)), ("256_224", (1, 3, 256, 256), (224, 224)),
                               ("128_224", (1, 3, 128, 128), (224, 224)), ("96x160_64", (1, 3, 96, 160), (64, 64))):
        img = seeded_noise(shape, 51) * 0.25 + 0.5
        out["rz_" + tag] = rz.resize(img, out_shape=target)[:, :, ::3, ::3].contiguous()
    save("clip_resize", **out)
    from oracle.clip_vit import CLIP_MEAN, CLIP_STD, VIT_CONFIGS
    for tag, size, n in (("tiny", 48, 2), ("tiny-odd", 40, 2), ("ViT-B-32", 256, 2)):
        res, patch, width, layers, heads, odim = VIT_CONFIGS[tag]
        m = ru.VisionTransformer(res, patch, width, layers, heads, odim).eval()
        m.load_state_dict(synth_like(m.state_dict(), 0))
        img = (seeded_noise((n, 3, size, size), 52) * 0.25 + 0.5).requires_grad_(True)
        mean = torch.tensor(CLIP_MEAN)[None, :, None, None]
        std = torch.tensor(CLIP_STD)[None, :, None, None]
        probe = seeded_noise((n, odim), 53)
        e = m((rz.resize(img, out_shape=(res, res)) - mean) / std)   # QuickGELU tower (ruclip/model.py:20-23)
        en = torch.nn.functional.normalize(e)
        (g,) = torch.autograd.grad((en * probe).sum(), img)
        save(f"clip_vit_{tag}", img=img.detach(), probe=probe, emb=e.detach(), emb_n=en.detach(),
             grad_sub=g[:, :, ::4, ::4].contiguous(), grad_mom=moments(g))


def _hf_vision_tower(cfg, sd, quick_gelu):
    """transformers' CLIPVisionModelWithProjection built from a config object (no download) and loaded with the open_clip-named
    synthetic weights ``sd`` -- an INDEPENDENT implementation of the tower open-clip-torch 2.0.2 ships (SURVEY §8c cross-check)."""
    # the reference loader's torchvision stand-in (oracle/_refimport.py) confuses transformers' optional-dependency probe: hide it
    hidden = {k: sys.modules.pop(k) for k in [k for k in sys.modules if k == "torchvision" or k.startswith("torchvision.")]}
    try:
        from transformers import CLIPVisionConfig, CLIPVisionModelWithProjection
    finally:
        sys.modules.update(hidden)
    res, patch, width, layers, heads, odim = cfg
    c = CLIPVisionConfig(hidden_size=width, intermediate_size=4 * width, num_hidden_layers=layers, num_attention_heads=heads,
                         image_size=res, patch_size=patch, projection_dim=odim, hidden_act="quick_gelu" if quick_gelu else "gelu",
                         layer_norm_eps=1e-5, attention_dropout=0.0)
    c._attn_implementation = "eager"
    m = CLIPVisionModelWithProjection(c).eval()
    hf = {"vision_model.embeddings.class_embedding": sd["class_embedding"],
          "vision_model.embeddings.patch_embedding.weight": sd["conv1.weight"],
          "vision_model.embeddings.position_embedding.weight": sd["positional_embedding"],
          "vision_model.pre_layrnorm.weight": sd["ln_pre.weight"], "vision_model.pre_layrnorm.bias": sd["ln_pre.bias"],
          "vision_model.post_layernorm.weight": sd["ln_post.weight"], "vision_model.post_layernorm.bias": sd["ln_post.bias"],
          "visual_projection.weight": sd["proj"].t().contiguous()}
    for i in range(layers):
        a, b = f"transformer.resblocks.{i}.", f"vision_model.encoder.layers.{i}."
        wq, wk, wv = sd[a + "attn.in_proj_weight"].chunk(3, dim=0)
        bq, bk, bv = sd[a + "attn.in_proj_bias"].chunk(3, dim=0)
        for nm, w_, b_ in (("q", wq, bq), ("k", wk, bk), ("v", wv, bv)):
            hf[b + f"self_attn.{nm}_proj.weight"], hf[b + f"self_attn.{nm}_proj.bias"] = w_, b_
        for src, dst in (("attn.out_proj", "self_attn.out_proj"), ("ln_1", "layer_norm1"), ("ln_2", "layer_norm2"),
                         ("mlp.c_fc", "mlp.fc1"), ("mlp.c_proj", "mlp.fc2")):
            hf[b + dst + ".weight"], hf[b + dst + ".bias"] = sd[a + src + ".weight"], sd[a + src + ".bias"]
    missing, unexpected = m.load_state_dict(hf, strict=False)
    assert not unexpected and all("position_ids" in k for k in missing), (missing, unexpected)
    return m


def gen_clip_hf():
    """ViT-L/14 (the benchmarked tower; exact GELU as the laion2b weights use, models/clip.py:21-27) and an exact-GELU tiny tower:
    resize (the reference's ResizeRight) -> normalise -> transformers CLIP vision tower with the name-keyed synthetic weights."""
    rz = R.ref("transforms.resize.resize_right")
    from oracle.clip_vit import CLIP_MEAN, CLIP_STD, VIT_CONFIGS, vit_state_dict_shapes
    from perceptor_amd.utils.synth import synth_state_dict
    mean = torch.tensor(CLIP_MEAN)[None, :, None, None]
    std = torch.tensor(CLIP_STD)[None, :, None, None]
    for tag, size, n, quick in (("tiny-odd", 40, 2, False), ("ViT-L-14", 256, 1, False), ("ViT-L-14", 256, 1, True)):
        cfg = VIT_CONFIGS[tag]
        sd = synth_state_dict(vit_state_dict_shapes(cfg), 0)
        m = _hf_vision_tower(cfg, sd, quick)
        img = (seeded_noise((n, 3, size, size), 52) * 0.25 + 0.5).requires_grad_(True)
        probe = seeded_noise((n, cfg[5]), 53)
        e = m(pixel_values=(rz.resize(img, out_shape=(cfg[0], cfg[0])) - mean) / std).image_embeds
        en = torch.nn.functional.normalize(e)
        (g,) = torch.autograd.grad((en * probe).sum(), img)
        # the image is seeded_noise((n, 3, size, size), 52) * 0.25 + 0.5: re-derived by the tests, not stored (786 KB at 256x256)
        save(f"clip_hf_{tag}_{'quickgelu' if quick else 'gelu'}", img_shape=np.array(img.shape), probe=probe, emb=e.detach(), emb_n=en.detach(),
             grad_sub=g[:, :, ::4, ::4].contiguous(), grad_mom=moments(g))


def _text_ids(n, t, vocab, seed):
    """Random prompts: BOS-like id, words, one EOT (= vocab - 1, the largest id) at a per-row position, zero padding behind it."""
    g = torch.Generator().manual_seed(seed)
    ids = torch.zeros((n, t), dtype=torch.int64)
    for i in range(n):
        L = int(torch.randint(2, t, (1,), generator=g))       # position of the EOT token
        ids[i, 0] = vocab - 2
        ids[i, 1:L] = torch.randint(1, vocab - 2, (L - 1,), generator=g)
        ids[i, L] = vocab - 1
    return ids


def _hf_text_tower(cfg, sd, quick_gelu):
    """transformers' CLIPTextModelWithProjection from a config object (no download), loaded with the open_clip-named weights."""
    hidden = {k: sys.modules.pop(k) for k in [k for k in sys.modules if k == "torchvision" or k.startswith("torchvision.")]}
    try:
        from transformers import CLIPTextConfig, CLIPTextModelWithProjection
    finally:
        sys.modules.update(hidden)
    ctx, vocab, width, layers, heads, odim = cfg
    c = CLIPTextConfig(vocab_size=vocab, hidden_size=width, intermediate_size=4 * width, num_hidden_layers=layers, num_attention_heads=heads,
                       max_position_embeddings=ctx, projection_dim=odim, hidden_act="quick_gelu" if quick_gelu else "gelu",
                       layer_norm_eps=1e-5, attention_dropout=0.0, eos_token_id=vocab - 1, bos_token_id=vocab - 2, pad_token_id=0)
    c._attn_implementation = "eager"
    m = CLIPTextModelWithProjection(c).eval()
    hf = {"text_model.embeddings.token_embedding.weight": sd["token_embedding.weight"],
          "text_model.embeddings.position_embedding.weight": sd["positional_embedding"],
          "text_model.final_layer_norm.weight": sd["ln_final.weight"], "text_model.final_layer_norm.bias": sd["ln_final.bias"],
          "text_projection.weight": sd["text_projection"].t().contiguous()}
    for i in range(layers):
        a, b = f"transformer.resblocks.{i}.", f"text_model.encoder.layers.{i}."
        wq, wk, wv = sd[a + "attn.in_proj_weight"].chunk(3, dim=0)
        bq, bk, bv = sd[a + "attn.in_proj_bias"].chunk(3, dim=0)
        for nm, w_, b_ in (("q", wq, bq), ("k", wk, bk), ("v", wv, bv)):
            hf[b + f"self_attn.{nm}_proj.weight"], hf[b + f"self_attn.{nm}_proj.bias"] = w_, b_
        for src, dst in (("attn.out_proj", "self_attn.out_proj"), ("ln_1", "layer_norm1"), ("ln_2", "layer_norm2"),
                         ("mlp.c_fc", "mlp.fc1"), ("mlp.c_proj", "mlp.fc2")):
            hf[b + dst + ".weight"], hf[b + dst + ".bias"] = sd[a + src + ".weight"], sd[a + src + ".bias"]
    missing, unexpected = m.load_state_dict(hf, strict=False)
    assert not unexpected and all("position_ids" in k for k in missing), (missing, unexpected)
    return m


def gen_clip_text():
    """CLIP text tower: (i) the reference's in-tree ruclip CLIP.encode_text (ruclip/model.py:204-228, QuickGELU) on a tiny config,
    (ii) transformers' CLIPTextModelWithProjection on a tiny exact-GELU config and on the ViT-L/14 text tower (QuickGELU: the
    encoder StableDiffusion conditions on, stable_diffusion.py:298-301) -- all with the name-keyed synthetic weights."""
    from oracle.clip_text import TEXT_CONFIGS, text_state_dict_shapes
    from perceptor_amd.utils.synth import synth_state_dict
    ru = R.ref("models.ruclip.model")
    cfg = TEXT_CONFIGS["tiny"]
    ctx, vocab, width, layers, heads, odim = cfg
    m = ru.CLIP(embed_dim=odim, image_resolution=32, vision_layers=1, vision_width=64, vision_patch_size=16, context_length=ctx,
                vocab_size=vocab, transformer_width=width, transformer_heads=heads, transformer_layers=layers, eos_id=vocab - 1).eval()
    sd = synth_state_dict(text_state_dict_shapes(cfg), 0)
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected and all(k.startswith("visual.") or k == "logit_scale" for k in missing), (missing, unexpected)
    ids = _text_ids(3, ctx, vocab, 61)
    with torch.no_grad():
        save("clip_text_ruclip_tiny", ids=ids, pooled=m.encode_text(ids))
    for tag, n, t, quick in (("tiny-wide", 3, 24, False), ("ViT-L-14", 2, 77, True)):
        cfg = TEXT_CONFIGS[tag]
        sd = synth_state_dict(text_state_dict_shapes(cfg), 0)
        m = _hf_text_tower(cfg, sd, quick)
        ids = _text_ids(n, t, cfg[1], 62)
        with torch.no_grad():
            o = m(input_ids=ids)
        save(f"clip_text_hf_{tag}_{'quickgelu' if quick else 'gelu'}", ids=ids, hidden=o.last_hidden_state, pooled=o.text_embeds)


def gen_clip_hf_model():
    """transformers' CLIPModel (both towers, one config object, no download) loaded DIRECTLY with transformers-named synthetic weights:
    the fixture for models.TransformersOpenAICLIP, whose state dict uses those names (models/transformers_openai_clip.py:58-68)."""
    hidden = {k: sys.modules.pop(k) for k in [k for k in sys.modules if k == "torchvision" or k.startswith("torchvision.")]}
    try:
        from transformers import CLIPConfig, CLIPModel, CLIPTextConfig, CLIPVisionConfig
    finally:
        sys.modules.update(hidden)
    rz = R.ref("transforms.resize.resize_right")
    from oracle.clip_vit import CLIP_MEAN, CLIP_STD
    from perceptor_amd.engine.text import hf_text_state_dict_shapes
    from perceptor_amd.engine.vit import hf_vision_state_dict_shapes
    from perceptor_amd.utils.synth import synth_state_dict
    vcfg, tcfg = (32, 8, 64, 2, 1, 32), (16, 96, 64, 2, 1, 32)
    vc = CLIPVisionConfig(hidden_size=64, intermediate_size=256, num_hidden_layers=2, num_attention_heads=1, image_size=32, patch_size=8,
                          projection_dim=32, hidden_act="quick_gelu", layer_norm_eps=1e-5)
    tc = CLIPTextConfig(vocab_size=96, hidden_size=64, intermediate_size=256, num_hidden_layers=2, num_attention_heads=1,
                        max_position_embeddings=16, projection_dim=32, hidden_act="quick_gelu", layer_norm_eps=1e-5, eos_token_id=95,
                        bos_token_id=94, pad_token_id=0)
    cfg = CLIPConfig(text_config=tc.to_dict(), vision_config=vc.to_dict(), projection_dim=32)
    cfg._attn_implementation = "eager"
    m = CLIPModel(cfg).eval()
    sd = synth_state_dict({**hf_vision_state_dict_shapes(vcfg), **hf_text_state_dict_shapes(tcfg)}, 0)
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected and all("position_ids" in k or k == "logit_scale" for k in missing), (missing, unexpected)
    img = (seeded_noise((2, 3, 40, 40), 52) * 0.25 + 0.5).requires_grad_(True)
    mean, std = torch.tensor(CLIP_MEAN)[None, :, None, None], torch.tensor(CLIP_STD)[None, :, None, None]
    vo = m.vision_model(pixel_values=(rz.resize(img, out_shape=(32, 32)) - mean) / std)
    ie = m.visual_projection(vo.pooler_output)
    ids = _text_ids(3, 16, 96, 63)
    with torch.no_grad():
        to = m.text_model(input_ids=ids)
        te = m.text_projection(to.pooler_output)
    ien, ten = torch.nn.functional.normalize(ie), torch.nn.functional.normalize(te)
    dist = (ten[:, None] - ien[None, :]).norm(dim=2).div(2).arcsin().square().mul(2)
    (g,) = torch.autograd.grad(dist.mean(), img)
    save("clip_hf_model_tiny", img=img.detach(), ids=ids, image_embeds=ie.detach(), image_hidden=vo.last_hidden_state.detach(),
         image_pooler=vo.pooler_output.detach(), text_embeds=te, text_hidden=to.last_hidden_state, text_pooler=to.pooler_output,
         distance=dist.detach(), grad=g)


def gen_sd_schedule():
    """StableDiffusion.schedule_indices (models/stable_diffusion/stable_diffusion.py:132-173, default rho = 3) cannot be run as that class
    (the module imports diffusers), but its body is statement for statement GuidedDiffusion.schedule_indices (guided_diffusion.py:58-96)
    over `self.schedule_alphas / schedule_sigmas`: the reference's GuidedDiffusion method is run here on the SD scaled-linear tables."""
    gd = R.ref("models.guided_diffusion.guided_diffusion")
    g = gd.GuidedDiffusion.__new__(gd.GuidedDiffusion)
    torch.nn.Module.__init__(g)
    betas = torch.linspace(0.00085 ** 0.5, 0.012 ** 0.5, 1000, dtype=torch.float32) ** 2
    ac = torch.cumprod(1.0 - betas, dim=0)
    g.schedule_alphas = torch.nn.Parameter(ac.sqrt(), requires_grad=False)
    g.schedule_sigmas = torch.nn.Parameter((1 - ac).sqrt(), requires_grad=False)
    save("sd_schedule", idx_50=g.schedule_indices(n_steps=50, rho=3.0), idx_500=g.schedule_indices(n_steps=500, rho=3.0),
         idx_20_500_20=g.schedule_indices(n_steps=20, from_index=500, to_index=20, rho=3.0))


def gen_sd_predictions():
    """The latent eps-form Predictions of the reference's StableDiffusion path (models/stable_diffusion/predictions.py:10-250), run as the
    reference class: DDIM (eta = 0 and > 0 with injected noise), reverse / noisy-reverse / resample steps, guidance, classifier-free
    guidance, forced variants, latent thresholding, Wasserstein statistics; encode / decode are identity callables here (the VAE is pinned
    separately).  The schedule is DDPMScheduler's scaled-linear table, restated (diffusers is absent): fp32 linspace of sqrt(beta)."""
    P = R.ref("models.stable_diffusion.predictions").Predictions
    betas = torch.linspace(0.00085 ** 0.5, 0.012 ** 0.5, 1000, dtype=torch.float32) ** 2
    ac = torch.cumprod(1.0 - betas, dim=0)
    alphas, sigmas = ac.sqrt(), (1 - ac).sqrt()
    x = seeded_noise((2, 4, 16, 16), 31)
    eps, eps2, noise, guide = (seeded_noise((2, 4, 16, 16), s_) for s_ in (32, 33, 34, 35))
    fi, ti, hi = torch.tensor([900, 37]), torch.tensor([850, 0]), torch.tensor([950, 400])
    ident = lambda t: t
    LT = sys.modules["lantern"].Tensor        # these two fields are annotated with the bare lantern.Tensor type: present the tables as that (sub)type
    mk = lambda e: P(from_diffused_latents=x, from_indices=fi, predicted_noise=e, schedule_alphas=alphas.as_subclass(LT),
                     schedule_sigmas=sigmas.as_subclass(LT), encode=ident, decode=ident)
    real = torch.randn_like
    torch.randn_like = lambda t, **kw: noise.to(t)
    try:
        p, p2 = mk(eps), mk(eps2)
        out = dict(x=x, eps=eps, eps2=eps2, noise=noise, guide=guide * 1e-6, fi=fi, ti=ti, hi=hi,
                   denoised=p.denoised_latents, step=p.step(ti), step_eta=p.step(ti, eta=0.7), reverse=p.reverse_step(hi),
                   resample_noise=p.resample_noise(ti), resample=p.resample(ti), noisy_reverse=p.noisy_reverse_step(hi),
                   guided=p.guided(guide * 1e-6, guidance_scale=0.5, clamp_value=1e-6).predicted_noise,
                   cfg=p.classifier_free_guidance(p2, guidance_scale=7.0).predicted_noise,
                   forced=p.forced_denoised_latents(x * 0.5).predicted_noise,
                   # (the [N] threshold only broadcasts for N = 1 upstream, as in the guided-diffusion class: one sample)
                   latent_thr=P(from_diffused_latents=x[:1], from_indices=fi[:1], predicted_noise=eps[:1] * 3, schedule_alphas=alphas.as_subclass(LT),
                                schedule_sigmas=sigmas.as_subclass(LT), encode=ident, decode=ident).latent_dynamic_threshold(0.95).predicted_noise,
                   wasserstein=torch.stack([p.wasserstein_distance(), p.wasserstein_square_distance()]))
    finally:
        torch.randn_like = real
    save("sd_predictions", **out)


def _ldm_unet_keys(cfg, sd):
    """diffusers key names (what the StableDiffusion engines / oracle use) -> the key names of the reference's vendored CompVis UNetModel
    (models/latent_diffusion/ldm/modules/diffusionmodules/openaimodel.py:413-1009), for layers_per_block = 2 ... any; the published
    checkpoint-conversion correspondence between the two layouts of the SAME network, restated."""
    L, nl = cfg.layers_per_block, len(cfg.block_out)
    res = (("norm1", "in_layers.0"), ("conv1", "in_layers.2"), ("time_emb_proj", "emb_layers.1"), ("norm2", "out_layers.0"),
           ("conv2", "out_layers.3"), ("conv_shortcut", "skip_connection"))
    pre = {"time_embedding.linear_1": "time_embed.0", "time_embedding.linear_2": "time_embed.2", "conv_in": "input_blocks.0.0",
           "conv_norm_out": "out.0", "conv_out": "out.2", "mid_block.resnets.0": "middle_block.0", "mid_block.attentions.0": "middle_block.1",
           "mid_block.resnets.1": "middle_block.2"}
    for i in range(nl):
        for j in range(L):
            pre[f"down_blocks.{i}.resnets.{j}"] = f"input_blocks.{1 + i * (L + 1) + j}.0"
            pre[f"down_blocks.{i}.attentions.{j}"] = f"input_blocks.{1 + i * (L + 1) + j}.1"
        pre[f"down_blocks.{i}.downsamplers.0.conv"] = f"input_blocks.{(i + 1) * (L + 1)}.0.op"
    ca = list(reversed(cfg.cross_attn))
    for i in range(nl):
        for j in range(L + 1):
            pre[f"up_blocks.{i}.resnets.{j}"] = f"output_blocks.{i * (L + 1) + j}.0"
            pre[f"up_blocks.{i}.attentions.{j}"] = f"output_blocks.{i * (L + 1) + j}.1"
        pre[f"up_blocks.{i}.upsamplers.0.conv"] = f"output_blocks.{i * (L + 1) + L}.{2 if ca[i] else 1}.conv"
    out = {}
    for k, v in sd.items():
        hit = max((p for p in pre if k.startswith(p + ".")), key=len)
        rest = k[len(hit) + 1:]
        if ".resnets." in hit or hit in ("mid_block.resnets.0", "mid_block.resnets.1"):
            for a, b in res:
                if rest.startswith(a + "."):
                    rest = b + rest[len(a):]
                    break
        out[pre[hit] + "." + rest] = v
    return out


def _ldm_vae_keys(cfg, sd, part):
    """diffusers AutoencoderKL key names -> the reference's vendored ldm Encoder / Decoder (ldm/modules/diffusionmodules/model.py:379-600)."""
    nl, out = len(cfg.block_out), {}
    res = (("conv_shortcut", "nin_shortcut"),)
    att = (("group_norm", "norm"), ("query", "q"), ("key", "k"), ("value", "v"), ("proj_attn", "proj_out"))
    for k, v in sd.items():
        if not k.startswith(part + "."):
            continue
        r = k[len(part) + 1:]
        r = r.replace("mid_block.resnets.0", "mid.block_1").replace("mid_block.resnets.1", "mid.block_2").replace("mid_block.attentions.0", "mid.attn_1")
        r = r.replace("conv_norm_out", "norm_out")
        for i in range(nl):
            for j in range(cfg.layers_per_block + 1):
                r = r.replace(f"up_blocks.{i}.resnets.{j}.", f"up.{nl - 1 - i}.block.{j}.").replace(f"down_blocks.{i}.resnets.{j}.", f"down.{i}.block.{j}.")
            r = r.replace(f"up_blocks.{i}.upsamplers.0.", f"up.{nl - 1 - i}.upsample.").replace(f"down_blocks.{i}.downsamplers.0.", f"down.{i}.downsample.")
        for a, b in res:
            r = r.replace(a, b)
        if "attn_1" in r:
            for a, b in att:
                r = r.replace("." + a + ".", "." + b + ".")
            if r.endswith(".weight") and v.ndim == 2:
                v = v[:, :, None, None]          # linear -> the 1x1 convolution the ldm AttnBlock holds
        out[r] = v
    return out


def gen_sd_ldm():
    """StableDiffusion's UNet and VAE as the REFERENCE's own tree holds them: the vendored CompVis latent-diffusion modules
    (models/latent_diffusion/ldm/...: openaimodel.UNetModel with SpatialTransformer = the original of diffusers' UNet2DConditionModel;
    model.Encoder / Decoder = the halves of AutoencoderKL).  diffusers 0.6.0 itself is absent; these are the same published network under the
    original key layout, so they pin oracle/sd.py's restatement with reference code: tiny UNet in full, the 860 M SD-v1 UNet at 16x16 latents
    with a 77-token context, tiny and SD-v1 VAE decoder / encoder."""
    from oracle import sd as osd
    from perceptor_amd.utils.synth import synth_state_dict
    om = R.ref("models.latent_diffusion.ldm.modules.diffusionmodules.openaimodel")
    mm = R.ref("models.latent_diffusion.ldm.modules.diffusionmodules.model")
    for tag, cfg, n, hw, tc in (("tiny", osd.SD_TINY, 2, 16, 7), ("v1", osd.SD_V1, 1, 16, 77)):
        bo = cfg.block_out
        ds_attn = [2 ** i for i, c in enumerate(cfg.cross_attn) if c]
        m = om.UNetModel(image_size=hw, in_channels=cfg.in_channels, out_channels=cfg.out_channels, model_channels=bo[0], attention_resolutions=ds_attn,
                         num_res_blocks=cfg.layers_per_block, channel_mult=[c // bo[0] for c in bo], num_heads=cfg.heads, use_spatial_transformer=True,
                         transformer_depth=1, context_dim=cfg.context_dim, use_checkpoint=False, legacy=False).eval()
        sd = synth_state_dict(osd.unet_state_dict_shapes(cfg), 0)
        m.load_state_dict(_ldm_unet_keys(cfg, sd), strict=True)
        x, ctx = seeded_noise((n, cfg.in_channels, hw, hw), 71), seeded_noise((n, tc, cfg.context_dim), 72)
        t = torch.tensor([981, 20][:n])
        with torch.no_grad():
            y = m(x, t, context=ctx)
        save(f"sd_ldm_unet_{tag}", eps=y, t=t, hw=np.array(hw), tc=np.array(tc))
    for tag, cfg, hw in (("tiny", osd.VAE_TINY, 16), ("v1", osd.VAE_V1, 8)):
        dd = dict(ch=cfg.block_out[0], out_ch=cfg.out_channels, ch_mult=tuple(c // cfg.block_out[0] for c in cfg.block_out), num_res_blocks=cfg.layers_per_block,
                  attn_resolutions=[], in_channels=cfg.out_channels, resolution=256, z_channels=cfg.latent_channels)
        sd = synth_state_dict({**osd.vae_encoder_state_dict_shapes(cfg), **osd.vae_decoder_state_dict_shapes(cfg)}, 0)
        dec, enc = mm.Decoder(**dd).eval(), mm.Encoder(**dd, double_z=True).eval()
        dec.load_state_dict(_ldm_vae_keys(cfg, sd, "decoder"), strict=True)
        enc.load_state_dict(_ldm_vae_keys(cfg, sd, "encoder"), strict=True)
        z = seeded_noise((1, cfg.latent_channels, hw, hw), 73)
        img = seeded_noise((1, 3, 8 * hw if tag == "v1" else 2 * hw, 8 * hw if tag == "v1" else 2 * hw), 74) * 0.5
        with torch.no_grad():
            # post_quant_conv / quant_conv are 1x1 convolutions of AutoencoderKL itself (ldm/models/autoencoder.py needs pytorch_lightning): plain conv2d here
            y = dec(torch.nn.functional.conv2d(z, sd["post_quant_conv.weight"], sd["post_quant_conv.bias"]))
            mom = torch.nn.functional.conv2d(enc(img), sd["quant_conv.weight"], sd["quant_conv.bias"])
        save(f"sd_ldm_vae_{tag}", dec=y, mean=mom[:, :cfg.latent_channels], logvar=mom[:, cfg.latent_channels:], hw=np.array(hw), img_hw=np.array(img.shape[-1]))


TOKENIZER_PROMPTS = [
    "a photograph of a playful cat", "painting of a dog", "", "  Hello,   World!  it's 2023 -- don't panic...",
    "An astronaut riding a horse on Mars; 4k, trending on artstation (highly detailed)", "fish &amp; chips &lt;3 #tasty @home 100% / 50$",
    "supercalifragilisticexpialidocious antidisestablishmentarianism", "the quick brown fox jumps over the lazy dog " * 12,
]


def gen_tokenizer():
    """Token ids of the reference's own CLIP tokenizer (models/slip/tokenizer.py:70-165, reading the merge list that lies next to it)
    for ASCII prompts.  The module imports ftfy (absent here): a stand-in whose fix_text is the identity is registered for the import
    -- the identity is what ftfy.fix_text returns for well-formed ASCII text, so no arithmetic of the path is replaced."""
    import types
    sys.modules.setdefault("ftfy", types.SimpleNamespace(fix_text=lambda t: t))
    tk = R.ref("models.slip.tokenizer").SimpleTokenizer()
    ids = torch.stack([tk(p) for p in TOKENIZER_PROMPTS])
    save("clip_tokenizer", ids=ids, lengths=np.array([len(tk.encode(p)) for p in TOKENIZER_PROMPTS]))


if __name__ == "__main__" and len(sys.argv) > 1:
    for name in sys.argv[1:]:
        globals()["gen_" + name]()
    sys.exit(0)

if __name__ == "__main__":
    which = sys.argv[1:] or ["sampling", "adm", "vdiff", "clip"]
    torch.manual_seed(0)
    for w in which:
        globals()["gen_" + w]()
