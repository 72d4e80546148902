"""GPU parity of the StableDiffusion engines (engine/sd.py) against the CPU fp32 oracle (oracle/sd.py) on identical name-keyed weights.

The oracle is pinned on the reference's vendored CompVis latent-diffusion UNet / VAE (tests/golden/sd_ldm_*.npz, tests/test_oracle_golden.py;
diffusers 0.6.0's own code is absent here, see oracle/sd.py), and test_sd_v1_and_vae_vs_reference_ldm_goldens below compares the HIP engines
with those reference outputs directly; the parameter inventory equals the published SD-v1 figures (tests/test_host_logic.py).
Tolerances (max-abs relative to max|output|, rel-L2): f16 <= 6e-3 / 4e-3, bf16 <= 4e-2 / 2.5e-2 (same budget as the ADM UNet's 16-bit modes).
"""
import pytest
import torch

pytestmark = pytest.mark.gpu

TOL = {"f16": (6e-3, 4e-3), "bf16": (4e-2, 2.5e-2)}


def _err(got, want):
    d = (got.double() - want.double())
    return float(d.abs().max() / want.abs().max()), float(d.norm() / want.double().norm())


def _unet_case(ocfg, n, hw, tc, dtype, seed=0):
    from oracle import sd as osd
    from perceptor_amd.engine import sd
    from perceptor_amd.utils.synth import seeded_noise, synth_state_dict
    cfg = sd.SdConfig(**ocfg.__dict__)
    w = synth_state_dict(sd.unet_state_dict_shapes(cfg), seed)
    x = seeded_noise((n, cfg.in_channels, hw, hw), 71)
    ctx = seeded_noise((n, tc, cfg.context_dim), 72)
    t = torch.tensor([981, 20, 500, 250][:n])
    with torch.no_grad():
        want = osd.unet_forward(w, ocfg, x, t, ctx)
    eng = sd.SdUnetEngine(cfg, w, "cuda", dtype)
    got = eng.forward(x.cuda(), t.cuda(), ctx.cuda()).cpu()
    return _err(got, want), eng


@pytest.mark.parametrize("dtype", ["f16", "bf16"])
def test_unet_tiny_and_mid_vs_oracle(dtype):
    from oracle import sd as osd
    for ocfg, n, hw, tc in ((osd.SD_TINY, 2, 16, 7), (osd.SD_MID, 2, 32, 13)):
        (emax, el2), _ = _unet_case(ocfg, n, hw, tc, dtype)
        assert emax < TOL[dtype][0] and el2 < TOL[dtype][1], (ocfg.block_out, emax, el2)


def test_unet_sd_v1_full_config_vs_oracle():
    """The 860 M-parameter SD-v1 UNet (all 686 tensors) at 32x32 latents with a 77-token context, f16 as the reference runs it."""
    from oracle import sd as osd
    (emax, el2), eng = _unet_case(osd.SD_V1, 1, 32, 77, "f16")
    assert emax < TOL["f16"][0] and el2 < TOL["f16"][1], (emax, el2)
    # determinism + the context k|v cache: a second call with the same context object reuses the projections bit-exactly
    from perceptor_amd.utils.synth import seeded_noise
    x, ctx, t = seeded_noise((1, 4, 32, 32), 71).cuda(), seeded_noise((1, 77, 768), 72).cuda(), torch.tensor([981]).cuda()
    a, b = eng.forward(x, t, ctx), eng.forward(x, t, ctx)
    assert torch.equal(a, b)
    with pytest.raises(ValueError):
        eng.forward(x[:, :3], t, ctx)
    with pytest.raises(ValueError):
        eng.forward(x, t, ctx[:, :, :100])


@pytest.mark.parametrize("dtype", ["f16", "bf16"])
def test_vae_decoder_vs_oracle(dtype):
    from oracle import sd as osd
    from perceptor_amd.engine import sd
    from perceptor_amd.utils.synth import seeded_noise, synth_state_dict
    for ocfg, hw in ((osd.VAE_TINY, 16), (osd.VAE_V1, 16)):
        cfg = sd.VaeConfig(**ocfg.__dict__)
        w = synth_state_dict(sd.vae_decoder_state_dict_shapes(cfg), 0)
        z = seeded_noise((1, 4, hw, hw), 73)
        with torch.no_grad():
            want = (osd.vae_decode(w, ocfg, z / 0.18215) + 1) / 2
        got = sd.VaeDecoderEngine(cfg, w, "cuda", dtype).forward(z.cuda()).cpu()
        d = (got - want)
        emax, el2 = float(d.abs().max() / (want - 0.5).abs().max()), float(d.norm() / (want - 0.5).norm())
        assert emax < TOL[dtype][0] and el2 < TOL[dtype][1], (ocfg.block_out, emax, el2)


TINY_TEXT = (16, 520, 32, 2, 1, 32)        # context, vocab, width (= the tiny UNet's context_dim), layers, heads, out_dim


def _tiny_model(fp16=True):
    from perceptor_amd import models
    from perceptor_amd.engine import sd
    cfg = sd.SdConfig(block_out=(32, 64, 64), cross_attn=(True, True, False), heads=2, context_dim=32)
    vae = sd.VaeConfig(block_out=(32, 64), layers_per_block=1)
    return models.StableDiffusion(fp16=fp16, config=cfg, vae_config=vae, text_config=TINY_TEXT).to("cuda")


def test_class_surface_step_cfg_decode_encode_vs_oracle():
    from oracle import clip_text, sd as osd
    from perceptor_amd.utils.synth import seeded_noise, synth_state_dict
    m = _tiny_model()
    sdk = m.state_dict()
    assert sum(k.startswith("unet.") for k in sdk) == 494 and "schedule_alphas" in sdk and any(k.startswith("vae.encoder.") for k in sdk)
    a_ref, s_ref = osd.schedule()
    assert torch.equal(m.schedule_alphas.cpu(), a_ref) and torch.equal(m.schedule_sigmas.cpu(), s_ref)
    ids = torch.tensor([[518, 5, 9, 300, 519] + [519] * 11, [518, 519] + [519] * 14])
    pos, neu = m.conditioning(token_ids=ids[:1]), m.conditioning(token_ids=ids[1:])
    tsd = synth_state_dict({k: v for k, v in clip_text.text_state_dict_shapes(TINY_TEXT).items()}, 0)
    want_h, _ = clip_text.text_forward(tsd, TINY_TEXT, ids, True)
    assert float((pos.encodings.cpu() - want_h[:1]).abs().max()) < 5e-3 * float(want_h.abs().max())
    x = seeded_noise((2, 4, 16, 16), 81).cuda()
    un, po = m.predictions_pair(x, 600, neu, pos)
    usd = synth_state_dict(osd.unet_state_dict_shapes(osd.SD_TINY), 0)
    t = torch.tensor([600, 600])
    with torch.no_grad():
        e_un = osd.unet_forward(usd, osd.SD_TINY, x.cpu(), t, want_h[1:].expand(2, -1, -1))
        e_po = osd.unet_forward(usd, osd.SD_TINY, x.cpu(), t, want_h[:1].expand(2, -1, -1))
    for got, want in ((un.predicted_noise, e_un), (po.predicted_noise, e_po)):
        emax, el2 = _err(got.cpu(), want)
        assert emax < TOL["f16"][0] and el2 < TOL["f16"][1], (emax, el2)
    sep = m.predictions(x, 600, pos)                     # the batched pair equals a separate call (rounding-level: tile choices depend on M)
    assert _err(sep.predicted_noise.cpu(), po.predicted_noise.cpu())[0] < 2e-3
    # algebra on the SAME predicted noise: cfg, step, denoised, forced, resample bounds (stable_diffusion/predictions.py)
    a, s, a2, s2 = a_ref[600], s_ref[600], a_ref[560], s_ref[560]
    strong = un.classifier_free_guidance(po, guidance_scale=7.0)
    e_cfg = osd.classifier_free_guidance(un.predicted_noise.cpu(), po.predicted_noise.cpu(), 7.0)
    assert float((strong.predicted_noise.cpu() - e_cfg).abs().max()) < 1e-5 * (1 + float(e_cfg.abs().max()))
    nxt = strong.step(560)
    want = osd.ddim_step(x.cpu(), e_cfg, a, s, a2, s2)
    assert float((nxt.cpu() - want).abs().max()) < 2e-5 * (1 + float(want.abs().max()))
    den = osd.denoised_latents(x.cpu(), e_cfg, a, s)
    assert float((strong.denoised_latents.cpu() - den).abs().max()) < 2e-5 * (1 + float(den.abs().max()))
    forced = strong.forced_denoised_latents(strong.denoised_latents)
    assert float((forced.predicted_noise - strong.predicted_noise).abs().max()) < 1e-4 * (1 + float(e_cfg.abs().max()))
    assert strong.reverse_step(700).shape == x.shape
    with pytest.raises(ValueError):
        strong.reverse_step(100)
    with pytest.raises(ValueError):
        strong.resample_noise(900)
    thr = strong.latent_dynamic_threshold(0.5)
    assert float(thr.predicted_noise.abs().max()) <= max(2.5, float(strong.predicted_noise.abs().flatten(1).quantile(0.5, dim=1).max())) + 1e-5
    # VAE: decode / encode against the oracle on the name-keyed weights
    vsd = synth_state_dict({**osd.vae_encoder_state_dict_shapes(osd.VAE_TINY), **osd.vae_decoder_state_dict_shapes(osd.VAE_TINY)}, 0)
    z = seeded_noise((1, 4, 16, 16), 82)
    with torch.no_grad():
        img = (osd.vae_decode(vsd, osd.VAE_TINY, z / 0.18215) + 1) / 2
    got = m.decode(z.cuda()).cpu()
    assert float((got - img).abs().max()) < 4e-2 * float((img - 0.5).abs().max())
    pic = (seeded_noise((1, 3, 32, 32), 83) * 0.25 + 0.5)
    with torch.no_grad():
        mean, _ = osd.vae_encode_moments(vsd, osd.VAE_TINY, pic * 2 - 1)
    lat = m.latents(pic.cuda()).cpu()
    assert float((lat - 0.18215 * mean).abs().max()) < 4e-2 * float((0.18215 * mean).abs().max())
    assert m.encode(pic.cuda(), method="sample").shape == lat.shape
    with pytest.raises(Exception):
        m.encode(torch.zeros(1, 3, 40, 32).cuda())
    with pytest.raises(ValueError):
        m.encode(pic.cuda(), method="nope")
    assert strong.dynamic_threshold(0.95).predicted_noise.shape == x.shape        # decode -> clamp -> encode round trip runs


def test_predictions_pair_context_cache_follows_the_prompt():
    """the (prompt pair -> context) cache of predictions_pair must not survive a new prompt whose encodings land on a recycled
    address, nor an in-place edit of the encodings; it must survive unchanged conditionings (one k|v projection per chain)."""
    import gc
    m = _tiny_model()
    x = torch.randn(2, 4, 16, 16, generator=torch.Generator().manual_seed(5)).cuda()
    ids_a = torch.tensor([[518, 5, 9, 300, 519] + [519] * 11])
    ids_b = torch.tensor([[518, 77, 41, 8, 12, 519] + [519] * 10])
    ids_n = torch.tensor([[518, 519] + [519] * 14])
    neu, pos = m.conditioning(token_ids=ids_n), m.conditioning(token_ids=ids_a)
    _, po_a = m.predictions_pair(x, 600, neu, pos)
    ctx_a = m._pair_ctx
    m.predictions_pair(x, 560, neu, pos)
    assert m._pair_ctx is ctx_a                                   # same conditionings: the context (and the engine's k|v cache) is reused
    want_a = po_a.predicted_noise.clone()
    del neu, pos, po_a
    gc.collect()
    torch.cuda.empty_cache()
    neu, pos = m.conditioning(token_ids=ids_n), m.conditioning(token_ids=ids_b)      # may reuse the freed addresses
    _, po_b = m.predictions_pair(x, 600, neu, pos)
    sep_b = m.predictions(x, 600, pos).predicted_noise
    assert _err(po_b.predicted_noise.cpu(), sep_b.cpu())[0] < 2e-3
    assert _err(po_b.predicted_noise.cpu(), want_a.cpu())[0] > 1e-2          # and it is not prompt A's answer
    pos.encodings.mul_(0)                                          # in-place edit: a new context
    _, po_0 = m.predictions_pair(x, 600, neu, pos)
    sep_0 = m.predictions(x, 600, pos).predicted_noise
    assert _err(po_0.predicted_noise.cpu(), sep_0.cpu())[0] < 2e-3
    assert _err(po_0.predicted_noise.cpu(), po_b.predicted_noise.cpu())[0] > 1e-3


def test_sample_loop_schedule_and_errors():
    from perceptor_amd import models
    from perceptor_amd.utils.tokenizer import ClipTokenizer
    m = _tiny_model()
    idx = m.schedule_indices(n_steps=50)
    assert idx.shape[1] == 2 and int(idx[0, 0]) == 999 and bool((idx[:, 0] > idx[:, 1]).all()) and bool((idx[1:, 0] == idx[:-1, 1]).all())
    with pytest.raises(ValueError):
        m.schedule_indices(from_index=10, to_index=20)
    with pytest.raises(ValueError):
        m.schedule_indices(n_steps=1000)                    # collapses below 0.9 * n_steps: ValueError here (AssertionError in GuidedDiffusion)
    with pytest.raises(FileNotFoundError):
        m.conditioning(["a cab"])                           # the merge list is data the caller provides
    m._tokenizer = ClipTokenizer(merges=[("a", "b"), ("ab", "c</w>"), ("c", "a")])
    assert m.tokenize(["abc"]).tolist()[0][:4] == [m._tokenizer.sot, 513, m._tokenizer.eot, m._tokenizer.eot]
    outs = list(m.sample("abc cab", n_steps=4, to_index=600, guidance_scale=3.0))
    assert len(outs) == len(m.schedule_indices(n_steps=4, to_index=600)) + 1
    assert outs[-1].predicted_noise.shape == (1, 4, 64, 64) and bool(torch.isfinite(outs[-1].predicted_noise).all())
    img = outs[-1].denoised_images
    assert img.shape == (1, 3, 128, 128) and bool(torch.isfinite(img).all())     # the tiny VAE up-samples x2 (two levels)
    with pytest.raises(ValueError):
        m.random_diffused_latents((1, 3, 500, 512))
    with pytest.raises(ValueError):
        next(m.sample("abc", from_index=500))
    with pytest.raises(RuntimeError):
        models.StableDiffusion(weights="pretrained")


@pytest.mark.parametrize("dtype", ["f16", "bf16"])
def test_sd_v1_and_vae_vs_reference_ldm_goldens(dtype):
    """HIP engines against outputs of the REFERENCE's vendored CompVis UNetModel / Encoder / Decoder (oracle/gen_golden.py: gen_sd_ldm):
    tiny UNet, the full 860 M SD-v1 UNet at 16x16 latents with a 77-token context, tiny and SD-v1 VAE."""
    import numpy as np
    from conftest import golden
    from oracle import sd as osd
    from perceptor_amd.engine import sd
    from perceptor_amd.utils.synth import seeded_noise, synth_state_dict
    for tag, ocfg in (("tiny", osd.SD_TINY), ("v1", osd.SD_V1)):
        g = golden(f"sd_ldm_unet_{tag}")
        cfg = sd.SdConfig(**ocfg.__dict__)
        n, hw, tc = g["eps"].shape[0], int(g["hw"]), int(g["tc"])
        eng = sd.SdUnetEngine(cfg, synth_state_dict(sd.unet_state_dict_shapes(cfg), 0), "cuda", dtype)
        x, ctx = seeded_noise((n, cfg.in_channels, hw, hw), 71), seeded_noise((n, tc, cfg.context_dim), 72)
        emax, el2 = _err(eng.forward(x.cuda(), g["t"].cuda(), ctx.cuda()).cpu(), g["eps"])
        assert emax < TOL[dtype][0] and el2 < TOL[dtype][1], (tag, emax, el2)
        del eng
    for tag, ocfg in (("tiny", osd.VAE_TINY), ("v1", osd.VAE_V1)):
        g = golden(f"sd_ldm_vae_{tag}")
        cfg = sd.VaeConfig(**ocfg.__dict__)
        w = synth_state_dict({**sd.vae_encoder_state_dict_shapes(cfg), **sd.vae_decoder_state_dict_shapes(cfg)}, 0)
        hw, ihw = int(g["hw"]), int(g["img_hw"])
        dec = sd.VaeDecoderEngine(cfg, w, "cuda", dtype).forward(seeded_noise((1, cfg.latent_channels, hw, hw), 73).cuda(), scale=1.0, to_images=False).cpu()
        emax, el2 = _err(dec, g["dec"])
        assert emax < TOL[dtype][0] and el2 < TOL[dtype][1], (tag, "dec", emax, el2)
        img = seeded_noise((1, 3, ihw, ihw), 74) * 0.5                       # the decoder-space x in [-1, 1]; the engine takes images in [0, 1]
        mean, logvar = sd.VaeEncoderEngine(cfg, w, "cuda", dtype).forward(((img + 1) / 2).cuda())
        emax, el2 = _err(mean.cpu(), g["mean"])
        assert emax < TOL[dtype][0] and el2 < TOL[dtype][1], (tag, "mean", emax, el2)
        assert _err(logvar.cpu(), g["logvar"])[1] < TOL[dtype][1]


def test_sd_predictions_class_vs_reference_class_golden():
    """models.stable_diffusion.Predictions against the REFERENCE class (models/stable_diffusion/predictions.py) on the same inputs
    (oracle/gen_golden.py: gen_sd_predictions; noise injected where the reference draws it)."""
    from conftest import golden
    from perceptor_amd.engine import sampler
    from perceptor_amd.models.stable_diffusion import Predictions
    from oracle import sd as osd
    g = golden("sd_predictions")
    a, s = (t.cuda() for t in osd.schedule())
    ident = lambda t: t
    mk = lambda e, sl=slice(None): Predictions(from_diffused_latents=g["x"][sl].cuda(), from_indices=g["fi"][sl].cuda(), predicted_noise=e[sl].cuda(),
                                                schedule_alphas=a, schedule_sigmas=s, encode=ident, decode=ident)
    p, p2 = mk(g["eps"]), mk(g["eps2"])
    close = lambda got, key, tol=3e-6: float((got.cpu() - g[key]).abs().max()) <= tol * (1 + float(g[key].abs().max()))
    real = sampler.randn_like
    sampler.randn_like = lambda t: g["noise"].to(t)
    try:
        assert close(p.denoised_latents, "denoised") and close(p.step(g["ti"]), "step") and close(p.step(g["ti"], eta=0.7), "step_eta")
        assert close(p.reverse_step(g["hi"]), "reverse") and close(p.resample_noise(g["ti"]), "resample_noise") and close(p.resample(g["ti"]), "resample")
        assert close(p.noisy_reverse_step(g["hi"]), "noisy_reverse")
    finally:
        sampler.randn_like = real
    assert close(p.guided(g["guide"].cuda(), guidance_scale=0.5, clamp_value=1e-6).predicted_noise, "guided")
    assert close(p.classifier_free_guidance(p2, guidance_scale=7.0).predicted_noise, "cfg", 1e-5)
    assert close(p.forced_denoised_latents(g["x"].cuda() * 0.5).predicted_noise, "forced", 1e-5)
    assert close(mk(g["eps"] * 3, slice(0, 1)).latent_dynamic_threshold(0.95).predicted_noise, "latent_thr")
    w = torch.stack([p.wasserstein_distance(), p.wasserstein_square_distance()])
    assert float((w.cpu() - g["wasserstein"]).abs().max()) <= 2e-6


def test_inpainting_checkpoint_surface_vs_oracle():
    """runwayml/stable-diffusion-inpainting: 9-channel UNet input = latents | binarised latent mask | latents of the masked image
    (conditioning.py:31-40), latent masks (blur + bilinear down-sampling), sample() with replace_diffused.  UNet values against the oracle
    on the concatenated input; the blur restates kornia's gaussian_blur2d (absent: that helper is parity-unpinned)."""
    from oracle import sd as osd
    from perceptor_amd import models
    from perceptor_amd.engine import sd
    from perceptor_amd.utils.synth import seeded_noise, synth_state_dict
    from perceptor_amd.utils.tokenizer import ClipTokenizer
    cfg = sd.SdConfig(in_channels=9, block_out=(32, 64, 64), cross_attn=(True, True, False), heads=2, context_dim=32)
    vae = sd.VaeConfig(block_out=(32, 64, 64, 64), layers_per_block=1)            # three down-samplings: latents at 1/8 resolution, as SD
    m = models.StableDiffusion("runwayml/stable-diffusion-inpainting", config=cfg, vae_config=vae, text_config=TINY_TEXT).to("cuda")
    img = (seeded_noise((1, 3, 128, 128), 91) * 0.2 + 0.5).clamp(0, 1).cuda()
    mask = torch.zeros(1, 1, 128, 128).cuda()
    mask[:, :, :, 48:] = 1.0
    lm = m.latent_masks(mask, 4.0)
    assert lm.shape == (1, 1, 16, 16) and float(lm[0, 0, 8, 0]) < 1e-3 and float(lm[0, 0, 8, 15]) > 0.999
    assert 0.0 < float(lm[0, 0, 8, 5]) < 0.5 < float(lm[0, 0, 8, 6]) < 1.0 and bool((lm[0, 0, 8, 1:] >= lm[0, 0, 8, :-1]).all())     # the edge at pixel 48 = latent column 6, blurred
    assert torch.equal(m.latent_masks(mask, None), torch.nn.functional.interpolate(mask, size=(16, 16), mode="bilinear"))
    with pytest.raises(ValueError):
        m.latent_masks(mask * 2, 4.0)
    with pytest.raises(ValueError):
        m.latent_masks(mask.expand(1, 3, 128, 128), 4.0)
    ids = torch.tensor([[518, 5, 9, 519] + [519] * 12])
    cond = m.conditioning(token_ids=ids, inpainting_masks=mask, inpainting_images=img)
    assert cond.inpainting_latents.shape == (1, 4, 16, 16) and cond.inpainting_latent_masks.shape == (1, 1, 16, 16)
    x = seeded_noise((1, 4, 16, 16), 92).cuda()
    got = m.predictions(x, 500, cond).predicted_noise.cpu()
    ocfg = osd.SdConfig(in_channels=9, block_out=(32, 64, 64), cross_attn=(True, True, False), heads=2, context_dim=32)
    w = synth_state_dict(osd.unet_state_dict_shapes(ocfg), 0)
    with torch.no_grad():
        want = osd.unet_forward(w, ocfg, cond.input(x).cpu(), torch.tensor([500]), cond.encodings.float().cpu())
    emax, el2 = _err(got, want)
    assert got.shape == (1, 4, 16, 16) and emax < TOL["f16"][0] and el2 < TOL["f16"][1], (emax, el2)
    m._tokenizer = ClipTokenizer(merges=[("a", "b"), ("ab", "c</w>"), ("c", "a")])
    outs = list(m.sample("abc", from_index=600, to_index=300, n_steps=4, init_image=img, inpainting_mask=mask, n_resample=1))
    last = outs[-1]
    assert last.predicted_noise.shape == (1, 4, 16, 16) and bool(torch.isfinite(last.predicted_noise).all())
    # replace_diffused: outside the mask the chain follows the (re-noised) original image's latents
    keep = cond.inpainting_latent_masks < 1e-3
    init = m.latents(img)
    a600 = float(m.schedule_alphas[int(last.from_indices[0])])
    assert bool(keep.any()) and float(((last.from_diffused_latents - init * a600) * keep).abs().max()) < 6.0     # noise-level distance, finite


def test_unet_and_vae_non_square_latents_vs_oracle():
    """Latents whose width is not a multiple of 32 (24 x 40): the convolutions leave the 32-pixel-wide tile kernels for the generic one."""
    from oracle import sd as osd
    from perceptor_amd.engine import sd
    from perceptor_amd.utils.synth import seeded_noise, synth_state_dict
    cfg = sd.SdConfig(**osd.SD_MID.__dict__)
    w = synth_state_dict(sd.unet_state_dict_shapes(cfg), 0)
    x, ctx, t = seeded_noise((2, 4, 24, 40), 75), seeded_noise((2, 9, cfg.context_dim), 76), torch.tensor([700, 3])
    with torch.no_grad():
        want = osd.unet_forward(w, osd.SD_MID, x, t, ctx)
    got = sd.SdUnetEngine(cfg, w, "cuda", "f16").forward(x.cuda(), t.cuda(), ctx.cuda()).cpu()
    emax, el2 = _err(got, want)
    assert emax < TOL["f16"][0] and el2 < TOL["f16"][1], (emax, el2)
    vcfg = sd.VaeConfig(**osd.VAE_TINY.__dict__)
    vw = synth_state_dict({**sd.vae_encoder_state_dict_shapes(vcfg), **sd.vae_decoder_state_dict_shapes(vcfg)}, 0)
    z = seeded_noise((1, 4, 12, 20), 77)
    with torch.no_grad():
        img = (osd.vae_decode(vw, osd.VAE_TINY, z / 0.18215) + 1) / 2
    out = sd.VaeDecoderEngine(vcfg, vw, "cuda", "bf16").forward(z.cuda()).cpu()
    assert out.shape == (1, 3, 24, 40) and float((out - img).abs().max()) < 4e-2 * float((img - 0.5).abs().max())
