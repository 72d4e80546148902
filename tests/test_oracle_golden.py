"""CPU: pin the oracle (oracle/*.py) against golden vectors produced by the reference code.

Fixtures come from oracle/gen_golden.py (reference modules imported in the build
container).  Tolerances: same fp32 ops in a different association order -> 1e-5
relative to the output scale unless stated.
"""
import pytest
import torch

from conftest import golden
from oracle import adm_unet, clip_vit, sampling, vdiff
from perceptor_amd.utils.synth import synth_state_dict


def _close(a, b, tol):
    scale = max(1.0, float(b.abs().max()))
    err = float((a.double() - b.double()).abs().max())
    assert err <= tol * scale, f"max abs err {err:.3e} > {tol * scale:.3e}"


def test_schedules_and_tables():
    g = golden("sampling")
    a, s = sampling.gd_tables()
    assert torch.equal(a, g["alphas"]) and torch.equal(s, g["sigmas"])
    assert torch.equal(sampling.schedule_indices(a, s, 50, rho=3.0), g["idx_50_r3"])
    assert torch.equal(sampling.schedule_indices(a, s, 50, rho=7.0), g["idx_50_r7"])
    assert torch.equal(sampling.schedule_indices(a, s, 250, rho=7.0), g["idx_250_r7"])
    assert torch.equal(sampling.schedule_indices(a, s, 20, from_index=400), g["idx_20_400"])
    assert g["idx_50_r3"][0].tolist() == [994, 988] and g["idx_50_r3"][-1].tolist() == [6, 0]
    _close(sampling.schedule_ts(50), g["ts_50"], 1e-6)
    _close(sampling.schedule_ts(500), g["ts_500"], 1e-6)
    with pytest.raises(ValueError):
        sampling.schedule_indices(a, s, 10, from_index=0, to_index=5)


def test_predictions_algebra():
    g = golden("sampling")
    a, s = g["alphas"], g["sigmas"]
    af, sf, at, st = a[g["fi"]], s[g["fi"]], a[g["ti"]], s[g["ti"]]
    img, eps, grad = g["img"], g["eps"], g["grad"]
    _close((sampling.eps_denoised_xs(img, eps, af, sf) + 1) / 2, g["eps_denoised"], 1e-6)
    _close(sampling.eps_step(img, eps, af, sf, at, st), g["eps_step"], 1e-6)
    eg = sampling.guided(eps, grad, sf)
    _close(eg, g["eps_guided"], 1e-6)
    _close(sampling.eps_step(img, eg, af, sf, at, st), g["eps_guided_step"], 1e-6)
    den = ((sampling.eps_denoised_xs(img, eps, af, sf) + 1) / 2).clamp(0, 1)
    _close(sampling.eps_forced_denoised_images(img, den, af, sf), g["eps_forced"], 1e-6)
    ft, tt = g["ft"], g["tt"]
    _close((sampling.v_denoised_xs(img, eps, ft) + 1) / 2, g["v_denoised"], 1e-6)
    _close(sampling.v_predicted_noise(img, eps, ft), g["v_eps"], 1e-6)
    _close(sampling.v_step(img, eps, ft, tt), g["v_step"], 1e-6)
    vg = sampling.guided(eps, grad, sampling.t_to_alpha_sigma(ft)[1])
    _close(vg, g["v_guided"], 1e-6)
    _close(sampling.v_step(img, vg, ft, tt), g["v_guided_step"], 1e-6)


def test_predictions_variants_f3():
    """Oracle restatement of the stochastic / sort / quantile variants and clamp_with_grad vs the reference classes (noise injected)."""
    g, g2 = golden("sampling"), golden("sampling2")
    a, s = g["alphas"], g["sigmas"]
    fi, ti, hi = g["fi"], g["ti"], g2["hi"]
    img, eps, noise = g["img"], g["eps"], g2["noise"]
    x0 = sampling.eps_denoised_xs(img, eps, a[fi], s[fi])
    _close(sampling.step_eta(x0, eps, a[fi], s[fi], a[ti], s[ti], 0.7, noise), g2["eps_step_eta"], 1e-6)
    _close(sampling.resample_noise(eps, s[fi], s[ti], noise), g2["eps_resample_noise"], 1e-6)
    _close(sampling.resample(x0, eps, a[fi], s[fi], s[ti], noise), g2["eps_resample"], 1e-6)
    _close(sampling.noisy_reverse_step(x0, eps, s[fi], a[hi], s[hi], noise), g2["eps_noisy_reverse"], 1e-6)
    _close(torch.stack([sampling.wasserstein(eps, 1), sampling.wasserstein(eps, 2)]), g2["eps_wasserstein"], 1e-6)
    ft, tt, ht = g["ft"], g["tt"], g2["ht"]
    (af, sf), (at, st), (ah, sh) = (sampling.t_to_alpha_sigma(t) for t in (ft, tt, ht))
    vx0, veps = sampling.v_denoised_xs(img, eps, ft), sampling.v_predicted_noise(img, eps, ft)
    _close(sampling.step_eta(vx0, veps, af, sf, at, st, 0.7, noise), g2["v_step_eta"], 1e-6)
    _close(sampling.resample_noise(veps, sf, st, noise), g2["v_resample_noise"], 1e-6)
    _close(sampling.resample(vx0, veps, af, sf, st, noise), g2["v_resample"], 1e-6)
    _close(sampling.noisy_reverse_step(vx0, veps, sf, ah, sh, noise), g2["v_noisy_reverse"], 1e-6)
    _close(torch.stack([sampling.wasserstein(veps, 1), sampling.wasserstein(veps, 2)]), g2["v_wasserstein"], 1e-6)
    big = g2["big"]
    _close(torch.stack([sampling.wasserstein(big, 1), sampling.wasserstein(big, 2)]), g2["big_wasserstein"], 1e-6)
    _close(torch.stack([sampling.quantile_abs(big, q) for q in (0.0, 0.5, 0.95, 0.999, 1.0)]), g2["big_quantiles"], 1e-6)
    assert torch.equal(sampling.clamp_with_grad_backward(g2["cwg_x"], g2["cwg_g"], 0.0, 1.0), g2["cwg_dx"])
    assert torch.equal(g2["cwg_x"].clamp(0.0, 1.0), g2["cwg_y"])


def test_philox_known_answers_and_randn_contract():
    """The oracle's Philox4x32-10 against Random123's known-answer vectors; the normal draw is a function of the global element index."""
    import numpy as np
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0), (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, want in kat:
        got = sampling.philox4x32_10(np.array([ctr], dtype=np.uint32), key)[0]
        assert tuple(int(v) for v in got) == want
    full = sampling.device_randn((4, 3, 8, 8), seed=99, stream=5)
    part = sampling.device_randn((2, 3, 8, 8), seed=99, stream=5, first_element=2 * 3 * 8 * 8 - 0)
    assert torch.equal(full[2:], part)
    odd = sampling.device_randn((7,), seed=99, stream=5, first_element=3)
    assert torch.equal(full.flatten()[3:10], odd)
    z = sampling.device_randn((1 << 16,), seed=1, stream=0)
    assert abs(float(z.mean())) < 0.02 and abs(float(z.std()) - 1) < 0.02
    assert not torch.equal(z, sampling.device_randn((1 << 16,), seed=1, stream=1))


def test_adm_tiny_fp16_checkpoint_weights():
    """fp32 weights with the torso convolutions cast to fp16 by the reference's own convert_to_fp16() (not bf16-representable)."""
    from conftest import fp16_torso_state_dict
    cfg = adm_unet.AdmConfig(64, 32, 1, (1, 2, 2), (2, 4), num_head_channels=16, use_scale_shift_norm=True, resblock_updown=True)
    g = golden("adm_tiny_a_fp16w")
    sd = fp16_torso_state_dict(adm_unet.state_dict_shapes(cfg), g)
    _close(adm_unet.adm_unet_forward(sd, cfg, g["x"], g["t"]), g["y"], 1e-5)


@pytest.mark.parametrize("tag,cfg", [
    ("a", adm_unet.AdmConfig(64, 32, 1, (1, 2, 2), (2, 4), num_head_channels=16, use_scale_shift_norm=True, resblock_updown=True)),
    ("b", adm_unet.AdmConfig(64, 32, 2, (1, 2), (2,), num_heads=2, use_new_attention_order=True)),
])
def test_adm_tiny(tag, cfg):
    g = golden(f"adm_tiny_{tag}")
    sd = synth_state_dict(adm_unet.state_dict_shapes(cfg), 0)
    _close(adm_unet.adm_unet_forward(sd, cfg, g["x"], g["t"]), g["y"], 1e-5)


def test_adm_pixelart_full_64():
    g = golden("adm_pixelart_64")
    cfg = adm_unet.pixelart_config()
    sd = synth_state_dict(adm_unet.state_dict_shapes(cfg), 0)
    y = adm_unet.adm_unet_forward(sd, cfg, g["x"], g["t"])
    _close(y[:, :, ::4, ::4], g["y_sub"], 1e-5)


def test_adm_standard_full_128():
    g = golden("adm_standard_128")
    cfg = adm_unet.openimages_config()
    shapes = adm_unet.state_dict_shapes(cfg)
    assert len(shapes) == 644  # GD-standard state dict has 644 tensors
    sd = synth_state_dict(shapes, 0)
    y = adm_unet.adm_unet_forward(sd, cfg, g["x"], g["t"])
    _close(y[:, :, ::4, ::4], g["y_sub"], 1e-5)
    f = y.flatten(1).double()
    _close(torch.stack([f.mean(1), f.std(1), f.norm(dim=1)], 1).float(), g["y_mom"], 1e-5)


def test_resize_matches_reference():
    g = golden("clip_resize")
    from perceptor_amd.utils.synth import seeded_noise
    for tag, shape, target in (("512_224", (1, 3, 512, 512), (224, 224)), ("256_224", (1, 3, 256, 256), (224, 224)),
                               ("128_224", (1, 3, 128, 128), (224, 224)), ("96x160_64", (1, 3, 96, 160), (64, 64))):
        img = seeded_noise(shape, 51) * 0.25 + 0.5
        _close(clip_vit.resize(img, target)[:, :, ::3, ::3], g["rz_" + tag], 2e-5)  # dense-matrix vs 14-tap gather: fp32 summation order


@pytest.mark.parametrize("tag", ["tiny", "tiny-odd", "ViT-B-32"])
def test_vit_embedding_and_image_gradient(tag):
    g = golden(f"clip_vit_{tag}")
    cfg = clip_vit.VIT_CONFIGS[tag]
    sd = synth_state_dict(clip_vit.vit_state_dict_shapes(cfg), 0)
    img = g["img"].clone().requires_grad_(True)
    e = clip_vit.encode_images(sd, cfg, img, quick_gelu=True, normalize=False)
    _close(e.detach(), g["emb"], 2e-5)
    en = torch.nn.functional.normalize(e)
    _close(en.detach(), g["emb_n"], 2e-5)
    (gr,) = torch.autograd.grad((en * g["probe"]).sum(), img)
    scale = float(g["grad_sub"].abs().max())
    assert float((gr[:, :, ::4, ::4] - g["grad_sub"]).abs().max()) <= 2e-4 * scale


@pytest.mark.parametrize("tag,quick", [("tiny-odd", False), ("ViT-L-14", False)])
def test_vit_vs_transformers_tower(tag, quick):
    """The oracle's exact-GELU path and the benchmarked ViT-L/14 against an independent implementation: transformers'
    CLIPVisionModelWithProjection with the same name-keyed weights (oracle/gen_golden.py: gen_clip_hf)."""
    from perceptor_amd.utils.synth import seeded_noise
    g = golden(f"clip_hf_{tag}_{'quickgelu' if quick else 'gelu'}")
    cfg = clip_vit.VIT_CONFIGS[tag]
    sd = synth_state_dict(clip_vit.vit_state_dict_shapes(cfg), 0)
    img = (seeded_noise(tuple(int(v) for v in g["img_shape"]), 52) * 0.25 + 0.5).requires_grad_(True)
    e = clip_vit.encode_images(sd, cfg, img, quick_gelu=quick, normalize=False)
    _close(e.detach(), g["emb"], 5e-5)
    en = torch.nn.functional.normalize(e)
    (gr,) = torch.autograd.grad((en * g["probe"]).sum(), img)
    scale = float(g["grad_sub"].abs().max())
    assert float((gr[:, :, ::4, ::4] - g["grad_sub"]).abs().max()) <= 5e-4 * scale


def test_spherical_loss_known_values():
    # identical unit vectors -> 0; orthogonal -> 2*asin(sqrt(2)/2)^2 = pi^2/8; antipodal -> pi^2/2
    e = torch.eye(4)[:2]
    t = torch.stack([e[0], e[1], -e[0]])
    d = (e[:, None] - t[None]).norm(dim=2).div(2).arcsin().square().mul(2)
    assert torch.allclose(d[0], torch.tensor([0.0, torch.pi**2 / 8, torch.pi**2 / 2]), atol=1e-5)
    assert torch.allclose(clip_vit.spherical_loss(e, t, torch.ones(3)), d.mean(), atol=1e-6)


def test_vdiff_yfcc2_full_128():
    g = golden("vdiff_yfcc_2_128")
    spec = vdiff.yfcc2_spec()
    sd = synth_state_dict(vdiff.state_dict_shapes(spec), 0)
    y = vdiff.vdiff_forward(sd, spec, g["x"], g["t"])
    _close(y[:, :, ::4, ::4], g["y_sub"], 1e-5)


def test_vdiff_cc12m1_full_64():
    g = golden("vdiff_cc12m_1_64")
    spec = vdiff.cc12m1_spec()
    sd = synth_state_dict(vdiff.state_dict_shapes(spec), 0)
    y = vdiff.vdiff_forward(sd, spec, g["x"], g["t"], g["clip_embed"])
    _close(y[:, :, ::2, ::2], g["y_sub"], 1e-5)


@pytest.mark.parametrize("name,res,spec_fn,gain", [("yfcc_1", 128, vdiff.yfcc1_spec, 1.0), ("wikiart", 64, vdiff.wikiart_spec, 0.6)])
def test_vdiff_yfcc1_wikiart(name, res, spec_fn, gain):
    g = golden(f"vdiff_{name}_{res}")
    spec = spec_fn()
    sd = synth_state_dict(vdiff.state_dict_shapes(spec), 0, gain=gain)
    y = vdiff.vdiff_forward(sd, spec, g["x"], g["t"])
    _close(y[:, :, ::2, ::2], g["y_sub"], 1e-5)


@pytest.mark.parametrize("fixture, tag, quick", [("clip_text_ruclip_tiny", "tiny", True), ("clip_text_hf_tiny-wide_gelu", "tiny-wide", False),
                                                   ("clip_text_hf_ViT-L-14_quickgelu", "ViT-L-14", True)])
def test_text_tower_vs_reference_and_transformers(fixture, tag, quick):
    """oracle/clip_text.py against the reference's in-tree ruclip CLIP.encode_text and against transformers'
    CLIPTextModelWithProjection (an independent implementation) on the same name-keyed weights (oracle/gen_golden.py: gen_clip_text)."""
    from oracle import clip_text
    g = golden(fixture)
    cfg = clip_text.TEXT_CONFIGS[tag]
    sd = synth_state_dict(clip_text.text_state_dict_shapes(cfg), 0)
    with torch.no_grad():
        hidden, pooled = clip_text.text_forward(sd, cfg, g["ids"], quick)
    assert float((pooled - g["pooled"]).abs().max()) < 2e-5 * (1 + float(g["pooled"].abs().max()))
    if "hidden" in g:
        assert float((hidden - g["hidden"]).abs().max()) < 2e-5 * (1 + float(g["hidden"].abs().max()))


def test_transformers_key_mapping_against_clipmodel_fixture():
    """engine.vit.from_hf_vision_state_dict / engine.text.from_hf_text_state_dict (host logic: transformers -> OpenAI-CLIP key names) +
    the oracle towers reproduce what transformers' CLIPModel computed from the SAME transformers-named weights (gen_clip_hf_model)."""
    from oracle import clip_text
    from perceptor_amd.engine.text import from_hf_text_state_dict, hf_text_state_dict_shapes
    from perceptor_amd.engine.vit import from_hf_vision_state_dict, hf_vision_state_dict_shapes
    g = golden("clip_hf_model_tiny")
    vcfg, tcfg = (32, 8, 64, 2, 1, 32), (16, 96, 64, 2, 1, 32)
    sd = synth_state_dict({**hf_vision_state_dict_shapes(vcfg), **hf_text_state_dict_shapes(tcfg)}, 0)
    with torch.no_grad():
        ie = clip_vit.encode_images(from_hf_vision_state_dict(sd), vcfg, g["img"], True, normalize=False)
        hidden, te = clip_text.text_forward(from_hf_text_state_dict(sd), tcfg, g["ids"], True)
    assert float((ie - g["image_embeds"]).abs().max()) < 2e-5 * (1 + float(g["image_embeds"].abs().max()))
    assert float((te - g["text_embeds"]).abs().max()) < 2e-5 * (1 + float(g["text_embeds"].abs().max()))
    assert float((hidden - g["text_hidden"]).abs().max()) < 2e-5 * (1 + float(g["text_hidden"].abs().max()))


@pytest.mark.parametrize("tag", ["tiny", "v1"])
def test_sd_unet_vs_reference_ldm(tag):
    """oracle/sd.py's UNet restatement against the REFERENCE's vendored CompVis UNetModel (the original of diffusers' UNet2DConditionModel,
    perceptor/models/latent_diffusion/ldm/modules/diffusionmodules/openaimodel.py) on the same name-keyed weights under the published key
    correspondence (oracle/gen_golden.py: gen_sd_ldm): tiny config in full, the 860 M-parameter SD-v1 configuration at 16x16 latents."""
    from oracle import sd as osd
    from perceptor_amd.utils.synth import seeded_noise
    g = golden(f"sd_ldm_unet_{tag}")
    cfg = osd.SD_TINY if tag == "tiny" else osd.SD_V1
    n, hw, tc = g["eps"].shape[0], int(g["hw"]), int(g["tc"])
    w = synth_state_dict(osd.unet_state_dict_shapes(cfg), 0)
    x, ctx = seeded_noise((n, cfg.in_channels, hw, hw), 71), seeded_noise((n, tc, cfg.context_dim), 72)
    with torch.no_grad():
        eps = osd.unet_forward(w, cfg, x, g["t"], ctx)
    assert float((eps - g["eps"]).abs().max()) < 3e-5 * (1 + float(g["eps"].abs().max()))


@pytest.mark.parametrize("tag", ["tiny", "v1"])
def test_sd_vae_vs_reference_ldm(tag):
    """VAE decoder and encoder of oracle/sd.py against the reference's vendored ldm Decoder / Encoder (ldm/modules/diffusionmodules/model.py)."""
    from oracle import sd as osd
    from perceptor_amd.utils.synth import seeded_noise
    g = golden(f"sd_ldm_vae_{tag}")
    cfg = osd.VAE_TINY if tag == "tiny" else osd.VAE_V1
    hw, ihw = int(g["hw"]), int(g["img_hw"])
    w = synth_state_dict({**osd.vae_encoder_state_dict_shapes(cfg), **osd.vae_decoder_state_dict_shapes(cfg)}, 0)
    with torch.no_grad():
        dec = osd.vae_decode(w, cfg, seeded_noise((1, cfg.latent_channels, hw, hw), 73))
        mean, logvar = osd.vae_encode_moments(w, cfg, seeded_noise((1, 3, ihw, ihw), 74) * 0.5)
    assert float((dec - g["dec"]).abs().max()) < 3e-5 * (1 + float(g["dec"].abs().max()))
    assert float((mean - g["mean"]).abs().max()) < 3e-5 * (1 + float(g["mean"].abs().max()))
    assert float((logvar - g["logvar"]).abs().max()) < 3e-5 * (1 + float(g["logvar"].abs().max()))
