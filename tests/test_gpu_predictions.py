"""GPU: every array of tests/golden/sampling.npz and sampling2.npz consumed through the drop-in CLASSES (Predictions eps-form and
v-form, clamp_with_grad) -- the reference classes produced the fixtures (oracle/gen_golden.py: gen_sampling, gen_sampling2).

Tolerance: these are fp32 elementwise updates; the HIP kernels fuse what the reference does in ~10 torch ops, so results differ by a few
fp32 roundings: 2e-6 * (max|ref| + 1).  Index / selection results (quantile order statistics, sort) are exact up to the final lerp."""
import numpy as np
import pytest
import torch

from conftest import golden

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _close(got, want, tol=2e-6):
    got, want = got.float().cpu(), want.float().cpu()
    assert got.shape == want.shape, (got.shape, want.shape)
    err = float((got - want).abs().max())
    assert err <= tol * (float(want.abs().max()) + 1), err


def _gd(g, img, idx, eps):
    from perceptor_amd.models.guided_diffusion.predictions import Predictions
    return Predictions(from_diffused_images=img.to(DEV), from_indices=idx.to(DEV), predicted_noise=eps.to(DEV),
                       schedule_alphas=g["alphas"].to(DEV), schedule_sigmas=g["sigmas"].to(DEV))


def _vd(img, ts, v):
    from perceptor_amd.models.velocity_diffusion.predictions import Predictions
    return Predictions(from_diffused_images=img.to(DEV), from_ts=ts.to(DEV), velocities=v.to(DEV))


def test_eps_form_class_vs_reference_fixtures():
    g = golden("sampling")
    p = _gd(g, g["img"], g["fi"], g["eps"])
    _close(p.denoised_images, g["eps_denoised"])
    _close(p.step(g["ti"]), g["eps_step"])
    pg = p.guided(g["grad"].to(DEV), guidance_scale=0.5, clamp_value=1e-6)
    _close(pg.predicted_noise, g["eps_guided"])
    _close(pg.step(g["ti"]), g["eps_guided_step"])
    _close(p.forced_denoised_images(p.denoised_images.clamp(0, 1)).predicted_noise, g["eps_forced"], 4e-6)
    _close(_gd(g, g["img"], g["ti"], g["eps"]).reverse_step(g["fi"]), g["eps_reverse"])
    # N == 1 (the reference's dynamic_threshold only runs there: its [N] threshold broadcasts against W)
    p1 = _gd(g, g["img"][:1], g["fi"][:1], g["eps"][:1] * 3)
    _close(p1.dynamic_threshold(0.95).predicted_noise, g["eps_dynthr"], 4e-6)
    assert torch.equal(p.forced_predicted_noise(g["grad"].to(DEV)).predicted_noise.cpu(), g["grad"])
    with pytest.raises(ValueError):
        p.reverse_step(g["ti"])
    with pytest.raises(ValueError):
        _gd(g, g["img"], g["ti"], g["eps"]).resample(g["fi"])


def test_v_form_class_vs_reference_fixtures():
    g = golden("sampling")
    v = _vd(g["img"], g["ft"], g["eps"])
    _close(v.denoised_images, g["v_denoised"])
    _close(v.predicted_noise, g["v_eps"])
    _close(v.step(g["tt"]), g["v_step"])
    vg = v.guided(g["grad"].to(DEV), guidance_scale=0.5, clamp_value=1e-6)
    _close(vg.velocities, g["v_guided"])
    _close(vg.step(g["tt"]), g["v_guided_step"])
    _close(v.forced_denoised_images(v.denoised_images.clamp(0, 1)).velocities, g["v_forced"], 4e-6)
    _close(v.forced_predicted_noise(g["eps"].to(DEV) * 0.5).velocities, g["v_forced_eps"], 4e-6)
    _close(v.static_threshold().velocities, g["v_static"], 4e-6)
    _close(_vd(g["img"][:1], g["ft"][:1], g["eps"][:1] * 3).dynamic_threshold(0.95).velocities, g["v_dynthr"], 4e-6)


def test_stochastic_variants_with_injected_noise(monkeypatch):
    """step(eta > 0), resample_noise, resample, noisy_reverse_step: the arithmetic around the noise, with the fixture's noise injected in
    place of the device generator (exactly how the fixture was made from the reference: torch.randn_like replaced)."""
    from perceptor_amd.engine import sampler
    g, g2 = golden("sampling"), golden("sampling2")
    noise = g2["noise"].to(DEV)
    monkeypatch.setattr(sampler, "randn_like", lambda t: noise.clone())
    p = _gd(g, g["img"], g["fi"], g["eps"])
    _close(p.step(g["ti"], eta=0.7), g2["eps_step_eta"], 4e-6)
    _close(p.resample_noise(g["ti"]), g2["eps_resample_noise"], 4e-6)
    _close(p.resample(g["ti"]), g2["eps_resample"], 4e-6)
    _close(p.noisy_reverse_step(g2["hi"]), g2["eps_noisy_reverse"], 4e-6)
    v = _vd(g["img"], g["ft"], g["eps"])
    _close(v.step(g["tt"], eta=0.7), g2["v_step_eta"], 4e-6)
    _close(v.resample_noise(g["tt"]), g2["v_resample_noise"], 4e-6)
    _close(v.resample(g["tt"]), g2["v_resample"], 4e-6)
    _close(v.noisy_reverse_step(g2["ht"]), g2["v_noisy_reverse"], 4e-6)
    _close(_vd(g["img"], g["tt"], g["eps"]).reverse_step(g["ft"]), g2["v_reverse"], 4e-6)


def test_wasserstein_and_quantile_kernels_vs_reference_fixtures():
    from perceptor_amd.engine import sampler
    g, g2 = golden("sampling"), golden("sampling2")
    p = _gd(g, g["img"], g["fi"], g["eps"])
    _close(torch.stack([p.wasserstein_distance(), p.wasserstein_square_distance()]), g2["eps_wasserstein"], 1e-5)
    v = _vd(g["img"], g["ft"], g["eps"])
    _close(torch.stack([v.wasserstein_distance(), v.wasserstein_square_distance()]), g2["v_wasserstein"], 1e-5)
    big = g2["big"].to(DEV)                                   # 6000 elements per row: crosses the 4096-element LDS block; row 1 is all ties
    pb = _gd(g, big * 0.2 + 0.5, torch.tensor([500, 20, 999]), big)
    _close(torch.stack([pb.wasserstein_distance(), pb.wasserstein_square_distance()]), g2["big_wasserstein"], 1e-5)
    assert torch.equal(sampler.sort_rows(big).cpu(), g2["big"].flatten(1).sort(dim=1)[0])          # sorting is exact
    for i, q in enumerate((0.0, 0.5, 0.95, 0.999, 1.0)):
        _close(sampler.quantile_abs(big, q), g2["big_quantiles"][i], 1e-6)


@pytest.mark.parametrize("n", [1, 5, 4096, 4097, 3 * 512 * 512])
def test_sort_and_quantile_properties_any_length(n):
    """Sortedness + permutation (checksum of the multiset) + agreement with torch.quantile at sizes from 1 to a full 512x512 sample."""
    from perceptor_amd.engine import sampler
    gen = torch.Generator().manual_seed(n)
    x = torch.randn(2, n, generator=gen)
    x[1] = (x[1] * 8).round() / 8                              # heavy ties
    xd = x.to(DEV)
    s = sampler.sort_rows(xd).cpu()
    assert torch.equal(s, x.sort(dim=1)[0])
    for q in (0.0, 0.3, 0.95, 1.0):
        got = sampler.quantile_abs(xd, q).cpu()
        want = torch.quantile(x.abs().double(), q, dim=1) if n > 1 << 24 else torch.quantile(x.abs(), q, dim=1)
        assert float((got - want.float()).abs().max()) <= 1e-6 * (float(want.abs().max()) + 1), (n, q, got, want)


def test_device_rng_matches_restated_generator_and_contract():
    """pmi_randn vs the numpy restatement of Philox4x32-10 + Box-Muller (pinned by Random123's known answers in the CPU suite); the raw
    generator against the same known answers on the device; seeding contract: each draw keys itself from torch's CPU generator (torch.manual_seed reproduces a run), a rank's
    sample_offset selects its slice of the global draw."""
    from oracle import sampling
    from perceptor_amd._hip import call, ptr
    from perceptor_amd.engine import sampler
    raw = torch.empty(4, dtype=torch.int32, device=DEV)
    for ctr, key, want in (((0, 0), 0, (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
                           ((0x85a308d3243f6a88, 0x0370734413198a2e), 0x299f31d0a4093822, (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))):
        call("pmi_philox4x32_10", ptr(raw), sampler._i64(ctr[0]), sampler._i64(ctr[1]), sampler._i64(key))
        assert tuple(int(v) & 0xFFFFFFFF for v in raw.cpu()) == want
    like = torch.empty(4, 3, 16, 16, device=DEV)
    torch.manual_seed(1234)
    a0, a1 = sampler.randn_like(like), sampler.randn_like(like)
    torch.manual_seed(1234)
    b0 = sampler.randn_like(like)
    assert torch.equal(a0, b0) and not torch.equal(a0, a1)
    torch.manual_seed(1234)                                     # the keys the two draws took from the CPU generator
    keys = [sampler.rng.next_key() for _ in range(2)]
    u64 = lambda v: v & ((1 << 64) - 1)
    ref0 = sampling.device_randn(like.shape, u64(keys[0][0]), u64(keys[0][1]))
    ref1 = sampling.device_randn(like.shape, u64(keys[1][0]), u64(keys[1][1]))
    assert float((a0.cpu() - ref0).abs().max()) < 2e-5 and float((a1.cpu() - ref1).abs().max()) < 2e-5
    torch.manual_seed(1234)
    sampler.rng.sample_offset = 2                               # this "rank" holds samples 2..3 of the batch of 4
    try:
        part = sampler.randn_like(like[:2])
    finally:
        sampler.rng.sample_offset = 0
    assert torch.equal(part, a0[2:])
    own = sampler.DeviceRng().manual_seed(7)                    # a private generator decouples the draws from torch's global one
    c0 = own.randn_like(like)
    assert torch.equal(c0, sampler.DeviceRng().manual_seed(7).randn_like(like)) and not torch.equal(c0, a0)
    z = sampler.randn_like(torch.empty(8, 3, 512, 512, device=DEV))
    assert abs(float(z.mean())) < 2e-3 and abs(float(z.std()) - 1) < 2e-3 and float(z.abs().max()) < 6.0
    kurt = float((z.double() ** 4).mean())
    assert abs(kurt - 3.0) < 0.02


def test_clamp_with_grad_forward_backward():
    from perceptor_amd.transforms import clamp_with_grad, ClampWithGrad
    g2 = golden("sampling2")
    x = g2["cwg_x"].to(DEV).requires_grad_()
    y = clamp_with_grad(x, 0.0, 1.0)
    y.backward(g2["cwg_g"].to(DEV))
    assert torch.equal(y.detach().cpu(), g2["cwg_y"])
    assert torch.equal(x.grad.cpu(), g2["cwg_dx"])
    # per-sample bounds (dynamic_threshold's use) and the module form
    lo, hi = torch.tensor([0.2, -0.1]), torch.tensor([0.7, 1.5])
    x2 = g2["cwg_x"].to(DEV).requires_grad_()
    y2 = clamp_with_grad(x2, lo, hi)
    y2.backward(g2["cwg_g"].to(DEV))
    xr = g2["cwg_x"]
    cl = torch.maximum(torch.minimum(xr, hi[:, None, None, None]), lo[:, None, None, None])
    assert torch.equal(y2.detach().cpu(), cl)
    assert torch.equal(x2.grad.cpu(), g2["cwg_g"] * (g2["cwg_g"] * (xr - cl) >= 0))
    assert torch.equal(ClampWithGrad().encode(g2["cwg_x"].to(DEV)).cpu(), g2["cwg_y"])
