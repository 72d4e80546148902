"""GPU: the UNet input-gradient path: what autograd computes through the reference's YFCC2Model when
losses/velocity_diffusion.py:33-61 (guided_resample_) backpropagates a loss on the denoised image to the noise.

  * adjoint kernels against torch autograd of the forward op (avg-pool, bilinear x2, GroupNorm(1, C)), and <A x, y> = <x, A^T y>;
  * engine backward on a tiny net of the family vs autograd through the fp32 oracle;
  * the full 968 M-parameter yfcc_2 at 128x128 vs the gradient the REFERENCE's autograd produced (tests/golden/vdiff_yfcc_2_128_grad.npz);
  * losses.VelocityDiffusion.guided_resample_ vs the same chain written with autograd over the oracle.
Tolerance: the gradient passes through the same 16-bit layers as the forward and back again, and every ReLU mask is taken from a
16-bit-rounded activation: a forward error of relative size e flips about a fraction e of the masks, a discrete error of relative size
~sqrt(e) in the gradient -- bf16 relative L2 <= 1e-1 (measured 6.1e-2 tiny, 6.7e-2 full yfcc_2), cosine >= 0.995 (0.998); f16 <= 4e-2
(2.1e-2) / 0.9995 (0.9998).  The consumer, Predictions.guided, clamps the gradient to +-clamp_value, i.e. uses its sign.
"""
import pytest
import torch
import torch.nn.functional as F

from conftest import golden

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


def _cos(a, b):
    return float(F.cosine_similarity(a.double().flatten(), b.double().flatten(), dim=0))


def _nhwc16(x, dtype):
    return x.permute(0, 2, 3, 1).contiguous().to(dtype).to(DEV)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_adjoint_kernels_vs_autograd(dtype):
    from perceptor_amd._hip import call, dtype_code, ptr
    dt = dtype_code("bf16" if dtype == torch.bfloat16 else "f16")
    g = torch.Generator().manual_seed(3)
    n, c, h, w = 2, 16, 6, 10
    x = torch.randn(n, c, h, w, generator=g).to(dtype).float().requires_grad_()
    for name, fwd, oshape in (("pmi_avgpool2_bwd", lambda t: F.avg_pool2d(t, 2), (n, c, h // 2, w // 2)),
                              ("pmi_upsample_bilinear2_bwd", lambda t: F.interpolate(t, scale_factor=2, mode="bilinear", align_corners=False),
                               (n, c, 2 * h, 2 * w))):
        dy = torch.randn(*oshape, generator=g).to(dtype).float()
        (ref,) = torch.autograd.grad((fwd(x) * dy).sum(), x)
        out = torch.empty((n, h, w, c), dtype=dtype, device=DEV)
        dyd = _nhwc16(dy, dtype)
        call(name, ptr(dyd), ptr(out), n, h, w, c, dt)
        got = out.float().cpu().permute(0, 3, 1, 2)
        assert float((got - ref).abs().max()) <= 2 ** (-7 if dtype == torch.bfloat16 else -10) * float(ref.abs().max()), name
    # GroupNorm(1, C) with affine, plus the residual path of the attention block
    gamma, beta = 1 + 0.1 * torch.randn(c, generator=g), 0.1 * torch.randn(c, generator=g)
    dy = torch.randn(n, c, h, w, generator=g).to(dtype).float()
    res = torch.randn(n, c, h, w, generator=g).to(dtype).float()
    (ref,) = torch.autograd.grad((F.group_norm(x, 1, gamma, beta, eps=1e-5) * dy).sum(), x)
    out = torch.empty((n, h, w, c), dtype=dtype, device=DEV)
    gam = gamma.to(DEV)
    xd, dyd, resd = _nhwc16(x.detach(), dtype), _nhwc16(dy, dtype), _nhwc16(res, dtype)     # named: a temporary's block is recycled by the next one
    from perceptor_amd import _hip
    part = torch.empty((n, _hip.lib().pmi_gn1_bwd_partials(h * w, c), 4), dtype=torch.float64, device=DEV)
    call("pmi_gn1_bwd", ptr(xd), ptr(dyd), ptr(gam), 0, 0.0, ptr(resd), ptr(out), ptr(part), n, h * w, c, 1e-5, dt)
    # a map large enough for several slices per sample, per-sample (FiLM) scale: vs autograd of group_norm * (1 + scale)
    n2, c2, h2, w2 = 2, 64, 48, 64
    xb = torch.randn(n2, c2, h2, w2, generator=g).to(dtype).float().requires_grad_()
    dyb = torch.randn(n2, c2, h2, w2, generator=g).to(dtype).float()
    film = 0.2 * torch.randn(n2, 2 * c2, generator=g)
    yb = F.group_norm(xb, 1, eps=1e-5) * (1 + film[:, :c2, None, None]) + film[:, c2:, None, None]
    (refb,) = torch.autograd.grad((yb * dyb).sum(), xb)
    xbd, dybd, filmd = _nhwc16(xb.detach(), dtype), _nhwc16(dyb, dtype), film.to(DEV)
    outb = torch.empty((n2, h2, w2, c2), dtype=dtype, device=DEV)
    P = _hip.lib().pmi_gn1_bwd_partials(h2 * w2, c2)
    assert P > 1
    partb = torch.empty((n2, P, 4), dtype=torch.float64, device=DEV)
    call("pmi_gn1_bwd", ptr(xbd), ptr(dybd), ptr(filmd), 2 * c2, 1.0, None, ptr(outb), ptr(partb), n2, h2 * w2, c2, 1e-5, dt)
    gotb = outb.float().cpu().permute(0, 3, 1, 2)
    assert float((gotb - refb).abs().max()) <= 2 ** (-6 if dtype == torch.bfloat16 else -9) * float(refb.abs().max())
    got = out.float().cpu().permute(0, 3, 1, 2)
    assert float((got - (ref + res)).abs().max()) <= 2 ** (-6 if dtype == torch.bfloat16 else -9) * float((ref + res).abs().max())
    a, b = torch.randn(4, 8, 8, 16, generator=g).to(dtype).to(DEV), torch.randn(4, 8, 8, 16, generator=g).to(dtype).to(DEV)
    s = torch.empty_like(a)
    call("pmi_add16", ptr(a), ptr(b), ptr(s), a.numel(), dt)
    assert torch.equal(s, (a.float() + b.float()).to(dtype))


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
@pytest.mark.parametrize("cond", [False, True])
def test_tiny_net_input_gradient_vs_oracle_autograd(cond, dtype):
    from oracle import vdiff as ov
    from perceptor_amd.engine import vdiff
    from perceptor_amd.utils.synth import seeded_noise, synth_state_dict
    spec = vdiff.make_spec("tiny", (3, 32, 32), [64, 128, 128], 2, 2, 4, 1, cond)
    sd = synth_state_dict(vdiff.state_dict_shapes(spec), 0)
    eng = vdiff.VDiffEngine(spec, sd, DEV, dtype)
    x = seeded_noise((2, 3, 32, 48), 5)
    t = torch.tensor([0.9, 0.3])
    ce = seeded_noise((2, 512), 6) if cond else None
    probe = seeded_noise((2, 3, 32, 48), 8)
    xr = x.clone().requires_grad_()
    with torch.enable_grad():
        v_ref = ov.vdiff_forward.__wrapped__(sd, ov.tiny_spec(cond), xr, t, ce)        # the oracle without its no_grad wrapper
        (g_ref,) = torch.autograd.grad((v_ref * probe).sum(), xr)
    img = ((x + 1) / 2).to(DEV)
    ced = ce.to(DEV) if cond else None
    v, tape = eng.forward_train(img, t.to(DEV), ced)
    assert _rel(v.cpu(), v_ref.detach()) <= 2e-2
    # training-mode forward keeps relu(conv2) as its own 16-bit tensor before the skip add (one more rounding than the fused inference epilogue)
    assert _rel(v.cpu(), eng.forward(img, t.to(DEV), ced).cpu()) <= (1e-2 if dtype == "bf16" else 2e-3)
    g_img = eng.backward(tape, probe.to(DEV), sd)
    g_x = g_img.cpu() / 2                                                              # images = (x + 1) / 2
    rel, cos = _rel(g_x, g_ref), _cos(g_x, g_ref)
    print(f"[parity] tiny v-net (cond={cond}) input gradient {dtype}: rel-L2={rel:.3e}, cos={cos:.5f}")
    assert rel <= (1e-1 if dtype == "bf16" else 4e-2) and cos >= (0.995 if dtype == "bf16" else 0.9995)


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
def test_yfcc2_full_input_gradient_vs_reference_autograd(dtype):
    from perceptor_amd import models
    from perceptor_amd.utils.synth import seeded_noise
    g0, g = golden("vdiff_yfcc_2_128"), golden("vdiff_yfcc_2_128_grad")
    m = models.VelocityDiffusion("yfcc_2", dtype=dtype).to(DEV)
    img = ((g0["x"] + 1) / 2).to(DEV).requires_grad_()
    probe = seeded_noise((1, 3, 128, 128), 46).to(DEV)
    with torch.enable_grad():                                                          # through the public surface: autograd.Function on velocities()
        v = m.velocities(img, g["t"].to(DEV))
        (v * probe).sum().backward()
    g_x = img.grad.cpu() / 2
    rel, cos = _rel(g_x[:, :, ::2, ::2], g["g_sub"]), _cos(g_x[:, :, ::2, ::2], g["g_sub"])
    print(f"[parity] yfcc_2@128 input gradient {dtype} vs reference autograd: rel-L2={rel:.3e}, cos={cos:.5f}")
    assert (rel <= 1e-1 and cos >= 0.995) if dtype == "bf16" else (rel <= 5e-2 and cos >= 0.999)     # (f16: ~sqrt(1e-3) of the ReLU masks' neighbourhood flips)
    f = g_x.flatten(1).double()
    assert torch.allclose(f.norm(dim=1).float(), g["g_mom"][:, 2], rtol=5e-2)


def test_cc12m1_full_input_gradient_vs_reference_autograd():
    """The CLIP-conditioned 603 M-parameter cc12m_1 (GroupNorm(1) + Modulation2d blocks) at 64x64 vs the reference's autograd."""
    from perceptor_amd import models
    from perceptor_amd.utils.synth import seeded_noise
    g0, g = golden("vdiff_cc12m_1_64"), golden("vdiff_cc12m_1_64_grad")
    m = models.VelocityDiffusion("cc12m_1_cfg", dtype="bf16").to(DEV)
    img = ((g0["x"] + 1) / 2).to(DEV).requires_grad_()
    probe = seeded_noise((1, 3, 64, 64), 47).to(DEV)
    with torch.enable_grad():
        v = m.velocities(img, g["t"].to(DEV), g0["clip_embed"][:, None, :].to(DEV))
        (v * probe).sum().backward()
    g_x = img.grad.cpu() / 2
    rel, cos = _rel(g_x, g["g"]), _cos(g_x, g["g"])
    print(f"[parity] cc12m_1@64 input gradient bf16 vs reference autograd: rel-L2={rel:.3e}, cos={cos:.5f}")
    assert rel <= 1.5e-1 and cos >= 0.99


def test_cc12m1_conditioning_gradient_vs_reference_autograd():
    """Upstream VelocityDiffusion.velocities keeps `conditioning` in the autograd graph (velocity_diffusion.py:96-109): d loss / d clip_embed
    through every Modulation2d, the mapping network and F.normalize, against the reference's autograd on the full 603 M-parameter net."""
    from perceptor_amd import models
    from perceptor_amd.utils.synth import seeded_noise
    g0, g = golden("vdiff_cc12m_1_64"), golden("vdiff_cc12m_1_64_grad")
    m = models.VelocityDiffusion("cc12m_1_cfg", dtype="bf16").to(DEV)
    img = ((g0["x"] + 1) / 2).to(DEV).requires_grad_()
    ce = g0["clip_embed"][:, None, :].to(DEV).requires_grad_()
    probe = seeded_noise((1, 3, 64, 64), 47).to(DEV)
    with torch.enable_grad():
        v = m.velocities(img, g["t"].to(DEV), ce)
        (v * probe).sum().backward()
    g_ce = ce.grad.cpu()[:, 0]
    rel, cos = _rel(g_ce, g["g_ce"]), _cos(g_ce, g["g_ce"])
    print(f"[parity] cc12m_1@64 conditioning gradient bf16 vs reference autograd: rel-L2={rel:.3e}, cos={cos:.5f}")
    assert rel <= 1.5e-1 and cos >= 0.99
    assert _rel(img.grad.cpu() / 2, g["g"]) <= 1.5e-1                     # the image gradient of the same call is unchanged
    # conditioning-only gradient (the image detached) takes the same path
    ce2 = g0["clip_embed"][:, None, :].to(DEV).requires_grad_()
    with torch.enable_grad():
        (m.velocities(img.detach(), g["t"].to(DEV), ce2) * probe).sum().backward()
    assert torch.equal(ce2.grad, ce.grad)


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
def test_wikiart_style_tiny_net_input_gradient(dtype):
    """The wikiart layer set (no attention norm, non-64-channel heads through the batched-GEMM attention backward, nearest upsampling and its
    adjoint, skip-first concat, log-SNR timestep features) on a tiny net vs autograd over the oracle."""
    from oracle import vdiff as ov
    from perceptor_amd.engine import vdiff
    from perceptor_amd.utils.synth import seeded_noise, synth_state_dict
    kw = dict(head_dim=32, attn_norm=False, up_mode="nearest", t_input="log_snr", skip_first=True)     # 32-channel heads: the batched-GEMM path, as 128 is
    spec = vdiff.make_spec("tinyw", (3, 32, 32), [64, 128, 128], 2, 2, 4, 1, False, **kw)
    sd = synth_state_dict(vdiff.state_dict_shapes(spec), 0, gain=0.7)
    eng = vdiff.VDiffEngine(spec, sd, DEV, dtype)
    ospec = dict(ov.tiny_spec(False), head_dim=32, up_mode="nearest", t_input="log_snr", skip_first=True)
    x = seeded_noise((2, 3, 32, 32), 5)
    t = torch.tensor([0.8, 0.3])
    probe = seeded_noise((2, 3, 32, 32), 8)
    xr = x.clone().requires_grad_()
    with torch.enable_grad():
        v_ref = ov.vdiff_forward.__wrapped__(sd, ospec, xr, t)
        (g_ref,) = torch.autograd.grad((v_ref * probe).sum(), xr)
    img = ((x + 1) / 2).to(DEV)
    v, tape = eng.forward_train(img, t.to(DEV))
    assert _rel(v.cpu(), v_ref.detach()) <= 2e-2
    g_x = eng.backward(tape, probe.to(DEV), sd).cpu() / 2
    rel, cos = _rel(g_x, g_ref), _cos(g_x, g_ref)
    print(f"[parity] tiny wikiart-style v-net input gradient {dtype}: rel-L2={rel:.3e}, cos={cos:.5f}")
    assert rel <= (1e-1 if dtype == "bf16" else 4e-2) and cos >= (0.995 if dtype == "bf16" else 0.9995)


def test_wikiart_full_input_gradient_vs_reference_autograd():
    from perceptor_amd import models
    from perceptor_amd.utils.synth import seeded_noise
    g0, g = golden("vdiff_wikiart_64"), golden("vdiff_wikiart_64_grad")
    m = models.VelocityDiffusion("wikiart", dtype="bf16", weight_gain=0.6).to(DEV)
    img = ((g0["x"] + 1) / 2).to(DEV).requires_grad_()
    probe = seeded_noise((1, 3, 64, 64), 48).to(DEV)
    with torch.enable_grad():
        v = m.velocities(img, g["t"].to(DEV))
        (v * probe).sum().backward()
    g_x = img.grad.cpu() / 2
    rel, cos = _rel(g_x, g["g"]), _cos(g_x, g["g"])
    print(f"[parity] wikiart@64 input gradient bf16 vs reference autograd: rel-L2={rel:.3e}, cos={cos:.5f}")
    assert rel <= 1.5e-1 and cos >= 0.99


def test_guided_resample_matches_autograd_chain_over_the_oracle():
    """losses.VelocityDiffusion.guided_resample_ (reference losses/velocity_diffusion.py:33-61) on a tiny net: the noise gradient
    against autograd over diffuse -> oracle UNet -> denoised_images, then the update rule with the device noise injected."""
    import math
    from oracle import sampling
    from oracle import vdiff as ov
    from perceptor_amd import losses, models
    from perceptor_amd.engine import sampler, vdiff
    from perceptor_amd.utils.synth import seeded_noise
    spec = vdiff.make_spec("tiny", (3, 32, 32), [64, 128, 128], 2, 2, 4, 1, False)
    m = models.VelocityDiffusion("yfcc_2", spec=spec, dtype="bf16").to(DEV)
    sd = {k: v.detach().cpu() for k, v in m.model.state_dict().items()}
    den0 = seeded_noise((2, 3, 32, 32), 12) * 0.2 + 0.5
    noise0 = seeded_noise((2, 3, 32, 32), 13)
    target = seeded_noise((2, 3, 32, 32), 14) * 0.2 + 0.5
    loss = losses.VelocityDiffusion(m, noise0.clone().to(DEV), from_ts=0.5, resample_ts=0.3)
    torch.manual_seed(77)
    key = sampler.rng.next_key()                                                        # the key the resample draw will take
    torch.manual_seed(77)
    captured = {}
    with loss.guided_resample_(den0.to(DEV), guidance_scale=0.5, clamp_value=1e-6) as dd:
        l = (dd - target.to(DEV)).square().mean()
        l.backward()
        captured["g"] = dd.grad.clone()
    # ---- the same chain with autograd over the oracle
    a, s = math.cos(0.5 * math.pi / 2), math.sin(0.5 * math.pi / 2)
    nz = noise0.clone().requires_grad_()
    with torch.enable_grad():
        x_d = (den0 * 2 - 1) * a + nz * s
        v = ov.vdiff_forward.__wrapped__(sd, ov.tiny_spec(False), x_d, torch.full((2,), 0.5))
        dd_ref = ((x_d * a - v * s) + 1) / 2
        (dd_ref - target).square().mean().backward()
    # reproduce the update: guided(-grad) then resample_noise(0.3) with the generator's noise
    u64 = lambda k: k & ((1 << 64) - 1)
    rn = sampling.device_randn((2, 3, 32, 32), u64(key[0]), u64(key[1]))
    eps_ref = (x_d * s + v * a).detach()
    v_g = sampling.guided(v.detach(), -nz.grad, torch.full((2,), s))
    eps_g = x_d.detach() * s + v_g * a
    sr = math.sin(0.3 * math.pi / 2)
    want = sampling.resample_noise(eps_g, torch.full((2,), s), torch.full((2,), sr), rn)
    got = loss.noise.data.cpu()
    # the guidance term is sign-like (clamp at 1e-6): compare where the reference gradient is clearly away from zero
    sure = nz.grad.abs() > 0.25 * nz.grad.abs().mean()
    frac = float(((got - want).abs() < 5e-2)[sure].float().mean())
    print(f"[parity] guided_resample_: noise update agrees on {frac:.4f} of the {int(sure.sum())} elements with |grad| > mean/4; "
          f"guidance step {float((eps_ref - eps_g).abs().max()):.3f}")
    assert frac >= 0.98
    assert float(loss.noise.grad.abs().max()) == 0.0                                   # zeroed at the end, as upstream


# ---- ADM UNet (GuidedDiffusion) input gradient, round 3 -----------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("case", [dict(c0=64, c1=0, film=True, act=2), dict(c0=64, c1=32, film=False, act=2), dict(c0=96, c1=0, film=False, act=0)])
def test_group_norm32_backward_vs_autograd(case, dtype):
    """pmi_gn_bwd_stats / _finalize / _apply against torch autograd of act(GroupNorm32(cat(x, x1)) * gamma (1 + scale) + beta' ...), one and two
    sources, with FiLM, with the gradient arriving over a second path added on the way out."""
    from perceptor_amd._hip import dtype_code
    from perceptor_amd.engine import ops
    dt = dtype_code("bf16" if dtype == torch.bfloat16 else "f16")
    g = torch.Generator().manual_seed(5)
    n, h, w, c0, c1 = 2, 12, 16, case["c0"], case["c1"]
    c = c0 + c1
    x = (torch.randn(n, c, h, w, generator=g) * 1.5 + 0.3).to(dtype).float().requires_grad_()
    gamma, beta = 1 + 0.1 * torch.randn(c, generator=g), 0.1 * torch.randn(c, generator=g)
    film = 0.2 * torch.randn(n, 2 * c, generator=g) if case["film"] else None
    y = F.group_norm(x, 32, gamma, beta, eps=1e-5)
    if film is not None:
        y = y * (1 + film[:, :c, None, None]) + film[:, c:, None, None]
    if case["act"] == 2:
        y = F.silu(y)
    dy = torch.randn(n, c, h, w, generator=g).to(dtype).float()
    extra = torch.randn(n, c, h, w, generator=g).to(dtype).float()
    (ref,) = torch.autograd.grad((y * dy).sum() + (x * extra).sum(), x)
    xd = _nhwc16(x.detach(), dtype)
    x0, x1 = (xd[..., :c0].contiguous(), xd[..., c0:].contiguous()) if c1 else (xd, None)
    ed = _nhwc16(extra, dtype)
    e0, e1 = (ed[..., :c0].contiguous(), ed[..., c0:].contiguous()) if c1 else (ed, None)
    filmd = film.to(DEV) if film is not None else None
    ca, cb, parts = ops.group_norm_coeffs_train(x0, gamma.to(DEV), beta.to(DEV), 32, dt, x1=x1, film=filmd, film_ld=2 * c if film is not None else 0)
    dx0, dx1 = ops.group_norm_backward(x0, _nhwc16(dy, dtype), ca, cb, parts, gamma.to(DEV), 32, dt, x1=x1, film=filmd,
                                       film_ld=2 * c if film is not None else 0, act=case["act"], gadd0=e0, gadd1=e1)
    got = torch.cat([dx0, dx1], dim=-1) if c1 else dx0
    got = got.float().cpu().permute(0, 3, 1, 2)
    tol = 2 ** (-6 if dtype == torch.bfloat16 else -9) * float(ref.abs().max())
    assert float((got - ref).abs().max()) <= tol, (float((got - ref).abs().max()), tol)


ADM_TINY = {
    "a": dict(image_size=64, model_channels=32, num_res_blocks=1, channel_mult=(1, 2, 2), attention_ds=(2, 4),
              num_head_channels=16, use_scale_shift_norm=True, resblock_updown=True),
    "b": dict(image_size=64, model_channels=32, num_res_blocks=2, channel_mult=(1, 2), attention_ds=(2,),
              num_heads=2, use_new_attention_order=True),
}


def _adm_grad(cfg_kw, dtype, x, t, probe, standard=False):
    from perceptor_amd.engine import adm
    from perceptor_amd.utils.synth import synth_state_dict
    cfg = adm.openimages_config() if standard else adm.AdmConfig(**cfg_kw)
    sd = synth_state_dict(adm.state_dict_shapes(cfg), 0)
    eng = adm.AdmEngine(cfg, sd, DEV, dtype)
    img = ((x + 1) / 2).to(DEV)
    y, tape = eng.forward_train(img, t.to(DEV), sd, out_channels=3)
    y_inf = eng.forward(img, t.to(DEV), out_channels=3)
    assert float((y - y_inf).abs().max()) <= 1e-6 + 2e-2 * float(y_inf.abs().max())      # the tape forward is the inference forward (attention path aside)
    g_img = eng.backward(tape, probe.to(DEV), sd)
    return g_img.cpu() / 2.0, y.cpu()          # d / d x with x = 2 img - 1


@pytest.mark.parametrize("dtype,tol_l2,tol_cos", [("bf16", 4e-2, 0.999), ("f16", 6e-3, 0.99995)])      # measured 1.1-1.9e-2 / 0.9998, 1.4-2.4e-3 / 0.999997
@pytest.mark.parametrize("tag", ["a", "b"])
def test_adm_tiny_input_gradient_vs_reference_autograd(tag, dtype, tol_l2, tol_cos):
    """Both tiny configs (both attention orders and ResBlock flavours, up / down ResBlocks, FiLM and additive timestep conditioning) against
    the gradient the REFERENCE's autograd produced (tests/golden/adm_tiny_*_grad.npz).  No ReLU masks on this path (SiLU is smooth), so the
    bounds are tighter than the v-diffusion nets': the error is the 16-bit rounding of ~60 layers forward and back."""
    from perceptor_amd.utils.synth import seeded_noise
    g0 = golden(f"adm_tiny_{tag}_grad")
    x = seeded_noise((2, 3, 64, 64), 31)
    probe = seeded_noise((2, 3, 64, 64), 61)
    got, _ = _adm_grad(ADM_TINY[tag], dtype, x, g0["t"], probe)
    rel, cos = _rel(got, g0["g"]), _cos(got, g0["g"])
    print(f"[grad] adm_tiny_{tag} {dtype}: rel-L2 {rel:.3e} cos {cos:.6f}")
    assert rel <= tol_l2 and cos >= tol_cos, (rel, cos)


@pytest.mark.parametrize("dtype,tol_l2,tol_cos", [("bf16", 3e-2, 0.9995), ("f16", 4e-3, 0.99999)])       # measured 1.14e-2 / 0.99994, 1.42e-3 / 0.999999
def test_adm_standard_128_input_gradient_vs_reference_autograd(dtype, tol_l2, tol_cos):
    """The shipped 558 M-parameter net at 128x128 (64-channel heads: the flash attention backward) vs the reference's autograd."""
    from perceptor_amd.utils.synth import seeded_noise
    g0 = golden("adm_standard_128_grad")
    x = seeded_noise((1, 3, 128, 128), 32)
    probe = seeded_noise((1, 3, 128, 128), 62)
    got, _ = _adm_grad(None, dtype, x, g0["t"], probe, standard=True)
    sub = got[:, :, ::2, ::2]
    rel, cos = _rel(sub, g0["g_sub"]), _cos(sub, g0["g_sub"])
    print(f"[grad] adm_standard_128 {dtype}: rel-L2 {rel:.3e} cos {cos:.6f}")
    assert rel <= tol_l2 and cos >= tol_cos, (rel, cos)
    f = got.flatten(1).double()
    assert abs(float(f.norm()) / float(g0["g_mom"][0, 2]) - 1) < 5e-2


def test_guided_diffusion_predicted_noise_is_differentiable():
    """models.GuidedDiffusion.predicted_noise(x.requires_grad_()) backpropagates like upstream (guided_diffusion.py:125-133); under no_grad or
    with a detached input it is the plain inference call; the precise / mixed modes return their own value and the f16 engine's gradient."""
    from perceptor_amd import models
    from perceptor_amd.engine import adm
    cfg = adm.AdmConfig(**ADM_TINY["a"])
    m = models.GuidedDiffusion(config=cfg, dtype="f16").to(DEV)
    img = (torch.rand(2, 3, 64, 64, generator=torch.Generator().manual_seed(1))).to(DEV).requires_grad_()
    eps = m.predicted_noise(img, torch.tensor([300, 20]))
    assert eps.requires_grad
    w = torch.randn(eps.shape, generator=torch.Generator().manual_seed(2)).to(DEV)
    (gr,) = torch.autograd.grad((eps * w).sum(), img)
    assert gr.shape == img.shape and torch.isfinite(gr).all() and float(gr.abs().max()) > 0
    # the class surface is the engine's forward_train / backward (whose gradient the tests above pin on the reference's autograd)
    eng = m.engine
    idx = torch.tensor([300, 20]).to(DEV)
    _, tape = eng.forward_train(img.detach(), idx, m.model.state_dict(), out_channels=3)
    want = eng.backward(tape, w, m.model.state_dict())
    assert torch.equal(gr, want)
    with torch.no_grad():
        assert not m.predicted_noise(img, 300).requires_grad
    # precise (and mixed) models: the value is the mode's own, the gradient the f16 engine's (lazily built twin)
    mp = models.GuidedDiffusion(config=cfg, dtype="precise").to(DEV)
    img2 = img.detach().clone().requires_grad_()
    eps_p = mp.predicted_noise(img2, torch.tensor([300, 20]))
    with torch.no_grad():
        assert torch.equal(eps_p.detach(), mp.predicted_noise(img2.detach(), torch.tensor([300, 20])))
    (gp,) = torch.autograd.grad((eps_p * w).sum(), img2)
    assert torch.equal(gp, gr)
