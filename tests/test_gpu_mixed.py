"""GPU: the mixed-precision mode (engine/adm_mixed.py) -- its kernels against fp32 torch on the same split operands, and the engine against the
reference golden / the precise engine with the contract's ABSOLUTE bound: eps max-abs error < 1e-3 (north_star).

Reference precision model: guided_diffusion.py:125-133 (autocast), unet.py:610-616 (fp16 torso), nn.py:17-19 (fp32 GroupNorm)."""
import pytest
import torch
import torch.nn.functional as F

from conftest import golden

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def to_split(x):
    """fp32 [..., C] -> hi + lo f16 pairs [..., 2C] in the [C/32][hi 32 | lo 32] layout (csrc/common.h F16X2)."""
    c = x.shape[-1]
    g = 32 if c % 32 == 0 else c
    hi = x.half()
    lo = (x - hi.float()).half()
    return torch.stack([hi.reshape(*x.shape[:-1], c // g, g), lo.reshape(*x.shape[:-1], c // g, g)], dim=-2).reshape(*x.shape[:-1], 2 * c).contiguous()


def from_split(s):
    c = s.shape[-1] // 2
    g = 32 if c % 32 == 0 else c
    v = s.float().reshape(*s.shape[:-1], c // g, 2, g)
    return (v[..., 0, :] + v[..., 1, :]).reshape(*s.shape[:-1], c)


def test_split_convert_round_trip():
    from perceptor_amd.engine import ops
    g = torch.Generator().manual_seed(0)
    x = torch.randn(3, 5, 7, 96, generator=g)
    s = to_split(x).to(DEV)
    p = ops.split_convert(s, False)
    assert torch.equal(p.cpu(), from_split(s.cpu()).half())
    s2 = ops.split_convert(p, True)
    assert torch.equal(from_split(s2.cpu()), p.cpu().float())


CASES = [
    dict(n=2, h=16, w=32, cin=64, cout=256, cfg=6),                            # 256-channel tiles
    dict(n=1, h=16, w=64, cin=128, cout=256, cfg=6, res=True),
    dict(n=2, h=16, w=32, cin=192, cout=256, cfg=6, split=128, res=True),      # skip-concat: two split sources
    dict(n=1, h=8, w=32, cin=64, cout=256, cfg=6, up=True, res_up=True),       # up ResBlock: gathered input, residual through the same up-sampling
    dict(n=2, h=16, w=32, cin=64, cout=128, cfg=7),                            # 128-channel tiles, two workgroups per CU
    dict(n=1, h=32, w=32, cin=128, cout=128, cfg=7, res=True, nbias=True),
    dict(n=1, h=16, w=32, cin=384, cout=128, cfg=7, split=256, res=True),
    dict(n=1, h=8, w=16, cin=64, cout=128, cfg=7, up=True, res_up=True),
    dict(n=1, h=16, w=16, cin=64, cout=128, cfg=-1, res=True),                 # W % 32 != 0: the generic fallback (always the doubled operand)
    dict(n=1, h=16, w=32, cin=128, cout=256, cfg=6, res=True, res_f32=True),   # fp32 residual: the 1x1 skip_connection's rows (weights-direct GEMM)
    dict(n=1, h=16, w=32, cin=384, cout=128, cfg=7, split=256, res=True, res_f32=True),
    dict(n=1, h=16, w=16, cin=64, cout=128, cfg=-1, res=True, res_f32=True),
]


@pytest.mark.parametrize("operand", ["single", "dbl"])
@pytest.mark.parametrize("case", CASES)
def test_conv3x3_mixed_vs_torch(case, operand):
    from perceptor_amd import _hip
    from perceptor_amd.engine import ops
    g = torch.Generator().manual_seed(7)
    n, h, w, cin, cout = case["n"], case["h"], case["w"], case["cin"], case["cout"]
    up = case.get("up", False)
    hin, win = (h, w)
    ho, wo = (2 * h, 2 * w) if up else (h, w)
    x = torch.randn(n, hin, win, cin, generator=g)
    xs = to_split(x)
    xv = from_split(xs)                                          # the value the kernel sees
    wt = (torch.randn(cout, cin, 3, 3, generator=g) / (9 * cin) ** 0.5).half().float()
    b = torch.randn(cout, generator=g) * 0.1
    ca = 1 + 0.2 * torch.randn(n, cin, generator=g)
    cb = 0.3 * torch.randn(n, cin, generator=g)
    y = F.silu(xv * ca[:, None, None, :] + cb[:, None, None, :])
    if operand == "single" and case["cfg"] >= 0:
        y = y.half().float()
    else:
        y = from_split(to_split(y))
    yn = y.permute(0, 3, 1, 2)
    if up:
        yn = F.interpolate(yn, scale_factor=2, mode="nearest")
    ref = F.conv2d(yn.double(), wt.double(), b.double(), padding=1).float().permute(0, 2, 3, 1)
    nb = None
    if case.get("nbias"):
        nb = torch.randn(n, cout, generator=g) * 0.2
        ref = ref + nb[:, None, None, :]
    res = None
    if case.get("res") or case.get("res_up"):
        rshape = (n, ho // 2, wo // 2, cout) if case.get("res_up") else (n, ho, wo, cout)
        rv = torch.randn(*rshape, generator=g)
        res = rv.clone() if case.get("res_f32") else to_split(rv)
        rv = rv if case.get("res_f32") else from_split(res)
        if case.get("res_up"):
            rv = F.interpolate(rv.permute(0, 3, 1, 2), scale_factor=2, mode="nearest").permute(0, 2, 3, 1)
        ref = ref + rv
    ml = ops.MixedLinear(wt, b, DEV, sources=(case["split"], cin - case["split"]) if "split" in case else None)
    sp = case.get("split")
    x0 = xs.to(DEV) if sp is None else to_split(x[..., :sp]).to(DEV)
    x1 = None if sp is None else to_split(x[..., sp:]).to(DEV)
    if case["cfg"] >= 0:
        _hip.lib().pmi_set_option(1, case["cfg"])
    ops.MIXED_TRACE = []
    try:
        out = ops.conv3x3_mixed(x0, ml, x1=x1, operand=operand, prologue=(ca.to(DEV), cb.to(DEV), _hip.ACT_SILU), up=up,
                                residual=res.to(DEV) if res is not None else None, res_up=bool(case.get("res_up")),
                                nbias=nb.to(DEV) if nb is not None else None)
        route = ops.MIXED_TRACE[-1][0]
    finally:
        _hip.lib().pmi_set_option(1, -1)
        ops.MIXED_TRACE = None
    assert route == ("wd" if case["cfg"] >= 0 else "fallback"), route
    got = from_split(out.cpu())
    scale = float(ref.abs().max())
    err = float((got - ref).abs().max())
    # dbl: fp32 accumulation order + the fast exp / rcp of the device SiLU; single: a device SiLU one fp32 ulp off the host's can round an
    # operand element to the neighbouring f16 (2^-11 relative on one of 9 Cin terms)
    tol = (2e-5 if (operand == "dbl" or case["cfg"] < 0) else 2e-4) * scale
    assert err <= tol, (err, tol, scale)
    st = getattr(out, "_pmi_stats", None)
    if case["cfg"] >= 0:
        assert st is not None
        s = st[0].double().sum(1).cpu()                      # [n, cout, 2]
        want_s, want_q = got.double().sum((1, 2)), (got.double() ** 2).sum((1, 2))
        assert torch.allclose(s[..., 0], want_s, rtol=1e-4, atol=1e-3 * ho * wo ** 0.5)
        assert torch.allclose(s[..., 1], want_q, rtol=1e-4, atol=1e-3)


def _standard(dtype):
    from perceptor_amd.engine import adm, adm_mixed
    from perceptor_amd.utils.synth import synth_state_dict
    cfg = adm.openimages_config()
    sd = synth_state_dict(adm.state_dict_shapes(cfg), 0)
    if dtype == "mixed":
        return adm_mixed.AdmMixedEngine(cfg, sd, DEV)
    return adm.AdmEngine(cfg, sd, DEV, dtype)


def test_adm_standard_128_mixed_vs_reference_golden():
    """The shipped 558 M-parameter net at 128x128 against the REFERENCE's output: absolute 1e-3 (measured 4-5e-4, the sweep's prediction)."""
    g = golden("adm_standard_128")
    eng = _standard("mixed")
    y = eng.forward(((g["x"] + 1) / 2).to(DEV), g["t"].to(DEV))
    err = float((y[:, :, ::4, ::4].cpu() - g["y_sub"]).abs().max())
    print(f"[parity] adm_standard_128 mixed: max|err|={err:.3e} (scale {float(g['y_sub'].abs().max()):.3f})")
    assert err < 1e-3, err


def test_adm_standard_256_mixed_vs_precise_engine_and_routes():
    """At 256x256 the four upper levels run the weights-direct mixed kernels (both operand forms, both tile sizes): absolute 1e-3 against the
    precise engine (itself 3e-6 from the reference golden), two samples / timesteps, and the routing is what the design says."""
    from perceptor_amd.engine import ops
    from perceptor_amd.utils.synth import seeded_noise
    x = seeded_noise((2, 3, 256, 256), 4321)
    t = torch.tensor([900, 80])
    img = ((x + 1) / 2).to(DEV)
    ref = _standard("precise").forward(img, t.to(DEV)).cpu()
    torch.cuda.empty_cache()
    eng = _standard("mixed")
    ops.MIXED_TRACE = []
    try:
        y = eng.forward(img, t.to(DEV)).cpu()
        trace = ops.MIXED_TRACE
    finally:
        ops.MIXED_TRACE = None
    err = float((y - ref).abs().max())
    rms = float((y - ref).pow(2).mean().sqrt())
    print(f"[parity] adm_standard_256 mixed vs precise: max|err|={err:.3e} rms={rms:.3e} (scale {float(ref.abs().max()):.3f})")
    assert err < 1e-3, err
    wd = [r for r in trace if r[0] == "wd"]
    assert {r[1] for r in wd} == {"single", "dbl"} and len(wd) >= 20, (len(wd), len(trace))     # (batch 2: the 1/8 level's grids are below the tile kernels' thresholds)
    y2 = eng.forward(img, t.to(DEV)).cpu()
    assert torch.equal(y, y2)                                  # deterministic (fixed-order statistics)


def test_adm_pixelart_64_mixed_vs_reference_golden():
    """The other shipped config (conv_resample stride-2 / nearest-up convolutions, additive timestep conditioning, one 128-channel head):
    no per-layer table, the by-level rule; absolute 1e-3 against the reference's output at 64x64."""
    from perceptor_amd.engine import adm, adm_mixed
    from perceptor_amd.utils.synth import synth_state_dict
    g = golden("adm_pixelart_64")
    cfg = adm.pixelart_config()
    sd = synth_state_dict(adm.state_dict_shapes(cfg), 0)
    eng = adm_mixed.AdmMixedEngine(cfg, sd, DEV)
    y = eng.forward(((g["x"] + 1) / 2).to(DEV), g["t"].to(DEV))
    err = float((y[:, :, ::4, ::4].cpu() - g["y_sub"]).abs().max())
    print(f"[parity] adm_pixelart_64 mixed: max|err|={err:.3e} (scale {float(g['y_sub'].abs().max()):.3f})")
    assert err < 1e-3, err


def test_full_size_512_mixed_vs_precise_engine_deterministic_and_chain_independent():
    """BASELINE configs[4]'s own size (512x512): every level of the shipped net on the weights-direct mixed kernels (128-channel tiles at full
    resolution, two-source 768 -> 256, the 32x32 maps, the plain-f16 levels below and both level boundaries).  The CPU oracle cannot run this
    size in test time: the reference here is the precise engine (3e-6 from the reference's golden at 128x128), absolute bound 1e-3; plus what
    must hold at any size: determinism and batch-permutation invariance, bit-exact."""
    from perceptor_amd.engine import ops
    from perceptor_amd.utils.synth import seeded_noise
    x = seeded_noise((2, 3, 512, 512), 777)
    t = torch.tensor([700, 150])
    img = ((x + 1) / 2).to(DEV)
    ref = _standard("precise").forward(img, t.to(DEV), out_channels=3).cpu()
    torch.cuda.empty_cache()
    eng = _standard("mixed")
    ops.MIXED_TRACE = []
    try:
        y = eng.forward(img, t.to(DEV), out_channels=3)
        trace = ops.MIXED_TRACE
    finally:
        ops.MIXED_TRACE = None
    # (at batch 2 the 64x64 and 32x32 maps are below the tile kernels' grid thresholds and take the generic split route; at the benchmark's batch 8 all do)
    assert all(r[0] == "wd" for r in trace if r[2][1] >= 128), [r for r in trace if r[0] != "wd" and r[2][1] >= 128]
    assert sum(r[0] == "wd" for r in trace) >= 40
    err = float((y.cpu() - ref).abs().max())
    rms = float((y.cpu() - ref).pow(2).mean().sqrt())
    print(f"[parity] adm_standard_512 mixed vs precise: max|err|={err:.3e} rms={rms:.3e} (scale {float(ref.abs().max()):.3f})")
    assert err < 1e-3, err
    assert torch.equal(y, eng.forward(img, t.to(DEV), out_channels=3))
    yp = eng.forward(img.flip(0).contiguous(), t.flip(0).to(DEV), out_channels=3)
    assert torch.equal(yp, y.flip(0)), "a chain's output depends on its position in the batch"
