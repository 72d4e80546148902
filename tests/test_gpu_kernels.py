"""GPU: each HIP kernel against a plain PyTorch fp32 reference of the same op.

Inputs are pre-rounded to the 16-bit compute type, so with fp32 accumulation the only
differences are summation order and the final 16-bit rounding of the output:
tolerance = 2 ulp of the output type relative to the output scale (+ small abs term).
"""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DTYPES = ["f16", "bf16"]
ULP = {"f16": 2.0 ** -10, "bf16": 2.0 ** -7}


def _dev():
    assert torch.cuda.is_available(), "gpu tests need a HIP device"
    return torch.device("cuda:0")


def _r(x, dtype):
    td = torch.float16 if dtype == "f16" else torch.bfloat16
    return x.to(td).float()


def _nhwc(x, dtype):
    td = torch.float16 if dtype == "f16" else torch.bfloat16
    return x.permute(0, 2, 3, 1).contiguous().to(td)


def _check(got, ref, dtype, extra=1.0):
    scale = float(ref.abs().max()) + 1e-6
    err = float((got.float() - ref.float()).abs().max())
    tol = 2.5 * ULP[dtype] * scale * extra
    assert err <= tol, f"max err {err:.4e} > tol {tol:.4e} (scale {scale:.3e})"


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", [
    dict(n=2, h=16, w=16, cin=64, cout=128, k=3),
    dict(n=1, h=12, w=20, cin=8, cout=32, k=3),            # first conv (padded 3->8 channels), ragged tiles
    dict(n=2, h=8, w=8, cin=192, cout=64, k=3, split=128),  # skip-concat two-pointer K loop
    dict(n=2, h=8, w=8, cin=64, cout=64, k=3, up=True),
    dict(n=2, h=8, w=12, cin=64, cout=64, k=3, res_up=True),  # up ResBlock tail: residual = nearest-up(skip)
    dict(n=1, h=16, w=16, cin=64, cout=72, k=3, stride=2),
    dict(n=2, h=8, w=8, cin=128, cout=256, k=1),
    dict(n=2, h=16, w=16, cin=384, cout=256, k=1, split=256),   # ResBlock skip_connection over a skip concat: two-source weights-direct GEMM
    dict(n=3, h=8, w=24, cin=768, cout=512, k=1, split=512),
    dict(n=1, h=16, w=16, cin=224, cout=320, k=1, split=128),   # second source with a K tail (96 channels), N tail
    dict(n=2, h=16, w=16, cin=384, cout=128, k=1, split=256),   # Cout = 128: the four-wave 128-column tiles, two sources
    dict(n=1, h=32, w=32, cin=128, cout=6, k=3, f32=True),  # output conv: 6 channels padded to 8, fp32 out
    # LDS-halo conv3x3 kernel (W % 32 == 0, Cin % 64 == 0): both tile configs, borders, concat, up, residual-up
    dict(n=2, h=16, w=32, cin=64, cout=128, k=3, force_cfg=1),
    dict(n=1, h=8, w=64, cin=128, cout=256, k=3, force_cfg=0),
    dict(n=2, h=16, w=32, cin=192, cout=256, k=3, split=128, force_cfg=0),
    dict(n=1, h=8, w=16, cin=64, cout=128, k=3, up=True, force_cfg=1),
    dict(n=1, h=16, w=32, cin=64, cout=384, k=3, res_up=True, force_cfg=2),
    dict(n=1, h=32, w=32, cin=128, cout=128, k=3, prologue=True, force_cfg=1),
    dict(n=2, h=8, w=32, cin=64, cout=256, k=3, prologue=True, force_cfg=0),
    dict(n=2, h=8, w=32, cin=128, cout=128, k=3, prologue=True, force_cfg=2),
    # weights-direct kernel (csrc/conv_wd.hip): config 4 = 8x32 px x 256 channels on 32x32x16 MFMA, 6 = the same tile on 16x16x32
    dict(n=1, h=8, w=64, cin=128, cout=256, k=3, force_cfg=4),
    dict(n=2, h=16, w=32, cin=192, cout=256, k=3, split=128, force_cfg=4),
    dict(n=2, h=24, w=32, cin=64, cout=512, k=3, prologue=True, force_cfg=4),
    dict(n=1, h=8, w=32, cin=64, cout=256, k=3, up=True, force_cfg=4),
    dict(n=1, h=16, w=64, cin=128, cout=256, k=3, res_up=True, prologue=True, force_cfg=4),
    dict(n=1, h=8, w=64, cin=128, cout=256, k=3, force_cfg=6),             # config 6: the 256-channel tile on v_mfma_f32_16x16x32
    dict(n=2, h=16, w=32, cin=192, cout=256, k=3, split=128, force_cfg=6),
    dict(n=2, h=24, w=32, cin=64, cout=512, k=3, prologue=True, force_cfg=6),
    dict(n=1, h=16, w=64, cin=128, cout=256, k=3, res_up=True, prologue=True, force_cfg=6),
    dict(n=1, h=8, w=64, cin=128, cout=128, k=3, force_cfg=7),             # config 7: 128-channel tile, 4 waves, 32-channel chunks
    dict(n=2, h=16, w=32, cin=192, cout=128, k=3, split=128, force_cfg=7),
    dict(n=2, h=24, w=32, cin=64, cout=384, k=3, prologue=True, force_cfg=7),
    dict(n=1, h=8, w=32, cin=64, cout=128, k=3, up=True, force_cfg=7),
    dict(n=1, h=16, w=64, cin=128, cout=128, k=3, res_up=True, prologue=True, force_cfg=7),
    dict(n=1, h=8, w=64, cin=128, cout=320, k=3, force_cfg=7),             # config 7 with a Cout tail: 320 = 2.5 tiles, two dead waves in the last one
    dict(n=2, h=16, w=32, cin=192, cout=192, k=3, split=128, prologue=True, force_cfg=7),
    dict(n=1, h=16, w=64, cin=128, cout=448, k=3, res_up=True, prologue=True, force_cfg=7),
    dict(n=2, h=16, w=64, cin=8, cout=128, k=3, force_cfg=8),             # config 8: first convolution (3 -> 8 padded input channels), K = 72 in 3 MFMA steps
    dict(n=1, h=24, w=32, cin=16, cout=320, k=3, force_cfg=8),            # Cout tail; K = 144: 5 steps, the last one half zero weights
    dict(n=1, h=8, w=96, cin=24, cout=192, k=3, force_cfg=8),             # yfcc's 19 -> 24 channels (K = 216: 7 steps; a step straddles two taps)
    dict(n=3, h=16, w=32, cin=32, cout=64, k=3, force_cfg=8),             # 32 channels: 9 full steps; one half-dead tile
    dict(n=8, h=128, w=128, cin=8, cout=128, k=3),                        # picked by itself at bench size
    dict(n=8, h=64, w=64, cin=320, cout=320, k=3, prologue=True),          # StableDiffusion level 0 (batch 8, 64x64 latents): picks the tail-tile config by itself
    dict(n=8, h=32, w=32, cin=512, cout=512, k=3, prologue=True),          # 32x32 maps at batch 8: 128-channel tiles with split-K 2 (raw slabs + reduce)
    dict(n=8, h=32, w=32, cin=1024, cout=512, k=3, split=512),             # split-K 4 over a two-source K
    dict(n=8, h=32, w=32, cin=512, cout=512, k=3, res_up=True),            # the reduce kernel's up-sampled residual
    dict(n=8, h=64, w=64, cin=64, cout=256, k=3, prologue=True),           # enough tiles for the halo kernel by itself
    dict(n=8, h=128, w=64, cin=64, cout=6, k=3, f32=True, prologue=True),  # last conv of the UNet: halo config 3 (<= 32 output channels), fp32 out
    dict(n=8, h=128, w=64, cin=128, cout=24, k=3),                         # config 3, 16-bit out with a 16-bit residual
    dict(n=2, h=16, w=16, cin=512, cout=128, k=3),                        # few tiles, long K: split-K slabs + reduce
    dict(n=1, h=8, w=8, cin=1024, cout=256, k=3, res_up=False),
    # small / odd-width maps with Cin % 128 == 0: the weights-direct GEMM's conv mode (csrc/gemm_wd.hip CONV), split-K, two sources, row tails
    dict(n=8, h=16, w=16, cin=1024, cout=1024, k=3),
    dict(n=8, h=8, w=8, cin=2048, cout=1024, k=3, split=1024),
    dict(n=3, h=12, w=20, cin=128, cout=256, k=3),                         # M = 720: a partly filled row tile; short K (no split)
    dict(n=4, h=16, w=16, cin=640, cout=320, k=3),                         # StableDiffusion: 128-column tiles with an N tail
    dict(n=32, h=4, w=4, cin=256, cout=128, k=3, f32=True),                # 4x4 maps (every pixel touches the border), fp32 out
    dict(n=8, h=16, w=16, cin=1536, cout=1024, k=3, split=1024),           # second source 512 channels
    dict(n=8, h=16, w=16, cin=1024, cout=1024, k=3, res_up=True),          # up ResBlock tail on a small map: the reduce kernel adds nearest-up(skip)
])
def test_igemm_conv(case, dtype):
    from perceptor_amd.engine import ops
    from perceptor_amd import _hip
    from perceptor_amd._hip import dtype_code
    dev = _dev()
    # small unit shapes would otherwise be routed to the generic kernel (grid-fill heuristic): force the halo tile configs
    _hip.lib().pmi_set_option(1, case.get("force_cfg", -1))
    g = torch.Generator().manual_seed(0)
    n, h, w, cin, cout, k = (case[z] for z in ("n", "h", "w", "cin", "cout", "k"))
    x = _r(torch.randn(n, cin, h, w, generator=g), dtype)
    wt = _r(torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5, dtype)
    b = torch.randn(cout, generator=g) * 0.1
    up, stride = case.get("up", False), case.get("stride", 1)
    pro = None
    if case.get("prologue"):
        pa, pb = 1 + 0.3 * torch.randn(n, cin, generator=g), 0.3 * torch.randn(n, cin, generator=g)
        pro = (pa.to(dev), pb.to(dev), 2)
        x_conv = _r(F.silu(x * pa[:, :, None, None] + pb[:, :, None, None]), dtype)   # the kernel rounds the activated patch to 16 bit
    else:
        x_conv = x
    xin = F.interpolate(x_conv, scale_factor=2, mode="nearest") if up else x_conv
    ref = F.conv2d(xin, wt, b, stride=stride, padding=k // 2)
    res = _r(torch.randn_like(ref), dtype)
    res_in = res
    if case.get("res_up"):
        res_in = _r(torch.randn(n, cout, h // 2, w // 2, generator=g), dtype)
        res = F.interpolate(res_in, scale_factor=2, mode="nearest")
    ref = F.relu(ref) + res
    dt = dtype_code(dtype)
    lin = ops.PackedLinear(wt, b, dt, dev)
    xs = _nhwc(x, dtype).to(dev)
    a0, a1 = xs, None
    if "split" in case:
        a0, a1 = xs[..., :case["split"]].contiguous(), xs[..., case["split"]:].contiguous()
    out = ops.igemm(a0, lin, a1=a1, prologue=pro, residual=_nhwc(res_in, dtype).to(dev) if not case.get("f32") else None,   # (the residual case keeps cout=24 on the generic kernel)
                    act=1, up=up, stride=stride, res_up=case.get("res_up", False), out_f32=case.get("f32", False))
    if case.get("f32"):
        ref = ref - res
        got = out[..., :cout].permute(0, 3, 1, 2).cpu()
        # with the fused prologue a few activations round to the neighbouring 16-bit value (fast exp/rcp in the kernel vs torch's silu):
        # tolerance is then a fraction of an input ulp instead of fp32 summation noise
        tol = (0.25 * ULP[dtype] if case.get("prologue") else 2e-5) * (float(ref.abs().max()) + 1)
        assert float((got - ref).abs().max()) <= tol
    else:
        _check(out[..., :cout].permute(0, 3, 1, 2).cpu(), ref, dtype)
    _hip.lib().pmi_set_option(1, -1)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", [dict(n=8, h=16, w=16, cin=1280, cout=1280), dict(n=8, h=8, w=8, cin=2560, cout=1280, split=1280),
                                  dict(n=2, h=8, w=8, cin=128, cout=256)])      # the last one: short K, no split-K -> the generic kernel adds the bias
def test_small_map_conv_with_per_sample_bias(case, dtype):
    """StableDiffusion's ResBlock conv1 + timestep projection (openaimodel.py `h + emb_out[..., None, None]`) on 16x16 / 8x8 maps: the
    weights-direct GEMM's conv mode with split-K, the per-sample bias added by the reduce kernel; vs fp32 torch on pre-rounded operands."""
    from perceptor_amd.engine import ops
    from perceptor_amd._hip import dtype_code
    dev = _dev()
    g = torch.Generator().manual_seed(3)
    n, h, w, cin, cout = (case[z] for z in ("n", "h", "w", "cin", "cout"))
    x = _r(torch.randn(n, cin, h, w, generator=g), dtype)
    wt = _r(torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5, dtype)
    b, nb = torch.randn(cout, generator=g) * 0.1, torch.randn(n, cout, generator=g)
    ref = F.silu(F.conv2d(x, wt, b, padding=1) + nb[:, :, None, None])
    lin = ops.PackedLinear(wt, b, dtype_code(dtype), dev, sources=(case["split"], cin - case["split"]) if "split" in case else None)
    xs = _nhwc(x, dtype).to(dev)
    a0, a1 = (xs[..., :case["split"]].contiguous(), xs[..., case["split"]:].contiguous()) if "split" in case else (xs, None)
    out = ops.igemm(a0, lin, a1=a1, nbias=nb.to(dev), act=2)
    _check(out[..., :cout].permute(0, 3, 1, 2).cpu(), ref, dtype)


@pytest.mark.parametrize("dtype", DTYPES)
def test_igemm_linear_ragged_and_nbias(dtype):
    from perceptor_amd.engine import ops
    from perceptor_amd._hip import dtype_code
    dev = _dev()
    g = torch.Generator().manual_seed(1)
    m, kdim, nout = 200, 136, 132
    x = _r(torch.randn(m, kdim, generator=g), dtype)
    wt = _r(torch.randn(nout, kdim, generator=g) / kdim ** 0.5, dtype)
    b = torch.randn(nout, generator=g)
    nb = torch.randn(4, nout, generator=g)
    ref = F.silu(x @ wt.T + b + nb.repeat_interleave(50, 0))
    lin = ops.PackedLinear(wt, b, dtype_code(dtype), dev)
    td = torch.float16 if dtype == "f16" else torch.bfloat16
    out = ops.igemm(x.to(td).to(dev).view(4, 5, 10, kdim), lin, nbias=nb.to(dev), act=2)
    _check(out.reshape(m, nout).cpu(), ref, dtype)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", [
    dict(m=2056, k=1024, n=1024, res="f32", f32=True),          # ViT-L/14 out_proj / c_proj shape: 144-row tiles, fp32 residual stream
    dict(m=2056, k=1024, n=4096, act=3),                          # c_fc with the exact-GELU epilogue, 16-bit out
    dict(m=2056, k=4096, n=1024, f32=True),                       # long K, few tiles: split-K slabs + reduce
    dict(m=300, k=256, n=256, res="16"),                          # ragged last tile, 16-bit residual
    dict(m=8192, k=512, n=1536),                                  # UNet attention qkv at 32x32
    dict(m=72, k=128, n=512, f32=True, bias=False),
    dict(m=4096, k=320, n=2560),                                  # SD level-0 GEGLU projection: K tail (320 = 2.5 chunks of 128)
    dict(m=4096, k=320, n=960, res="16"),                         # N tail: 960 = 3.75 tiles of 256, the last tile's two dead waves store nothing
    dict(m=1000, k=640, n=1920, act=3),                           # both tails, ragged rows
    dict(m=2048, k=2560, n=640, res="f32"),                       # fp32 residual, 16-bit out (the transformer blocks' last feed-forward GEMM), N tail
    dict(m=333, k=1280, n=1184, f32=True, res="f32"),             # N = 37 blocks of 32, fp32 out
    dict(m=5000, k=256, n=128, res="16"),                         # N = 128: four-wave 128-column tiles (256 threads, two workgroups per CU)
    dict(m=700, k=384, n=128, act=3, f32=True),
])
def test_gemm_wd(case, dtype):
    """Weights-direct GEMM (csrc/gemm_wd.hip) against fp32 torch on operands pre-rounded to the compute type."""
    from perceptor_amd.engine import ops
    from perceptor_amd._hip import IgemmArgs, dtype_code
    dev = _dev()
    g = torch.Generator().manual_seed(3)
    m, k, n = case["m"], case["k"], case["n"]
    x = _r(torch.randn(m, k, generator=g), dtype)
    wt = _r(torch.randn(n, k, generator=g) / k ** 0.5, dtype)
    b = torch.randn(n, generator=g) * 0.1 if case.get("bias", True) else None
    ref = x @ wt.T + (b if b is not None else 0)
    act = case.get("act", 0)
    if act == 3:
        ref = F.gelu(ref)
    td = torch.float16 if dtype == "f16" else torch.bfloat16
    res = None
    if case.get("res") == "f32":
        res = torch.randn(m, n, generator=g)
        ref = ref + res
        res = res.to(dev)
    elif case.get("res") == "16":
        res = _r(torch.randn(m, n, generator=g), dtype)
        ref = ref + res
        res = res.to(td).to(dev)
    lin = ops.PackedLinear(wt, b, dtype_code(dtype), dev)
    import ctypes as C
    from perceptor_amd import _hip
    out = ops.igemm(x.to(td).to(dev), lin, residual=res, act=act, out_f32=case.get("f32", False))
    a = IgemmArgs()          # the call above must have gone to the weights-direct kernel
    a.taps, a.stride, a.M, a.N, a.K, a.C0, a.batch, a.Bf = 1, 1, m, n, k, k, 1, 1
    assert _hip.lib().pmi_gemm_wd_eligible(C.byref(a)) == 1
    if case.get("f32"):
        assert float((out.cpu() - ref).abs().max()) <= 3e-5 * (float(ref.abs().max()) + 1) * (k / 1024) ** 0.5 + 1e-6
    else:
        _check(out.cpu(), ref, dtype)


@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_wd_k_tail_does_not_read_the_next_row(dtype):
    """K = 320 is 2.5 staging chunks of 128; the tail half-chunk must read zeros, not the first channels of the next row (another
    sample): a NaN / Inf planted there would survive the multiplication by the zero-padded weights.  Rows are views into a wider buffer
    whose columns K..K+63 hold NaN -- exactly the bytes the unmasked tail read."""
    from perceptor_amd.engine import ops
    from perceptor_amd._hip import dtype_code
    dev = _dev()
    g = torch.Generator().manual_seed(11)
    m, k, n, ld = 1024, 320, 960, 384
    td = torch.float16 if dtype == "f16" else torch.bfloat16
    x = _r(torch.randn(m, k, generator=g), dtype)
    wt = _r(torch.randn(n, k, generator=g) / k ** 0.5, dtype)
    buf = torch.full((m, ld), float("nan"), dtype=td, device=dev)
    buf[:, :k] = x.to(td).to(dev)
    lin = ops.PackedLinear(wt, None, dtype_code(dtype), dev)
    out = ops.igemm(buf[:, :k], lin)
    assert torch.isfinite(out).all()
    _check(out.cpu(), x @ wt.T, dtype)
    # contiguous rows: the bytes behind row r are row r + 1; NaN in row r + 1 must not reach row r
    x2 = x.clone()
    x2[1::2] = float("nan")
    out2 = ops.igemm(x2.to(td).to(dev), lin)
    assert torch.isfinite(out2[0::2]).all()
    _check(out2[0::2].cpu(), (x @ wt.T)[0::2], dtype)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("m, k, f", [(4096, 320, 1280), (1000, 640, 2560), (300, 1280, 5120), (40, 320, 1280)])
def test_geglu_linear_fused_epilogue_and_fallback(m, k, f, dtype):
    """GEGLU feed-forward projection (stable_diffusion/attention.py:346-348): value * gelu(gate) from the weights-direct GEMM's epilogue
    (weights packed as 16 value | 16 gate column groups); m = 40 is below the kernel's row minimum: GEMM + the interleaved gate pass."""
    from perceptor_amd.engine import ops
    from perceptor_amd._hip import dtype_code
    dev = _dev()
    g = torch.Generator().manual_seed(5)
    x = _r(torch.randn(m, k, generator=g), dtype)
    wt = _r(torch.randn(2 * f, k, generator=g) / k ** 0.5, dtype)
    b = torch.randn(2 * f, generator=g) * 0.1
    h = x @ wt.T + b
    ref = h[:, :f] * F.gelu(h[:, f:])
    wi, bi = ops.interleave_geglu(wt, b)
    lin = ops.PackedLinear(wi, bi, dtype_code(dtype), dev)
    td = torch.float16 if dtype == "f16" else torch.bfloat16
    out = ops.geglu_linear(x.to(td).to(dev), lin)
    assert out.shape == (m, f)
    if m >= 64:
        _check(out.cpu(), ref, dtype)
    else:   # the projection is rounded to 16 bit before the gate pass: one more rounding than the fused epilogue
        assert float((out.float().cpu() - ref).abs().max()) <= 4 * ULP[dtype] * float(ref.abs().max())


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", [dict(c=64, g=32, film=True, pool=False), dict(c=256, g=32, film=False, pool=True),
                                  dict(c=96, g=1, film=True, pool=False, affine=False), dict(c=384, g=32, split=256)])
def test_group_norm(case, dtype):
    from perceptor_amd.engine import ops
    from perceptor_amd._hip import dtype_code
    dev = _dev()
    g = torch.Generator().manual_seed(2)
    n, h, w, c = 3, 16, 24, case["c"]
    x = _r(torch.randn(n, c, h, w, generator=g) * 2 + 0.5, dtype)
    affine = case.get("affine", True)
    gamma = 1 + 0.1 * torch.randn(c, generator=g) if affine else None
    beta = 0.1 * torch.randn(c, generator=g) if affine else None
    ref = F.group_norm(x, case["g"], gamma, beta, eps=1e-5)
    film = None
    if case.get("film"):
        film = torch.randn(n, 2 * c + 8, generator=g) * 0.3
        ref = ref * (1 + film[:, :c, None, None]) + film[:, c:2 * c, None, None]
    ref = F.silu(ref)
    if case.get("pool"):
        ref = F.avg_pool2d(ref, 2)
    xs = _nhwc(x, dtype).to(dev)
    x0, x1 = xs, None
    if "split" in case:
        x0, x1 = xs[..., :case["split"]].contiguous(), xs[..., case["split"]:].contiguous()
    out = ops.group_norm(x0, gamma.to(dev) if affine else None, beta.to(dev) if affine else None, case["g"], dtype_code(dtype),
                         x1=x1, film=film.to(dev) if film is not None else None, film_ld=2 * c + 8, act=2, pool=case.get("pool", False))
    _check(out.permute(0, 3, 1, 2).cpu(), ref, dtype, extra=2.0)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", [dict(t=64, heads=2, d=64, order=0), dict(t=256, heads=3, d=64, order=1),
                                  dict(t=16, heads=2, d=64, order=1), dict(t=100, heads=2, d=64, order=0),
                                  dict(t=64, heads=4, d=16, order=0), dict(t=50, heads=2, d=32, order=1),
                                  dict(t=36, heads=1, d=128, order=0)])
def test_attention(case, dtype):
    from perceptor_amd.engine import ops
    from perceptor_amd._hip import dtype_code
    dev = _dev()
    g = torch.Generator().manual_seed(3)
    n, t, heads, d = 2, case["t"], case["heads"], case["d"]
    c = heads * d
    qkv = _r(torch.randn(n, t, 3 * c, generator=g), dtype)
    if case["order"] == 0:
        q, k, v = qkv.view(n, t, heads, 3, d).permute(3, 0, 2, 1, 4)
    else:
        q, k, v = qkv.view(n, t, 3, heads, d).permute(2, 0, 3, 1, 4)
    att = torch.softmax((q @ k.transpose(-1, -2)) * d ** -0.5, dim=-1)
    ref = (att @ v).permute(0, 2, 1, 3).reshape(n, t, c)
    td = torch.float16 if dtype == "f16" else torch.bfloat16
    out = ops.attention(qkv.to(td).to(dev), heads, case["order"], dtype_code(dtype))
    _check(out.cpu(), ref, dtype, extra=4.0)   # probabilities are rounded to 16 bit before P.V


@pytest.mark.parametrize("dtype", DTYPES)
def test_pool_upsample_prep_finish(dtype):
    from perceptor_amd.engine import ops
    from perceptor_amd._hip import call, dtype_code, ptr
    dev = _dev()
    dt = dtype_code(dtype)
    g = torch.Generator().manual_seed(4)
    x = _r(torch.randn(2, 16, 12, 20, generator=g), dtype)
    xs = _nhwc(x, dtype).to(dev)
    _check(ops.avgpool2(xs, dt).permute(0, 3, 1, 2).cpu(), F.avg_pool2d(x, 2), dtype)
    _check(ops.upsample_bilinear2(xs, dt).permute(0, 3, 1, 2).cpu(),
           F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=False), dtype)
    img = torch.rand(2, 3, 8, 12, generator=g)
    planes = torch.randn(2, 16, generator=g)
    td = torch.float16 if dtype == "f16" else torch.bfloat16
    out = torch.empty(2, 8, 12, 24, dtype=td, device=dev)
    img_d, planes_d = img.to(dev), planes.to(dev)   # keep alive: ptr() of a temporary may be recycled
    call("pmi_prep_input", ptr(img_d), ptr(planes_d), 16, ptr(out), 2, 8, 12, 24, dt)
    ref = torch.cat([img * 2 - 1, planes[:, :, None, None].expand(2, 16, 8, 12), torch.zeros(2, 5, 8, 12)], 1)
    _check(out.permute(0, 3, 1, 2).cpu(), ref, dtype)
    y = torch.randn(2, 8, 12, 8, generator=g)
    o = torch.empty(2, 3, 8, 12, device=dev)
    y_d = y.to(dev)
    call("pmi_finish_output", ptr(y_d), 8, ptr(o), 2, 8, 12, 3)
    assert torch.equal(o.cpu(), y[..., :3].permute(0, 3, 1, 2))


def test_sampler_updates_match_golden():
    from conftest import golden
    from perceptor_amd._hip import call, ptr
    dev = _dev()
    g = {k: v.to(dev) for k, v in golden("sampling").items()}
    a, s = g["alphas"], g["sigmas"]
    af, sf, at, st = (z.contiguous() for z in (a[g["fi"]], s[g["fi"]], a[g["ti"]], s[g["ti"]]))
    nxt, den = torch.empty_like(g["img"]), torch.empty_like(g["img"])
    call("pmi_ddim_eps_step", ptr(g["img"]), ptr(g["eps"]), ptr(af), ptr(sf), ptr(at), ptr(st), ptr(nxt), ptr(den), 2, 3 * 16 * 16)
    assert float((nxt - g["eps_step"]).abs().max()) < 2e-6 * float(g["eps_step"].abs().max() + 1)
    assert float((den - g["eps_denoised"]).abs().max()) < 2e-6 * float(g["eps_denoised"].abs().max() + 1)
    gd = torch.empty_like(g["eps"])
    call("pmi_guided_update", ptr(g["eps"]), ptr(g["grad"]), ptr(sf), 0.5, 1e-6, ptr(gd), 2, 3 * 16 * 16)
    assert float((gd - g["eps_guided"]).abs().max()) < 1e-6
    import math
    ft, tt = g["ft"], g["tt"]
    vaf, vsf = torch.cos(ft * math.pi / 2).contiguous(), torch.sin(ft * math.pi / 2).contiguous()
    vat, vst = torch.cos(tt * math.pi / 2).contiguous(), torch.sin(tt * math.pi / 2).contiguous()
    call("pmi_ddim_v_step", ptr(g["img"]), ptr(g["eps"]), ptr(vaf), ptr(vsf), ptr(vat), ptr(vst), ptr(nxt), ptr(den), 2, 3 * 16 * 16)
    assert float((nxt - g["v_step"]).abs().max()) < 2e-6 * float(g["v_step"].abs().max() + 1)
    assert float((den - g["v_denoised"]).abs().max()) < 2e-6 * float(g["v_denoised"].abs().max() + 1)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", [dict(h=16, w=32, cin=64, cout=256, force_cfg=0), dict(h=32, w=32, cin=64, cout=128, force_cfg=1), dict(h=16, w=16, cin=64, cout=192),
                                  dict(h=16, w=64, cin=64, cout=256, force_cfg=4), dict(h=16, w=64, cin=64, cout=256, force_cfg=6),
                                  dict(h=16, w=64, cin=64, cout=128, force_cfg=7), dict(h=16, w=64, cin=64, cout=320, force_cfg=7),
                                  dict(h=16, w=64, cin=8, cout=320, force_cfg=8), dict(h=8, w=16, cin=64, cout=64, k=1)])
def test_fused_output_statistics_feed_groupnorm(case, dtype):
    """conv epilogue statistics (halo + generic kernels) -> GroupNorm coefficients == standalone statistics pass,
    also for a concat of two producers with a group that spans both."""
    from perceptor_amd.engine import ops
    from perceptor_amd._hip import dtype_code
    dev = _dev()
    dt = dtype_code(dtype)
    g = torch.Generator().manual_seed(7)
    n, h, w, cin, cout, k = 3, case["h"], case["w"], case["cin"], case["cout"], case.get("k", 3)
    x = _nhwc(torch.randn(n, cin, h, w, generator=g), dtype).to(dev)
    lin = ops.PackedLinear(torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5, torch.randn(cout, generator=g), dt, dev)
    lin2 = ops.PackedLinear(torch.randn(64, cin, k, k, generator=g) / (cin * k * k) ** 0.5, torch.randn(64, generator=g), dt, dev)
    from perceptor_amd import _hip
    _hip.lib().pmi_set_option(1, case.get("force_cfg", -1))
    y = ops.igemm(x, lin, want_stats=True)
    _hip.lib().pmi_set_option(1, -1)
    y2 = ops.igemm(x, lin2, want_stats=True)
    assert hasattr(y, "_pmi_stats") and hasattr(y2, "_pmi_stats")
    gamma, beta = (1 + 0.1 * torch.randn(cout + 64, generator=g)).to(dev), (0.1 * torch.randn(cout + 64, generator=g)).to(dev)
    a_f, b_f = ops.group_norm_coeffs(y, gamma, beta, 32, dt, x1=y2)                       # fused statistics, concat of two producers
    yc, y2c = y.clone(), y2.clone()                                                         # clones carry no statistics
    a_s, b_s = ops.group_norm_coeffs(yc, gamma, beta, 32, dt, x1=y2c)                     # standalone statistics kernel
    ref = F.group_norm(torch.cat([y, y2], -1).float().permute(0, 3, 1, 2), 32, gamma, beta, eps=1e-5)
    got = (torch.cat([y, y2], -1).float() * a_f[:, None, None, :] + b_f[:, None, None, :]).permute(0, 3, 1, 2)
    # statistics of the un-rounded fp32 outputs vs of the stored 16-bit tensor: differ by rounding noise only
    assert float((a_f - a_s).abs().max()) <= 4 * ULP[dtype] * float(a_s.abs().max())
    assert float((got - ref).abs().max()) <= 4 * ULP[dtype] * float(ref.abs().max())


def test_gemm_routing_fuzz():
    """Seeded sweep over the shape space the GEMM routing rules cut up (weights-direct 256- / 128-column tiles, N and K tails, two-source
    K, split-K, generic kernel): whichever kernel a shape lands on, the result matches fp32 torch on pre-rounded operands."""
    from perceptor_amd.engine import ops
    from perceptor_amd._hip import dtype_code
    import random
    dev = _dev()
    rnd = random.Random(11)
    g = torch.Generator().manual_seed(12)
    dtype = "bf16"
    dt = dtype_code(dtype)
    for it in range(28):
        m = rnd.choice([64, 72, 200, 513, 1000, 2056, 4099])
        n = 32 * rnd.randint(1, 40)
        two = it % 3 == 0
        k0 = 128 * rnd.randint(1, 6) if two else 32 * rnd.randint(1, 48)
        k1 = 32 * rnd.randint(1, 12) if two else 0
        k = k0 + k1
        x = _r(torch.randn(m, k, generator=g), dtype)
        wt = _r(torch.randn(n, k, generator=g) / k ** 0.5, dtype)
        b = torch.randn(n, generator=g) * 0.1
        res_kind = rnd.choice([None, "16", "f32"])
        out_f32 = rnd.choice([False, True]) if res_kind != "16" else False
        act = rnd.choice([0, 0, 2, 3])
        ref = x @ wt.T + b
        ref = F.silu(ref) if act == 2 else F.gelu(ref) if act == 3 else ref
        res = None
        if res_kind == "f32":
            res = torch.randn(m, n, generator=g)
            ref = ref + res
            res = res.to(dev)
        elif res_kind == "16":
            res = _r(torch.randn(m, n, generator=g), dtype)
            ref = ref + res
            res = res.to(torch.bfloat16).to(dev)
        lin = ops.PackedLinear(wt, b, dt, dev, sources=(k0, k1) if two else None)
        xd = x.to(torch.bfloat16).to(dev)
        a0, a1 = (xd[:, :k0].contiguous(), xd[:, k0:].contiguous()) if two else (xd, None)
        out = ops.igemm(a0, lin, a1=a1, residual=res, act=act, out_f32=out_f32)
        tag = (it, m, n, k0, k1, res_kind, out_f32, act)
        if out_f32:
            assert float((out.cpu() - ref).abs().max()) <= 3e-5 * (float(ref.abs().max()) + 1) * max(1.0, (k / 1024) ** 0.5) + 1e-6, tag
        else:
            d = (out.float().cpu() - ref).abs()
            assert float(d.max()) <= 2.5 * ULP[dtype] * (float(ref.abs().max()) + 1e-6), tag
