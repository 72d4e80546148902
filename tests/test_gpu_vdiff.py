"""GPU parity: HIP v-diffusion engine vs oracle (tiny nets, full tensor) and vs the reference's golden
vectors (full-size yfcc_2 @128x128, cc12m_1 @64x64).  Same precision model and tolerances as test_gpu_adm.py;
these nets have no normalisation between convs (yfcc_2), so the bound is relative to max|v|."""
import pytest
import torch

from conftest import golden

pytestmark = pytest.mark.gpu

TOL_MAX = {"f16": 4e-3, "bf16": 2.5e-2}     # x max|ref|; measured on MI355X: f16 1.0-2.7e-3, bf16 0.9-1.8e-2 (the < 1e-3 ABSOLUTE contract is
TOL_L2 = {"f16": 2.5e-3, "bf16": 2e-2}      # asserted in precise mode: tests/test_gpu_precise.py)


def _compare(got, ref, dtype, tag):
    got, ref = got.float().cpu(), ref.float().cpu()
    scale = float(ref.abs().max())
    err = float((got - ref).abs().max())
    l2 = float((got - ref).norm() / ref.norm())
    print(f"[parity] {tag} {dtype}: max|err|={err:.3e} (scale {scale:.3f}), rel-L2={l2:.3e}")
    assert err <= TOL_MAX[dtype] * scale, (tag, err, scale)
    assert l2 <= TOL_L2[dtype], (tag, l2)


@pytest.mark.parametrize("dtype", ["f16", "bf16"])
@pytest.mark.parametrize("cond", [False, True])
def test_vdiff_tiny_vs_oracle(cond, dtype):
    from oracle import vdiff as ov
    from perceptor_amd.engine import vdiff
    from perceptor_amd.utils.synth import seeded_noise, synth_state_dict
    spec = vdiff.make_spec("tiny", (3, 32, 32), [64, 128, 128], 2, 2, 4, 1, cond)
    sd = synth_state_dict(vdiff.state_dict_shapes(spec), 0)
    eng = vdiff.VDiffEngine(spec, sd, "cuda:0", dtype)
    x = seeded_noise((3, 3, 32, 48), 5)
    t = torch.tensor([0.9, 0.3, 0.05])
    ce = seeded_noise((3, 512), 6) if cond else None
    v = eng.forward(((x + 1) / 2).cuda(), t.cuda(), ce.cuda() if cond else None)
    _compare(v, ov.vdiff_forward(sd, ov.tiny_spec(cond), x, t, ce), dtype, f"vdiff tiny cond={cond}")


@pytest.mark.parametrize("dtype", ["f16", "bf16"])
def test_yfcc2_full_128_vs_reference_golden(dtype):
    from perceptor_amd import models
    g = golden("vdiff_yfcc_2_128")
    m = models.VelocityDiffusion("yfcc_2", dtype=dtype).to("cuda")
    v = m.velocities(((g["x"] + 1) / 2).cuda(), g["t"].cuda())
    _compare(v[:, :, ::4, ::4], g["y_sub"], dtype, "yfcc_2@128 vs reference golden")


@pytest.mark.parametrize("dtype", ["f16", "bf16"])
def test_cc12m1_full_64_vs_reference_golden(dtype):
    from perceptor_amd import models
    g = golden("vdiff_cc12m_1_64")
    m = models.VelocityDiffusion("cc12m_1_cfg", dtype=dtype).to("cuda")
    v = m.velocities(((g["x"] + 1) / 2).cuda(), g["t"].cuda(), conditioning=g["clip_embed"][None].cuda())
    _compare(v[:, :, ::2, ::2], g["y_sub"], dtype, "cc12m_1@64 vs reference golden")
    pred = m(((g["x"] + 1) / 2).cuda(), g["t"].cuda(), conditioning=g["clip_embed"][None].cuda())
    nxt = pred.step(torch.tensor([0.6]))
    assert torch.isfinite(nxt).all() and nxt.shape == (1, 3, 64, 64)


@pytest.mark.parametrize("dtype", ["f16", "bf16"])
@pytest.mark.parametrize("name,res,gain", [("yfcc_1", 128, 1.0), ("wikiart", 64, 0.6)])
def test_yfcc1_wikiart_full_vs_reference_golden(name, res, gain, dtype):
    """yfcc_1 (481 M) and wikiart (no-norm attention with 128-channel heads, nearest upsampling, log-SNR features)."""
    from perceptor_amd import models
    g = golden(f"vdiff_{name}_{res}")
    m = models.VelocityDiffusion(name, dtype=dtype, weight_gain=gain).to("cuda")
    v = m.velocities(((g["x"] + 1) / 2).cuda(), g["t"].cuda())
    _compare(v[:, :, ::2, ::2], g["y_sub"], dtype, f"{name}@{res} vs reference golden")
