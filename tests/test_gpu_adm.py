"""GPU parity: HIP ADM-UNet engine vs the CPU fp32 oracle and the reference's golden vectors.

Tolerances (stated per north_star "within a stated fp32 tolerance"): weights are bf16-exact
(perceptor_amd/utils/synth.py), so the only error source is rounding activations to the 16-bit
MFMA input type through ~20-150 layers.  Bounds of the single-pass 16-bit modes, relative to max|eps|, about 1.5-2x the
measured values:   f16 (11-bit mantissa): 4e-3        bf16 (8-bit mantissa): 2.5e-2
plus a relative-L2 bound (2.5e-3 / 2e-2).  The contract's absolute 1e-3 is met and asserted by the precise mode
(tests/test_gpu_precise.py); these modes trade it for twice the MFMA throughput.
"""
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


def _engine(cfg_kw, dtype):
    from perceptor_amd.engine import adm
    from perceptor_amd.utils.synth import synth_state_dict
    cfg = adm.AdmConfig(**cfg_kw)
    sd = synth_state_dict(adm.state_dict_shapes(cfg), 0)
    return cfg, sd, adm.AdmEngine(cfg, sd, "cuda:0", dtype)


TINY = {
    "a": dict(image_size=64, model_channels=32, num_res_blocks=1, channel_mult=(1, 2, 2), attention_ds=(2, 4),
              num_head_channels=16, use_scale_shift_norm=True, resblock_updown=True),
    "b": dict(image_size=64, model_channels=32, num_res_blocks=2, channel_mult=(1, 2), attention_ds=(2,),
              num_heads=2, use_new_attention_order=True),
}


@pytest.mark.parametrize("dtype", ["f16", "bf16"])
@pytest.mark.parametrize("tag", ["a", "b"])
def test_adm_tiny_vs_golden_and_oracle(tag, dtype):
    from oracle import adm_unet
    g = golden(f"adm_tiny_{tag}")
    cfg, sd, eng = _engine(TINY[tag], dtype)
    images = ((g["x"] + 1) / 2).cuda()
    y = eng.forward(images, g["t"].cuda())
    _compare(y, g["y"], dtype, f"adm_tiny_{tag} vs reference golden")
    ocfg = adm_unet.AdmConfig(**TINY[tag])
    _compare(y, adm_unet.adm_unet_forward(sd, ocfg, g["x"], g["t"]), dtype, f"adm_tiny_{tag} vs oracle")


@pytest.mark.parametrize("dtype", ["f16", "bf16"])
def test_adm_d64_heads_ragged_batch(dtype):
    """64-channel heads (flash kernel) + batch 3 + non-square image, vs the oracle."""
    from oracle import adm_unet
    from perceptor_amd.utils.synth import seeded_noise
    kw = dict(image_size=64, model_channels=64, num_res_blocks=1, channel_mult=(1, 2), attention_ds=(1, 2),
              num_head_channels=64, use_scale_shift_norm=True, resblock_updown=True)
    cfg, sd, eng = _engine(kw, dtype)
    x = seeded_noise((3, 3, 32, 48), 77)
    t = torch.tensor([999, 0, 250])
    y = eng.forward(((x + 1) / 2).cuda(), t.cuda())
    _compare(y, adm_unet.adm_unet_forward(sd, adm_unet.AdmConfig(**kw), x, t), dtype, "adm_d64")


@pytest.mark.parametrize("dtype", ["f16", "bf16"])
def test_adm_standard_full_128(dtype):
    """The shipped 558 M-parameter 'standard' net at 128x128: strided golden slice + moments."""
    from perceptor_amd.engine import adm
    from perceptor_amd.utils.synth import synth_state_dict
    g = golden("adm_standard_128")
    cfg = adm.openimages_config()
    sd = synth_state_dict(adm.state_dict_shapes(cfg), 0)
    eng = adm.AdmEngine(cfg, sd, "cuda:0", dtype)
    y = eng.forward(((g["x"] + 1) / 2).cuda(), g["t"].cuda())
    _compare(y[:, :, ::4, ::4], g["y_sub"], dtype, "adm_standard_128 vs reference golden")
    f = y.flatten(1).double().cpu()
    mom = torch.stack([f.mean(1), f.std(1), f.norm(dim=1)], 1).float()
    assert torch.allclose(mom[:, 1:], g["y_mom"][:, 1:], rtol=2e-2), (mom, g["y_mom"])
