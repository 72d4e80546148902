"""GPU: the ``checkpoint=`` ingestion path and state-dict round trips of the three wrappers (VERDICT r1 test gap f).

A real GuidedDiffusion checkpoint is a ``torch.save``d dict with the reference's key names in which ``convert_to_fp16()`` left the torso
convolutions as fp16 tensors and everything else fp32 (unet.py:610-616; guided_diffusion.py:25-41 loads it).  The file written here has
exactly that layout (dtypes included), with the values of tests/golden/adm_tiny_a_fp16w.npz, whose outputs the REFERENCE computed.
"""
import pytest
import torch

from conftest import fp16_torso_state_dict, golden

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_guided_diffusion_checkpoint_with_fp16_torso(tmp_path):
    from perceptor_amd import models
    from perceptor_amd.engine import adm
    from test_gpu_adm import TINY
    g = golden("adm_tiny_a_fp16w")
    cfg = adm.AdmConfig(**TINY["a"])
    sd = fp16_torso_state_dict(adm.state_dict_shapes(cfg), g)
    fp16_keys = set(g["rounded_keys"].tolist())
    ckpt = {k: (v.half() if k in fp16_keys else v) for k, v in sd.items()}             # the on-disk dtypes of a real checkpoint
    assert any(v.dtype == torch.float16 for v in ckpt.values()) and any(v.dtype == torch.float32 for v in ckpt.values())
    path = tmp_path / "tiny_a.pt"
    torch.save(ckpt, path)
    images, t = ((g["x"] + 1) / 2).to(DEV), g["t"].to(DEV)
    ref = g["y"][:, :3]
    scale = float(g["y"].abs().max())
    for dtype, bound in (("precise", None), ("f16", 4e-3 * scale), ("bf16", 2.5e-2 * scale)):
        m = models.GuidedDiffusion("standard", checkpoint=str(path), dtype=dtype, config=cfg).to(DEV)
        err = float((m.predicted_noise(images, t).cpu() - ref).abs().max())
        print(f"[parity] checkpoint= (fp16 torso) {dtype}: max|err|={err:.3e} (scale {scale:.3f})")
        assert err < (1e-3 if bound is None else bound)
    # state_dict() -> file -> load_state_dict() into a model built from other weights: same bits out
    other = models.GuidedDiffusion("standard", dtype="f16", config=cfg, seed=5).to(DEV)
    before = other.predicted_noise(images, t)
    torch.save(m.state_dict(), tmp_path / "sd.pt")
    m16 = models.GuidedDiffusion("standard", checkpoint=str(path), dtype="f16", config=cfg).to(DEV)
    want = m16.predicted_noise(images, t)
    other.load_state_dict(torch.load(tmp_path / "sd.pt", weights_only=True))
    after = other.predicted_noise(images, t)
    assert not torch.equal(before, want) and torch.equal(after, want)
    with pytest.raises(RuntimeError):
        bad = dict(ckpt)
        bad.pop(next(iter(bad)))
        torch.save(bad, tmp_path / "bad.pt")
        models.GuidedDiffusion("standard", checkpoint=str(tmp_path / "bad.pt"), config=cfg)


def test_velocity_diffusion_and_clip_checkpoints(tmp_path):
    from perceptor_amd import models
    g = golden("vdiff_cc12m_1_64")
    m = models.VelocityDiffusion("cc12m_1_cfg", dtype="f16").to(DEV)
    torch.save({k: v.cpu() for k, v in m.model.state_dict().items()}, tmp_path / "v.pt")
    m2 = models.VelocityDiffusion("cc12m_1_cfg", checkpoint=str(tmp_path / "v.pt"), dtype="f16").to(DEV)
    x = ((g["x"] + 1) / 2).to(DEV)
    cond = g["clip_embed"].to(DEV) if "clip_embed" in g else None
    args = (x, g["t"].to(DEV)) + ((cond,) if cond is not None else ())
    assert torch.equal(m.velocities(*args), m2.velocities(*args))
    gc = golden("clip_vit_tiny")
    cfg = (32, 8, 64, 2, 1, 32)
    c = models.OpenCLIP("tiny", "synthetic", quick_gelu=True, config=cfg).to(DEV)
    # an open_clip checkpoint holds the whole model: the image tower under "visual." plus text-side tensors the HIP path ignores
    full = {k: v.cpu() for k, v in c.model.state_dict().items()}
    full["logit_scale"] = torch.tensor(4.6)
    full["token_embedding.weight"] = torch.zeros(8, 4)
    torch.save(full, tmp_path / "clip.pt")
    c2 = models.OpenCLIP("tiny", "synthetic", quick_gelu=True, config=cfg, checkpoint=str(tmp_path / "clip.pt"), seed=3).to(DEV)
    img = gc["img"].to(DEV)
    assert torch.equal(c.encode_images(img), c2.encode_images(img))
    e = c2.encode_images(img, normalize=False).cpu()
    assert float((e - gc["emb"]).norm() / gc["emb"].norm()) <= 1e-2
