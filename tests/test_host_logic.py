"""CPU: host-side logic of the drop-in classes (schedules, shapes, packing, resize tables, error behaviour)
against the reference's golden vectors.  No HIP compute."""
import os

import pytest
import torch

from conftest import golden


def test_schedule_indices_and_tables_match_reference():
    from perceptor_amd import models
    g = golden("sampling")
    m = models.GuidedDiffusion("pixelart")
    assert torch.equal(m.schedule_alphas.data, g["alphas"]) and torch.equal(m.schedule_sigmas.data, g["sigmas"])
    assert torch.equal(m.schedule_indices(n_steps=50, rho=3.0), g["idx_50_r3"])
    assert torch.equal(m.schedule_indices(n_steps=250, rho=7.0), g["idx_250_r7"])
    assert torch.equal(m.schedule_indices(n_steps=20, from_index=400, to_index=0), g["idx_20_400"])
    with pytest.raises(ValueError):
        m.schedule_indices(from_index=0, to_index=10)
    with pytest.raises(ValueError):
        m.random_diffused((1, 3, 30, 32))
    with pytest.raises(ValueError):
        m.indices(torch.zeros(2, 2))
    with pytest.raises(RuntimeError):           # no CPU fallback
        m.predicted_noise(torch.zeros(1, 3, 64, 64), 10)
    assert m.shape == (3, 256, 256) and m.alphas(5).shape == (1, 1, 1, 1)


def test_schedule_ts_matches_reference():
    from perceptor_amd import models
    g = golden("sampling")
    assert torch.allclose(models.VelocityDiffusion.schedule_ts(n_steps=50), g["ts_50"], atol=1e-6)
    assert torch.allclose(models.VelocityDiffusion.schedule_ts(), g["ts_500"], atol=1e-6)
    s = torch.tensor([0.1, 0.7])
    t = models.VelocityDiffusion.sigmas_to_ts(s)
    assert torch.allclose(torch.sin(t * torch.pi / 2), s, atol=1e-6)   # reference test_convert_sigma_ts


def test_state_dict_surface_and_key_compatibility():
    from oracle import adm_unet, vdiff as ov
    from perceptor_amd.engine import adm, vdiff
    assert adm.state_dict_shapes(adm.openimages_config()) == adm_unet.state_dict_shapes(adm_unet.openimages_config())
    assert adm.state_dict_shapes(adm.pixelart_config()) == adm_unet.state_dict_shapes(adm_unet.pixelart_config())
    assert vdiff.state_dict_shapes(vdiff.yfcc2_spec()) == ov.state_dict_shapes(ov.yfcc2_spec())
    assert vdiff.state_dict_shapes(vdiff.cc12m1_spec()) == ov.state_dict_shapes(ov.cc12m1_spec())
    from perceptor_amd import models
    m = models.GuidedDiffusion("pixelart")
    sd = m.model.state_dict()
    assert len(sd) == 426 and "input_blocks.0.0.weight" in sd and sd["out.2.weight"].shape == (6, 128, 3, 3)
    m.model.load_state_dict(sd)
    with pytest.raises(RuntimeError):
        m.model.load_state_dict({"bogus": torch.zeros(1)})


def test_weight_packing_layout():
    from perceptor_amd.engine.ops import PackedLinear
    w = torch.arange(2 * 3 * 3 * 3, dtype=torch.float32).reshape(2, 3, 3, 3)
    p = PackedLinear(w, torch.tensor([1.0, 2.0]), 0, "cpu", cin_pad=8)
    assert p.w.shape == (4, 72) and p.taps == 9 and p.K == 72 and p.n_p == 4
    # k = tap * Cin_pad + c with tap = ky*3 + kx
    assert float(p.w[1, (1 * 3 + 2) * 8 + 2]) == float(w[1, 2, 1, 2])
    assert float(p.w[0, 5]) == 0.0 and p.b.tolist() == [1.0, 2.0, 0.0, 0.0]


def test_fragment_orders_of_the_weights_direct_kernels():
    """The host-side fragment packings the weights-direct kernels stream (element (n, k) lands in the lane / register the MFMA A operand
    expects): config 8's first-convolution order and the GEMM order, against the plain [n][k] packing."""
    from perceptor_amd.engine.ops import PackedLinear
    g = torch.Generator().manual_seed(0)
    lin = PackedLinear(torch.randn(64, 19, 3, 3, generator=g), torch.zeros(64), 0, "cpu", cin_pad=24)     # K = 216: 7 steps of 32, 8 zero columns
    f = lin.frag_c8()
    assert f.shape == (2, 7, 2, 4, 16, 8)                      # [N/32][step][16-channel block][k quarter][channel][8 k]
    w = torch.zeros(64, 7 * 32, dtype=lin.w.dtype)
    w[:, :lin.K] = lin.w
    for nb, s_, cb, q, r in [(0, 0, 0, 0, 0), (1, 6, 1, 3, 15), (0, 3, 1, 2, 7), (1, 6, 0, 3, 4)]:
        assert torch.equal(f[nb, s_, cb, q, r], w[nb * 32 + cb * 16 + r, 32 * s_ + 8 * q: 32 * s_ + 8 * q + 8])
    assert float(f[:, 6, :, 3].abs().max()) == 0.0            # k 216..223: past 9 taps x 24 channels
    lin = PackedLinear(torch.randn(64, 160, generator=g), None, 0, "cpu")                                   # K = 160 -> zero-padded to 256
    f = lin.frag_gemm()
    assert f.shape == (2, 2, 4, 2, 4, 16, 8)                   # [N/32][K/128][k-step][16-column block][k quarter][column][8 k]
    for nb, ch, ks, cb, q, r in [(0, 0, 0, 0, 0, 0), (1, 1, 0, 1, 3, 15), (0, 0, 3, 1, 2, 9)]:
        k0 = 128 * ch + 32 * ks + 8 * q
        assert torch.equal(f[nb, ch, ks, cb, q, r], lin.w[nb * 32 + cb * 16 + r, k0:k0 + 8])
    assert float(f[:, 1, 1:].abs().max()) == 0.0               # k >= 160


def test_resize_tables_equal_dense_oracle_operator():
    from oracle import clip_vit
    from perceptor_amd.transforms.resize import band_tables
    for i, o, method in ((512, 224, "lanczos3"), (256, 224, "lanczos3"), (128, 224, "cubic"), (160, 64, "lanczos3")):
        idx, w, idx_t, w_t = band_tables(i, o, method)
        dense = torch.zeros(o, i)
        for j in range(o):
            for k in range(idx.shape[1]):
                if idx[j, k] >= 0:
                    dense[j, idx[j, k]] += w[j, k]
        assert torch.allclose(dense, clip_vit.resize_matrix(i, o, method), atol=1e-7)
        dense_t = torch.zeros(i, o)
        for r in range(i):
            for k in range(idx_t.shape[1]):
                if idx_t[r, k] >= 0:
                    dense_t[r, idx_t[r, k]] += w_t[r, k]
        assert torch.equal(dense_t, dense.T)


def test_wrapper_errors_and_records():
    from perceptor_amd import losses, models
    from perceptor_amd.models.guided_diffusion import Predictions
    with pytest.raises(ValueError):
        models.OpenCLIP("ViT-B-32", "not-a-tag")
    with pytest.raises(RuntimeError):
        models.OpenCLIP("ViT-B-32", "laion2b_s34b_b79k")       # valid pair, but weights are not downloadable here
    with pytest.raises(KeyError):
        models.VelocityDiffusion("nope")
    c = models.CLIP("ViT-B-32", weights="synthetic")
    assert c.architecture == "ViT-B-32-quickgelu" and c.quick_gelu and c.image_size == (224, 224)
    l = losses.CLIP("ViT-L-14", weights="synthetic")
    assert l.multiplier == 0.01 and l.mul_(2.0).multiplier == 0.02
    with pytest.raises(ValueError):
        l.add_text_off_()
    z = torch.zeros(1, 3, 4, 4)
    p = Predictions(from_diffused_images=z, from_indices=torch.tensor([3]), predicted_noise=z,
                    schedule_alphas=torch.ones(10), schedule_sigmas=torch.ones(10))
    with pytest.raises(AttributeError):
        p.predicted_noise = z
    assert p.replace(predicted_noise=z + 1).predicted_noise.sum() == 48
    d = models.OpenCLIP.spherical_distance(torch.eye(3)[:1], torch.eye(3)[1:2])
    assert torch.allclose(d, torch.tensor([[torch.pi**2 / 8]]))


# ---- state-dict compatibility of the wrappers (.to() / .parameters() / .state_dict() must cover the network) ----
_TINY_ADM = dict(image_size=64, model_channels=32, num_res_blocks=1, channel_mult=(1, 2), attention_ds=(2,), num_head_channels=16,
                 use_scale_shift_norm=True, resblock_updown=True)


def test_guided_diffusion_state_dict_covers_the_unet_and_round_trips(tmp_path):
    import torch
    from perceptor_amd import models
    from perceptor_amd.engine import adm
    cfg = adm.AdmConfig(**_TINY_ADM)
    m = models.GuidedDiffusion("standard", config=cfg)
    shapes = adm.state_dict_shapes(cfg)
    sd = m.state_dict()
    assert set(sd) == {"model." + k for k in shapes} | {"schedule_alphas", "schedule_sigmas"}      # the reference wrapper's keys
    assert all(tuple(sd["model." + k].shape) == tuple(s) for k, s in shapes.items())
    assert sum(p.numel() for p in m.parameters()) == sum(v.numel() for v in sd.values())
    assert not any(p.requires_grad for p in m.parameters())                                          # frozen, as guided_diffusion.py:41
    # torch.save / load into a model with other weights reproduces every tensor
    path = tmp_path / "wrapper.pt"
    torch.save(sd, path)
    m2 = models.GuidedDiffusion("standard", config=cfg, seed=1)
    assert not torch.equal(m2.state_dict()["model.out.2.weight"], sd["model.out.2.weight"])
    m2.load_state_dict(torch.load(path, weights_only=True))
    assert all(torch.equal(v, m2.state_dict()[k]) for k, v in sd.items())
    # a reference checkpoint (UNet keys, fp16 torso convolutions as unet.py:610-616 leaves them) through checkpoint=
    ref_ckpt = {k[len("model."):]: (v.half() if v.ndim >= 3 and not k.startswith("model.out.") else v) for k, v in sd.items() if k.startswith("model.")}
    torch.save(ref_ckpt, tmp_path / "unet.pt")
    m3 = models.GuidedDiffusion("standard", config=cfg, checkpoint=str(tmp_path / "unet.pt"))
    got = m3.model.state_dict()
    assert all(torch.equal(got[k], v.float()) for k, v in ref_ckpt.items())
    with __import__("pytest").raises(RuntimeError):
        m3.load_state_dict({"model.bogus": torch.zeros(1)})


def test_velocity_diffusion_and_clip_state_dicts():
    import torch
    from perceptor_amd import models
    from perceptor_amd.engine import vdiff, vit
    spec = vdiff.make_spec("tiny", (3, 32, 32), [64, 128, 128], 2, 2, 4, 1, True)
    m = models.VelocityDiffusion("tiny", spec=spec)
    assert set(m.state_dict()) == {"model." + k for k in vdiff.state_dict_shapes(spec)}
    m2 = models.VelocityDiffusion("tiny", spec=spec, seed=3)
    m2.load_state_dict(m.state_dict())
    assert all(torch.equal(a, b) for a, b in zip(m.parameters(), m2.parameters()))
    c = models.OpenCLIP("tiny-test", "synthetic", config=(32, 8, 64, 2, 1, 32))
    assert set(c.state_dict()) == {"model.visual." + k for k in vit.vit_state_dict_shapes((32, 8, 64, 2, 1, 32))}
    assert c.device.type == "cpu" and c.engine is None


REF_BPE = "/root/reference/perceptor/models/slip/bpe_simple_vocab_16e6.txt.gz"


@pytest.mark.skipif(not os.path.exists(REF_BPE), reason="the CLIP merge list is data that is not part of this repository (build container only)")
def test_clip_tokenizer_matches_reference_ids():
    """perceptor_amd.utils.tokenizer against ids the reference's own tokenizer produced (oracle/gen_golden.py: gen_tokenizer), reading the
    merge list (data, 48 894 merges) from where it lies in the reference tree; skipped where that tree is absent."""
    from oracle.gen_golden import TOKENIZER_PROMPTS
    from perceptor_amd.utils.tokenizer import ClipTokenizer
    g = golden("clip_tokenizer")
    tk = ClipTokenizer(REF_BPE)
    assert tk.vocab_size == 49408 and (tk.sot, tk.eot) == (49406, 49407)
    for i, p in enumerate(TOKENIZER_PROMPTS):
        assert len(tk.encode(p)) == int(g["lengths"][i])
    ids = tk(TOKENIZER_PROMPTS)
    # the reference tokenizer cuts an over-long prompt without restoring the end token (slip/tokenizer.py:163-165); open_clip.tokenize,
    # which models/open_clip.py:102 calls, sets the last id to the end token: identical everywhere else
    long_rows = g["lengths"] + 2 > 77
    assert torch.equal(ids[~long_rows], g["ids"][~long_rows])
    assert torch.equal(ids[long_rows][:, :-1], g["ids"][long_rows][:, :-1]) and bool((ids[long_rows][:, -1] == tk.eot).all())
    hf = tk(TOKENIZER_PROMPTS[:3], pad="eot")
    assert bool((hf[2, 1:] == tk.eot).all()) and int(hf[2, 0]) == tk.sot


def test_clip_tokenizer_small_merge_list():
    from perceptor_amd.utils.tokenizer import ClipTokenizer
    tk = ClipTokenizer(merges=[("a", "b"), ("ab", "c</w>"), ("c", "a")])
    base = 512
    assert tk.encode("abc") == [base + 1]                       # a b c</w> -> ab c</w> -> abc</w>
    assert tk.encode("cab") == [base + 2, tk.ids["b</w>"]]      # (a, b) outranks (c, a) but "b</w>" is not "b": only c a merges
    assert tk("abc", context_length=4).tolist() == [[tk.sot, base + 1, tk.eot, 0]]
    with pytest.raises(FileNotFoundError):
        ClipTokenizer("/nonexistent/bpe.txt.gz")


def test_sd_parameter_inventory_matches_published_figures():
    """The SD-v1 UNet has 686 state-dict tensors and 859 520 964 parameters, the VAE decoder (+ post_quant_conv) 49 490 199 (published
    figures of the CompVis/runwayml v1 checkpoints): pins the restated architecture's shape inventory, engine and oracle independently."""
    from oracle import sd as osd
    from perceptor_amd.engine import sd
    count = lambda S: sum(int(torch.Size(s).numel()) for s in S.values())
    for S in (osd.unet_state_dict_shapes(osd.SD_V1), sd.unet_state_dict_shapes(sd.SD_V1)):
        assert len(S) == 686 and count(S) == 859_520_964
    assert osd.unet_state_dict_shapes(osd.SD_V1) == sd.unet_state_dict_shapes(sd.SD_V1)
    for S in (osd.vae_decoder_state_dict_shapes(osd.VAE_V1), sd.vae_decoder_state_dict_shapes(sd.VAE_V1)):
        assert count(S) == 49_490_199
    a, s = osd.schedule()
    assert a.shape == (1000,) and abs(float(a[0]) ** 2 - (1 - 0.00085)) < 1e-6 and abs(float(a[-1]) ** 2 - 0.0046602) < 1e-5
    assert torch.allclose(a ** 2 + s ** 2, torch.ones(1000), atol=1e-6)


def test_sd_schedule_indices_match_reference_method_on_sd_tables():
    """StableDiffusion.schedule_indices against the reference's schedule_indices body run on the scaled-linear tables (gen_sd_schedule)."""
    from perceptor_amd import models
    g = golden("sd_schedule")
    from perceptor_amd.engine import sd
    m = models.StableDiffusion(config=sd.SdConfig(block_out=(32, 64), cross_attn=(True, False), heads=2, context_dim=32, layers_per_block=1),
                               vae_config=sd.VaeConfig(block_out=(32, 64), layers_per_block=1), text_config=(16, 96, 32, 1, 1, 32))
    assert torch.equal(m.schedule_indices(n_steps=50).cpu(), g["idx_50"])
    assert torch.equal(m.schedule_indices().cpu(), g["idx_500"])
    assert torch.equal(m.schedule_indices(n_steps=20, from_index=500, to_index=20).cpu(), g["idx_20_500_20"])
