"""GPU parity of the CLIP text tower (engine/text.py) and of OpenCLIP.encode_texts / encode_tokens.

Fixtures: the reference's in-tree ruclip CLIP.encode_text and transformers' CLIPTextModelWithProjection on the name-keyed weights
(oracle/gen_golden.py: gen_clip_text).  GEMM operands are 16-bit with an fp32 residual stream: rel-L2 <= 1e-2 in bf16 (measured 2-4e-3),
<= 2e-3 in f16 (measured 3-6e-4).
"""
import pytest
import torch

from conftest import golden

pytestmark = pytest.mark.gpu

TINY_VIT = (32, 8, 64, 2, 1, 32)


def _rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


@pytest.mark.parametrize("fixture, tag, quick", [("clip_text_ruclip_tiny", "tiny", True), ("clip_text_hf_tiny-wide_gelu", "tiny-wide", False)])
@pytest.mark.parametrize("precision", ["bf16", "fp16"])
def test_encode_tokens_through_the_class(fixture, tag, quick, precision):
    from oracle.clip_text import TEXT_CONFIGS
    from perceptor_amd import models
    g = golden(fixture)
    m = models.OpenCLIP("tiny", "synthetic", precision, quick_gelu=quick, config=TINY_VIT, text_config=TEXT_CONFIGS[tag]).to("cuda")
    pooled = m.encode_tokens(g["ids"], normalize=False).cpu()
    tol = 1e-2 if precision == "bf16" else 2e-3
    assert _rel(pooled, g["pooled"]) < tol
    pn = m.encode_tokens(g["ids"].cuda()).cpu()
    assert _rel(pn, torch.nn.functional.normalize(g["pooled"])) < tol
    if "hidden" in g:
        hidden, _ = m.text.forward(g["ids"])
        assert _rel(hidden.cpu(), g["hidden"]) < tol


@pytest.mark.parametrize("dtype, tol", [("bf16", 1e-2), ("f16", 2e-3)])
def test_vit_l14_text_tower_vs_transformers(dtype, tol):
    """The ViT-L/14 text encoder (the one StableDiffusion conditions on): all 77 hidden states and the pooled embedding."""
    from perceptor_amd.engine.text import TEXT_CONFIGS, TextEngine, text_state_dict_shapes
    from perceptor_amd.utils.synth import synth_state_dict
    g = golden("clip_text_hf_ViT-L-14_quickgelu")
    cfg = TEXT_CONFIGS["ViT-L-14"]
    eng = TextEngine(cfg, synth_state_dict(text_state_dict_shapes(cfg), 0), "cuda", dtype, quick_gelu=True)
    hidden, pooled = eng.forward(g["ids"])
    assert _rel(hidden.cpu(), g["hidden"]) < tol and _rel(pooled.cpu(), g["pooled"]) < tol
    # causal: changing tokens behind position p leaves hidden[:, :p+1] bit-identical
    ids2 = g["ids"].clone()
    ids2[:, 40:] = 7
    h2, _ = eng.forward(ids2)
    assert torch.equal(h2[:, :40], hidden[:, :40])


def test_encode_texts_strings_and_errors():
    from oracle import clip_text
    from perceptor_amd import models
    from perceptor_amd.utils.synth import synth_state_dict
    from perceptor_amd.utils.tokenizer import ClipTokenizer
    tcfg = (16, 520, 64, 2, 1, 32)
    m = models.OpenCLIP("tiny", "synthetic", quick_gelu=True, config=TINY_VIT, text_config=tcfg).to("cuda")
    with pytest.raises(FileNotFoundError):
        m.encode_texts(["a cab"])                      # no merge list given: loud, not faked
    m._tokenizer = ClipTokenizer(merges=[("a", "b"), ("ab", "c</w>"), ("c", "a")])
    ids = m.tokenize(["abc cab", "b"])
    assert ids.shape == (2, 16) and int(ids[0, 0]) == m._tokenizer.sot
    sd = synth_state_dict(clip_text.text_state_dict_shapes(tcfg), 0)
    _, want = clip_text.text_forward(sd, tcfg, ids, True)
    got = m.encode_texts(["abc cab", "b"]).cpu()
    assert _rel(got, torch.nn.functional.normalize(want)) < 1e-2
    with pytest.raises(ValueError):
        m.text.forward(torch.full((1, 17), 1, dtype=torch.int64))
    with pytest.raises(ValueError):
        m.text.forward(torch.full((1, 4), 9999, dtype=torch.int64))
    with pytest.raises(RuntimeError):
        models.OpenCLIP("tiny", "synthetic", config=TINY_VIT).to("cuda").encode_tokens(ids)     # no text_config: no text tower


def test_transformers_openai_clip_surface_vs_clipmodel_fixture():
    """models.TransformersOpenAICLIP (transformers-named state dict) against transformers' CLIPModel on the same weights: features,
    encodings, spherical distance and the image gradient through it (models/transformers_openai_clip.py:88-134, tests :140-152)."""
    from perceptor_amd import models
    g = golden("clip_hf_model_tiny")
    m = models.TransformersOpenAICLIP(config=(32, 8, 64, 2, 1, 32), text_config=(16, 96, 64, 2, 1, 32)).to("cuda")
    assert "vision_model.encoder.layers.1.self_attn.q_proj.weight" in m.state_dict() and "text_projection.weight" in m.state_dict()
    with torch.no_grad():
        ie = m.encode_images(g["img"])
        te = m.encode_token_ids(g["ids"])
    assert _rel(ie.unnormalized_encodings.cpu(), g["image_embeds"]) < 1e-2 and _rel(te.unnormalized_encodings.cpu(), g["text_embeds"]) < 1e-2
    assert _rel(ie.features.last_hidden_state.cpu(), g["image_hidden"]) < 1e-2 and _rel(ie.features.pooler_output.cpu(), g["image_pooler"]) < 1e-2
    assert _rel(te.features.last_hidden_state.cpu(), g["text_hidden"]) < 1e-2 and _rel(te.features.pooler_output.cpu(), g["text_pooler"]) < 1e-2
    assert float((ie.encodings.norm(dim=1) - 1).abs().max()) < 1e-5
    d = m.spherical_distance(te, ie)
    assert float((d.cpu() - g["distance"]).abs().max()) < 2e-2 * float(g["distance"].abs().max())
    img = g["img"].cuda().requires_grad_(True)
    with torch.enable_grad():
        ieg = m.encode_images(img)
        m.spherical_distance(te, ieg).mean().backward()
    # the gradient path returns features too, as upstream: the detached hidden state / pooled class token of the same pass
    assert _rel(ieg.features.last_hidden_state.cpu(), g["image_hidden"]) < 1e-2 and _rel(ieg.features.pooler_output.cpu(), g["image_pooler"]) < 1e-2
    assert not ieg.features.last_hidden_state.requires_grad
    cos = torch.nn.functional.cosine_similarity(img.grad.cpu().double().flatten(), g["grad"].double().flatten(), dim=0)
    assert float(cos) > 0.999 and _rel(img.grad.cpu(), g["grad"]) < 3e-2
    with pytest.raises(NotImplementedError):
        models.TransformersOpenAICLIP("M-CLIP/XLM-Roberta-Large-Vit-L-14")


def test_loss_add_texts_end_to_end():
    """losses.OpenCLIP.add_texts_ (losses/open_clip.py:60-66 -> models.OpenCLIP.encode_texts) feeding the image loss: text prompts through the
    tokenizer and the HIP text tower become the targets of the spherical loss, whose gradient reaches the images."""
    from perceptor_amd import losses
    from perceptor_amd.utils.synth import seeded_noise
    from perceptor_amd.utils.tokenizer import ClipTokenizer
    loss = losses.OpenCLIP("tiny", "synthetic", quick_gelu=True, config=TINY_VIT, text_config=(16, 520, 64, 2, 1, 32)).to("cuda")
    loss.model._tokenizer = ClipTokenizer(merges=[("a", "b"), ("ab", "c</w>"), ("c", "a")])
    loss.add_texts_(["abc cab", "b"], weights=[1.0, 0.5])
    assert loss.encodings.shape == (2, 32) and float((loss.encodings.norm(dim=1) - 1).abs().max()) < 1e-5
    want = loss.model.encode_texts(["abc cab", "b"])
    assert torch.equal(loss.encodings.data, want)
    img = (seeded_noise((2, 3, 40, 40), 52) * 0.25 + 0.5).cuda()
    val, grad = loss.loss_and_grad(img)
    assert bool(torch.isfinite(val)) and grad.shape == img.shape and float(grad.abs().max()) > 0
    x = img.clone().requires_grad_(True)
    with torch.enable_grad():
        loss(x).backward()
    cos = torch.nn.functional.cosine_similarity(x.grad.flatten(), grad.flatten(), dim=0)
    assert float(cos) > 0.9999
