"""GPU: BASELINE.json configs[3] at its full per-GPU size (StableDiffusion v1 latent UNet, 512x512 = 64x64 latents, batch 4 with
classifier-free guidance = UNet batch 8, 77-token context, f16) through size-independent properties -- the CPU oracle cannot run this
size in test time (value parity of the same engine against the oracle: the full 860 M UNet at 32x32 latents, tests/test_gpu_sd.py):

  * determinism: the same inputs twice give bit-identical predicted noise and latents;
  * chain independence (the basis of the replica sharding): permuting the batch permutes the outputs bit-exactly, and a rank holding
    chains [2r, 2r+2) reproduces its slice of the 4-chain result (rounding-level: tile / split-K choices depend on the row count);
  * classifier-free guidance with scale 1 returns the conditioned prediction, with scale 0 the unconditioned one;
  * the latent algebra at full size: x = denoised * alpha + eps * sigma reproduces the input, a DDIM step to the same index is the identity;
  * the pair evaluation (one UNet pass of batch 8) agrees with two separate passes of batch 4; HIP-graph replay of a CFG step is bit-exact;
  * everything finite, the text context matters (conditioned != unconditioned).
"""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def workload():
    from perceptor_amd import models
    from perceptor_amd.utils.synth import seeded_noise
    dev = torch.device("cuda:0")
    m = models.StableDiffusion().to(dev)
    ids = torch.full((2, 77), 49407, dtype=torch.int64)
    ids[:, 0] = 49406
    ids[1, 1:9] = torch.tensor([1125, 539, 320, 2368, 525, 320, 4558, 267])
    neu, pos = m.conditioning(token_ids=ids[:1]), m.conditioning(token_ids=ids[1:])
    lat = seeded_noise((4, 4, 64, 64), 1234).to(dev)
    return m, neu, pos, lat, m.schedule_indices(n_steps=50)


def test_c4_pair_is_deterministic_chain_independent_and_matches_separate_passes(workload):
    m, neu, pos, lat, sched = workload
    fi = sched[3][0]
    un0, po0 = m.predictions_pair(lat, fi, neu, pos)
    un1, po1 = m.predictions_pair(lat, fi, neu, pos)
    assert po0.predicted_noise.shape == (4, 4, 64, 64) and bool(torch.isfinite(po0.predicted_noise).all())
    assert torch.equal(po0.predicted_noise, po1.predicted_noise) and torch.equal(un0.predicted_noise, un1.predicted_noise)
    assert float((po0.predicted_noise - un0.predicted_noise).abs().max()) > 1e-3, "the prompt does not reach the UNet"
    perm = torch.tensor([2, 0, 3, 1], device=lat.device)
    _, po_p = m.predictions_pair(lat[perm].contiguous(), fi, neu, pos)
    assert torch.equal(po_p.predicted_noise, po0.predicted_noise[perm]), "a chain's output depends on its position in the batch"
    scale = float(po0.predicted_noise.abs().max())
    sep = m.predictions(lat, fi, pos).predicted_noise                       # batch 4 alone: other tile / split-K choices, rounding-level only
    assert float((sep - po0.predicted_noise).abs().max()) <= 4e-3 * scale
    for r in range(2):
        _, po_r = m.predictions_pair(lat[2 * r:2 * r + 2].contiguous(), fi, neu, pos)
        assert float((po_r.predicted_noise - po0.predicted_noise[2 * r:2 * r + 2]).abs().max()) <= 4e-3 * scale


def test_c4_cfg_and_latent_algebra_at_full_size(workload):
    m, neu, pos, lat, sched = workload
    fi, ti = sched[3]
    un, po = m.predictions_pair(lat, fi, neu, pos)
    assert float((un.classifier_free_guidance(po, 1.0).predicted_noise - po.predicted_noise).abs().max()) <= 1e-6 * (1 + float(po.predicted_noise.abs().max()))
    assert torch.equal(un.classifier_free_guidance(po, 0.0).predicted_noise, un.predicted_noise)
    strong = un.classifier_free_guidance(po, 7.0)
    a, s = strong.from_alphas, strong.from_sigmas
    recon = strong.denoised_latents * a + strong.predicted_noise * s
    assert float((recon - lat).abs().max()) <= 2e-5 * (1 + float(lat.abs().max()))
    assert float((strong.step(fi) - lat).abs().max()) <= 2e-5 * (1 + float(lat.abs().max()))
    nxt = strong.step(ti)
    assert nxt.shape == lat.shape and bool(torch.isfinite(nxt).all())


def test_c4_graph_replay_matches_eager_and_decode_runs(workload):
    from perceptor_amd.engine.graph import GraphedStep
    m, neu, pos, lat, sched = workload
    fi, ti = sched[5]

    def step(x, f, t):
        un, po = m.predictions_pair(x, f, neu, pos)
        return un.classifier_free_guidance(po, 7.0).step(t)

    eager = step(lat, fi, ti)
    g = GraphedStep(step, lat, fi, ti)
    assert torch.equal(g(lat, fi, ti), eager)
    img = m.decode(eager[:1])
    assert img.shape == (1, 3, 512, 512) and bool(torch.isfinite(img).all())
    back = m.latents(img)
    assert back.shape == (1, 4, 64, 64) and bool(torch.isfinite(back).all())
