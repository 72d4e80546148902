#!/usr/bin/env python3
"""Per-shape table of every pmi_igemm launch in one StableDiffusion CFG step (bench config c4), HIP events around each launch."""
import collections, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from perceptor_amd import models
from perceptor_amd.engine import ops
from perceptor_amd.utils.synth import seeded_noise

steps = 3
dev = torch.device("cuda:0")
m = models.StableDiffusion().to(dev)
ids = torch.full((2, 77), 49407, dtype=torch.int64); ids[:, 0] = 49406; ids[1, 1:5] = torch.tensor([1125, 539, 320, 2368])
neu, pos = m.conditioning(token_ids=ids[:1]), m.conditioning(token_ids=ids[1:])
lat = seeded_noise((4, 4, 64, 64), 1234).to(dev)


def step(lat, fi, ti):
    un, po = m.predictions_pair(lat, fi, neu, pos)
    return un.classifier_free_guidance(po, 7.0).step(ti)


lat = step(lat, 999, 980)
torch.cuda.synchronize()
ops.GEMM_TRACE = []
for i in range(steps):
    lat = step(lat, 980 - 20 * i, 960 - 20 * i)
torch.cuda.synchronize()
tr, ops.GEMM_TRACE = ops.GEMM_TRACE, None
agg = collections.OrderedDict()
for desc, fl, e0, e1 in tr:
    d = agg.setdefault(desc, [0, 0.0, 0.0])
    d[0] += 1; d[1] += e0.elapsed_time(e1); d[2] += fl
tot = sum(v[1] for v in agg.values()) / steps
print(f"total GEMM-launch time per step {tot:.2f} ms")
for desc, (n, ms, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{ms / steps:8.3f} ms/step  x{n // steps:4d}  {ms / n * 1e3:8.1f} us  {fl / ms / 1e9:7.1f} TFLOP/s  {desc}")

# ---- the pieces outside the per-step loop: VAE decode / encode at 512x512, the text encoder (once per prompt) ----
def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


z = lat[:1]
img = m.decode(z)
print(f"VAE decode 1x4x64x64 -> 1x3x512x512: {timeit(lambda: m.decode(z)):.2f} ms;  batch 4: {timeit(lambda: m.decode(lat)):.2f} ms")
print(f"VAE encode 1x3x512x512 -> latents: {timeit(lambda: m.latents(img)):.2f} ms")
print(f"text encoder (ViT-L/14 text, 2 prompts x 77 tokens): {timeit(lambda: m.token_encodings(ids)):.2f} ms")
