#!/usr/bin/env python3
"""Per-shape table of every pmi_igemm launch in one denoising step of a bench config (HIP events around each launch)."""
import argparse, collections, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from perceptor_amd import models, losses
from perceptor_amd.engine import ops
from perceptor_amd.utils.synth import seeded_noise

p = argparse.ArgumentParser(); p.add_argument("--config", default="c5"); p.add_argument("--steps", type=int, default=3)
a = p.parse_args()
dev = torch.device("cuda:0")
model_name, res, nb, clip_arch = bench.CONFIGS[a.config]
is_v = model_name not in ("standard", "pixelart")
model = (models.VelocityDiffusion(model_name, dtype="bf16") if is_v else models.GuidedDiffusion(model_name, dtype="bf16")).to(dev)
clip_loss = None
if clip_arch:
    clip_loss = losses.OpenCLIP(clip_arch, "synthetic", dtype="bf16").to(dev)
    clip_loss.add_encodings_(torch.nn.functional.normalize(seeded_noise((2, clip_loss.model.output_dim), 7)).to(dev))
cond = seeded_noise((1, 1, 512), 11).to(dev) if model_name.startswith("cc12m") else None
images = (seeded_noise((nb, 3, res, res), 1234) * 0.5 + 0.5).to(dev)
sched = model.schedule_ts(n_steps=50).to(dev) if is_v else model.schedule_indices(n_steps=50, rho=7.0)

def step(images, i):
    fi, ti = sched[i]
    pred = model.predictions(images, fi, cond) if is_v else model.predictions(images, fi)
    if clip_loss is not None:
        _, g = clip_loss.loss_and_grad(pred.denoised_images, n_total=nb)
        pred = pred.guided(g, guidance_scale=0.5, clamp_value=1e-6)
    return pred.step(ti)

images = step(images, 0)
torch.cuda.synchronize()
ops.GEMM_TRACE = []
for i in range(a.steps):
    images = step(images, 1 + i)
torch.cuda.synchronize()
tr, ops.GEMM_TRACE = ops.GEMM_TRACE, None
agg = collections.OrderedDict()
for desc, fl, e0, e1 in tr:
    d = agg.setdefault(desc, [0, 0.0, 0.0])
    d[0] += 1; d[1] += e0.elapsed_time(e1); d[2] += fl
tot = sum(v[1] for v in agg.values()) / a.steps
print(f"total GEMM-launch time per step {tot:.2f} ms")
for desc, (n, ms, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{ms / a.steps:8.3f} ms/step  x{n // a.steps:4d}  {ms / n * 1e3:8.1f} us  {fl / ms / 1e9:7.1f} TFLOP/s  {desc}")
