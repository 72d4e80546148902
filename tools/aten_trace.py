#!/usr/bin/env python3
"""torch (aten) GPU kernels launched inside one denoising step of a bench config, with the Python line that issued them: the launches that
are torch ops rather than kernels of libperceptor_hip.so.   python tools/aten_trace.py [--config c5] [--dtype bf16]"""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from torch.profiler import ProfilerActivity, profile

p = argparse.ArgumentParser()
p.add_argument("--config", default="c5"); p.add_argument("--dtype", default="bf16")
a = p.parse_args()
from perceptor_amd import losses, models
from perceptor_amd.utils.synth import seeded_noise
dev = torch.device("cuda:0")
model_name, res, nb, clip_arch = bench.CONFIGS[a.config]
model = models.GuidedDiffusion(model_name, dtype=a.dtype).to(dev)
clip_loss = None
if clip_arch:
    clip_loss = losses.OpenCLIP(clip_arch, "synthetic", dtype="bf16").to(dev)
    clip_loss.add_encodings_(torch.nn.functional.normalize(seeded_noise((2, clip_loss.model.output_dim), 7)).to(dev))
images = (seeded_noise((nb, 3, res, res), 1234) * 0.5 + 0.5).to(dev)
sched = model.schedule_indices(n_steps=50, rho=7.0)


def step(images, i):
    fi, ti = sched[i]
    pred = model.predictions(images, fi)
    if clip_loss is not None:
        _, grad = clip_loss.loss_and_grad(pred.denoised_images, n_total=nb)
        pred = pred.guided(grad, guidance_scale=0.5, clamp_value=1e-6)
    return pred.step(ti)


for i in range(2):
    images = step(images, i)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    images = step(images, 2)
    torch.cuda.synchronize()
rows = {}
for ev in prof.events():
    if not ev.name.startswith("aten::") or ev.device_time_total <= 0 and ev.self_device_time_total <= 0:
        continue
    if ev.self_device_time_total <= 0:
        continue
    src = next((s for s in ev.stack if "perceptor_amd" in s or "bench" in s or "aten_trace" in s), ev.stack[0] if ev.stack else "?")
    key = (ev.name, str(ev.input_shapes)[:80], src.strip()[:110])
    r = rows.setdefault(key, [0, 0.0])
    r[0] += 1; r[1] += ev.self_device_time_total
tot = sum(r[1] for r in rows.values())
print(f"aten ops with device time in one step: {sum(r[0] for r in rows.values())} launches, {tot / 1e3:.3f} ms")
for k, r in sorted(rows.items(), key=lambda kv: -kv[1][1]):
    print(f"{r[1]:9.1f} us {r[0]:4d} x  {k[0]:28s} {k[1]:80s} {k[2]}")
