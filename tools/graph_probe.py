#!/usr/bin/env python3
"""Try capturing one denoising step of a bench config into a HIP graph and compare replay vs eager timing/results."""
import argparse, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from perceptor_amd import models, losses
from perceptor_amd.utils.synth import seeded_noise

p = argparse.ArgumentParser(); p.add_argument("--config", default="c1"); p.add_argument("--steps", type=int, default=20)
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
images0 = (seeded_noise((nb, 3, res, res), 1234) * 0.5 + 0.5).to(dev)
sched = model.schedule_ts(n_steps=50).to(dev) if is_v else model.schedule_indices(n_steps=50, rho=7.0)

def step(images, fi, ti):
    pred = model.predictions(images, fi, cond) if is_v else model.predictions(images, fi)
    if clip_loss is not None:
        _, g = clip_loss.loss_and_grad(pred.denoised_images, n_total=nb)
        pred = pred.guided(g, guidance_scale=0.5, clamp_value=1e-6)
    return pred.step(ti)

# eager reference
img = images0.clone()
for i in range(3):
    img = step(img, sched[i][0], sched[i][1])
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(3, 3 + a.steps):
    img = step(img, sched[i][0], sched[i][1])
torch.cuda.synchronize()
t_eager = (time.perf_counter() - t0) / a.steps
ref = img.clone()

# graph: static input buffers, one capture, replay per step
s_img = images0.clone(); s_f = sched[0][0].clone(); s_t = sched[0][1].clone()
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for i in range(2):
        out = step(s_img, s_f, s_t)
torch.cuda.current_stream().wait_stream(side)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    s_out = step(s_img, s_f, s_t)
img = images0.clone()
def run(i, img):
    s_img.copy_(img); s_f.copy_(sched[i][0]); s_t.copy_(sched[i][1])
    g.replay()
    return s_out
for i in range(3):
    img = run(i, img).clone()
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(3, 3 + a.steps):
    img = run(i, img).clone()
torch.cuda.synchronize()
t_graph = (time.perf_counter() - t0) / a.steps
err = float((img - ref).abs().max())
print(f"{a.config}: eager {t_eager * 1e3:.3f} ms/step, graph {t_graph * 1e3:.3f} ms/step, max |graph - eager| = {err:.3e}")
