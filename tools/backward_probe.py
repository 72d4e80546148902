#!/usr/bin/env python3
"""Time the v-diffusion UNet input-gradient path (engine/vdiff.py: forward_train / backward) against the inference forward.
    python tools/backward_probe.py [--model yfcc_2] [--res 512] [--batch 8] [--dtype bf16]"""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from perceptor_amd import models
from perceptor_amd.utils.synth import seeded_noise

p = argparse.ArgumentParser()
p.add_argument("--model", default="yfcc_2"); p.add_argument("--res", type=int, default=512); p.add_argument("--batch", type=int, default=8)
p.add_argument("--dtype", default="bf16"); p.add_argument("--iters", type=int, default=5)
a = p.parse_args()
dev = "cuda:0"
m = models.VelocityDiffusion(a.model, dtype=a.dtype).to(dev)
img = (seeded_noise((a.batch, 3, a.res, a.res), 1) * 0.5 + 0.5).to(dev)
t = torch.full((a.batch,), 0.5, device=dev)
ce = seeded_noise((a.batch, 512), 2).to(dev) if m.spec["cond"] else None
probe = seeded_noise((a.batch, 3, a.res, a.res), 3).to(dev)
sd = m.model.state_dict()
eng = m.engine


def timed(fn):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        out = fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / a.iters, out


t_fwd, _ = timed(lambda: eng.forward(img, t, ce))
t_trn, (v, tape) = timed(lambda: eng.forward_train(img, t, ce))
t_bwd, g = timed(lambda: eng.backward(tape, probe, sd))
print(f"{a.model} {a.res}x{a.res} batch {a.batch} {a.dtype}: forward {t_fwd:.1f} ms, training-mode forward {t_trn:.1f} ms, "
      f"input-gradient backward {t_bwd:.1f} ms (x{t_bwd / t_fwd:.2f} of the forward); peak memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
