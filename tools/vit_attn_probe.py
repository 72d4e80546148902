#!/usr/bin/env python3
"""Timing of the ViT attention kernels alone (pmi_vit_attn_fwd / pmi_vit_attn_bwd, csrc/attn.hip) at the benchmark's CLIP shape."""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from perceptor_amd._hip import call, ptr, dtype_code

p = argparse.ArgumentParser()
p.add_argument("--n", type=int, default=8); p.add_argument("--t", type=int, default=257); p.add_argument("--heads", type=int, default=16)
p.add_argument("--iters", type=int, default=50); p.add_argument("--dtype", default="bf16")
a = p.parse_args()
dev = torch.device("cuda:0"); dt = dtype_code(a.dtype); tdt = torch.bfloat16 if a.dtype == "bf16" else torch.float16
n, t, heads = a.n, a.t, a.heads
m, width = n * t, heads * 64
g = torch.Generator().manual_seed(0)
qkv = (torch.randn(m, 3 * width, generator=g) * 0.5).to(tdt).to(dev)
da = torch.randn(m, width, generator=g).to(tdt).to(dev)
tp32 = (t + 31) // 32 * 32
aws = torch.empty((6, n * heads, tp32, 64), dtype=tdt, device=dev); lse = torch.empty((n * heads, tp32), dtype=torch.float32, device=dev)
out = torch.empty((m, width), dtype=tdt, device=dev)
bws = torch.empty((2, n * heads, tp32, 64), dtype=tdt, device=dev); delta = torch.empty((n * heads, tp32), dtype=torch.float32, device=dev)
dqkv = torch.empty((m, 3 * width), dtype=tdt, device=dev)
fwd = lambda: call("pmi_vit_attn_fwd", ptr(qkv), ptr(aws), ptr(lse), ptr(out), n, t, heads, 0.125, dt)
bwd = lambda: call("pmi_vit_attn_bwd", ptr(aws), ptr(lse), ptr(out), ptr(da), ptr(bws), ptr(delta), ptr(dqkv), n, t, heads, 0.125, dt)
for name, f in (("fwd (qkv_split + attention)", fwd), ("bwd (dO prep + dK/dV + dQ)", bwd)):
    f(); torch.cuda.synchronize()
    for r in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            f()
        e1.record(); torch.cuda.synchronize()
        print(f"{name}: {e0.elapsed_time(e1) / a.iters * 1e3:.1f} us", flush=True)
print("checksums", float(out.float().abs().sum()), float(dqkv.float().abs().sum()))
