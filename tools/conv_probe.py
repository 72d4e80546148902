#!/usr/bin/env python3
"""Micro-benchmark of one conv3x3 shape through pmi_igemm (A/B: LDS-halo kernel vs generic implicit GEMM)."""
import argparse, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from perceptor_amd.engine import ops
from perceptor_amd._hip import dtype_code

p = argparse.ArgumentParser()
p.add_argument("--n", type=int, default=8); p.add_argument("--hw", type=int, default=256)
p.add_argument("--cin", type=int, default=256); p.add_argument("--cout", type=int, default=256)
p.add_argument("--iters", type=int, default=20); p.add_argument("--dtype", default="bf16")
p.add_argument("--halo", type=int, default=1); p.add_argument("--pro", type=int, default=0)
p.add_argument("--rounds", type=int, default=3); p.add_argument("--cfg", type=int, default=-1); p.add_argument("--glds", type=int, default=1); p.add_argument("--dbg", type=int, default=0); p.add_argument("--persist", type=int, default=1)
a = p.parse_args()
dev = torch.device("cuda:0")
dt = dtype_code(a.dtype)
td = torch.bfloat16 if a.dtype == "bf16" else torch.float16
g = torch.Generator().manual_seed(0)
x = torch.randn(a.n, a.hw, a.hw, a.cin, generator=g).to(td).to(dev)
w = torch.randn(a.cout, a.cin, 3, 3, generator=g) / (a.cin * 9) ** 0.5
lin = ops.PackedLinear(w, torch.zeros(a.cout), dt, dev)
pro = None
if a.pro:
    pro = (torch.ones(a.n, a.cin, device=dev), torch.zeros(a.n, a.cin, device=dev), 2)
flops = 2.0 * a.n * a.hw * a.hw * a.cin * a.cout * 9
from perceptor_amd import _hip
_hip.lib().pmi_set_option(1, a.cfg)
_hip.lib().pmi_set_option(2, a.glds)
_hip.lib().pmi_set_option(3, a.persist)
STAMP = None
for halo in ([a.halo] if a.halo in (0, 1) else [0, 1]):
    ops.set_halo(bool(halo))
    out = ops.igemm(x, lin, prologue=pro)
    torch.cuda.synchronize()
    for r in range(a.rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            ops.igemm(x, lin, prologue=pro, out=out)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.iters
        print(f"halo={halo} cfg={a.cfg} glds={a.glds} dbg={a.dbg} persist={a.persist} n={a.n} hw={a.hw} cin={a.cin} cout={a.cout} pro={a.pro}: {ms:.3f} ms  {flops / ms / 1e9:.1f} TFLOP/s", flush=True)
