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
p.add_argument("--mixed", default="", help="single | dbl: the mixed-mode convolution (split input / output, fused prologue) through ops.conv3x3_mixed")
p.add_argument("--rounds", type=int, default=3); p.add_argument("--cfg", type=int, default=-1); p.add_argument("--dbg", type=int, default=0); p.add_argument("--stamps", type=int, default=0); p.add_argument("--res", type=int, default=0); p.add_argument("--stats", type=int, default=0)
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
    pro = (torch.ones(a.n, a.cin, device=dev), torch.zeros(a.n, a.cin, device=dev), {1: 2, 2: 0, 3: 1}[a.pro])   # --pro 1: SiLU, 2: affine only, 3: ReLU
flops = 2.0 * a.n * a.hw * a.hw * a.cin * a.cout * 9
from perceptor_amd import _hip
_hip.lib().pmi_set_option(1, a.cfg)
STAMP = None
if a.mixed:
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
    xs = ops.split_convert(x.to(torch.float16), True)
    ml = ops.MixedLinear(w, torch.zeros(a.cout), dev)
    pro = (torch.ones(a.n, a.cin, device=dev), torch.zeros(a.n, a.cin, device=dev), 2)
    res = ops.split_convert(torch.randn(a.n, a.hw, a.hw, a.cout, generator=g).to(torch.float16).to(dev), True) if a.res else None
    run = lambda: ops.conv3x3_mixed(xs, ml, operand=a.mixed, prologue=pro, residual=res)
    run(); torch.cuda.synchronize()
    for r in range(a.rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            run()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.iters
        print(f"mixed-{a.mixed} cfg={a.cfg} n={a.n} hw={a.hw} cin={a.cin} cout={a.cout} res={a.res}: {ms:.3f} ms  {flops / ms / 1e9:.1f} TFLOP/s (algorithmic)", flush=True)
for halo in ([] if a.mixed else [a.halo] if a.halo in (0, 1) else [0, 1]):
    ops.set_halo(bool(halo))
    res = torch.randn(a.n, a.hw, a.hw, a.cout, generator=g).to(td).to(dev) if a.res else None
    out = ops.igemm(x, lin, prologue=pro, residual=res, want_stats=bool(a.stats))
    torch.cuda.synchronize()
    for r in range(a.rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            ops.igemm(x, lin, prologue=pro, out=out, residual=res, want_stats=bool(a.stats))
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.iters
        print(f"halo={halo} cfg={a.cfg} dbg={a.dbg} n={a.n} hw={a.hw} cin={a.cin} cout={a.cout} pro={a.pro}: {ms:.3f} ms  {flops / ms / 1e9:.1f} TFLOP/s", flush=True)

if a.stamps:
    # needs conv3x3.hip built with -DPMI_STAMPS: per-workgroup wall_clock64 (100 MHz) at entry / first barrier / main loop end /
    # epilogue end / exit, plus HW_ID and XCC_ID
    ops.set_halo(True)
    ws = torch.zeros(1 << 20, dtype=torch.int64, device=dev)   # [0, 2^19): phase stamps, 8 per workgroup; [2^19, ...): 4 per wave
    ops.DEBUG_WS = ws
    if a.mixed:
        run()
    else:
        ops.igemm(x, lin, prologue=pro, out=out, residual=res, want_stats=bool(a.stats))
    torch.cuda.synchronize()
    ops.DEBUG_WS = None
    st = ws.cpu()[:1 << 19].view(-1, 8)
    st = st[st[:, 0] > 0]
    t = st[:, :5].double()
    t0 = t[:, 0].min()
    us = lambda v: float(v) * 0.01
    print(f"workgroups {len(st)}  kernel span {us(t[:, 4].max() - t0):.1f} us")
    for nm, k0, k1 in (("prologue", 0, 1), ("mainloop", 1, 2), ("epilogue", 2, 3), ("stats+exit", 3, 4), ("total", 0, 4)):
        d = t[:, k1] - t[:, k0]
        print(f"  {nm:10s} mean {us(d.mean()):8.2f} us  min {us(d.min()):8.2f}  max {us(d.max()):8.2f}")
    if (st[:, 6] > 0).any():
        t6, t7 = st[:, 6].double(), st[:, 7].double()
        print(f"  epilogue split: acc->LDS {us((t6 - t[:, 2]).mean()):.2f} us, write-out {us((t7 - t6).mean()):.2f} us, stats/drain {us((t[:, 3] - t7).mean()):.2f} us")
    cw = ws.cpu()[(1 << 19):(1 << 19) + len(st) * 32].view(-1, 4).double()     # per wave: s_memtime at loop start / end, wall clock at both
    cw = cw[cw[:, 0] > 0]
    if len(cw):
        dcyc, dwall = cw[:, 1] - cw[:, 0], (cw[:, 3] - cw[:, 2]) * 0.01
        print(f"  per-wave main loop: mean {float(dwall.mean()):.2f} us  min {float(dwall.min()):.2f}  max {float(dwall.max()):.2f};  shader clock in the loop {float((dcyc / dwall).median()) / 1e3:.3f} GHz")
        w8 = dwall.view(-1, 8) if len(dwall) % 8 == 0 else None
        if w8 is not None:
            print("  main loop by wave id (us): " + " ".join(f"{float(v):.1f}" for v in w8.mean(0)))
    hw = st[:, 5]
    cu = ((hw >> 32) & 0xf) * 1024 + ((hw >> 13) & 0x7) * 64 + ((hw >> 8) & 0xf) * 4 + ((hw >> 12) & 1)   # xcc, se, cu, sh (ids only used to group)
    ids = cu.unique()
    busy = torch.zeros(len(ids), dtype=torch.float64)
    gaps = []
    for i, c in enumerate(ids):
        rows = t[cu == c]
        rows = rows[rows[:, 0].argsort()]
        busy[i] = (rows[:, 4] - rows[:, 0]).sum()
        if len(rows) > 1:
            gaps.append((rows[1:, 0] - rows[:-1, 4]).mean())
    span = t[:, 4].max() - t0
    print(f"  distinct CU ids {len(ids)}  mean busy fraction {float(busy.mean() / span):.3f}  mean gap between consecutive workgroups on a CU {us(torch.tensor(gaps).mean()) if gaps else 0:.2f} us")
    print(f"  first start spread {us(t[:, 0].sort().values[min(255, len(t) - 1)] - t0):.2f} us; last-wave tail: 90th pct end {us(t[:, 4].quantile(0.9) - t0):.1f} us vs span")
