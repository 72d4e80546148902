#!/usr/bin/env python3
"""Micro-benchmark of plain GEMM shapes through pmi_igemm (generic 128x128x64 kernel)."""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from perceptor_amd.engine import ops
from perceptor_amd._hip import dtype_code

p = argparse.ArgumentParser()
p.add_argument("--shapes", default="2056x4096x1024,2056x4096x64,2056x4096x4096,2056x1024x4096,2048x4096x1024,4096x4096x1024,8192x4096x1024,2056x3072x1024,2056x1024x1024")
p.add_argument("--wd", type=int, default=1); p.add_argument("--lt", type=int, default=1); p.add_argument("--f32", type=int, default=0); p.add_argument("--iters", type=int, default=50); p.add_argument("--splitk", type=int, default=1); p.add_argument("--stamps", type=int, default=0)
a = p.parse_args()
dev = torch.device("cuda:0")
dt = dtype_code("bf16")
ops.SPLITK_ENABLED = bool(a.splitk)
ops.GEMM_WD_ENABLED = bool(a.wd)

for sh in a.shapes.split(","):
    M, N, K = map(int, sh.split("x"))
    x = torch.randn(M, K).to(torch.bfloat16).to(dev)
    lin = ops.PackedLinear(torch.randn(N, K) / K ** 0.5, torch.zeros(N), dt, dev)
    out = ops.igemm(x, lin, out_f32=bool(a.f32))
    torch.cuda.synchronize()
    best = 1e9
    for r in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            ops.igemm(x, lin, out=out)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / a.iters)
    tiles = ((M + 127) // 128) * ((N + 127) // 128)
    print(f"M={M} N={N} K={K} tiles={tiles}: {best * 1e3:8.1f} us  {2.0 * M * N * K / best / 1e9:7.1f} TFLOP/s", flush=True)
    if a.stamps:   # needs igemm.hip built with -DPMI_STAMPS
        ws = torch.zeros(1 << 18, dtype=torch.int64, device=dev)
        ops.DEBUG_WS = ws
        ops.igemm(x, lin, out=out)
        torch.cuda.synchronize()
        ops.DEBUG_WS = None
        st = ws.cpu().view(-1, 8)
        st = st[st[:, 0] > 0][:, :4].double()
        t0 = st[:, 0].min()
        us = lambda v: float(v) * 0.01
        print(f"   workgroups {len(st)} span {us(st[:, 3].max() - t0):.1f} us | prologue {us((st[:, 1] - st[:, 0]).mean()):.2f}  mainloop {us((st[:, 2] - st[:, 1]).mean()):.2f}"
              f"  epilogue {us((st[:, 3] - st[:, 2]).mean()):.2f} us | start spread: 50% {us(st[:, 0].median() - t0):.2f} 95% {us(st[:, 0].quantile(0.95) - t0):.2f} max {us(st[:, 0].max() - t0):.2f}")
