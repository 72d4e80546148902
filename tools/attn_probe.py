#!/usr/bin/env python3
"""Time pmi_attn_flash against the batched-GEMM attention path on the StableDiffusion shapes and check both agree.
usage: python tools/attn_probe.py   (on the GPU box)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from perceptor_amd import _hip
from perceptor_amd.engine import ops

dt = _hip.DT_F16
torch.manual_seed(0)


def timeit(fn, n=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for n, t, c, heads, tk in ((8, 4096, 320, 8, None), (8, 1024, 640, 8, None), (8, 256, 1280, 8, None), (8, 64, 1280, 8, None),
                           (8, 4096, 320, 8, 77), (8, 1024, 640, 8, 77), (1, 4096, 512, 1, None)):
    d = c // heads
    if tk is None:
        qkv = (torch.randn(n, t, 3 * c, device="cuda") * 1.0).half()
        ops.FLASH_ENABLED = False
        ref = ops.attention(qkv, heads, 1, dt)
        t_ref = timeit(lambda: ops.attention(qkv, heads, 1, dt))
        ops.FLASH_ENABLED = True
        res = {}
        for qt in ((0, 1, 2) if d <= 64 else (0, 1)):
            _hip.lib().pmi_set_option(9, qt)
            out = ops.attention(qkv, heads, 1, dt)
            res[qt] = (timeit(lambda: ops.attention(qkv, heads, 1, dt)), float((out.float() - ref.float()).abs().max()))
        fl = 4.0 * n * heads * t * t * d
    else:
        q = torch.randn(n, t, c, device="cuda").half()
        kv = torch.randn(n, tk, 2 * c, device="cuda").half()
        ops.FLASH_ENABLED = False
        ref = ops.cross_attention(q, kv, heads, dt)
        t_ref = timeit(lambda: ops.cross_attention(q, kv, heads, dt))
        ops.FLASH_ENABLED = True
        res = {}
        for qt in ((0, 1, 2) if d <= 64 else (0, 1)):
            _hip.lib().pmi_set_option(9, qt)
            out = ops.cross_attention(q, kv, heads, dt)
            res[qt] = (timeit(lambda: ops.cross_attention(q, kv, heads, dt)), float((out.float() - ref.float()).abs().max()))
        fl = 4.0 * n * heads * t * tk * d
    _hip.lib().pmi_set_option(9, 0)
    print(f"N={n} T={t} Tk={tk} C={c} heads={heads} d={d}: gemm path {t_ref:.3f} ms; " +
          "; ".join(f"flash {'lds4' if qt == 0 else 'qt' + str(qt)} {ms:.3f} ms ({fl / ms / 1e9:.0f} TFLOP/s) maxdiff {df:.2e}" for qt, (ms, df) in res.items()), flush=True)
