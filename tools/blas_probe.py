#!/usr/bin/env python3
"""How fast are the library GEMMs (torch -> hipBLASLt/rocBLAS) on the ViT shapes, next to pmi_igemm?"""
import torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
dev = torch.device("cuda:0")
for M, N, K in [(2056, 4096, 1024), (2056, 1024, 4096), (2056, 3072, 1024), (2056, 1024, 1024), (2056, 1024, 3072), (400, 3072, 768), (400, 768, 3072), (2097152, 128, 256)]:
    a = torch.randn(M, K, device=dev, dtype=torch.bfloat16); w = torch.randn(N, K, device=dev, dtype=torch.bfloat16)
    b = torch.randn(N, device=dev, dtype=torch.bfloat16)
    for _ in range(3): torch.nn.functional.linear(a, w, b)
    torch.cuda.synchronize()
    best = 1e9
    for r in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): torch.nn.functional.linear(a, w, b)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 50)
    print(f"torch linear M={M} N={N} K={K}: {best * 1e3:8.1f} us  {2.0 * M * N * K / best / 1e9:7.1f} TFLOP/s", flush=True)
