#!/usr/bin/env python3
"""Numerical feasibility of Winograd F(2x2, 3x3) with 16-bit MFMA operands (CPU, fp32 reference).

direct:   conv(round16(x), round16(w)) with fp32 accumulation                         -> what the HIP kernels do today
winograd: U = G w G^T and V = B^T d B computed in fp32, ROUNDED to 16 bit (they are the MFMA operands), fp32 accumulation over
          channels, output transform in fp32
Both are compared with the fp32 convolution of the unrounded tensors."""
import torch, torch.nn.functional as F

def rnd(t, dt): return t.to(dt).float()
G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]])
Bt = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1.]])
At = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1.]])

def winograd(x, w, dt):
    n, c, h, wd = x.shape
    k = w.shape[0]
    U = rnd(torch.einsum("ij,kcjl,ml->kcim", G, w, G), dt)                       # [k, c, 4, 4]
    xp = F.pad(x, (1, 1, 1, 1))
    tiles = xp.unfold(2, 4, 2).unfold(3, 4, 2)                                      # [n, c, th, tw, 4, 4]
    V = rnd(torch.einsum("ij,nctujl,ml->nctuim", Bt, tiles, Bt), dt)
    M = torch.einsum("kcim,nctuim->nktuim", U, V)
    Y = torch.einsum("ij,nktujl,ml->nktuim", At, M, At)                            # [n, k, th, tw, 2, 2]
    return Y.permute(0, 1, 2, 4, 3, 5).reshape(n, k, h, wd)

torch.manual_seed(0)
for cin in (128, 512):
    x = torch.randn(1, cin, 32, 32) * torch.rand(1, cin, 1, 1) * 2     # per-channel scales like post-SiLU activations
    x = F.silu(x)
    w = torch.randn(64, cin, 3, 3) / (cin * 9) ** 0.5
    ref = F.conv2d(x, w, padding=1)
    for name, dt in (("f16", torch.float16), ("bf16", torch.bfloat16)):
        d = F.conv2d(rnd(x, dt), rnd(w, dt), padding=1)
        wg = winograd(rnd(x, dt), w, dt)
        e = lambda y: float((y - ref).norm() / ref.norm())
        print(f"Cin={cin} {name}: direct rel-L2 {e(d):.2e}   winograd rel-L2 {e(wg):.2e}   ratio {e(wg) / e(d):.2f}")
