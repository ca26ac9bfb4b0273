#!/usr/bin/env python
"""Weight-gradient time (GEMM + split-K sums + weight-norm finalisation) of single 3x3 stride-1 layers at the config-5
generator / discriminator shapes; `dx` adds the input gradient for comparison."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'compatibility-family-learning_amd')]
import torch
from cfl import hipabi as H
from cfl import hipgan as G
shapes = [(300, 8, 8, 512, 1024), (300, 16, 16, 256, 512), (300, 4, 4, 64, 2048), (300, 32, 32, 32, 32), (300, 16, 16, 64, 64), (300, 8, 8, 128, 128), (300, 4, 4, 256, 256)]
for (B, Hh, W, Ci, Co) in shapes:
    conv = H.make_conv(B, Hh, W, Ci, Co, 3, 3, 1, None)
    ws = H.conv_workspace(conv, 'cuda')
    x = torch.randn(B, Hh, W, Ci, device='cuda'); V = torch.randn(3, 3, Ci, Co, device='cuda') * 0.05
    g = torch.ones(Co, device='cuda')
    dy = torch.randn(B, Hh, W, Co, device='cuda')
    dV, dg, db = torch.empty_like(V), torch.empty_like(g), torch.empty_like(g)
    dx = torch.empty_like(x)
    gf = 2.0 * B * Hh * W * Co * 9 * Ci / 1e9
    for what in ('dw', 'dx'):
        kw = dict(dV=dV, dg=dg, db=db) if what == 'dw' else dict(dx=dx)
        for _ in range(3): G.conv_bwd(conv, x, V, g, None, dy, ws, **kw)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        n = 20
        for _ in range(n): G.conv_bwd(conv, x, V, g, None, dy, ws, **kw)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
        print('%s %dx%dx%dx%d->%d: %.3f ms  %.1f TF/s' % (what, B, Hh, W, Ci, Co, dt * 1e3, gf / dt / 1e3))
