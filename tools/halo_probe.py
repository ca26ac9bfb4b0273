#!/usr/bin/env python
"""Forward time of single 3x3 stride-1 layers (config-5 generator / discriminator shapes) on the halo kernel
(csrc/conv_halo.h); CFL_DEBUG_NOHALO=1 in the environment times the gathered GEMM instead."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'compatibility-family-learning_amd')]
import torch
from cfl import hipabi as H
shapes = [(300, 8, 8, 512, 1024), (300, 16, 16, 256, 512), (500, 32, 32, 32, 32), (500, 16, 16, 64, 64), (500, 8, 8, 128, 128),
          (500, 4, 4, 256, 256), (300, 4, 4, 64, 2048)]
for abl in [0]:
    for (B, Hh, W, Ci, Co) in shapes:
        conv = H.make_conv(B, Hh, W, Ci, Co, 3, 3, 1, None)
        ws = H.conv_workspace(conv, 'cuda')
        x = torch.randn(B, Hh, W, Ci, device='cuda'); V = torch.randn(3, 3, Ci, Co, device='cuda') * 0.05
        g = torch.ones(Co, device='cuda'); b = torch.zeros(Co, device='cuda')
        y = torch.empty(B, Hh, W, Co, device='cuda')
        for _ in range(3): H.conv2d_wn_fwd(conv, x, V, g, b, ws, y=y)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        n = 20
        for _ in range(n): H.conv2d_wn_fwd(conv, x, V, g, b, ws, y=y)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
        gf = 2.0 * B * Hh * W * Co * 9 * Ci / 1e9
        print('abl %2d fwd %dx%dx%dx%d->%d: %.3f ms  %.1f TF/s' % (abl, B, Hh, W, Ci, Co, dt * 1e3, gf / dt / 1e3))
