#!/usr/bin/env python
"""Step time of the MrCGAN post-epoch step at BASELINE config 5 shape
(64x64x3 images, L=64, K=2, z=20, batch 100, srgan, lambda_gp 0.5, m_prj 0.2, m_enc 0.05)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'compatibility-family-learning_amd')]
import numpy as np, torch
from cfl.models.mrcgan import GanPhase
B = int(os.environ.get('B', 100)); L, zd = 64, 20
shape = (64, 64, 3)
dev = torch.device('cuda')
ph = GanPhase('srgan', shape, 'tanh', zd, L, B, dev, np.random.RandomState(0), lambda_gp=0.5, lambda_dra=0.5,
              m_enc=0.05, m_prj=0.2)
g = torch.Generator(device=dev); g.manual_seed(0)
N = int(np.prod(shape))
batch = [torch.tanh(torch.randn(B, N, device=dev, generator=g))] + \
        [0.3 * torch.randn(B, L, device=dev, generator=g) for _ in range(4)] + \
        [torch.randn(B, zd, device=dev, generator=g), torch.rand(B, 1, device=dev, generator=g)]
for _ in range(2): ph.step(*batch)
torch.cuda.synchronize(); t0 = time.perf_counter()
n = int(os.environ.get('N', 5))
for _ in range(n): ph.step(*batch)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
print('MrCGAN step B=%d: %.1f ms -> %.1f images/s' % (B, dt * 1e3, B / dt), ph.read_scalars())
# host side alone: how long does the python loop take to ENQUEUE a step (the stream is left to drain afterwards)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(n): ph.step(*batch)
host = (time.perf_counter() - t0) / n
torch.cuda.synchronize(); both = (time.perf_counter() - t0) / n
print('host enqueue %.1f ms/step, with drain %.1f ms/step' % (host * 1e3, both * 1e3))
