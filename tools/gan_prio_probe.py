#!/usr/bin/env python
"""MrCGAN step with its critical chain (G forward -> D forward -> g-loss backward through D -> G backward) on a HIGH-priority
stream against the default stream, alternating (round 5 re-test of round 4's experiment: the gradient-penalty chain now starts
at the top of the step)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'compatibility-family-learning_amd')]
import numpy as np, torch
from cfl.models.mrcgan import GanPhase
B, L, zd = 100, 64, 20
shape = (64, 64, 3)
dev = torch.device('cuda')
ph = GanPhase('srgan', shape, 'tanh', zd, L, B, dev, np.random.RandomState(0), lambda_gp=0.5, lambda_dra=0.5, m_enc=0.05, m_prj=0.2)
g = torch.Generator(device=dev); g.manual_seed(0)
N = int(np.prod(shape))
batch = [torch.tanh(torch.randn(B, N, device=dev, generator=g))] + \
        [0.3 * torch.randn(B, L, device=dev, generator=g) for _ in range(4)] + \
        [torch.randn(B, zd, device=dev, generator=g), torch.rand(B, 1, device=dev, generator=g)]
print('priority range', torch.cuda.Stream.priority_range())
hi = torch.cuda.Stream(device=dev, priority=-1)


def run(stream, n=20):
    ctx = torch.cuda.stream(stream) if stream is not None else torch.cuda.stream(torch.cuda.default_stream())
    with ctx:
        for _ in range(3):
            ph.step(*batch)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            ph.step(*batch)
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for i in range(3):
    print('default stream %.2f ms   high-priority stream %.2f ms' % (run(None), run(hi)), flush=True)
