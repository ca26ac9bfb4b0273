#!/usr/bin/env python
"""Where does the host time of one MrCGAN post-epoch step go (cProfile over 20 steps; config 5 shape)?"""
import cProfile, pstats, os, sys
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
for _ in range(3): ph.step(*batch)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    ph.step(*batch)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(28)
