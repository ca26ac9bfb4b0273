#!/usr/bin/env python
"""Experiment (round 5): the MrCGAN post-epoch step captured in ONE hipGraph (torch.cuda.graph) and replayed -- how much of the
13 ms step is host enqueue (~470 launches from Python, 9-11 ms of host time per step)?  Timing only: the Adam step size is
baked into the captured launches, so the replayed trajectory is not the training trajectory."""
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
for _ in range(3): ph.step(*batch)
torch.cuda.synchronize()
n = 20
t0 = time.perf_counter()
for _ in range(n): ph.step(*batch)
torch.cuda.synchronize()
eager = (time.perf_counter() - t0) / n
print('eager   %.2f ms/step' % (eager * 1e3))

def join_prep():
    for net in (ph.gen, ph.disc):
        ev = getattr(net, '_prep_event', None)
        if ev is not None:
            torch.cuda.current_stream().wait_event(ev)

join_prep(); torch.cuda.synchronize()
for net in (ph.gen, ph.disc):
    net._prep_event = None
graph = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    with torch.cuda.graph(graph, stream=s):
        ph.step(*batch)
        join_prep()
torch.cuda.synchronize()
for _ in range(3): graph.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n): graph.replay()
torch.cuda.synchronize()
rep = (time.perf_counter() - t0) / n
print('graph   %.2f ms/step (replay)' % (rep * 1e3), ph.read_scalars()['d_total_loss'])
