#!/usr/bin/env python
"""Does a captured hipGraph of the 4-kernel step run faster than direct launches?  (experiment)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'compatibility-family-learning_amd')]
import numpy as np, torch
from cfl import hipabi as H
from cfl.engine import PairEngine
from oracle import cfl_oracle as O
B, D, K, L = 512, 4096, 3, 20
cfg = O.EncoderCfg(D=D, L=L, K=K)
params = O.init_encoder_params(cfg, np.random.RandomState(0), np.float32)
eng = PairEngine(D, L, K, 'pcd', weight_norm=False, has_bias=True, norm=H.make_norm(1 / 58.388599),
                 loss=H.make_loss(), params=params, batch_size=B)
g = torch.Generator(device='cuda'); g.manual_seed(1)
nb = 12
pool = [tuple(torch.randn(B, D, generator=g, device='cuda').abs_() * 13 for _ in range(4)) for _ in range(nb)]
for i in range(50): eng.step(pool[i % nb])
torch.cuda.synchronize()
def timed(fn, n=2000):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n): fn(i)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
print('direct launches: %.2f us/step' % timed(lambda i: eng.step(pool[i % nb])))
s = torch.cuda.Stream()
graphs = []
with torch.cuda.stream(s):
    for b in range(nb):
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=s):
            eng.step(pool[b])
        graphs.append(gr)
torch.cuda.synchronize()
print('graph replay (one graph per pool batch): %.2f us/step' % timed(lambda i: graphs[i % nb].replay()))
# 8 steps per graph
with torch.cuda.stream(s):
    gr8 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr8, stream=s):
        for b in range(nb): eng.step(pool[b])
torch.cuda.synchronize()
print('graph replay (12 steps per graph): %.2f us/step' % (timed(lambda i: gr8.replay(), 200) / nb))
