"""Diagnostic: cost of small torch ops / allocations while the stream is idle vs deeply queued."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'compatibility-family-learning_amd')):
    sys.path.insert(0, p)
from cfl import hipabi as H
from cfl.engine import PairEngine
sys.path.insert(0, ROOT)
import bench

dev = torch.device('cuda')
D, B = 4096, 512
eng = PairEngine(D, 20, 3, norm=H.make_norm(1 / 58.4), params=bench.init_params(D, 20, 3), batch_size=B)
teacher = bench.teacher_of(D, dev)
pool = [bench.make_block(B, D, dev, i, teacher) for i in range(8)]


def queue(n):
    for i in range(n):
        eng.step(pool[i % 8])


def t(label, fn, n=200, depth=0):
    torch.cuda.synchronize()
    queue(depth)
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    dt = (time.perf_counter() - t0) / n
    torch.cuda.synchronize()
    print('%-46s depth %4d: %8.1f us per call' % (label, depth, dt * 1e6))


x = torch.zeros(512, device=dev)
pin = torch.empty(64, 17).pin_memory()
for depth in (0, 300):
    t('torch.empty(512)', lambda: torch.empty(512, device=dev), depth=depth)
    t('(x > 0).float().mean()', lambda: (x > 0).float().mean(), depth=depth)
    t('eng.scores(dense 512)', lambda: eng.scores(pool[0][0], pool[0][1]), depth=depth)
    t('pinned slot copy_ non_blocking', lambda: pin[3, :16].copy_(eng.scalars, non_blocking=True), depth=depth)
    t('Event record', lambda: torch.cuda.Event().record(), depth=depth)
    t('eng.step', lambda: eng.step(pool[0]), depth=depth)
