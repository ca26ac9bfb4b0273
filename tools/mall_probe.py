"""Diagnostic: does the projection launch run faster when its batch was read once just before (i.e. sits in the 256 MiB
Infinity Cache) than when it comes cold from HBM?  Library HIP-event profile per kernel."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'compatibility-family-learning_amd')):
    sys.path.insert(0, p)
from cfl import hipabi as H
from cfl.engine import PairEngine
D, L, K, B = 4096, 20, 3, 512
rng = np.random.RandomState(0)
params = {'outputs/W': (rng.randn(D, L) * 0.02).astype(np.float32), 'outputs/b': np.zeros(L, np.float32),
          'proto/W': (rng.randn(D, L * K) * 0.02).astype(np.float32), 'proto/b': np.zeros(L * K, np.float32)}
eng = PairEngine(D, L, K, 'pcd', weight_norm=False, has_bias=True, norm=H.make_norm(1.0 / 58.4),
                 loss=H.make_loss(pos_weight=0.25, lambda_m=0.5), lr=1e-3, device='cuda', params=params)
NB = 48
pool = [[torch.rand(B, D, device='cuda') for _ in range(4)] for _ in range(NB)]
flat = [torch.cat([t.view(-1) for t in b]) for b in pool]   # separate copies used only for touching? no: touch the batch itself
side = torch.cuda.Stream()


def run(label, touch):
    for i in range(60):
        eng.step(pool[i % NB])
    torch.cuda.synchronize()
    H.profile_enable(True)
    for i in range(300):
        b = pool[i % NB]
        if touch == 'same-stream':
            for t in b:
                t.sum()
        eng.step(b)
    torch.cuda.synchronize()
    H.profile_enable(False)
    prof = H.profile_read()
    print('%-40s ' % label + '  '.join('%s %.2f' % (k, 1e3 * ms / n) for k, (ms, n) in prof.items()))


run('cold (pool of %d MiB)' % (NB * 32), None)
run('batch summed right before the step', 'same-stream')
run('cold again', None)
