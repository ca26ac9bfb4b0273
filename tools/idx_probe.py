"""Diagnostic: the fused step fed from dense batch blocks vs from a resident table by index (random rows, sorted
rows, a small table), per kernel (library HIP-event profile) and per step (event-free loop)."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'compatibility-family-learning_amd')):
    sys.path.insert(0, p)
from cfl import hipabi as H  # noqa: E402
from cfl.engine import PairEngine  # noqa: E402

D, L, K, B = 4096, 20, 3, 512
rng = np.random.RandomState(0)
params = {'outputs/W': (rng.randn(D, L) * 0.02).astype(np.float32), 'outputs/b': np.zeros(L, np.float32),
          'proto/W': (rng.randn(D, L * K) * 0.02).astype(np.float32), 'proto/b': np.zeros(L * K, np.float32)}
eng = PairEngine(D, L, K, 'pcd', weight_norm=False, has_bias=True, norm=H.make_norm(1.0 / 58.4),
                 loss=H.make_loss(pos_weight=0.25, lambda_m=0.5), lr=1e-3, device='cuda', params=params)
NROWS = int(sys.argv[1]) if len(sys.argv) > 1 else 40000
table = torch.rand(NROWS, D, device='cuda')
NB = 48


def streams(kind, nrows):
    out = []
    for _ in range(NB):
        idx = [rng.randint(0, nrows, size=B).astype(np.int32) for _ in range(4)]
        if kind == 'sorted':
            idx = [np.sort(i) for i in idx]
        if kind == 'contig':
            s0 = rng.randint(0, nrows - 4 * B)
            idx = [np.arange(s0 + j * B, s0 + (j + 1) * B, dtype=np.int32) for j in range(4)]
        out.append(H.IndexStreams.from_tensors([torch.tensor(i, device='cuda') for i in idx]))
    return out


def run(label, batches):
    for i in range(100):
        eng.step(batches[i % NB])
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(20):
        t0 = time.perf_counter()
        for i in range(40):
            eng.step(batches[i % NB])
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 40)
    H.profile_enable(True)
    for i in range(300):
        eng.step(batches[i % NB])
    torch.cuda.synchronize()
    H.profile_enable(False)
    prof = H.profile_read()
    print('%-34s %.2f us/step | ' % (label, 1e6 * best) +
          '  '.join('%s %.2f' % (k, 1e3 * ms / n) for k, (ms, n) in prof.items()))


dense = [[torch.rand(B, D, device='cuda') for _ in range(4)] for _ in range(NB)]
run('dense blocks', dense)
run('indexed, random rows of %d' % NROWS, [(table, s) for s in streams('random', NROWS)])
run('indexed, sorted random rows', [(table, s) for s in streams('sorted', NROWS)])
run('indexed, contiguous rows', [(table, s) for s in streams('contig', NROWS)])
run('indexed, random rows of 4096', [(table, s) for s in streams('random', 4096)])
run('dense blocks', dense)
