"""Diagnostic: fused gradient tail vs finalize kernel, first difference."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'compatibility-family-learning_amd')):
    sys.path.insert(0, p)
from cfl import hipabi as H  # noqa: E402
from cfl.engine import PairEngine  # noqa: E402
from oracle import cfl_oracle as O  # noqa: E402


def run(B, D, K, L, P=None, steps=20):
    rng = np.random.RandomState(5)
    cfg = O.EncoderCfg(D=D, L=L, K=K)
    params = O.init_encoder_params(cfg, rng, np.float32)
    pool = [[torch.from_numpy(np.abs(rng.randn(B, D)).astype(np.float32) * 3).cuda() for _ in range(4)]
            for _ in range(3)]
    res = {}
    for mode in ('fused', 'finalize'):
        os.environ['CFL_DEBUG_NOFUSE'] = '0' if mode == 'fused' else '1'
        if P:
            os.environ['CFL_DEBUG_P'] = str(P)
        H.reload_env()
        eng = PairEngine(D, L, K, norm=H.make_norm(1 / 8.0), loss=H.make_loss(), params=params, batch_size=B)
        snaps = []
        for it in range(steps):
            eng.fwd_bwd(pool[it % 3])
            snaps.append((eng.grad.clone(), eng.scalars.clone()))
            eng.step(pool[it % 3])
            snaps.append((eng.grad.clone(), eng.theta.clone()))
        res[mode] = snaps
    os.environ.pop('CFL_DEBUG_P', None)
    for i, (a, b) in enumerate(zip(res['fused'], res['finalize'])):
        for j, (x, y) in enumerate(zip(a, b)):
            if not torch.equal(x, y):
                d = (x != y).nonzero().flatten()
                print('B=%d D=%d P=%s: first difference at snap %d tensor %d: %d entries differ, first idx %s, max abs diff %.3e (max %.3e)'
                      % (B, D, P, i, j, d.numel(), d[:8].tolist(), float((x - y).abs().max()), float(y.abs().max())))
                return
    print('B=%d D=%d P=%s: identical' % (B, D, P))


run(1024, 2048, 3, 20)
run(1024, 2048, 3, 20, P=2)
run(1024, 2048, 3, 20, P=8)
run(512, 4096, 3, 20)
run(512, 4096, 3, 20, P=4)
run(512, 4096, 3, 20, P=8)
run(2048, 1024, 3, 20)
