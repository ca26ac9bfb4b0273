#!/usr/bin/env python
"""Gradient error of the bf16x3 matrix-core path vs the exact-fp32 MFMA path, both against the
fp64 oracle, at the headline shape (B=512, D=4096, K=3, L=20)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'compatibility-family-learning_amd')]
import numpy as np, torch
from cfl import hipabi as H
from cfl.engine import PairEngine
from oracle import cfl_oracle as O
B, D, K, L = 512, 4096, 3, 20
cfg = O.EncoderCfg(D=D, L=L, K=K)
rng = np.random.RandomState(0)
params = O.init_encoder_params(cfg, rng, np.float32)
batch = [np.abs(rng.randn(B, D)).astype(np.float32) * 13 for _ in range(4)]
p64 = {k: v.astype(np.float64) for k, v in params.items()}
xs = tuple(b.astype(np.float64) / 58.388599 for b in batch)
sc_, grads, _, _, _ = O.train_step_loss_and_grads(cfg, O.LossCfg(), p64, 1e-6, xs)
dev = [torch.from_numpy(b).cuda() for b in batch]
out = {}
for mode in ('x3', 'fp32'):
    os.environ['CFL_EXACT_FP32'] = '0' if mode == 'x3' else '1'
    H.reload_env()
    eng = PairEngine(D, L, K, norm=H.make_norm(1 / 58.388599), params=params, batch_size=B)
    eng.fwd_bwd(dev)
    torch.cuda.synchronize()
    g, _, _ = H.unpack_theta(eng.shape, eng.grad)
    out[mode] = g
for k in ('outputs/W', 'proto/W'):
    ref = grads[k]
    sc = np.abs(ref).max()
    for mode in ('x3', 'fp32'):
        err = np.abs(out[mode][k].astype(np.float64) - ref)
        print('%-10s %-5s max|err|/max|g| = %.3e   rms err/rms g = %.3e' % (k, mode, err.max() / sc, np.sqrt((err ** 2).mean()) / np.sqrt((ref ** 2).mean())))
