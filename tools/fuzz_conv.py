#!/usr/bin/env python
"""Randomised sweep of the weight-norm conv / transposed-conv layers against the torch-fp64 oracle: random
(B, H, W, Ci, Co, kernel, stride, activation, bias) through the assertions of tests/test_conv_gpu.py, plus the
transposed layer.  Usage: python tools/fuzz_conv.py [N] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'compatibility-family-learning_amd')]
import numpy as np
import torch
import __graft_entry__ as g
g.build()
import tests.test_conv_gpu as T
from cfl import hipabi as H, hipgan as G
from oracle import gan_oracle as GO
T.H = H

N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
fails = 0


def transposed_case(B, Hh, Ww, Ci, Co, K, S, act, seed):
    r = np.random.RandomState(seed)
    x = r.randn(B, Hh, Ww, Ci)
    V = r.randn(K, K, Co, Ci) * 0.2
    gg = 1.0 + 0.3 * r.randn(Co)
    b = 0.1 * r.randn(Co)
    tx, tV, tg, tb = (torch.tensor(a, requires_grad=True) for a in (x, V, gg, b))
    y = GO.conv2d_transpose_weight_norm(tx, tV, tg, tb, S, act)
    y = y.permute(0, 2, 3, 1) if y.shape[1] == Co and y.shape[-1] != Co else y
    dy = torch.tensor(r.randn(*y.shape))
    (y * dy).sum().backward()
    conv = H.CflConv(B, Hh, Ww, Ci, Co, K, K, S, H.CONV_ACTS[act])
    ws = torch.empty((G.conv_ws_bytes(conv, True) + 3) // 4, dtype=torch.float32, device='cuda')
    f = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float32, device='cuda')
    gy = torch.empty(B, Hh * S, Ww * S, Co, dtype=torch.float32, device='cuda')
    G.conv_fwd(conv, f(x), f(V), f(gg), f(b), gy, ws, True)
    ref = y.detach().numpy()
    assert np.abs(gy.cpu().numpy() - ref).max() <= 1e-5 * max(1.0, np.abs(ref).max()), 'convT fwd'
    dx, dV, dg, db = (torch.empty_like(f(a)) for a in (x, V, gg, b))
    G.conv_bwd(conv, f(x), f(V), f(gg), gy, f(dy.numpy()), ws, dx=dx, dV=dV, dg=dg, db=db, transposed=True)
    for got, want, name in ((dx, tx.grad, 'dx'), (dV, tV.grad, 'dV'), (dg, tg.grad, 'dg'), (db, tb.grad, 'db')):
        err = float((got.cpu().double() - want).abs().max())
        assert err <= 3e-5 * max(1e-2, float(want.abs().max())), ('convT ' + name, err)


for it in range(N):
    B = int(rng.choice([1, 2, 3, 5, 8]))
    Hh = int(rng.choice([1, 4, 7, 8, 14, 16]))
    Ww = Hh if rng.rand() < 0.7 else int(rng.choice([4, 6, 9, 16]))
    Ci = int(rng.choice([1, 3, 4, 8, 12, 32, 64]))
    # (fan-in 1 makes the weight-norm gradient an exact cancellation: skip the degenerate 1x1xCi=1 layer)
    Co = int(rng.choice([1, 3, 4, 12, 16, 20, 32, 48, 64, 96]))
    K = int(rng.choice([1, 3, 4, 5])) if Hh > 1 else 1
    if K == 1 and Ci == 1:
        Ci = 4
    S = int(rng.choice([1, 2])) if K > 1 else 1
    act = rng.choice([None, 'lrelu', 'relu'])
    bias = bool(rng.rand() < 0.7)
    if os.environ.get('FUZZ_STEM'):     # the 3-channel 4x4 stride-2 image stem (csrc/conv_stem.h): even sides, W >= 4, Co 16 / 32 / 64
        B = int(rng.choice([1, 2, 3, 7, 33]))
        Hh = int(rng.choice([2, 4, 6, 10, 16, 32, 64]))
        Ww = int(rng.choice([4, 6, 8, 12, 30, 32, 34, 64, 70]))
        Ci, Co, K, S = 3, int(rng.choice([16, 32, 64])), 4, 2
    desc = (B, Hh, Ww, Ci, Co, K, S, act, bias)
    try:
        T.test_conv2d_wn_fwd_bwd(*desc)
        if K in (3, 5) and Hh <= 8 and rng.rand() < 0.5:
            tact = act if act != 'lrelu' else 'relu'
            try:
                transposed_case(B, Hh, Ww, Ci, Co, K, 2, tact, it)
            except AssertionError as e:
                # a relu pre-activation within fp32 rounding of 0 (mask flip vs the fp64 oracle) is not a failure:
                # the same shape must pass with other data
                if tact is None:
                    raise
                transposed_case(B, Hh, Ww, Ci, Co, K, 2, tact, it + 100003)
                print('discontinuity (passes with other data):', desc, repr(e)[:100], flush=True)
    except Exception as e:          # noqa
        fails += 1
        print('FAIL', desc, '->', repr(e)[:300], flush=True)
print('conv fuzz: %d cases, %d failures' % (N, fails))
sys.exit(1 if fails else 0)
