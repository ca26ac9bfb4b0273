#!/usr/bin/env python
"""Projection-kernel forms side by side (run on the GPU box): the scoring call of dist_eval (8192 pairs) and training
steps at large batches, under the CFL_DEBUG_PROJ_STREAM / CFL_DEBUG_PROJ_MIX overrides, plus a bit-for-bit check of
the scores against the chunk-at-a-time form.  Experiment helper, not part of the product.

    python tools/proj_probe.py [--train 2048,8192] [--pairs 8192] > gpurun_out/proj_probe.jsonl
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'compatibility-family-learning_amd')]
import numpy as np  # noqa
import torch  # noqa
from cfl import hipabi as H  # noqa
from cfl.engine import PairEngine  # noqa

ap = argparse.ArgumentParser()
ap.add_argument('--train', default='1024,2048,8192')
ap.add_argument('--pairs', default='8192,32768')
ap.add_argument('--input-size', type=int, default=4096)
ap.add_argument('--num-components', type=int, default=3)
ap.add_argument('--latent-size', type=int, default=20)
ap.add_argument('--variants', default='-1:0:-1:-1,1:-1:-1:-1,0:0:-1:1', help='stream:mix:ring:x3 overrides, comma separated')
a = ap.parse_args()
D, K, L = a.input_size, a.num_components, a.latent_size


def xavier(rng, fi, fo):
    lim = np.sqrt(6.0 / (fi + fo))
    return rng.uniform(-lim, lim, (fi, fo)).astype(np.float32)


rng = np.random.RandomState(0)
params = {'outputs/W': xavier(rng, D, L), 'outputs/b': np.zeros(L, np.float32),
          'proto/W': xavier(rng, D, L * K), 'proto/b': np.zeros(L * K, np.float32)}


def set_variant(stream, mix, ring, x3=-1):
    os.environ['CFL_DEBUG_PROJ_STREAM'] = str(stream)
    os.environ['CFL_DEBUG_PROJ_MIX'] = str(mix)
    os.environ['CFL_DEBUG_PROJ_RING'] = str(ring)
    os.environ['CFL_DEBUG_PROJ_X3'] = str(x3)
    H.reload_env()


def events_us(fn, n):
    st = torch.cuda.current_stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record(st)
    for i in range(n):
        fn(i)
    e1.record(st)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


variants = [(tuple(int(x) for x in v.split(':')) + (-1,))[:4] for v in a.variants.split(',')]
g = torch.Generator(device='cuda')
g.manual_seed(1)

# ---- scoring calls ---------------------------------------------------------------------------------------
for n in [int(x) for x in a.pairs.split(',') if x]:
    eng = PairEngine(D, L, K, 'pcd', weight_norm=False, has_bias=True, norm=H.make_norm(1 / 58.388599),
                     loss=H.make_loss(), params=params, batch_size=None)
    nsets = max(2, (600 << 20) // (8 * n * D))
    sets = [(torch.randn(n, D, generator=g, device='cuda').abs_() * 13, torch.randn(n, D, generator=g, device='cuda').abs_() * 13)
            for _ in range(nsets)]
    ref = None
    for stream, mix, ring, x3 in variants:
        set_variant(stream, mix, ring, x3)
        eng._ws = {}
        sc = eng.scores(*sets[0]).clone()
        torch.cuda.synchronize()
        if ref is None:
            ref = sc
        same = bool(torch.equal(ref, sc))
        maxdiff = float((ref - sc).abs().max())
        for i in range(4):
            eng.scores(*sets[i % nsets])
        t = events_us(lambda i: eng.scores(*sets[i % nsets]), 100)
        H.profile_enable(True)
        for i in range(50):
            eng.scores(*sets[i % nsets])
        torch.cuda.synchronize()
        H.profile_enable(False)
        prof = H.profile_read()
        print(json.dumps({'what': 'scores', 'pairs': n, 'stream': stream, 'mix': mix, 'ring': ring, 'x3': x3, 'max_abs_diff_to_first': maxdiff, 'score_scale': float(ref.abs().max()), 'call_us': round(t, 2),
                          'hbm_frac': round(8.0 * D * n / t / 1e3 / 8000.0, 4), 'bit_identical_to_first': same,
                          'kernels_us': {k: round(1e3 * ms / c, 2) for k, (ms, c) in prof.items()}}), flush=True)
    del sets, eng
    torch.cuda.empty_cache()

# ---- training steps ----------------------------------------------------------------------------------------
for B in [int(x) for x in a.train.split(',') if x]:
    nb = max(2, (600 << 20) // (16 * B * D))
    pool = [tuple(torch.randn(B, D, generator=g, device='cuda').abs_() * 13 for _ in range(4)) for _ in range(nb)]
    for stream, mix, ring, x3 in variants:
        set_variant(stream, mix, ring, x3)
        eng = PairEngine(D, L, K, 'pcd', weight_norm=False, has_bias=True, norm=H.make_norm(1 / 58.388599),
                         loss=H.make_loss(), params=params, batch_size=B)
        for i in range(10):
            eng.step(pool[i % nb])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        nsteps = 100
        for i in range(nsteps):
            eng.step(pool[i % nb])
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / nsteps * 1e6
        H.profile_enable(True)
        for i in range(50):
            eng.step(pool[i % nb])
        torch.cuda.synchronize()
        H.profile_enable(False)
        prof = H.profile_read()
        sc = eng.read_scalars()
        print(json.dumps({'what': 'train', 'B': B, 'stream': stream, 'mix': mix, 'ring': ring, 'x3': x3, 'step_us': round(wall, 2),
                          'loss_after': sc['total'],
                          'kernels_us': {k: round(1e3 * ms / c, 2) for k, (ms, c) in prof.items()}}), flush=True)
        del eng
    del pool
    torch.cuda.empty_cache()
