#!/usr/bin/env python
"""A train of dist_eval-sized scoring calls (cfl_pair_scores on 8192 pairs of 4096-d rows, pcd K=3 L=20): the
workload profiled for `roofline_eval` (rocprofv3 --kernel-trace --stats / --pmc passes of tools/measure_eval.sh).
Prints one JSON line with the event-timed call.  Experiment / measurement helper, not part of the product."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'compatibility-family-learning_amd')]
import numpy as np  # noqa
import torch  # noqa
from cfl import hipabi as H  # noqa
from cfl.engine import PairEngine  # noqa

ap = argparse.ArgumentParser()
ap.add_argument('--pairs', type=int, default=8192)
ap.add_argument('--calls', type=int, default=200)
ap.add_argument('--input-size', type=int, default=4096)
ap.add_argument('--num-components', type=int, default=3)
ap.add_argument('--latent-size', type=int, default=20)
ap.add_argument('--indexed', action='store_true', help='score by index from a resident table (what dist_eval does)')
a = ap.parse_args()
D, K, L, n = a.input_size, a.num_components, a.latent_size, a.pairs
rng = np.random.RandomState(0)
lim = lambda fi, fo: np.sqrt(6.0 / (fi + fo))
params = {'outputs/W': rng.uniform(-lim(D, L), lim(D, L), (D, L)).astype(np.float32), 'outputs/b': np.zeros(L, np.float32),
          'proto/W': rng.uniform(-lim(D, L * K), lim(D, L * K), (D, L * K)).astype(np.float32),
          'proto/b': np.zeros(L * K, np.float32)}
eng = PairEngine(D, L, K, 'pcd', weight_norm=False, has_bias=True, norm=H.make_norm(1 / 58.388599), loss=H.make_loss(),
                 params=params, batch_size=None)
g = torch.Generator(device='cuda')
g.manual_seed(1)
nsets = max(2, (600 << 20) // (8 * n * D))     # > 256 MiB Infinity Cache between two uses of a row
if a.indexed:
    table = torch.randn(nsets * 2 * n, D, generator=g, device='cuda').abs_() * 13
    perm = torch.randperm(table.shape[0], generator=g, device='cuda').to(torch.int32)
    sets = [(table, H.IndexStreams.from_tensors([perm[2 * i * n:(2 * i + 1) * n].contiguous(),
                                                  perm[(2 * i + 1) * n:(2 * i + 2) * n].contiguous()])) for i in range(nsets)]
else:
    sets = [(torch.randn(n, D, generator=g, device='cuda').abs_() * 13, torch.randn(n, D, generator=g, device='cuda').abs_() * 13)
            for _ in range(nsets)]
for i in range(2 * nsets):
    eng.scores(*sets[i % nsets])
torch.cuda.synchronize()
st = torch.cuda.current_stream()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(st)
for i in range(a.calls):
    eng.scores(*sets[i % nsets])
e1.record(st)
torch.cuda.synchronize()
t = e0.elapsed_time(e1) * 1e-3 / a.calls
print(json.dumps({'pairs_per_call': n, 'calls': a.calls, 'indexed': bool(a.indexed), 'avg_call_us': round(t * 1e6, 3),
                  'pairs_per_s': round(n / t, 1), 'achieved_GBps': round(8.0 * D * n / t / 1e9, 1),
                  'frac_of_8TBps': round(8.0 * D * n / t / 1e9 / 8000.0, 4)}))
