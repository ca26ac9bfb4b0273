#!/usr/bin/env python
"""Per-kernel timing of the bench workload (HIP-event hooks of the library).
Usage: [CFL_HIP_LIB=...so] [CFL_DEBUG_S=..] [CFL_DEBUG_P=..] python tools/kernel_probe.py [--steps N] [--tag T]
Experiment helper, not part of the product."""
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
from oracle import cfl_oracle as O  # noqa

ap = argparse.ArgumentParser()
ap.add_argument('--steps', type=int, default=300)
ap.add_argument('--batch-size', type=int, default=512)
ap.add_argument('--input-size', type=int, default=4096)
ap.add_argument('--num-components', type=int, default=3)
ap.add_argument('--latent-size', type=int, default=20)
ap.add_argument('--tag', default='')
ap.add_argument('--dist-type', default='pcd')
ap.add_argument('--weight-norm', action='store_true')
ap.add_argument('--caffe-margin', type=float, default=None)
a = ap.parse_args()
B, D, K, L = a.batch_size, a.input_size, a.num_components, a.latent_size
cfg = O.EncoderCfg(D=D, L=L, K=K, dist_type=a.dist_type, style='cfl' if a.weight_norm else 'dist')
params = O.init_encoder_params(cfg, np.random.RandomState(0), np.float32)
eng = PairEngine(D, L, K, a.dist_type, weight_norm=a.weight_norm, has_bias=cfg.has_bias,
                 norm=H.make_norm(1 / 58.388599), loss=H.make_loss(caffe_margin=a.caffe_margin,
                 use_threshold=a.dist_type != 'siamese' or not a.caffe_margin), params=params, batch_size=B)
g = torch.Generator(device='cuda'); g.manual_seed(1)
nb = max(2, (384 << 20) // (16 * B * D))
pool = [tuple(torch.randn(B, D, generator=g, device='cuda').abs_() * 13 for _ in range(4)) for _ in range(nb)]
for i in range(50):
    eng.step(pool[i % nb])
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(a.steps):
    eng.step(pool[i % nb])
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / a.steps * 1e6
H.profile_enable(True)
for i in range(a.steps):
    eng.step(pool[i % nb])
torch.cuda.synchronize()
H.profile_enable(False)
prof = H.profile_read()
print(json.dumps({'tag': a.tag, 'S': os.environ.get('CFL_DEBUG_S'), 'P': os.environ.get('CFL_DEBUG_P'),
                  'step_us': round(wall, 2),
                  'kernels_us': {k: round(1e3 * ms / n, 2) for k, (ms, n) in prof.items()}}))
