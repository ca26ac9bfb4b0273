#!/usr/bin/env python
"""The data-parallel step on a ONE-RANK process group (CFL_FORCE_DP=1), for rocprofv3: N steps of the headline shape through
PairEngine.step with the exchange named by CFL_DP_EXCHANGE (allreduce | oneshot).  Usage: [CFL_DP_EXCHANGE=oneshot] python
tools/dp_probe.py [steps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'compatibility-family-learning_amd')]
os.environ['CFL_FORCE_DP'] = '1'
os.environ.setdefault('MASTER_PORT', str(36000 + os.getpid() % 2000))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from cfl import engine, hipabi as H  # noqa: E402
from cfl.engine import PairEngine  # noqa: E402
from oracle import cfl_oracle as O  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
assert engine.init_from_env() == 1
D, L, K, B = 4096, 20, 3, int(os.environ.get('B', 512))
cfg = O.EncoderCfg(D=D, L=L, K=K)
p = O.init_encoder_params(cfg, np.random.RandomState(0), np.float32)
eng = PairEngine(D, L, K, 'pcd', norm=H.make_norm(1 / 58.388599), loss=H.make_loss(), lr=1e-3, device='cuda', params=p, batch_size=B)
g = torch.Generator(device='cuda'); g.manual_seed(1)
nb = max(2, (384 << 20) // (16 * B * D))
pool = [tuple(torch.randn(B, D, generator=g, device='cuda').abs_() * 13 for _ in range(4)) for _ in range(nb)]
for i in range(50):
    eng.step(pool[i % nb])
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(steps):
    eng.step(pool[i % nb])
torch.cuda.synchronize()
print('%s: %.2f us per step (one-rank group, native=%s)' % (os.environ.get('CFL_DP_EXCHANGE', 'allreduce'),
                                                            1e6 * (time.perf_counter() - t0) / steps, eng.dp_native() is not None))
engine.finalize()
