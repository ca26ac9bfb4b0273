#!/usr/bin/env python
"""Per-wave timeline of the mid (row math) launch from in-kernel cycle-counter stamps (diagnostic build lib/stamps.so,
-DCFL_STAMPS).  Usage: python tools/mid_stamp_probe.py [config3]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'compatibility-family-learning_amd')]
os.environ['CFL_HIP_LIB'] = os.path.join(ROOT, 'compatibility-family-learning_amd', 'lib', 'stamps.so')
import numpy as np  # noqa
import torch  # noqa
from cfl import hipabi as H  # noqa
from cfl.engine import PairEngine  # noqa
from oracle import cfl_oracle as O  # noqa

c3 = len(sys.argv) > 1 and sys.argv[1] == 'config3'
B, D, K, L = (512, 1024, 1, 256) if c3 else (512, 4096, 3, 20)
cfg = O.EncoderCfg(D=D, L=L, K=K, dist_type='siamese' if c3 else 'pcd', style='cfl' if c3 else 'dist')
eng = PairEngine(D, L, K, cfg.dist_type, weight_norm=cfg.weight_norm, has_bias=cfg.has_bias, norm=H.make_norm(1 / 58.388599),
                 loss=H.make_loss(use_threshold=not c3, caffe_margin=100.0 if c3 else None),
                 params=O.init_encoder_params(cfg, np.random.RandomState(0), np.float32), batch_size=B)
g = torch.Generator(device='cuda'); g.manual_seed(1)
nb = 12
pool = [tuple(torch.randn(B, D, generator=g, device='cuda').abs_() * 13 for _ in range(4)) for _ in range(nb)]
for i in range(30):
    eng.step(pool[i % nb])
torch.cuda.synchronize()
lib = H.lib()
lib.cfl_debug_clear_stamps()
eng.step(pool[5])
torch.cuda.synchronize()
n = 16384 * 8
buf = (C.c_ulonglong * n)()
lib.cfl_debug_read_stamps(buf, C.c_size_t(n))
st = np.frombuffer(buf, dtype=np.uint64).reshape(-1, 8).astype(np.int64)
st = st[(st[:, 0] > 0) & (st[:, 5] > 0)][:, :6]
t0 = st[:, 0].min()
rel = st - t0
names = ['entry', 'slabs summed', 'distance done', 'loss + rowq', 'stores issued', 'stores acked']
print('waves with stamps:', st.shape[0], '(cycle-counter ticks)')
for i, nme in enumerate(names):
    c = rel[:, i]
    print('%-16s min %8d  p10 %8d  median %8d  p90 %8d  max %8d' % (nme, c.min(), np.percentile(c, 10), np.median(c), np.percentile(c, 90), c.max()))
d = np.diff(rel, axis=1)
print('phase durations (median / p90):')
for i in range(5):
    print('  %-16s -> %-16s  %8d %8d' % (names[i], names[i + 1], np.median(d[:, i]), np.percentile(d[:, i], 90)))
print('wave lifetime median', int(np.median(rel[:, 5] - rel[:, 0])), ' kernel span (first entry -> last ack)', int(rel[:, 5].max()))
