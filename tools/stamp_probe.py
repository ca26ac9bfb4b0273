#!/usr/bin/env python
"""Per-wave timeline of the proj kernel from in-kernel s_memtime stamps (diagnostic build
lib/stamps.so, -DCFL_STAMPS).  Experiment helper."""
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

B, D, K, L = 512, 4096, 3, 20
cfg = O.EncoderCfg(D=D, L=L, K=K)
eng = PairEngine(D, L, K, norm=H.make_norm(1 / 58.388599),
                 params=O.init_encoder_params(cfg, np.random.RandomState(0), np.float32), batch_size=B)
g = torch.Generator(device='cuda'); g.manual_seed(1)
nb = 12
pool = [tuple(torch.randn(B, D, generator=g, device='cuda').abs_() * 13 for _ in range(4)) for _ in range(nb)]
for i in range(30):
    eng.step(pool[i % nb])
torch.cuda.synchronize()
lib = H.lib()
lib.cfl_debug_clear_stamps()
eng.fwd_bwd(pool[5])
torch.cuda.synchronize()
n = 16384 * 8
buf = (C.c_ulonglong * n)()
lib.cfl_debug_read_stamps(buf, C.c_size_t(n))
st = np.frombuffer(buf, dtype=np.uint64).reshape(-1, 8).astype(np.int64)
st = st[st[:, 0] > 0]
t0 = st[:, 0].min()
rel = st - t0
names = ['entry', 'q0 frag ready', 'q0 mfma done', 'q1 done', 'q2 done', 'q3 done', 'after barrier', 'end']
print('waves with stamps:', st.shape[0], ' (s_memtime ticks; 100 MHz realtime? constant-rate counter)')
for i, nme in enumerate(names):
    c = rel[:, i]
    print('%-16s min %8d  p10 %8d  median %8d  p90 %8d  max %8d' % (nme, c.min(), np.percentile(c, 10), np.median(c), np.percentile(c, 90), c.max()))
d = np.diff(rel, axis=1)
print('phase durations (median / p90):')
for i in range(7):
    print('  %-16s -> %-16s  %8d %8d' % (names[i], names[i + 1], np.median(d[:, i]), np.percentile(d[:, i], 90)))
# split by job: z index is the slowest dim -> first half src (NT=4), second half dst (NT=2)
half = st.shape[0] // 2
for nm, sl in (('src', slice(0, half)), ('dst', slice(half, None))):
    dd = d[sl]
    print(nm, 'median phases:', [int(np.median(dd[:, i])) for i in range(7)], ' wave lifetime median', int(np.median(rel[sl, 7] - rel[sl, 0])))
