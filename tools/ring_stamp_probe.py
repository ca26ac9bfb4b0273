#!/usr/bin/env python
"""Per-wave timeline of the ring projection kernel from in-kernel s_memtime stamps (diagnostic build lib/stamps.so,
-DCFL_STAMPS; one scoring call of 8192 pairs).  Experiment helper."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'compatibility-family-learning_amd')]
os.environ['CFL_HIP_LIB'] = os.path.join(ROOT, 'compatibility-family-learning_amd', 'lib', 'stamps.so')
os.environ['CFL_DEBUG_PROJ_RING'] = '1'
import numpy as np  # noqa
import torch  # noqa
from cfl import hipabi as H  # noqa
from cfl.engine import PairEngine  # noqa

D, K, L, n = 4096, 3, 20, 8192
rng = np.random.RandomState(0)
params = {'outputs/W': rng.uniform(-.03, .03, (D, L)).astype(np.float32), 'outputs/b': np.zeros(L, np.float32),
          'proto/W': rng.uniform(-.03, .03, (D, L * K)).astype(np.float32), 'proto/b': np.zeros(L * K, np.float32)}
eng = PairEngine(D, L, K, norm=H.make_norm(1 / 58.388599), params=params, batch_size=None)
g = torch.Generator(device='cuda'); g.manual_seed(1)
sets = [(torch.randn(n, D, generator=g, device='cuda').abs_() * 13, torch.randn(n, D, generator=g, device='cuda').abs_() * 13) for _ in range(3)]
for i in range(6):
    eng.scores(*sets[i % 3])
torch.cuda.synchronize()
lib = H.lib()
lib.cfl_debug_clear_stamps()
eng.scores(*sets[0])
torch.cuda.synchronize()
N = 16384 * 8
buf = (C.c_ulonglong * N)()
lib.cfl_debug_read_stamps(buf, C.c_size_t(N))
st = np.frombuffer(buf, dtype=np.uint64).reshape(-1, 8, 8).astype(np.int64)[:256]   # [block][wave][slot]
t0 = st[st > 0].min()
def show(name, arr, names):
    print(name)
    for i, nm in enumerate(names):
        c = arr[:, :, i].ravel(); c = c[c > 0] - t0
        if c.size:
            print('  %-28s min %8d  median %8d  p90 %8d  max %8d' % (nm, c.min(), np.median(c), np.percentile(c, 90), c.max()))
show('consumers (waves 0-3), cycles since the first stamp of the launch', st[:, :4],
     ['entry', 'after first barrier', 'unit0 step0 frags consumed', 'unit0 loop done', 'unit0 stored', 'end'])
show('loaders (waves 4-7)', st[:, 4:],
     ['entry', 'pointers ready', 'prologue issued', 'step 0 landed', 'loop t=0 waited', 'unit0 last wait', 'unit0 last barrier', 'end'])
