#!/usr/bin/env python
"""Per-layer-call time table of the MrCGAN post-epoch step at the config-5 shape (64x64x3, L=64, K=2, z=20, B=100,
srgan): every WNLayer.fwd / bwd call bracketed by events (ms-scale calls: the perturbation is small), summed by
(layer, call kind, shape) over N steps.  GF = fp32-equivalent GEMM flops of the call."""
import collections
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'compatibility-family-learning_amd')]
import numpy as np  # noqa: E402
import torch  # noqa: E402
from cfl.models import gan_blocks  # noqa: E402
from cfl.models.mrcgan import GanPhase  # noqa: E402

B = int(os.environ.get('B', 100)); L, zd = 64, 20
shape = (64, 64, 3)
dev = torch.device('cuda')
ph = GanPhase('srgan', shape, 'tanh', zd, L, B, dev, np.random.RandomState(0), lambda_gp=0.5, lambda_dra=0.5,
              m_enc=0.05, m_prj=0.2)
g = torch.Generator(device=dev); g.manual_seed(0)
N = int(np.prod(shape))
batch = [torch.tanh(torch.randn(B, N, device=dev, generator=g))] + \
        [0.3 * torch.randn(B, L, device=dev, generator=g) for _ in range(4)] + \
        [torch.randn(B, zd, device=dev, generator=g), torch.rand(B, 1, device=dev, generator=g)]
for _ in range(2):
    ph.step(*batch)
torch.cuda.synchronize()
records = []
WN = gan_blocks.WNLayer
orig_fwd, orig_bwd = WN.fwd, WN.bwd


def fwd(self, x, ws, *a, **k):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    y = orig_fwd(self, x, ws, *a, **k)
    e1.record()
    oh, ow = y.shape[1], y.shape[2]
    gf = 2.0 * y.shape[0] * oh * ow * self.co * self.kh * self.kw * self.ci / (self.stride ** 2 if self.kind == 'convt' else 1) / 1e9
    records.append((self.scope.split('/', 1)[0][:3] + ':' + self.scope.split('/', 1)[1], 'fwd', tuple(x.shape), gf, e0, e1))
    return y


def bwd(self, x, y, dy, ws, need_dx=True, need_dw=True, *a, **k):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    r = orig_bwd(self, x, y, dy, ws, need_dx, need_dw, *a, **k)
    e1.record()
    oh, ow = self.out_hw(x.shape[1], x.shape[2])
    one = 2.0 * x.shape[0] * oh * ow * self.co * self.kh * self.kw * self.ci / (self.stride ** 2 if self.kind == 'convt' else 1) / 1e9
    kind = 'bwd' + ('_dx' if need_dx else '') + ('_dw' if need_dw else '')
    records.append((self.scope.split('/', 1)[0][:3] + ':' + self.scope.split('/', 1)[1], kind, tuple(x.shape),
                    one * (int(need_dx) + int(need_dw)), e0, e1))
    return r


WN.fwd, WN.bwd = fwd, bwd
n = int(os.environ.get('N', 3))
t0 = time.perf_counter()
for _ in range(n):
    ph.step(*batch)
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / n
agg = collections.OrderedDict()
for name, kind, shp, gf, e0, e1 in records:
    k = (name, kind, shp)
    a = agg.setdefault(k, [0, 0.0, 0.0])
    a[0] += 1; a[1] += e0.elapsed_time(e1); a[2] += gf
tot = sum(a[1] for a in agg.values()) / n
print('step %.2f ms wall (with events); layer calls %.2f ms; other %.2f ms' % (wall * 1e3, tot, wall * 1e3 - tot))
print('%-52s %-10s %-22s %5s %9s %8s %8s' % ('layer', 'call', 'x shape', 'n/st', 'ms/step', 'GF/call', 'TF/s'))
for (name, kind, shp), (c, ms, gf) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print('%-52s %-10s %-22s %5.1f %9.3f %8.2f %8.1f' % (name, kind, 'x'.join(map(str, shp)), c / n, ms / n, gf / c,
                                                          gf / ms if ms else 0))
