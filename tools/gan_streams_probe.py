#!/usr/bin/env python
"""Does the MrCGAN step time depend on HOW MANY streams the process created before the step's own (the HIP runtime deals streams
to its 4 hardware queues in creation order)?  N dummy streams first, then tools/gan_probe.py's measurement."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'compatibility-family-learning_amd')]
import numpy as np, torch
from cfl.models.mrcgan import GanPhase
n_dummy = int(os.environ.get('DUMMY', 0))
dev = torch.device('cuda')
torch.zeros(1, device=dev)
dummies = [torch.cuda.Stream(device=dev) for _ in range(n_dummy)]
for s in dummies:                      # (a stream gets its hardware queue at first use)
    with torch.cuda.stream(s):
        torch.zeros(1, device=dev)
torch.cuda.synchronize()
B, L, zd = 100, 64, 20
shape = (64, 64, 3)
ph = GanPhase('srgan', shape, 'tanh', zd, L, B, dev, np.random.RandomState(0), lambda_gp=0.5, lambda_dra=0.5, m_enc=0.05, m_prj=0.2)
g = torch.Generator(device=dev); g.manual_seed(0)
N = int(np.prod(shape))
batch = [torch.tanh(torch.randn(B, N, device=dev, generator=g))] + \
        [0.3 * torch.randn(B, L, device=dev, generator=g) for _ in range(4)] + \
        [torch.randn(B, zd, device=dev, generator=g), torch.rand(B, 1, device=dev, generator=g)]
for _ in range(3): ph.step(*batch)
torch.cuda.synchronize(); t0 = time.perf_counter()
n = 20
for _ in range(n): ph.step(*batch)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
def sid(st):
    return None if st is None else (st.stream_id >> 5, st.stream_id, hex(st.cuda_stream))


names = {'gen.prep': getattr(ph.gen, '_prep_stream', None), 'disc.prep': getattr(ph.disc, '_prep_stream', None),
         'gen.ws.side': ph.gen.ws.side_stream, 'disc.ws.side': ph.disc.ws.side_stream}
for k, (st, ws) in enumerate(getattr(ph.disc, '_chains', [])):
    names['chain%d' % (k + 1)] = st
    names['chain%d.side' % (k + 1)] = ws.side_stream
print('tuning', getattr(ph, 'stream_tuning', None))
print('dummy streams %d: MrCGAN step %.2f ms   ' % (n_dummy, dt * 1e3) +
      '  '.join('%s=%s' % (k, sid(v)[0] if v is not None else None) for k, v in names.items()))
