#!/usr/bin/env python
"""Which of the MrCGAN step's streams share a hardware queue, and what does that do to the step time?  For a series of candidate
stream sets (as GanPhase._tune_streams builds them) the partition of [main, gen.side, disc.side, chain1, chain1.side, chain2,
chain2.side, gen.prep, disc.prep] into hardware queues is found by a pairwise probe (a 1 ms spin on stream A, a tiny kernel on
stream B launched right after: B ends behind the spin <=> same queue), and the step is timed."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'compatibility-family-learning_amd')]
os.environ['CFL_GAN_TUNE_STREAMS'] = '0'
import numpy as np, torch
from cfl.models.mrcgan import GanPhase
B, L, zd = 100, 64, 20
shape = (64, 64, 3)
dev = torch.device('cuda')
ph = GanPhase('srgan', shape, 'tanh', zd, L, B, dev, np.random.RandomState(0), lambda_gp=0.5, lambda_dra=0.5, m_enc=0.05, m_prj=0.2)
g = torch.Generator(device=dev); g.manual_seed(0)
N = int(np.prod(shape))
batch = [torch.tanh(torch.randn(B, N, device=dev, generator=g))] + \
        [0.3 * torch.randn(B, L, device=dev, generator=g) for _ in range(4)] + \
        [torch.randn(B, zd, device=dev, generator=g), torch.rand(B, 1, device=dev, generator=g)]
for _ in range(2):
    ph.step(*batch)
torch.cuda.synchronize()
names = ['main', 'g.side', 'd.side', 'c1', 'c1.side', 'c2', 'c2.side', 'g.prep', 'd.prep']
spin = int(2.0e6)      # ~1 ms of s_sleep cycles
tiny = torch.zeros(64, device=dev)


def same_queue(a, b):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(b):
        e0.record()
    with torch.cuda.stream(a):
        torch.cuda._sleep(spin)
    with torch.cuda.stream(b):
        tiny.add_(1.0)
        e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) > 0.4


def partition(streams):
    groups = []
    for i, s in enumerate(streams):
        for grp in groups:
            if same_queue(streams[grp[0]], s):
                grp.append(i)
                break
        else:
            groups.append([i])
    return groups


spacers = []
for c in range(int(os.environ.get('CANDIDATES', 12))):
    if c:
        spacers.append(torch.cuda.Stream(device=dev))
    st = ph._stream_set()
    if os.environ.get('STRUCT') == '4q':      # main (+ G's weight gradients inline) | chain1 | chain2 | ONE stream for D's weight gradients and the cache preparations
        shared = st['chain_side'][0]
        st['chain_side'][1] = shared
        st['prep'] = [shared, shared]
        st['gen_side'] = None
        st['disc_side'] = shared
    ph._install_streams(st)
    streams = [torch.cuda.default_stream(dev), st['gen_side'] or torch.cuda.default_stream(dev), st['disc_side'], st['chain'][0],
               st['chain_side'][0], st['chain'][1], st['chain_side'][1], st['prep'][0], st['prep'][1]]
    for _ in range(2):
        ph.step(*batch, apply=False)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(6):
        ph.step(*batch, apply=False)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 6 * 1e3
    groups = partition(streams)
    print('%.2f ms  ' % dt + ' | '.join(','.join(names[i] for i in grp) for grp in groups), flush=True)
