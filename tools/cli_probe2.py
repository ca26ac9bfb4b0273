"""Diagnostic: per-call host time of the chunked training loop (run on the GPU box): which calls are slow."""
import os
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'compatibility-family-learning_amd')):
    sys.path.insert(0, p)
from cfl.input_data import ResidentFeatures, load_data_sets  # noqa: E402
from cfl.models.dist import construct_model  # noqa: E402
from cfl.ops import normalizer, unnormalizer  # noqa: E402
from cfl.synthetic import make_dataset  # noqa: E402

D, B, NV = 4096, 512, 58.388599
print('cpus', os.cpu_count(), 'affinity', len(os.sched_getaffinity(0)))
root = tempfile.mkdtemp(prefix='cfl_probe_')
make_dataset(os.path.join(root, 'syn'), D=D, n_items=20000, n_pos=200000, n_neg=200000,
             splits=(('train', 1.0), ('val', 0.1), ('test', 0.02)))
data = load_data_sets(os.path.join(root, 'syn'), D, seed=633)
model, aux = construct_model(input_shape=(D,), latent_size=20, normalize_value=NV, lr=1e-3, beta1=0.9, beta2=0.999,
                             num_components=3, batch_size=B, data=data, reg_const=0.0,
                             data_normalizer=normalizer(NV, 0., None, None),
                             data_unnormalizer=unnormalizer(NV, 0.), seed=633, device='cuda')
tr = ResidentFeatures(aux.train, 'cuda')
eng = model.engine


def cg(name):
    for base in ('/sys/fs/cgroup', '/sys/fs/cgroup/cpu', '/sys/fs/cgroup/cpu,cpuacct'):
        try:
            return open(os.path.join(base, name)).read().strip().replace('\n', ' ; ')
        except OSError:
            pass
    return None


print('cpu.max', cg('cpu.max'), '| cfs_quota', cg('cpu.cfs_quota_us'), cg('cpu.cfs_period_us'))
print('cpu.stat', cg('cpu.stat'))
import threading


def loop(n, chunk, label, prefetch=True, ahead=None):
    c0, st0 = os.times(), cg('cpu.stat')
    import collections
    inflight = collections.deque()
    ds = tr.dataset
    if not prefetch:
        saved = ds.prefetch_reshuffle
        ds.prefetch_reshuffle = lambda b: None
    tw, ts, ti, evs = [], [], [], []
    torch.cuda.synchronize()
    t_all = time.perf_counter()
    i = 0
    while i < n:
        t0 = time.perf_counter()
        win = tr.next_windows(B, min(chunk, n - i))
        t1 = time.perf_counter()
        if win is None:
            b = tr.next_indexed(B)
            t2 = time.perf_counter()
            eng.step(b)
            ti.append((t2 - t1, time.perf_counter() - t2))
            i += 1
            continue
        tw.append(t1 - t0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        t1 = time.perf_counter()
        eng.step_windows(win)
        ts.append((time.perf_counter() - t1, win.nsteps))
        e1.record()
        evs.append((e0, e1, win.nsteps, t1 - t_all))
        if ahead is not None:
            inflight.append((e1, win.nsteps))
            while sum(k for _, k in inflight) > ahead:
                inflight.popleft()[0].synchronize()
        i += win.nsteps
    t_host = time.perf_counter() - t_all
    torch.cuda.synchronize()
    t_drain = time.perf_counter() - t_all
    if not prefetch:
        ds.prefetch_reshuffle = saved
    tw, ti = np.array(tw), np.array(ti).reshape(-1, 2)
    per = np.array([t / k for t, k in ts])
    print('%-34s host %.1f us/step drain %.1f | next_windows n=%d sum %.1f ms max %.2f ms | step_windows n=%d sum %.1f ms '
          'median %.1f us/step max call %.2f ms | wraps n=%d next_indexed sum %.1f ms step sum %.1f ms'
          % (label, 1e6 * t_host / n, 1e6 * t_drain / n, len(tw), 1e3 * tw.sum(), 1e3 * tw.max(), len(ts),
             1e3 * sum(t for t, _ in ts), 1e6 * np.median(per), 1e3 * max(t for t, _ in ts), len(ti),
             1e3 * ti[:, 0].sum(), 1e3 * ti[:, 1].sum()))
    busy = [(a.elapsed_time(b), k) for a, b, k, _ in evs]
    gaps = [evs[j][1].elapsed_time(evs[j + 1][0]) for j in range(len(evs) - 1)]
    print('    GPU: inside windows %.1f ms (median %.1f us/step, max %.1f us/step), between windows %.1f ms (max %.2f ms)'
          % (sum(t for t, _ in busy), 1e3 * np.median([t / k for t, k in busy]), 1e3 * max(t / k for t, k in busy),
             sum(gaps), max(gaps)))
    slow = sorted(((1e3 * t / k, j) for j, (t, k) in enumerate(busy)), reverse=True)[:5]
    print('    slowest windows on the GPU (us/step, index of %d):' % len(busy), [(round(a, 1), j) for a, j in slow],
          ' largest gaps (ms, after index):', sorted(((round(g, 2), j) for j, g in enumerate(gaps)), reverse=True)[:5])
    hs = sorted(((round(1e3 * t, 2), j) for j, (t, k) in enumerate(ts)), reverse=True)[:3]
    print('    slowest host calls (ms, index):', hs)
    c1 = os.times()
    print('    process cpu: user %.3f s sys %.3f s over %.3f s wall; threads %d' % (c1.user - c0.user, c1.system - c0.system, t_drain, threading.active_count()))
    print('    cpu.stat before', st0)
    print('    cpu.stat after ', cg('cpu.stat'))
    big = sorted(((t, k) for t, k in ts), reverse=True)[:6]
    print('    slowest step_windows calls (ms, steps):', [(round(1e3 * t, 2), k) for t, k in big])


loop(400, 25, 'warm')
loop(4000, 25, 'default threads, unbounded')
from cfl.engine import quiet_host_threads  # noqa: E402
quiet_host_threads()
loop(4000, 25, 'quiet threads, unbounded')
loop(4000, 25, 'quiet threads, <= 200 ahead', ahead=200)
loop(4000, 400, 'quiet threads, chunks of 400')
loop(4000, 25, 'quiet threads, no prefetch', prefetch=False)
