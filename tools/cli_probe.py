"""Diagnostic: where the host time of the cfl.bin.train_dist inner loop goes (run on the GPU box)."""
import cProfile
import os
import pstats
import sys
import tempfile
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'compatibility-family-learning_amd')):
    sys.path.insert(0, p)
from cfl.bin.train_dist import train_steps  # noqa: E402
from cfl.input_data import ResidentFeatures, load_data_sets  # noqa: E402
from cfl.models.dist import construct_model  # noqa: E402
from cfl.ops import normalizer, unnormalizer  # noqa: E402
from cfl.synthetic import make_dataset  # noqa: E402

D, B, NV = 4096, 512, 58.388599
root = tempfile.mkdtemp(prefix='cfl_probe_')
make_dataset(os.path.join(root, 'syn'), D=D, n_items=20000, n_pos=200000, n_neg=200000,
             splits=(('train', 1.0), ('val', 0.1), ('test', 0.02)))
data = load_data_sets(os.path.join(root, 'syn'), D, seed=633)
model, aux = construct_model(input_shape=(D,), latent_size=20, normalize_value=NV, lr=1e-3, beta1=0.9, beta2=0.999,
                             num_components=3, batch_size=B, data=data, reg_const=0.0,
                             data_normalizer=normalizer(NV, 0., None, None),
                             data_unnormalizer=unnormalizer(NV, 0.), seed=633, device='cuda')
tr, va = ResidentFeatures(aux.train, 'cuda'), ResidentFeatures(aux.val, 'cuda')
eng = model.engine


def timed(label, fn, n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print('%-44s host %.2f us/step, with drain %.2f us/step' % (label, 1e6 * (t1 - t0) / n, 1e6 * (t2 - t0) / n))


train_steps(model, tr, va, B, None, 100, None)
n = 2000
timed('step_windows chunks of 25, no read-back', lambda: train_steps(model, tr, va, B, None, n, None), n)
timed('step_windows chunks of 25 + read-backs', lambda: train_steps(model, tr, va, B, None, n, lambda i, s, v: None), n)


def single():
    for _ in range(n):
        eng.step(tr.next_indexed(B))


timed('single indexed steps (python per step)', single, n)
win = tr.next_windows(B, 200)
timed('one step_windows call of %d steps' % win.nsteps, lambda: eng.step_windows(win), win.nsteps)


def readback():
    for _ in range(50):
        model.scalars()
        model.batch_accuracy(va.next_indexed(B))


timed('scalars + validation batch (per call)', readback, 50)
pr = cProfile.Profile()
pr.enable()
train_steps(model, tr, va, B, None, 1000, lambda i, s, v: None)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(18)
