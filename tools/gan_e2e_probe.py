#!/usr/bin/env python
"""End-to-end MrCGAN post-epoch loop at the config-5 shape through the CLI (experiments/dyadic/run_gen.sh in synthetic form:
64x64x3 images + 1024-d latents, L = 64, K = 2, B = 100, srgan, lambda_gp 0.5): where an ITERATION of cfl.bin.train's post
epoch goes -- batch assembly on the host (record reads, PNG decoding, latent parsing), input preparation, the GPU step.
Usage: python tools/gan_e2e_probe.py [n_items] [n_pairs]"""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'compatibility-family-learning_amd')]
import torch  # noqa: E402
from cfl.bin import train  # noqa: E402
from cfl.models import cfl as M  # noqa: E402
from cfl.synthetic import make_double_dataset  # noqa: E402

# (E2E_DUMMY=<n>: n streams created first -- the step time depends on how the runtime deals the step's streams to its 4 hardware
# queues, and that on how many streams existed before them; LEDGER round 5 items 25-26)
_dummies = [torch.cuda.Stream() for _ in range(int(os.environ.get('E2E_DUMMY', 0)))]
for _s in _dummies:
    with torch.cuda.stream(_s):
        torch.zeros(1, device='cuda')
n_items = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
n_pairs = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
tmp = tempfile.mkdtemp(prefix='gan_e2e_')
root = os.path.join(tmp, 'data')
t0 = time.perf_counter()
make_double_dataset(os.path.join(root, 'dy'), image_shape=(64, 64, 3), latent_dim=1024, n_items=n_items, n_pos=n_pairs,
                    n_neg=n_pairs, k=2, seed=5)
print('dataset written in %.1f s' % (time.perf_counter() - t0), flush=True)
base = ['--data-name', 'dy', '--data-root', root, '--checkpoint-root', os.path.join(tmp, 'ck'), '--log-root',
        os.path.join(tmp, 'logs'), '--model-type', 'linear', '--data-type', 'tanh', '--data-mean', '0.5', '--data-norm', '0.5',
        '--data-directed', '--latent-norm', '31.9098', '--data-is-image', '--data-is-double', '--raw-latent', '--latent-shape',
        '1024', '--input-shape', '64', '64', '3', '--dist-type', 'pcd', '--lambda-m', '0.5', '--use-threshold',
        '--num-components', '2', '--latent-size', '64', '--batch-size', '100', '--seed', '3']
train.main(base + ['--epochs', '1', '--reset'])
acc = {'data': 0.0, 'step': 0.0, 'n': 0}
orig_post = M.CFL.post_step


def post_step(self, *a, **k):
    t = time.perf_counter()
    r = orig_post(self, *a, **k)
    acc['step'] += time.perf_counter() - t
    acc['n'] += 1
    return r


M.CFL.post_step = post_step
if os.environ.get('E2E_SKIP_STEP'):          # host side alone: everything but the enqueue of the GAN step itself
    from cfl.models import mrcgan
    mrcgan.GanPhase.step = lambda self, *a, **k: self.scalars
if os.environ.get('E2E_SKIP_INPUTS'):        # the step on constant inputs: no batch assembly / input preparation per iteration
    _cache = {}
    _orig_inputs = M.CFL.gan_inputs
    M.CFL.gan_inputs = lambda self, *a, **k: _cache.setdefault('x', _orig_inputs(self, *a, **k))
orig_epoch = M.CFL._post_epoch


def post_epoch(self, *a, **k):
    torch.cuda.synchronize()
    t = time.perf_counter()
    n0, s0 = acc['n'], acc['step']
    r = orig_epoch(self, *a, **k)
    torch.cuda.synchronize()
    # the LAST post epoch counts (the first one also decodes every record once and tunes the step's stream placement)
    acc['epoch'], acc['n_last'], acc['step_last'] = time.perf_counter() - t, acc['n'] - n0, acc['step'] - s0
    return r


M.CFL._post_epoch = post_epoch
gan = ['--m-prj', '0.2', '--m-enc', '0.05', '--d-lr', '0.0002', '--d-beta1', '0.5', '--g-lr', '0.0002', '--g-beta1', '0.5',
       '--gan', '--gan-type', 'srgan', '--lambda-gp', '0.5']
train.main(base + gan + ['--load-pre-weights', '--epochs', '1', '--post-epochs', '3', '--disable-eval'])
n = max(acc.get('n_last', 0), 1)
print('post epochs: %d iterations, %.1f ms per iteration end to end; post_step (input preparation + enqueue of the GPU step) %.1f ms; '
      'the rest (host batch assembly: record reads, PNG decoding, latents) %.1f ms'
      % (n, 1e3 * acc['epoch'] / n, 1e3 * acc['step_last'] / n, 1e3 * (acc['epoch'] - acc['step_last']) / n))
