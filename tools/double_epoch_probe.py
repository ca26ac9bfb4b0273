#!/usr/bin/env python
"""Distance epochs of cfl.bin.train on an image + latent dataset (the first call of experiments/dyadic/run_gen.sh in synthetic
form: 64x64x3 PNGs + 1024-d raw latents, L = 64, K = 2, B = 100): ms per training iteration with the latents resident in HBM
(default: indexed kernels, fused epoch loop), resident with one Python iteration per step (CFL_FUSED_EPOCHS=0), and with host
batches (CFL_DOUBLE_RESIDENT=0); the difference of a 3-epoch and a 1-epoch call, so that set-up does not count.  Usage: python tools/double_epoch_probe.py [n_items] [n_pairs]"""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'compatibility-family-learning_amd')]
import torch  # noqa: E402
from cfl.bin import train  # noqa: E402
from cfl.synthetic import make_double_dataset  # noqa: E402

n_items = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
n_pairs = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
tmp = tempfile.mkdtemp(prefix='dbl_epoch_')
root = os.path.join(tmp, 'data')
make_double_dataset(os.path.join(root, 'dy'), image_shape=(64, 64, 3), latent_dim=1024, n_items=n_items, n_pos=n_pairs,
                    n_neg=n_pairs, k=2, seed=5)
base = ['--data-name', 'dy', '--data-root', root, '--checkpoint-root', os.path.join(tmp, 'ck'), '--log-root',
        os.path.join(tmp, 'logs'), '--model-type', 'linear', '--data-type', 'tanh', '--data-mean', '0.5', '--data-norm', '0.5',
        '--data-directed', '--latent-norm', '31.9098', '--data-is-image', '--data-is-double', '--raw-latent', '--latent-shape',
        '1024', '--input-shape', '64', '64', '3', '--dist-type', 'pcd', '--lambda-m', '0.5', '--use-threshold',
        '--num-components', '2', '--latent-size', '64', '--batch-size', '100', '--seed', '3', '--disable-eval']
def wall(epochs):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    train.main(base + ['--epochs', str(epochs), '--reset'])
    torch.cuda.synchronize()
    return time.perf_counter() - t0


nb = n_pairs // 100
for tag, env in (('resident latents, fused epoch loop', {'CFL_DOUBLE_RESIDENT': '1', 'CFL_FUSED_EPOCHS': '1'}),
                 ('resident latents, one Python iteration per step', {'CFL_DOUBLE_RESIDENT': '1', 'CFL_FUSED_EPOCHS': '0'}),
                 ('host batches (the reference\'s data path)', {'CFL_DOUBLE_RESIDENT': '0', 'CFL_FUSED_EPOCHS': '0'})):
    os.environ.update(env)
    t1, t3 = wall(1), wall(3)
    print('%s: %.3f ms per training iteration (two epochs of %d iterations: %.2f s; the 1-epoch call %.2f s)'
          % (tag, 1e3 * (t3 - t1) / (2 * nb), nb, t3 - t1, t1), flush=True)
