#!/usr/bin/env python
"""Distance epochs of cfl.bin.train --model-type conv (ConvPCD: BASELINE config 0's model; experiments/fashion_30/run.sh) on a
synthetic 28x28x1 pair set in vector format, B = 100: ms per training iteration = (3-epoch call - 1-epoch call) / 2 epochs."""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'compatibility-family-learning_amd')]
import torch  # noqa: E402
from cfl.bin import train  # noqa: E402
from cfl.synthetic import make_dataset  # noqa: E402

n_pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
tmp = tempfile.mkdtemp(prefix='conv_epoch_')
root = os.path.join(tmp, 'data')
make_dataset(os.path.join(root, 'img'), D=784, n_items=10000, n_pos=n_pairs, n_neg=n_pairs, k=1, latent=8, seed=3, scale=0.25)
base = ['--data-name', 'img', '--data-root', root, '--checkpoint-root', os.path.join(tmp, 'ck'), '--log-root',
        os.path.join(tmp, 'logs'), '--model-type', 'conv', '--data-type', 'sigmoid', '--dist-type', 'pcd', '--use-threshold',
        '--reg-const', '5e-4', '--num-components', '1', '--latent-size', '30', '--input-shape', '28', '28', '1', '--batch-size', '100',
        '--seed', '10', '--disable-eval', '--reset']


def wall(epochs):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    train.main(base + ['--epochs', str(epochs)])
    torch.cuda.synchronize()
    return time.perf_counter() - t0


nb = n_pairs // 100
t1, t3 = wall(1), wall(3)
print('conv model: %.3f ms per training iteration (two epochs of %d iterations: %.2f s; the 1-epoch call %.2f s)'
      % (1e3 * (t3 - t1) / (2 * nb), nb, t3 - t1, t1))
