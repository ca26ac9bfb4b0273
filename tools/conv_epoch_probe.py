#!/usr/bin/env python
"""Distance epochs of cfl.bin.train --model-type conv (ConvPCD: BASELINE config 0's model; experiments/fashion_30/run.sh) on a
synthetic 28x28x1 pair set in vector format, B = 100: wall time per training iteration of epochs 2 .. 3 of a 3-epoch run
(every CFL.train_step call counted and stamped; synchronised at the end), to be read beside the kernel time per iteration of the
same command under `rocprofv3 --kernel-trace --stats` (tools/r06_run3.sh -> profiles/r06_conv_epoch.md)."""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'compatibility-family-learning_amd')]
import torch  # noqa: E402
from cfl.bin import train  # noqa: E402
from cfl.models import cfl as M  # noqa: E402
from cfl.synthetic import make_dataset  # noqa: E402

n_pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
tmp = tempfile.mkdtemp(prefix='conv_epoch_')
root = os.path.join(tmp, 'data')
make_dataset(os.path.join(root, 'img'), D=784, n_items=10000, n_pos=n_pairs, n_neg=n_pairs, k=1, latent=8, seed=3, scale=0.25)
base = ['--data-name', 'img', '--data-root', root, '--checkpoint-root', os.path.join(tmp, 'ck'), '--log-root',
        os.path.join(tmp, 'logs'), '--model-type', 'conv', '--data-type', 'sigmoid', '--dist-type', 'pcd', '--use-threshold',
        '--reg-const', '5e-4', '--num-components', '1', '--latent-size', '30', '--input-shape', '28', '28', '1', '--batch-size', '100',
        '--seed', '10', '--disable-eval', '--reset']
stamps = []
orig = M.CFL.train_step


def stamped(self, batch):
    stamps.append(time.perf_counter())
    return orig(self, batch)


M.CFL.train_step = stamped
nb = n_pairs // 100
train.main(base + ['--epochs', '3'])
torch.cuda.synchronize()
t_end = time.perf_counter()
n = len(stamps)
first = stamps[nb] if n > nb else stamps[0]          # first iteration of epoch 2 (epoch 1 carries the one-time costs)
print('conv model: %d train_step calls (%d per epoch); epochs 2-3: %.3f ms wall per training iteration (%.2f s for %d iterations)'
      % (n, nb, 1e3 * (t_end - first) / max(n - nb, 1), t_end - first, n - nb))
