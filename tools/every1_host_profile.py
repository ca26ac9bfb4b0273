#!/usr/bin/env python
"""Host profile of cfl.bin.train_dist.train_steps at --scalar-every 1 (the reference cadence) on a synthetic dataset."""
import cProfile, pstats, os, sys, tempfile, shutil, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'compatibility-family-learning_amd')]
import torch
from cfl.bin.train_dist import train_steps
from cfl.input_data import ResidentFeatures, load_data_sets
from cfl.models.dist import construct_model
from cfl.ops import normalizer, unnormalizer
from cfl.synthetic import make_dataset
from cfl.engine import quiet_host_threads
B, D = int(os.environ.get('B', 512)), 4096
KK, LL = int(os.environ.get('K', 3)), int(os.environ.get('L', 20))
root = tempfile.mkdtemp(prefix='cfl_prof_')
try:
    quiet_host_threads()
    make_dataset(os.path.join(root, 'syn'), D=D, n_items=20000, n_pos=100000, n_neg=100000, splits=(('train', 1.0), ('val', 0.2), ('test', 0.02)))
    data = load_data_sets(os.path.join(root, 'syn'), D, seed=633)
    model, aux = construct_model(input_shape=(D,), latent_size=LL, normalize_value=58.388599, lr=1e-3, beta1=0.9, beta2=0.999,
                                 num_components=KK, batch_size=B, data=data, reg_const=0.0,
                                 data_normalizer=normalizer(58.388599, 0., None, None), data_unnormalizer=unnormalizer(58.388599, 0.),
                                 seed=633, device=torch.device('cuda'))
    tr, va = ResidentFeatures(aux.train, model.device), ResidentFeatures(aux.val, model.device)
    seen = []
    cb = lambda i, s, v: seen.append((i, s['total'], v))
    train_steps(model, tr, va, B, None, 100, cb, scalar_every=1)
    torch.cuda.synchronize()
    t0 = time.perf_counter(); train_steps(model, tr, va, B, None, 1000, cb, scalar_every=1); torch.cuda.synchronize()
    print('every1: %.2f us/step' % ((time.perf_counter() - t0) / 1000 * 1e6))
    pr = cProfile.Profile(); pr.enable()
    train_steps(model, tr, va, B, None, 1000, cb, scalar_every=1)
    pr.disable(); torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats('tottime').print_stats(18)
finally:
    shutil.rmtree(root, ignore_errors=True)
