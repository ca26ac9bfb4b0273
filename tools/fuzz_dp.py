#!/usr/bin/env python
"""Randomised shapes through the ONE-SHOT data-parallel step on a one-rank group (the script of
tests/test_dp_step_gpu.py::test_one_shot_exchange_on_every_model_family_equals_the_fused_step with random cases): parameters,
Adam slots, planes and scalars must equal the fused single-GPU step bit for bit, with the push fused into the weight-gradient
launch or -- plans without a half-tile weight gradient -- through the separate push kernel.
Usage: python tools/fuzz_dp.py [N] [seed]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'compatibility-family-learning_amd')]
import numpy as np  # noqa: E402
import tests.test_dp_step_gpu as T  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
cases = []
for _ in range(N):
    style = str(rng.choice(['dist', 'cfl']))
    dtype = 'pcd' if style == 'dist' else str(rng.choice(['pcd', 'monomer', 'siamese'], p=[0.5, 0.25, 0.25]))
    D = int(rng.choice([128, 192, 256, 512, 1024, 1088, 2048, 4096, 8192] if os.environ.get('FUZZ_ANY_D') else [512, 1024, 2048, 4096, 8192]))
    if dtype == 'siamese':
        L, K = int(rng.randint(2, 300)), 1
    else:
        K = int(rng.randint(1, 9))
        L = int(rng.randint(2, max(3, min(80, 600 // K))))
    B = int(rng.choice([2, 7, 64, 100, 129, 256, 512, 1024, 2048]))
    if B * D > 2048 * 4096:
        B = 512
    lkw = {}
    if rng.rand() < 0.4:
        lkw['reg_const'] = float(rng.choice([1e-4, 1e-3]))
    if style == 'cfl':
        if rng.rand() < 0.4:
            lkw['pos_weight'] = float(rng.choice([0.0625, 0.5, 2.0]))
        if dtype == 'siamese':
            lkw.update(use_threshold=bool(rng.rand() < 0.5), caffe_margin=float(rng.choice([5.0, 100.0])))
    directed = bool(style == 'cfl' and rng.rand() < 0.25)
    cases.append((style, dtype, D, L, K, B, lkw, directed, 4))
env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK'):
    env.pop(k, None)
code = T._FAMILIES % dict(root=ROOT, port=str(38600 + os.getpid() % 1000), cases=cases)
r = subprocess.run([sys.executable, '-c', code], env=env, cwd=ROOT, capture_output=True, text=True, timeout=3000)
if r.returncode:
    print(r.stdout[-3000:], r.stderr[-5000:])
    sys.exit(1)
res = eval([ln for ln in r.stdout.splitlines() if ln.startswith('RESULT')][-1][len('RESULT '):])
fails = 0
for case, (style, dtype, K, directed, pushed, same, worst, planes_same, lost) in zip(cases, res):
    ok = lost == 0 and same and worst <= 2e-6 and planes_same in (True, None)
    fails += not ok
    if not ok:
        print('FAIL', case, dict(pushed=pushed, same=same, worst=worst, planes_same=planes_same, lost=lost))
print('dp fuzz: %d cases (%d with the fused push), %d failures' % (len(res), sum(1 for x in res if x[4]), fails))
sys.exit(1 if fails else 0)
