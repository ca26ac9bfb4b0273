#!/usr/bin/env python
"""Round 5: the training step that carries a validation batch as extra scoring rows (cfl_pair_train_val_steps_idx_planes) at the
headline shape: GPU time per iteration of the bare library loop (no Python per iteration), its kernel intervals, and the
separate-scoring form beside it.  Usage: [CFL_DEBUG_S=..] python tools/val_fused_probe.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'compatibility-family-learning_amd')]
import numpy as np, torch
from argparse import Namespace
from cfl import hipabi as H
from cfl.engine import PairEngine
from oracle import cfl_oracle as O
D, L, K, B = 4096, 20, 3, 512
rng = np.random.RandomState(0)
eng = PairEngine(D, L, K, norm=H.make_norm(1 / 58.388599), loss=H.make_loss(), lr=1e-3,
                 params=O.init_encoder_params(O.EncoderCfg(D=D, L=L, K=K), rng, np.float32), batch_size=B)
g = torch.Generator(device='cuda'); g.manual_seed(1)
n_items = 40000
table = torch.randn(n_items, D, generator=g, device='cuda').abs_() * 13
vtable = torch.randn(n_items // 4, D, generator=g, device='cuda').abs_() * 13
npairs = 200000
mk = lambda n, hi: torch.randint(0, hi, (n, 2), generator=g, device='cuda', dtype=torch.int32)
pos, neg, vpos, vneg = mk(npairs, n_items), mk(npairs, n_items), mk(npairs, n_items // 4), mk(npairs, n_items // 4)
CH = 16
ring = torch.empty(CH, H.S_COUNT + 2 * B, dtype=torch.float32).pin_memory()
slots = [ring[i].data_ptr() for i in range(CH)]
def chunk(head, fused):
    win = Namespace(table=table, pos_pairs=pos, neg_pairs=neg, pos_head=head, neg_head=head, batch_rows=B, shard_lo=0, rows=B, nsteps=CH, switched=None)
    vwin = Namespace(table=vtable, pos_pairs=vpos, neg_pairs=vneg, pos_head=head, neg_head=head, batch_rows=B, switched=None)
    if fused:
        eng.step_windows_val(win, vwin, [True] * CH, slots)
    else:
        eng.step_windows(win)
for fused in (True, False, True, False):
    for w in range(5): chunk(w * CH * B, fused)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 100
    for w in range(n): chunk((w % 20) * CH * B, fused)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / (n * CH)
    H.profile_enable(True)
    for w in range(10): chunk(w * CH * B, fused)
    torch.cuda.synchronize(); H.profile_enable(False)
    prof = H.profile_read()
    print('%-28s %.2f us/iteration   intervals %s' % ('train + fused validation' if fused else 'train only', dt * 1e6,
          {k: round(1e3 * ms / c, 2) for k, (ms, c) in prof.items()}), 'plan', H.plan_describe(eng.shape, B)['S'])
print('ring slot 0: total %.4f  first scores %s' % (ring[0][0].item(), ring[0][H.S_COUNT:H.S_COUNT + 3].tolist()))
