"""N > 1 path on CPU: two processes over gloo exercise the same host logic the GPU
ranks run over RCCL -- contiguous row shards of one seeded global batch, one
all-reduce(sum) of the flat fp32 gradient in the device layout, scale 1/world --
with the NumPy oracle standing in for the per-rank HIP step."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out):
    for p in (ROOT, os.path.join(ROOT, 'compatibility-family-learning_amd')):
        if p not in sys.path:
            sys.path.insert(0, p)
    from cfl import engine, hipabi as H
    from oracle import cfl_oracle as O
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        rng = np.random.RandomState(0)                     # identical on every rank
        cfg = O.EncoderCfg(D=64, L=6, K=3, dist_type='pcd', style='cfl')
        lcfg = O.LossCfg(pos_weight=0.25, lambda_m=0.5)
        p = O.init_encoder_params(cfg, rng, np.float64)
        B = 16
        batch = tuple(rng.randn(B, cfg.D) for _ in range(4))
        thr = np.float64(0.3)
        lo, hi = engine.shard_rows(B)
        assert (lo, hi) == (rank * B // world, (rank + 1) * B // world)
        shard = tuple(b[lo:hi] for b in batch)
        _, g, _, dthr, _ = O.train_step_loss_and_grads(cfg, lcfg, p, thr, shard)
        sh = H.make_shape(cfg.D, cfg.L, cfg.K, 'pcd', True, True)
        flat = H.pack_theta(sh, g, None, dthr, 'cpu')       # flat gradient, device layout
        scale = engine.reduce_gradients(flat)              # all_reduce(SUM) over gloo
        assert scale == 1.0 / world and engine.world_size() == world and engine.rank() == rank
        flat *= scale
        if rank == 0:
            _, gf, _, dthr_f, _ = O.train_step_loss_and_grads(cfg, lcfg, p, thr, batch)
            ref = H.pack_theta(sh, gf, None, dthr_f, 'cpu')
            out.put(float((flat - ref).abs().max()) / float(ref.abs().max()))
    finally:
        dist.destroy_process_group()


def test_sharded_gradients_equal_full_batch_gradient():
    ctx = mp.get_context('spawn')
    out = ctx.SimpleQueue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert out.get() < 1e-6       # fp32 flat buffers; summation order differs from one rank


def test_shard_rows_cover_the_batch_exactly():
    from cfl import engine
    for world in (1, 2, 4, 8):
        spans = [engine.shard_rows(512, r, world) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == 512
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
    import pytest
    with pytest.raises(ValueError):
        engine.shard_rows(100, 0, 8)


def test_one_shot_exchange_slices_tile_the_buffer():
    """cfl/dp_exchange.py: the reduce-scatter / sharded-Adam / all-gather exchange gives every rank a slice of whole
    float4s; the owned parameter ranges tile [0, n_adam) exactly once for every world size, including buffers that do
    not divide evenly and worlds larger than the number of float4s."""
    from cfl.dp_exchange import owned_range, slice_floats
    for n_adam in (64, 4096, 393280, 393216 + 64):
        n = n_adam + 64                                   # [gradient | 16 scalars + pad]
        for world in (1, 2, 3, 4, 7, 8, 16):
            sl = slice_floats(n, world)
            assert sl % 4 == 0 and sl * world >= n and sl * (world - 1) < n + 4 * world
            covered = np.zeros(n_adam, np.int32)
            for r in range(world):
                lo, hi = owned_range(n_adam, sl, r)
                assert 0 <= lo <= hi <= n_adam and lo % 4 == 0
                covered[lo:hi] += 1
            assert (covered == 1).all()
