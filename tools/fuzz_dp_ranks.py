#!/usr/bin/env python
"""Random model families / shapes through BOTH data-parallel exchanges with several ranks sharing the one GPU (gloo control plane):
after every step the one-shot exchange (push fused into the weight-gradient launch where the plan allows, sharded Adam + all-gather)
must leave the parameters where the all-reduce path leaves them, on every rank, with no lost hand-off.
(More than two ranks on ONE GPU with weight-normalised heads can stall for 15-20 s per step -- profiles/r06_handoff_timeout.txt -- the
in-launch hand-offs are bounded by 60 s of wall clock, so such runs are slow, not wrong.)
Usage: python tools/fuzz_dp_ranks.py [world] [N] [seed]        (also imported by tests/test_data_parallel_gpu.py: CASES / worker)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, 'compatibility-family-learning_amd')):
    if _p not in sys.path:
        sys.path.insert(0, _p)
import numpy as np  # noqa: E402


def random_cases(n, seed, world):
    rng = np.random.RandomState(seed)
    cases = []
    for _ in range(n):
        style = str(rng.choice(['dist', 'cfl']))
        dtype = 'pcd' if style == 'dist' else str(rng.choice(['pcd', 'monomer', 'siamese'], p=[0.5, 0.25, 0.25]))
        # FUZZ_FUSABLE=1: shapes whose plan has a half-tile weight gradient (D / 32 x column jobs in [256, 640]): the fused push
        D = int(rng.choice([2048, 4096, 8192] if os.environ.get('FUZZ_FUSABLE') else [128, 256, 512, 1024, 1088, 2048, 4096]))
        if dtype == 'siamese':
            L, K = int(rng.randint(2, 200)), 1
        else:
            K = int(rng.randint(1, 7))
            L = int(rng.randint(2, max(3, min(40, 300 // K))))
        B = int(rng.choice([1, 3, 16, 50, 64, 128, 256])) * world        # global batch: equal shares
        lkw = {}
        if rng.rand() < 0.4:
            lkw['reg_const'] = float(rng.choice([1e-4, 1e-3]))
        if style == 'cfl':
            if rng.rand() < 0.4:
                lkw['pos_weight'] = float(rng.choice([0.0625, 0.5, 2.0]))
            if dtype == 'siamese':
                lkw.update(use_threshold=bool(rng.rand() < 0.5), caffe_margin=float(rng.choice([5.0, 100.0])))
        directed = bool(style == 'cfl' and rng.rand() < 0.25)
        cases.append((style, dtype, D, L, K, B, lkw, directed, 5))
    return cases


def worker(rank, world, port, cases, out):
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0', CFL_DP_MAX_BLOCKS='64')
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from cfl import engine, hipabi as H
        from cfl.engine import PairEngine
        from oracle import cfl_oracle as O
        res = []
        for style, dtype, D, L, K, B, lkw, directed, steps in cases:
            rng = np.random.RandomState(3)                       # identical on every rank
            cfg = O.EncoderCfg(D=D, L=L, K=K, dist_type=dtype, style=style)

            def params():
                p = O.init_encoder_params(cfg, rng, np.float32)
                for k in p:
                    p[k] = (p[k] + 0.05 * rng.randn(*p[k].shape).astype(np.float32) * (0.1 if k.endswith('/W') else 1.0)).astype(np.float32)
                return p
            p0 = params()
            p1 = params() if directed else None

            def mk(exchange):
                os.environ['CFL_DP_EXCHANGE'] = exchange
                return PairEngine(D, L, K, dtype, weight_norm=cfg.weight_norm, has_bias=cfg.has_bias, directed=directed,
                                  norm=H.make_norm(1.0 / 31.9098), loss=H.make_loss(**lkw), lr=1e-3, device='cuda', params=p0,
                                  params_dst=p1, thr=40.0 if dtype == 'siamese' else 0.5, batch_size=B)
            a, b = mk('allreduce'), mk('oneshot')
            lo, hi = engine.shard_rows(B)
            worst = 0.0
            for _ in range(steps):
                full = [torch.from_numpy((np.abs(rng.randn(B, D)) * 8.0).astype(np.float32)).cuda() for _ in range(4)]
                shard = [x[lo:hi].contiguous() for x in full]
                a.step(shard)
                b.step(shard)
                scale = max(1e-3, float(a.theta.abs().max()))
                worst = max(worst, float((a.theta - b.theta).abs().max()) / scale)
            try:
                sa = a.read_scalars()
            except H.CflHipError as e:
                raise RuntimeError('all-reduce engine, case %r: %s' % ((style, dtype, D, L, K, B, lkw, directed), e))
            try:
                sb = b.read_scalars()
            except H.CflHipError as e:
                raise RuntimeError('one-shot engine (lost %d), case %r: %s' % (int(b._oneshot.lost.item()),
                                                                              (style, dtype, D, L, K, B, lkw, directed), e))
            worst = max(worst, max(abs(sa[k] - sb[k]) / max(1.0, abs(sa[k])) for k in sa))
            th = b.theta.clone()
            dist.all_reduce(th, op=dist.ReduceOp.MAX)
            same = bool(torch.equal(th, b.theta))                # every rank holds the same parameters
            b.sync_state()
            slots = max(float((a.m - b.m).abs().max() / max(1e-12, float(a.m.abs().max()))),
                        float((a.v - b.v).abs().max() / max(1e-20, float(a.v.abs().max()))))
            res.append((bool(H.dp_push_fusable(b.shape, hi - lo)), worst, same, slots, int(b._oneshot.lost.item())))
            del a, b
            torch.cuda.empty_cache()
        if rank == 0:
            out.put(res)
        dist.barrier()
    finally:
        dist.destroy_process_group()


def run(world, cases, port):
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    out = ctx.SimpleQueue()
    procs = [ctx.Process(target=worker, args=(r, world, port, cases, out)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(1800)
        assert p.exitcode == 0, p.exitcode
    return out.get()


if __name__ == '__main__':
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    seed = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    cases = random_cases(n, seed, world)
    if os.environ.get('FUZZ_CASES'):            # explicit cases (a repro): a Python literal list of case tuples
        cases = eval(os.environ['FUZZ_CASES'])
    res = run(world, cases, 29500 + os.getpid() % 2000)
    fails = 0
    for case, (pushed, worst, same, slots, lost) in zip(cases, res):
        ok = lost == 0 and same and worst <= 5e-6 and slots <= 1e-4
        fails += not ok
        if not ok:
            print('FAIL', case, dict(pushed=pushed, worst=worst, same=same, slots=slots, lost=lost))
    print('dp ranks fuzz: world %d, %d cases (%d with the fused push), %d failures; worst %.2e' %
          (world, len(res), sum(1 for r in res if r[0]), fails, max(r[1] for r in res)))
    sys.exit(1 if fails else 0)
