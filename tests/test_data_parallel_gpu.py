"""The N > 1 path on the GPU (-m gpu; the box has ONE GPU, so the two ranks share it and talk over gloo --
functional coverage of exactly the code the RCCL ranks run, not a measurement):

* PairEngine.step on the two row shards of a global batch == the single-GPU step on the whole batch
  (gradient, scalars, Adam result), for dense and for indexed batches;
* dist_eval under two ranks (every rank scores a shard of each pair list, rank 0 gathers and computes AUC /
  accuracy, both ranks get the numbers) == the single-GPU evaluation, bit for bit;
* `python bench.py --gpus 2` starts its own two ranks, trains shards of one seeded global pool, and prints one
  JSON line with n_gpus == ranks_seen == 2.
"""
import json
import os
import subprocess
import time
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out):
    for p in (ROOT, os.path.join(ROOT, 'compatibility-family-learning_amd')):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from cfl import engine, hipabi as H
        from cfl.engine import PairEngine
        rng = np.random.RandomState(0)                       # identical on every rank
        D, L, K, B = 256, 6, 3, 64
        params = {'outputs/W': (rng.randn(D, L) * 0.05).astype(np.float32), 'outputs/b': np.zeros(L, np.float32),
                  'outputs/g': np.ones(L, np.float32), 'proto/W': (rng.randn(D, L * K) * 0.05).astype(np.float32),
                  'proto/b': np.zeros(L * K, np.float32), 'proto/g': np.ones(L * K, np.float32)}
        mk = lambda: PairEngine(D, L, K, 'pcd', weight_norm=True, has_bias=True, norm=H.make_norm(0.25),
                                loss=H.make_loss(pos_weight=0.25, lambda_m=0.5, reg_const=1e-3), lr=1e-3,
                                device='cuda', params=params)
        eng = mk()
        table = torch.tensor(np.abs(rng.randn(500, D)).astype(np.float32), device='cuda')
        lo, hi = engine.shard_rows(B)
        worst = 0.0
        ref = mk() if rank == 0 else None
        for it in range(300):
            idx = [torch.tensor(rng.randint(0, 500, size=B).astype(np.int32), device='cuda') for _ in range(4)]
            full = [table[i.long()].contiguous() for i in idx]
            if rank == 0:
                # the single-GPU twin starts every step from the data-parallel engine's state: 300 single steps are compared,
                # not two free-running trajectories (whose summation-order differences compound through Adam)
                ref.theta.copy_(eng.theta); ref.m.copy_(eng.m); ref.v.copy_(eng.v)
                ref.beta1_power, ref.beta2_power = eng.beta1_power, eng.beta2_power
            if it % 2 == 0:
                eng.step([x[lo:hi].contiguous() for x in full])                      # dense shard
            else:
                eng.step((table, H.IndexStreams.from_tensors([i[lo:hi].contiguous() for i in idx])))
            assert eng.world_size == world
            if it % 10 and it < 290:
                continue                                   # (every step is taken; every tenth and the last ten are compared)
            s = eng.read_scalars()
            if rank == 0:
                # the single-GPU step on the whole global batch (no collective: fwd_bwd + Adam by hand)
                ref.fwd_bwd(full)
                rs = ref.read_scalars()
                g_dp = eng.grad * (1.0 / world)
                worst = max(worst, float((g_dp - ref.grad).abs().max() / ref.grad.abs().max()))
                ref.apply_adam(1.0)
                worst = max(worst, float((eng.theta - ref.theta).abs().max()))
                for k in ('total', 'thres', 'loss_pos', 'loss_neg', 'cd', 'accuracy', 'mean_d_pos', 'threshold'):
                    worst = max(worst, abs(s[k] - rs[k]) / max(1.0, abs(rs[k])))
        # every rank holds the same weights
        th = eng.theta.clone()
        dist.all_reduce(th, op=dist.ReduceOp.MAX)
        same = bool(torch.equal(th, eng.theta))
        if rank == 0:
            out.put((worst, same))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_sharded_steps_equal_the_single_gpu_step():
    ctx = mp.get_context('spawn')
    out = ctx.SimpleQueue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    worst, same = out.get()
    assert same
    # (300 steps, each compared from the same state.  The data-parallel step projects on the bf16 matrix cores from kept planes,
    # the twin's fwd_bwd with the exact-fp32 kernel, and the row sums are split over two ranks: observed 2.1e-6, bar 5e-6,
    # SURVEY 8(e) allows 1e-5)
    assert worst < 5e-6, worst


def _eval_worker(rank, world, port, path, out):
    for p in (ROOT, os.path.join(ROOT, 'compatibility-family-learning_amd')):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from cfl import engine, hipgan, input_data, ops, utils
        from cfl.models.dist import construct_model
        data = input_data.load_data_sets(path, 200, seed=5)
        model, _ = construct_model(input_shape=(200,), latent_size=6, num_components=2, lr=2e-3, beta1=0.9,
                                   beta2=0.999, batch_size=50, normalize_value=16.0, reg_const=1e-3,
                                   data_normalizer=ops.normalizer(16.0, 0.), data=data, seed=3)
        res = input_data.ResidentFeatures(data.train, model.device)
        for _ in range(5):
            model.engine.step(res.next_indexed(50, engine.shard_rows(50)))
        ev = utils.dist_eval(None, model, 64, data.val)                 # collective: 175 + 174 pairs, ragged shards
        one = hipgan.auc(utils._resident_scores(model, data.val, False, True),
                         utils._resident_scores(model, data.val, True, True))
        out.put((rank, ev.auc, ev.accuracy, one[0], one[1]))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_sharded_evaluation_equals_the_single_gpu_evaluation(tmp_path):
    from cfl.synthetic import make_dataset
    path = str(tmp_path / 'toy')
    make_dataset(path, D=200, n_items=300, n_pos=701, n_neg=699, k=2, latent=6, seed=1, scale=4.0)
    ctx = mp.get_context('spawn')
    out = ctx.SimpleQueue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_eval_worker, args=(r, 2, port, path, out)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    got = sorted(out.get() for _ in range(2))
    for rank, auc, acc, auc1, acc1 in got:
        assert (auc, acc) == (auc1, acc1), got          # same scores, same sort: the same two numbers
    assert got[0][1:] == got[1][1:] and 0.0 < got[0][1] < 1.0


def test_bench_starts_its_own_ranks():
    env = dict(os.environ, CFL_DIST_BACKEND='gloo', CFL_BENCH_LEG_SECONDS='0.2', CFL_DP_MAX_BLOCKS='64')
    env.pop('WORLD_SIZE', None)
    env.pop('RANK', None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '5', '--warmup', '2',
                        '--repeats', '3', '--pool-mib', '64', '--no-cpu-baseline', '--no-kernel-profile',
                        '--no-cli-loop'], env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['ranks_seen'] == 2 and out['config']['global_batch_rows'] == 1024
    assert out['config']['parallelism'] == 'dp2' and out['scaling'] == 'weak' and out['value'] > 0
    assert np.isfinite(out['config']['final_loss'])
    # round 6: the attribution of the scaling number travels in the same line -- both exchanges, two batch sizes per GPU, the
    # step without any collective; the one-shot legs come from a CHILD job started by rank 0 (isolation: a failure of that path
    # must not take the line down)
    legs = out['dp_scaling']
    for k in ('b512_allreduce', 'b512_without_collective', 'b2048_allreduce', 'b2048_without_collective'):
        assert legs[k]['us_per_step'] > 0, (k, legs[k])
    g5 = legs['config5_mrcgan_b100_per_gpu']
    assert g5['ms_per_step'] > 0 and g5['rows_per_gpu'] == 100 and np.isfinite(g5['d_total_loss']), g5
    one = legs['oneshot']
    assert one.get('dp_leg') == 'oneshot', one
    for k in ('b512', 'b2048'):
        assert one[k]['us_per_step'] > 0 and one[k]['lost_handoffs'] == 0, one
        assert one[k]['driven_by'].startswith('library'), one
    # a mismatching launcher is an error, not a silent single-GPU run
    env2 = dict(env, WORLD_SIZE='1', RANK='0')
    r2 = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup',
                         '1'], env=env2, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r2.returncode != 0 and '--gpus 2 but WORLD_SIZE 1' in r2.stderr


def test_bench_line_survives_a_dying_one_shot_child_job():
    """The one-shot exchange has never run on more than one GPU: `bench.py --gpus N` measures it in a CHILD job so that a crash or a
    hang there cannot take the headline line down.  Injected: the child's ranks die at once."""
    env = dict(os.environ, CFL_DIST_BACKEND='gloo', CFL_BENCH_LEG_SECONDS='0.1', CFL_DP_MAX_BLOCKS='64', CFL_BENCH_FAIL_DP_LEG='1')
    env.pop('WORLD_SIZE', None)
    env.pop('RANK', None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '5', '--warmup', '2',
                        '--repeats', '3', '--pool-mib', '64', '--restore-steps', '20', '--no-cpu-baseline', '--no-kernel-profile',
                        '--no-cli-loop'], env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['value'] > 0
    assert 'error' in out['dp_scaling']['oneshot'], out['dp_scaling']['oneshot']
    assert out['dp_scaling']['b2048_allreduce']['us_per_step'] > 0


def test_bench_line_survives_wedged_data_parallel_legs():
    """Last resort of `bench.py --gpus N`: when the dp_scaling legs do not come back (injected: the child job sleeps), every rank's
    watchdog fires, rank 0 prints the line with the headline it has already measured, and the job ends with exit code 0."""
    env = dict(os.environ, CFL_DIST_BACKEND='gloo', CFL_BENCH_LEG_SECONDS='0.1', CFL_DP_MAX_BLOCKS='64', CFL_BENCH_FAIL_DP_LEG='hang')
    env.pop('WORLD_SIZE', None)
    env.pop('RANK', None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '5', '--warmup', '2',
                        '--repeats', '3', '--pool-mib', '64', '--restore-steps', '20', '--no-cpu-baseline', '--no-kernel-profile',
                        '--no-cli-loop', '--dp-legs-timeout', '8'], env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['value'] > 0
    assert 'did not finish' in out['dp_scaling']['error'], out['dp_scaling']


def _gan_worker(rank, world, port, out):
    for p in (ROOT, os.path.join(ROOT, 'compatibility-family-learning_amd')):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), CFL_GAN_TUNE_STREAMS='0', CFL_DP_MAX_BLOCKS='64')
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        import tests.test_arith_goldens as T
        res = {}
        for name in ('gan_sr_double', 'gan_conv_mnist', 'cgan_conv_t', 'cgan_sr'):
            case = T.R.case_by_name(name)
            models = {}
            for kind in ('sharded', 'replica'):
                os.environ['CFL_GAN_DP_SHARD'] = '1' if kind == 'sharded' else '0'
                m = T.build_product_model(case)
                T.load_initial(m, case)
                models[kind] = m
            sh, rep = models['sharded'], models['replica']
            assert sh._gan_shard == (rank, world) and sh.gan_phase.B == case['batch_size'] // world
            assert rep._gan_shard is None and rep.gan_phase.B == case['batch_size']
            df = T._Diffs(case)
            worst_s = 0.0
            for step in range(case['steps']):
                inp = T.R.inputs(case, step)
                lab = T.labeled(case, inp['batch'])
                draws = (inp['z'], inp['eps'], inp['c'])
                for m in (sh, rep):
                    if case.get('cgan'):
                        m.post_step(lab, draws=draws)
                    else:
                        m.post_step(lab, T.unl(case, inp['unlabeled'][0]), T.unl(case, inp['unlabeled'][1]), draws=draws)
                s, sr = sh.gan_phase.read_scalars(), rep.gan_phase.read_scalars()
                for k in s:      # the logged numbers are GLOBAL-batch means on every rank
                    worst_s = max(worst_s, abs(s[k] - sr[k]) / max(1.0, abs(sr[k])))
                # ... and they sit where the goldens of the reference's own graph put them (the bars of test_hip_matches_reference_graph)
                for k in ('d_total_loss', 'g_total_loss', 'd_loss_real', 'd_loss_fake', 'd_loss_neg', 'd_grad_loss', 'd_loss_d',
                          'g_loss', 'g_loss_d', 'g_loss_d_neg', 'g_loss_int'):
                    if k in s:
                        df.scalar(step, k, s[k], 2e-3 if step else 5e-5)
            df.finish()
            # variables after the case's steps: sharded == whole batch up to the summation order (two partial sums instead of
            # one; Adam's early steps move every entry by ~lr whatever the gradient's size, so an entry whose tiny gradient
            # changes sign between the two orders differs by up to 2 lr per step: counted, not maxed)
            frac, worst_w, same = 1.0, 0.0, True
            for a, b in ((sh.gan_phase.disc, rep.gan_phase.disc), (sh.gan_phase.gen, rep.gan_phase.gen)):
                d = (a.pool.theta - b.pool.theta).abs()
                frac = min(frac, float((d <= 2e-6).float().mean()))
                worst_w = max(worst_w, float(d.max()))
                th = a.pool.theta.clone()
                dist.all_reduce(th, op=dist.ReduceOp.MAX)
                same = same and bool(torch.equal(th, a.pool.theta))
            res[name] = (worst_s, frac, worst_w, same, 2 * case['steps'] * 1e-3)
        if rank == 0:
            out.put(res)
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_post_epoch_step_sharded_over_two_ranks_equals_the_whole_batch():
    """MrCGAN post epochs under data parallelism (round 6): every rank runs the G / D step on ITS rows of the global batch (X_hat's
    std from the global batch; --cgan: its rows of both halves), one all-reduce of [d gradient | g gradient | scalars], two Adams.
    Against a replica model running the whole batch in the same process, on the four GAN golden cases, with the reference
    graph's golden scalars as the third witness; every rank ends with the same weights bit for bit."""
    ctx = mp.get_context('spawn')
    out = ctx.SimpleQueue()
    port = 29500 + (os.getpid() + 977) % 2000
    procs = [ctx.Process(target=_gan_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
        assert p.exitcode == 0
    res = out.get()
    assert set(res) == {'gan_sr_double', 'gan_conv_mnist', 'cgan_conv_t', 'cgan_sr'}
    for name, (worst_s, frac, worst_w, same, cap) in res.items():
        assert same, name
        assert worst_s < 2e-5, (name, worst_s)
        assert frac > 0.995 and worst_w <= cap, (name, frac, worst_w)


def _oneshot_worker(rank, world, port, out):
    for p in (ROOT, os.path.join(ROOT, 'compatibility-family-learning_amd')):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0',
                      CFL_DP_MAX_BLOCKS='64')     # (the ranks share ONE GPU here: few polling blocks per rank, csrc/cfl_dp.hip)
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from argparse import Namespace
        from cfl import engine, hipabi as H
        from cfl.engine import PairEngine
        rng = np.random.RandomState(0)                       # identical on every rank
        D, L, K, B = 256, 6, 3, 64
        params = {'outputs/W': (rng.randn(D, L) * 0.05).astype(np.float32), 'outputs/b': np.zeros(L, np.float32),
                  'proto/W': (rng.randn(D, L * K) * 0.05).astype(np.float32), 'proto/b': np.zeros(L * K, np.float32)}

        def mk(exchange):
            os.environ['CFL_DP_EXCHANGE'] = exchange
            return PairEngine(D, L, K, 'pcd', weight_norm=False, has_bias=True, norm=H.make_norm(0.25),
                              loss=H.make_loss(reg_const=1e-3), lr=1e-3, device='cuda', params=params)
        a, b = mk('allreduce'), mk('oneshot')
        assert a._oneshot is None and b._oneshot is not None
        table = torch.tensor(np.abs(rng.randn(500, D)).astype(np.float32), device='cuda')
        lo, hi = engine.shard_rows(B)
        worst = 0.0
        for it in range(100 if world == 2 else 24):
            idx = [torch.tensor(rng.randint(0, 500, size=B).astype(np.int32), device='cuda') for _ in range(4)]
            batch = (table, H.IndexStreams.from_tensors([i[lo:hi].contiguous() for i in idx]))
            a.step(batch)
            b.step(batch)
            sa, sb = a.read_scalars(), b.read_scalars()
            # (the one-shot exchange leaves every rank ITS slice of the gradient sum: reduce-scatter, not all-reduce)
            glo, ghi = b._oneshot.owned()
            worst = max(worst, float((a.theta - b.theta).abs().max()),
                        float((a.grad[glo:ghi] - b.grad[glo:ghi]).abs().max() / a.grad.abs().max()))
            for k in ('total', 'accuracy', 'mean_d_pos'):
                worst = max(worst, abs(sa[k] - sb[k]) / max(1.0, abs(sa[k])))
        # windows of device pair lists through step_windows (K steps per engine call, the exchange inside the loop)
        # == the same steps taken one by one
        n = 4 * B
        pos = torch.tensor(rng.randint(0, 500, size=(n, 2)).astype(np.int32), device='cuda')
        neg = torch.tensor(rng.randint(0, 500, size=(n, 2)).astype(np.int32), device='cuda')
        win = Namespace(table=table, pos_pairs=pos, neg_pairs=neg, pos_head=0, neg_head=B, batch_rows=B, shard_lo=lo,
                        rows=hi - lo, nsteps=3, switched=[False, True, False])
        c = mk('oneshot')
        try:                                                  # an unsynced checkpoint cannot be written
            b.state_dict()
            raise AssertionError('state_dict() from sharded, unsynced Adam slots must raise')
        except H.CflHipError:
            pass
        b.sync_state()                                        # collective: the Adam slots are sharded over the ranks
        if rank == 0:                                         # the complete slots equal the all-reduce path's (replicated) ones
            worst = max(worst, float((a.m - b.m).abs().max() / a.m.abs().max()), float((a.v - b.v).abs().max() / a.v.abs().max()))
        c.load_state_dict(b.state_dict())
        b.step_windows(win)
        for i in range(3):
            ph, nh = i * B + lo, B + i * B + lo
            cols = (1, 0) if win.switched[i] else (0, 1)
            streams = H.IndexStreams.from_tensors([pos[ph:ph + hi - lo, cols[0]].contiguous(), pos[ph:ph + hi - lo, cols[1]].contiguous(),
                                                   neg[nh:nh + hi - lo, cols[0]].contiguous(), neg[nh:nh + hi - lo, cols[1]].contiguous()])
            c.step((table, streams))
        windows_equal = bool(torch.equal(b.theta, c.theta)) and b.global_step == c.global_step
        # ... and with the validation fetch inside the iterations (round 6: every rank carries the whole validation batch as extra
        # rows of its own launches; the ring slot gets the GLOBAL scalar sums from the exchange's last kernel): the same
        # parameters as the plain windows, scores equal to a scoring call with the pre-update weights, scalars equal on every rank
        vwin = Namespace(table=table, pos_pairs=neg, neg_pairs=pos, pos_head=0, neg_head=B, batch_rows=B, switched=None)
        ring = torch.zeros(2, H.S_COUNT + 2 * B).pin_memory()
        pre = b.scores_pos_neg(table, H.IndexStreams.from_tensors([neg[0:B, 0].contiguous(), neg[0:B, 1].contiguous(),
                                                                   pos[B:2 * B, 0].contiguous(), pos[B:2 * B, 1].contiguous()])).cpu()
        b.step_windows_val(win, vwin, [True, False, True], [ring[0].data_ptr(), ring[1].data_ptr()])
        c.step_windows(win)
        torch.cuda.synchronize()
        val_ok = bool(torch.equal(b.theta, c.theta)) and \
            float((ring[0, H.S_COUNT:] - pre).abs().max()) < 1e-4 * max(1.0, float(pre.abs().max()))
        sums = ring[:, :H.S_COUNT].clone()
        dist.all_reduce(sums, op=dist.ReduceOp.MAX)
        val_ok = val_ok and bool(torch.equal(sums, ring[:, :H.S_COUNT])) and float(ring[1, 0]) != 0.0
        windows_equal = windows_equal and val_ok
        th = b.theta.clone()
        dist.all_reduce(th, op=dist.ReduceOp.MAX)
        same = bool(torch.equal(th, b.theta))                 # every rank holds the same weights, bit for bit
        lost = int(b._oneshot.lost.item())
        if rank == 0:
            out.put((worst, same, windows_equal, lost))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_oneshot_exchange_equals_allreduce_and_windows_equal_single_steps():
    """CFL_DP_EXCHANGE=oneshot (csrc/cfl_dp.hip, cfl/dp_exchange.py): reduce-scatter by direct stores into the owners'
    slot arrays (fine-grained device memory mapped through hipIpc), TF-Adam on the owned slice of theta / m / v (sharded
    Adam slots), all-gather of the updated theta slices by direct stores.  Two ranks on the one GPU: parameters / owned
    gradient sums / scalars / gathered Adam slots equal the all-reduce path to 1e-6, both ranks hold identical parameters,
    no hand-off was lost; and PairEngine.step_windows under data parallelism (K steps per call, the exchange inside the
    loop) equals the same steps taken one at a time, bit for bit."""
    ctx = mp.get_context('spawn')
    out = ctx.SimpleQueue()
    port = 33500 + os.getpid() % 2000
    procs = [ctx.Process(target=_oneshot_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    worst, same, windows_equal, lost = out.get()
    assert lost == 0
    assert same and windows_equal
    assert worst < 1e-6, worst


def test_eight_ranks_share_one_gpu():
    """EIGHT ranks (the size of the node the scaling bench runs on) on the one GPU -- IPC-handle fan-out to
    seven peers, slice boundaries at n / 8 (not multiples of a weight tile), eight rows per rank, the fused push of eight
    processes' weight-gradient launches into one another's slots, flag fan-in of eight -- against the all-reduce path: same
    parameters / owned gradient sums / scalars to 1e-6, identical parameters on every rank, no lost hand-off, windowed
    iterations (with the validation fetch) equal to single steps."""
    world = 8
    ctx = mp.get_context('spawn')
    out = ctx.SimpleQueue()
    port = 37500 + os.getpid() % 2000
    procs = [ctx.Process(target=_oneshot_worker, args=(r, world, port, out)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(900)
        assert p.exitcode == 0
    worst, same, windows_equal, lost = out.get()
    assert lost == 0
    assert same and windows_equal
    assert worst < 1e-6, worst
