#!/usr/bin/env python
"""bench.py -- training throughput of the cfl pair-distance hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): Amazon also_viewed / Monomer-style 4096-d
image features, PCD K=3, latent_size L=20 (the reference default, cfl/utils.py:520),
batch 512 rows per GPU per step, model = `Dist` of cfl/models/dist.py (plain FC
heads + biases, learned-threshold sigmoid-CE), normalize_value 58.388599, TF-Adam
lr 1e-3.  A "step" = forward + backward + Adam over one row batch of 4 x [512, 4096]
fp32 inputs already resident in HBM.  Batches rotate through a pool larger than
the 256 MiB Infinity Cache so the input stream really comes from HBM.

N > 1: one process per GPU (torchrun), weak scaling (512 rows per GPU), one RCCL
all-reduce of the flat fp32 gradient per step.

Prints ONE JSON line (rank 0) with the driver's contract fields plus `roofline`
(dominant kernel, HIP-event timed) and `cpu_baseline` (NumPy oracle on host cores).
"""
import argparse
import json
import os
import sys
import time

# must be in the environment before the HIP runtime starts (multi-process RCCL needs dmabuf IPC on this pool)
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, 'compatibility-family-learning_amd')
for _p in (ROOT, PKG):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
FP32_MFMA_PEAK_TF = 157.3  # dense fp32 MFMA peak (same guide)
NORMALIZE_VALUE = 58.388599
NOISE = float(os.environ.get("CFL_BENCH_NOISE", "0.3"))  # target-side noise of the planted positives


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=2000)
    ap.add_argument('--warmup', type=int, default=200)
    ap.add_argument('--batch-size', type=int, default=512)
    ap.add_argument('--input-size', type=int, default=4096)
    ap.add_argument('--num-components', type=int, default=3)
    ap.add_argument('--latent-size', type=int, default=20)
    ap.add_argument('--pool-mib', type=int, default=384,
                    help='resident batch pool per GPU (> 256 MiB Infinity Cache)')
    ap.add_argument('--cpu-seconds', type=float, default=10.0)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-kernel-profile', action='store_true')
    return ap.parse_args()


def make_pool(B, D, nbatches, device, seed):
    """Synthetic post-ReLU-like CNN features: |N(0,1)| scaled so that max ~ the
    Monomer normalize_value; positive targets planted by a hidden linear
    teacher, negative targets planted from unrelated sources."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    gt = torch.Generator(device=device)
    gt.manual_seed(20261002)                  # the hidden teacher is the same for every pool / rank / eval split
    s = NORMALIZE_VALUE / 4.5
    pool = []
    teacher = torch.randn(D, 64, generator=gt, device=device) / D ** 0.5
    back = torch.randn(64, D, generator=gt, device=device) / 8.0
    for _ in range(nbatches):
        ps = torch.randn(B, D, generator=g, device=device).abs_() * s
        pd = ((ps @ teacher) @ back + NOISE * s * torch.randn(B, D, generator=g, device=device)).abs_()
        ns = torch.randn(B, D, generator=g, device=device).abs_() * s
        # negatives: the target of an UNRELATED source, so that both targets have the same marginal and only the
        # pairing separates the classes
        other = torch.randn(B, D, generator=g, device=device).abs_() * s
        ndd = ((other @ teacher) @ back + NOISE * s * torch.randn(B, D, generator=g, device=device)).abs_()
        pool.append((ps.contiguous(), pd.contiguous(), ns.contiguous(), ndd.contiguous()))
    return pool


def cpu_baseline(args, seconds):
    """The oracle (a NumPy port of the reference step: fwd + analytic bwd + TF-Adam,
    fp32) timed on this box's host cores on a bounded sample of the same workload."""
    from oracle import cfl_oracle as O
    rng = np.random.RandomState(0)
    B, D = args.batch_size, args.input_size
    cfg = O.EncoderCfg(D=D, L=args.latent_size, K=args.num_components)
    tr = O.OracleTrainer(cfg, O.LossCfg(), lr=1e-3, dtype=np.float32)
    batch = tuple((np.abs(rng.randn(B, D)) * (1 / 4.5)).astype(np.float32) for _ in range(4))
    tr.step(batch)
    n, t0 = 0, time.perf_counter()
    while True:
        tr.step(batch)
        n += 1
        el = time.perf_counter() - t0
        if el >= seconds and n >= 3:
            break
    try:
        from threadpoolctl import threadpool_info
        cores = max([i.get('num_threads', 1) for i in threadpool_info()] or [1])
    except Exception:
        cores = os.cpu_count()
    return {'value': round(n * B / el, 1), 'unit': 'triplets/s', 'cores': int(cores),
            'kind': 'port',
            'sample': '%d steps of batch %d (%.1f s) of the same 4096-d K=%d L=%d step, '
                      'NumPy fp32 oracle (fwd+bwd+TF-Adam)' % (
                          n, B, el, args.num_components, args.latent_size)}


def eval_auc(args, eng, device, seed):
    """The second half of BASELINE.json's metric: AUC of the just-trained weights on held-out synthetic pairs,
    scored by the HIP scoring path (cfl_pair_scores) and by the fp64 oracle on the same weights and pairs
    (a bounded sample: 2 x 2048 pairs).  The oracle is only the checker here."""
    from oracle import cfl_oracle as O
    from cfl import hipabi as H
    B, D = 2048, args.input_size
    ps, pd, ns, nd = make_pool(B, D, 1, device, seed)[0]
    sp = eng.scores(ps, pd).cpu().numpy().astype(np.float64)
    sn = eng.scores(ns, nd).cpu().numpy().astype(np.float64)
    hip = O.dist_eval(sp, sn)
    p, _, thr = H.unpack_theta(eng.shape, eng.theta)
    cfg = O.EncoderCfg(D=D, L=args.latent_size, K=args.num_components)
    p64 = {k: v.astype(np.float64) for k, v in p.items()}
    f = lambda t: t.cpu().numpy().astype(np.float64) / NORMALIZE_VALUE
    op = O.pair_scores(cfg, p64, np.float64(thr), f(ps), f(pd))
    on = O.pair_scores(cfg, p64, np.float64(thr), f(ns), f(nd))
    ora = O.dist_eval(op, on)
    return {'pairs': 2 * B, 'hip': round(hip['auc'], 6), 'oracle_fp64': round(ora['auc'], 6),
            'abs_diff': float(abs(hip['auc'] - ora['auc'])), 'accuracy_hip': round(hip['accuracy'], 6),
            'accuracy_oracle': round(ora['accuracy'], 6),
            'max_abs_score_diff': float(max(np.abs(sp - op).max(), np.abs(sn - on).max()))}


def _trace(msg):
    if os.environ.get('CFL_BENCH_TRACE'):
        print('[bench rank %s] %s' % (os.environ.get('RANK', '0'), msg), file=sys.stderr, flush=True)


def main():
    args = parse()
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    # one process per GPU; CFL_DIST_BACKEND=gloo (+ fewer GPUs than ranks) is a functional test mode of the
    # data-parallel path on a single-GPU box, not a measurement
    backend = os.environ.get('CFL_DIST_BACKEND', 'nccl')
    ndev = torch.cuda.device_count()
    dev_index = local_rank if (backend == 'nccl' or local_rank < ndev) else local_rank % max(ndev, 1)
    torch.cuda.set_device(dev_index)
    device = torch.device('cuda', dev_index)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if backend == 'nccl':
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    if args.gpus != world and rank == 0 and world > 1:
        print('warning: --gpus %d but WORLD_SIZE %d' % (args.gpus, world), file=sys.stderr)

    _trace('process group ready (backend %s, device %s)' % (backend if world > 1 else '-', device))
    from cfl import hipabi as H
    from cfl.engine import PairEngine
    from oracle import cfl_oracle as O  # initial weights only (Xavier, seed 0)

    B, D, K, L = args.batch_size, args.input_size, args.num_components, args.latent_size
    cfg = O.EncoderCfg(D=D, L=L, K=K)
    params = O.init_encoder_params(cfg, np.random.RandomState(0), np.float32)
    eng = PairEngine(D, L, K, 'pcd', weight_norm=False, has_bias=True,
                     norm=H.make_norm(1.0 / NORMALIZE_VALUE), loss=H.make_loss(),
                     lr=1e-3, device=device, params=params, batch_size=B)

    batch_bytes = 4 * B * D * 4
    nb = max(2, (args.pool_mib * (1 << 20) + batch_bytes - 1) // batch_bytes)
    pool = make_pool(B, D, nb, device, seed=633 + rank)

    def run(nsteps, start):
        for i in range(nsteps):
            eng.step(pool[(start + i) % nb])

    _trace('pool ready (%d batches)' % nb)
    run(args.warmup, 0)
    torch.cuda.synchronize()
    _trace('warm-up done')
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(args.steps, args.warmup)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    _trace('timed region done: %.3f s' % elapsed)
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    scal = eng.read_scalars()

    out = None
    if rank == 0:
        rows = args.steps * B * world
        out = {
            'metric': 'triplets/sec',
            'value': round(rows / elapsed, 1),
            'unit': 'triplets/s',
            'n_gpus': world,
            'steps': args.steps,
            'warmup': args.warmup,
            'ms_per_step': round(1e3 * elapsed / args.steps, 5),
            'higher_is_better': True,
            'scaling': 'weak',
            'vs_baseline': None,
            'dtype': 'f32',
            'data': 'synthetic',
            'config': {
                'workload': 'Monomer-style 4096-d image features, PCD K=%d L=%d, batch %d rows '
                            '(4x[%d,%d] f32) per GPU per step, Dist model (FC heads+bias, '
                            'thr-BCE), fwd+bwd+TF-Adam' % (K, L, B, B, D),
                'input_size': D, 'num_components': K, 'latent_size': L,
                'batch_rows_per_gpu': B, 'global_batch_rows': B * world,
                'pool_mib_per_gpu': int(nb * batch_bytes >> 20),
                'parallelism': 'dp%d' % world if world > 1 else 'single',
                'final_loss': round(scal['total'], 6),
                'arithmetic': 'fp32 everywhere; the weight-gradient contraction runs on the bf16 matrix cores with '
                              'every fp32 operand split exactly in three bf16 values (six partial products, fp32 '
                              'accumulate; error vs fp64 equal to the fp32-MFMA kernel, tests/test_hip_parity.py); '
                              'CFL_EXACT_FP32=1 selects k-ordered fp32 MFMA for it',
            },
        }

    # ---- roofline of the dominant kernel: HIP events on the launch stream, a
    # second pass of the same K steps (events perturb the step time slightly, so
    # they are kept out of the pass that produces `value`) ----------------------
    # With N > 1 every step contains a collective, so EVERY rank runs these extra steps (a rank-0-only
    # pass would dead-lock in the all-reduce); only rank 0 records events and reports.
    if not args.no_kernel_profile and rank != 0 and world > 1:
        run(min(args.steps, 500), args.warmup + args.steps)
        run(100, 0)
        torch.cuda.synchronize()
    if rank == 0 and not args.no_kernel_profile:
        H.profile_enable(True)
        run(min(args.steps, 500), args.warmup + args.steps)
        torch.cuda.synchronize()
        H.profile_enable(False)
        prof = H.profile_read()
        # dispatch latency contained in every event interval: event-time 64-float Adam launches
        # queued behind real steps (so the queue never runs dry and the host is not the limiter)
        tiny = [torch.zeros(64, device=device) for _ in range(4)]
        run(100, 0)
        H.profile_enable(True)
        for _ in range(200):
            H.adam_tf(tiny[0], tiny[1], tiny[2], tiny[3], 1e-3, 0.9, 0.999)
        torch.cuda.synchronize()
        H.profile_enable(False)
        ov = H.profile_read().get('adam', (0.0, 1))
        event_overhead_us = round(1e3 * ov[0] / max(ov[1], 1), 3)
        kern = {k: {'avg_us': round(1e3 * ms / n, 3), 'launches': int(n)} for k, (ms, n) in prof.items()}
        dom = max(('proj', 'grad'), key=lambda k: prof.get(k, (0, 1))[0])
        raw_s = prof[dom][0] / prof[dom][1] * 1e-3
        # The event pairs perturb the stream: the intervals of one step add up to more than the step takes in
        # the timed region above (no events).  That excess is bracketing overhead; it is split evenly over the
        # step's launches and removed, which is what makes the figure agree with the rocprofv3 kernel durations
        # committed under profiles/ (single-process runs only: with N > 1 the step also contains the all-reduce).
        step_kernels = [k for k in ('colnorm', 'proj', 'mid', 'grad', 'finalize') if k in prof]
        sum_intervals_s = sum(prof[k][0] / prof[k][1] for k in step_kernels) * 1e-3
        excess_s = max(sum_intervals_s - elapsed / args.steps, 0.0) / len(step_kernels) if world == 1 else 0.0
        avg_s = max(raw_s - excess_s, 1e-9)
        alg_bytes = 16.0 * D * B                       # 4 fp32 vectors per row, read once
        alg_flops = 4.0 * D * L * (K + 1) * B          # one of fwd / dW: half of 8*D*L*(K+1)
        traffic = None
        tp = os.path.join(ROOT, 'profiles', 'traffic.json')
        if os.path.exists(tp):
            try:
                traffic = json.load(open(tp)).get(dom)
            except Exception:
                traffic = None
        out['roofline'] = {
            'kernel': ('cfl_grad_x3_kernel' if dom == 'grad' and os.environ.get('CFL_EXACT_FP32', '0') in ('', '0')
                       else 'cfl_%s_kernel' % dom),
            'bound': 'hbm',
            'achieved': round(alg_bytes / avg_s / 1e9, 1),
            'peak': HBM_PEAK_GBS,
            'unit': 'GB/s',
            'frac': round(alg_bytes / avg_s / 1e9 / HBM_PEAK_GBS, 4),
            'traffic': traffic,
            'avg_launch_us': round(avg_s * 1e6, 3),
            'avg_event_interval_us': round(raw_s * 1e6, 3),
            'event_excess_per_launch_us': round(excess_s * 1e6, 3),
            'algorithmic_bytes_per_launch': alg_bytes,
            'mfma_f32': {'achieved_tflops': round(alg_flops / avg_s / 1e12, 2),
                         'peak_tflops': FP32_MFMA_PEAK_TF,
                         'frac': round(alg_flops / avg_s / 1e12 / FP32_MFMA_PEAK_TF, 4)},
            'kernels': kern,
            'timing': 'hipEvent pairs around every launch on the launch stream, separate pass of %d steps (`kernels` '
                      'lists the raw intervals).  The intervals of a step sum to more than ms_per_step of the '
                      'event-free timed region; that excess (bracketing overhead) is split evenly over the launches '
                      'and subtracted: avg_launch_us = interval - excess.  Compare profiles/*kernel_stats.csv' %
                      min(args.steps, 500),
            'event_interval_of_64_float_kernel_us': event_overhead_us,
        }
    if rank == 0 and not args.no_cpu_baseline and world == 1:
        out['eval_auc'] = eval_auc(args, eng, device, seed=99)
        out['cpu_baseline'] = cpu_baseline(args, args.cpu_seconds)
    elif rank == 0:
        out['cpu_baseline'] = None
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
