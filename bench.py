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

N > 1: one process per GPU, weak scaling (512 rows per GPU): every step trains ONE
global batch of 512*N rows -- block r of global batch j is seeded by (j, r) and lives on
rank r -- with one RCCL all-reduce of the flat fp32 buffer [gradient | loss scalars] per
step and the Adam apply replicated.  `python bench.py --gpus N` starts its own N ranks
(a parent that never touches the GPU spawns `python -m torch.distributed.run ...` and
relays rank 0's line); under torchrun (WORLD_SIZE set) it is one of the ranks.

Timing: W untimed warm-up steps (at least one pass over the pool), then R repeats of
EXACTLY K steps, each repeat bracketed by barrier + synchronize on both sides and reduced
with MAX over ranks; `ms_per_step` / `value` come from the MEDIAN repeat (min / max are
in the line), so the timed work is ~6 s (--timed-seconds) whatever K is.

Prints ONE JSON line (rank 0) with the driver's contract fields plus `roofline`
(dominant kernel, HIP-event timed), `roofline_eval` (scoring kernels), `cli_loop`
(the cfl.bin.train_dist inner loop on a synthetic features.b) and `cpu_baseline`
(NumPy oracle on the host cores).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

# must be in the environment before the HIP runtime starts (multi-process RCCL needs dmabuf IPC on this pool)
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, 'compatibility-family-learning_amd')
for _p in (ROOT, PKG):
    if _p not in sys.path:
        sys.path.insert(0, _p)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
FP32_MFMA_PEAK_TF = 157.3  # dense fp32 MFMA peak (same guide)
BF16X3_PEAK_TF = 2500.0 / 6.0  # fp32-equivalent ceiling of the bf16x3 arithmetic: dense bf16 peak / six partial products
NORMALIZE_VALUE = 58.388599
NOISE = float(os.environ.get("CFL_BENCH_NOISE", "0.3"))  # target-side noise of the planted positives


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=50)
    ap.add_argument('--repeats', type=int, default=0, help='timed repeats of the K-step region (0 = auto: >= 50 '
                    'and enough for --timed-seconds of timed work)')
    ap.add_argument('--timed-seconds', type=float, default=7.0, help='target length of the timed headline region (auto '
                    'repeats): longer than the 5 s period of an external utilisation sampler')
    ap.add_argument('--batch-size', type=int, default=512)
    ap.add_argument('--input-size', type=int, default=4096)
    ap.add_argument('--num-components', type=int, default=3)
    ap.add_argument('--latent-size', type=int, default=20)
    ap.add_argument('--pool-mib', type=int, default=384,
                    help='resident batch pool per GPU (> 256 MiB Infinity Cache)')
    ap.add_argument('--cpu-seconds', type=float, default=10.0)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-kernel-profile', action='store_true')
    ap.add_argument('--no-cli-loop', action='store_true')
    ap.add_argument('--no-other-configs', action='store_true')
    ap.add_argument('--no-dp-form', action='store_true', help='skip the one-rank RCCL leg (dp_form)')
    ap.add_argument('--cli-items', type=int, default=40000, help='items of the synthetic features.b of the cli_loop leg')
    ap.add_argument('--cli-pairs', type=int, default=200000)
    ap.add_argument('--restore-steps', type=int, default=1500, help='the timed repeats start from the training state after this '
                    'many steps (restored before every repeat): the timed model is a model IN training, not a converged one')
    ap.add_argument('--dp-leg', default='', help='(internal) N > 1: run only the named data-parallel leg and print its JSON '
                    '(`oneshot`: the one-shot exchange at B = 512 and 2048 per GPU; started by rank 0 of the main job as a '
                    'child job so that a failure of that path cannot take the headline line down)')
    ap.add_argument('--no-live-traffic', action='store_true', help='quote profiles/traffic.json instead of measuring the HBM '
                    'traffic of the step with two rocprofv3 --pmc child runs')
    ap.add_argument('--dp-legs-timeout', type=int, default=900, help='N > 1: seconds after which the line is printed without '
                    'the dp_scaling legs (they wedged)')
    ap.add_argument('--no-dp-legs', action='store_true', help='N > 1: skip the extra data-parallel legs (dp_scaling)')
    return ap.parse_args()


# ---------------------------------------------------------------------------------------------------------
# N > 1 without a launcher: the parent starts the ranks and never initialises the GPU itself
# ---------------------------------------------------------------------------------------------------------
_live_children = []      # process groups started by spawn_ranks (the watchdog of the dp_scaling legs kills what is left of them)


def spawn_ranks(gpus, argv, key='"metric"', timeout=None, extra_env=None):
    """start `gpus` ranks of this script under torch.distributed.run (a fresh rendezvous port); returns (return code, the
    JSON line containing `key` or None).  The child's other output goes to stderr.  On a timeout the whole process group of
    the launcher is killed."""
    import signal
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ, MASTER_ADDR='127.0.0.1')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_PORT', 'GROUP_RANK', 'LOCAL_WORLD_SIZE', 'ROLE_RANK', 'ROLE_WORLD_SIZE',
              'TORCHELASTIC_RUN_ID', 'TORCHELASTIC_RESTART_COUNT', 'TORCHELASTIC_MAX_RESTARTS', 'TORCHELASTIC_USE_AGENT_STORE'):
        env.pop(k, None)                                  # (a child job started BY a rank must not inherit that rank's identity)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')     # dmabuf IPC: RCCL / cross-process tensor sharing needs it here
    env.update(extra_env or {})
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, start_new_session=True)
    _live_children.append(proc.pid)
    try:
        out, _ = proc.communicate(timeout=timeout)
    except subprocess.TimeoutExpired:
        try:
            os.killpg(proc.pid, signal.SIGKILL)          # exactly the process group this call started
        except OSError:
            pass
        out, _ = proc.communicate()
        _live_children.remove(proc.pid)
        print('bench.py: child job timed out after %s s' % timeout, file=sys.stderr)
        return 124, None
    _live_children.remove(proc.pid)
    line = None
    for ln in out.splitlines():
        if ln.startswith('{') and key in ln:
            line = ln
        else:
            print(ln, file=sys.stderr)
    return proc.returncode, line


def launch_ranks(args):
    rc, line = spawn_ranks(args.gpus, sys.argv[1:])
    if line is not None:
        print(line, flush=True)
    elif rc == 0:
        print('bench.py: the ranks printed no result line', file=sys.stderr)
        return 1
    return rc


# ---------------------------------------------------------------------------------------------------------
# synthetic data
# ---------------------------------------------------------------------------------------------------------
def teacher_of(D, device):
    """the hidden teacher is the same for every pool / rank / eval split"""
    import torch
    gt = torch.Generator(device=device)
    gt.manual_seed(20261002)
    teacher = torch.randn(D, 64, generator=gt, device=device) / D ** 0.5
    back = torch.randn(64, D, generator=gt, device=device) / 8.0
    return teacher, back


def make_block(B, D, device, seed, teacher):
    """One row block of a batch.  Synthetic post-ReLU-like CNN features: |N(0,1)| scaled so that max ~ the
    Monomer normalize_value; positive targets planted by a hidden linear teacher, negative
    targets planted from unrelated sources, so that both targets have the same marginal and only the pairing
    separates the classes."""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    s = NORMALIZE_VALUE / 4.5
    t, back = teacher
    r = lambda: torch.randn(B, D, generator=g, device=device)
    ps = r().abs_() * s
    pd = ((ps @ t) @ back + NOISE * s * r()).abs_()
    ns = r().abs_() * s
    other = r().abs_() * s
    nd = ((other @ t) @ back + NOISE * s * r()).abs_()
    return (ps.contiguous(), pd.contiguous(), ns.contiguous(), nd.contiguous())


def block_seed(batch_index, block):
    """block `block` (= rank) of global batch `batch_index`: the global batch is the concatenation of its blocks"""
    return 633 + 1000003 * batch_index + block


def init_params(D, L, K, seed=0):
    """Xavier-uniform weights, zero biases: the initial variables of `Dist` (cfl/models/dist.py:43-68)"""
    import numpy as np
    from cfl.models.base import xavier_uniform
    rng = np.random.RandomState(seed)
    return {'outputs/W': xavier_uniform(rng, D, L), 'outputs/b': np.zeros(L, np.float32),
            'proto/W': xavier_uniform(rng, D, L * K), 'proto/b': np.zeros(L * K, np.float32)}


def cpu_baseline(args, seconds):
    """The oracle (a NumPy port of the reference step: fwd + analytic bwd + TF-Adam,
    fp32) timed on this box's host cores on a bounded sample of the same workload."""
    import numpy as np
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


def cpu_loader_baseline(args, seconds=3.0, n_items=6000):
    """The reference's batch assembly timed beside the step (optional leg): next_labeled_batch reads every one
    of its 4 * B vectors with a seek + fromfile on features.b (cfl/input_data.py:212-228 through :542-589 of the reference),
    restated in oracle/loader_oracle.py.  Rows/s of that loader alone on a synthetic features.b (page cache warm: the best
    case for it) -- the reference's real bottleneck at the throughput the GPU step reaches."""
    import shutil
    import tempfile
    import numpy as np
    from oracle import loader_oracle as LO
    B, D = args.batch_size, args.input_size
    tmp = tempfile.mkdtemp(prefix='bench_loader_')
    try:
        path = os.path.join(tmp, 'features.b')
        rng = np.random.RandomState(0)
        LO.write_features(path, rng, n_items, D)
        pos = rng.randint(0, n_items, size=(64 * B, 2))
        neg = rng.randint(0, n_items, size=(64 * B, 2))
        LO.labeled_batch_by_seek(path, pos[:B], neg[:B], D)
        n, t0 = 0, time.perf_counter()
        while True:
            k = n % 64
            LO.labeled_batch_by_seek(path, pos[k * B:(k + 1) * B], neg[k * B:(k + 1) * B], D)
            n += 1
            el = time.perf_counter() - t0
            if el >= seconds and n >= 2:
                break
        return {'value': round(n * B / el, 1), 'unit': 'triplets/s', 'cores': 1, 'kind': 'port',
                'sample': '%d batches of %d rows (%.1f s): 4 x %d seek + read calls of %d bytes each per batch on a %d-item '
                          'features.b (page cache warm), oracle/loader_oracle.py' % (n, B, el, B, 4 * D, n_items)}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def eval_auc(args, eng, device, teacher):
    """The second half of BASELINE.json's metric: AUC of the just-trained weights on held-out synthetic pairs,
    scored by the HIP scoring path (cfl_pair_scores) and by the fp64 oracle on the same weights and pairs
    (a bounded sample: 2 x 2048 pairs).  The oracle is only the checker here."""
    import numpy as np
    from oracle import cfl_oracle as O
    from cfl import hipabi as H
    B, D = 2048, args.input_size
    ps, pd, ns, nd = make_block(B, D, device, 99, teacher)
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


def _rocprof_avg_us(kernel_name):
    """Average launch duration (us) of `kernel_name` in the committed rocprofv3 summary of this command
    (profiles/kernel_stats.csv = the latest profiles/r*_kernel_stats.csv, copied by tools/measure.sh): the figure the
    event-timed interval of this run is to be compared with."""
    import csv
    path = os.path.join(ROOT, 'profiles', 'kernel_stats.csv')
    try:
        with open(path) as f:
            for row in csv.DictReader(f):
                if row.get('Name', '').split('(')[0].strip() == kernel_name:
                    return round(float(row['AverageNs']) / 1e3, 3)
    except Exception:
        return None
    return None


def traffic_live(args, timeout=240):
    """HBM traffic per launch of the step's kernels MEASURED IN THIS RUN: two child runs of
    tools/kernel_probe.py (the same step on the same shapes) under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` -- separate
    passes, the program directly after `--`, no trace domains -- reduced as MI355X_MICROARCH.md's HBM section prescribes
    (2 x FETCH_SIZE + WRITE_SIZE in KiB: gfx950 tallies the 128-byte read requests at 64 bytes).  None when rocprofv3 is missing
    or a pass fails (the committed builder-box file is quoted then)."""
    import shutil
    import tempfile
    prof = shutil.which('rocprofv3') or ('/opt/rocm/bin/rocprofv3' if os.path.exists('/opt/rocm/bin/rocprofv3') else None)
    if prof is None:
        return None
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import pmc_traffic
    tmp = tempfile.mkdtemp(prefix='bench_pmc_', dir='/tmp')
    try:
        dirs = {}
        for counter in ('FETCH_SIZE', 'WRITE_SIZE'):
            d = os.path.join(tmp, counter)
            cmd = [prof, '--pmc', counter, '--output-format', 'csv', '-d', d, '-o', 'run', '--', sys.executable,
                   os.path.join(ROOT, 'tools', 'kernel_probe.py'), '--steps', '40', '--batch-size', str(args.batch_size),
                   '--input-size', str(args.input_size), '--num-components', str(args.num_components), '--latent-size',
                   str(args.latent_size)]
            r = subprocess.run(cmd, cwd='/tmp', env=dict(os.environ, TMPDIR='/tmp'), stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                               timeout=timeout)
            if r.returncode != 0:
                print('bench.py: rocprofv3 --pmc %s failed (rc %d): %s' % (counter, r.returncode, r.stderr[-300:]), file=sys.stderr)
                return None
            dirs[counter] = d
        fetch, nf = pmc_traffic.collect(dirs['FETCH_SIZE'], 'FETCH_SIZE')
        write, _ = pmc_traffic.collect(dirs['WRITE_SIZE'], 'WRITE_SIZE')
        if not fetch:
            return None
        return {k: int(round((2.0 * fetch[k] + write.get(k, 0.0)) * 1024)) for k in fetch}
    except Exception as e:          # noqa: BLE001 (a side measurement must not take the headline line down)
        print('bench.py: live PMC traffic: %r' % (e,), file=sys.stderr)
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def _eval_traffic():
    tp = os.path.join(ROOT, 'profiles', 'traffic_eval.json')
    try:
        return json.load(open(tp))
    except Exception:
        return None


def roofline_eval(args, eng, pool, device):
    """The scoring path (dist_eval / dist_predict: cfl_pair_scores, proj + mid): 8*D algorithmic bytes per scored
    pair (two fp32 vectors read once).  Whole-call throughput, HIP events on the launch stream around a train of
    scoring calls over the pool's source / target batches (the same HBM-resident rows the training legs use)."""
    import torch
    B, D = args.batch_size, args.input_size
    calls = [(b[0], b[1]) for b in pool] + [(b[2], b[3]) for b in pool]
    for xs, xt in calls[:4]:
        eng.scores(xs, xt)
    torch.cuda.synchronize()
    st = torch.cuda.current_stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = max(1, 2000 // len(calls))
    e0.record(st)
    for _ in range(reps):
        for xs, xt in calls:
            eng.scores(xs, xt)
    e1.record(st)
    torch.cuda.synchronize()
    n_calls = reps * len(calls)
    per_call_s = e0.elapsed_time(e1) * 1e-3 / n_calls
    alg = 8.0 * D * B
    # the same path at the call size dist_eval / dist_predict use (cfl.utils.RESIDENT_EVAL_ROWS pairs per call): two
    # sets of fresh rows (2 x 1.07 GB at the headline shape: far past the 256 MiB Infinity Cache), alternated
    from cfl.utils import RESIDENT_EVAL_ROWS
    big = None
    nbig = RESIDENT_EVAL_ROWS
    if nbig > B:
        g = torch.Generator(device=device)
        g.manual_seed(77)
        mk = lambda: torch.randn(nbig, D, generator=g, device=device).abs_().mul_(NORMALIZE_VALUE / 4.5)
        sets = [(mk(), mk()) for _ in range(2)]
        for xs, xt in sets:
            eng.scores(xs, xt)
        torch.cuda.synchronize()
        f0, f1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ncalls = 100
        f0.record(st)
        for i in range(ncalls):
            eng.scores(*sets[i & 1])
        f1.record(st)
        torch.cuda.synchronize()
        t = f0.elapsed_time(f1) * 1e-3 / ncalls
        big = {'pairs_per_call': nbig, 'avg_call_us': round(t * 1e6, 3), 'pairs_per_s': round(nbig / t, 1),
               'achieved': round(8.0 * D * nbig / t / 1e9, 1), 'frac': round(8.0 * D * nbig / t / 1e9 / HBM_PEAK_GBS, 4)}
        del sets
        eng._ws = {k: v for k, v in eng._ws.items() if k[1] != 1}
        torch.cuda.empty_cache()
    return {'kernels': 'cfl_proj_kernel + cfl_mid_row_kernel (one cfl_pair_scores call)', 'bound': 'hbm', 'dist_eval_call': big,
            'achieved': round(alg / per_call_s / 1e9, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
            'frac': round(alg / per_call_s / 1e9 / HBM_PEAK_GBS, 4), 'traffic': _eval_traffic(),
            'traffic_source': 'profiles/traffic_eval.json: builder box, rocprofv3 --pmc passes of tools/score_loop.py (one dist_eval-sized call: proj + mid); NOT measured in this run',
            'pairs_per_call': B, 'calls': n_calls, 'avg_call_us': round(per_call_s * 1e6, 3),
            'pairs_per_s': round(B / per_call_s, 1), 'algorithmic_bytes_per_call': alg}


def other_configs(device, args=None):
    """The other BASELINE.json configurations on the same kernels, measured by THIS run (so that the driver, not a
    builder-run probe, produces them): event-free step time over a pool larger than the Infinity Cache, median of 5
    repeats of 100 steps.  Weights are Xavier-uniform (RandomState(0)); inputs |N(0,1)| * 13.
      config3   experiments/dyadic/run.sh:40-50  1024-d, siamese L=256, weight-norm, hinge margin 100, pos_weight .0625, B=512
      config4   Polyvore: 2048-d, pcd K=5 L=20, weight-norm, pos_weight .25, B=1024
      reference experiments/monomer/run.sh:3-11 shape: 4096-d, pcd K=4 L=10, B=100 (the reference's own batch size)
      config5   MrCGAN post-epoch step, 64x64x3, latent 64, K=2, z=20, srgan, lambda_gp 0.5, B=100 (ms per step)"""
    import numpy as np
    import torch
    from cfl import hipabi as H
    from cfl.engine import PairEngine
    from cfl.models.base import xavier_uniform
    out = {}

    def params_of(D, L, K, dist, wn, rng):
        p = {'outputs/W': xavier_uniform(rng, D, L)}
        if dist != 'siamese':
            p['proto/W'] = xavier_uniform(rng, D, L * K)
        if wn:
            for k in list(p):
                p[k.replace('/W', '/g')] = np.ones(p[k].shape[1], np.float32)
        if (dist == 'pcd') or not wn:
            for k in [k for k in p if k.endswith('/W')]:
                p[k.replace('/W', '/b')] = np.zeros(p[k].shape[1], np.float32)
        return p

    g = torch.Generator(device=device)
    g.manual_seed(5)
    for name, D, L, K, dist, wn, B, lkw, nv in (
            ('config3_siamese_hinge', 1024, 256, 1, 'siamese', True, 512,
             dict(use_threshold=False, caffe_margin=100.0, pos_weight=0.0625), 31.9098),
            ('config4_polyvore_pcd_k5', 2048, 20, 5, 'pcd', True, 1024, dict(pos_weight=0.25), 1.0),
            ('reference_shape_k4_l10_b100', 4096, 10, 4, 'pcd', False, 100, dict(), NORMALIZE_VALUE)):
        try:
            eng = PairEngine(D, L, K, dist, weight_norm=wn, has_bias=(dist == 'pcd') or not wn,
                             norm=H.make_norm(1.0 / nv), loss=H.make_loss(**lkw), lr=1e-3, device=device,
                             params=params_of(D, L, K, dist, wn, np.random.RandomState(0)),
                             thr=30.0 if dist == 'siamese' else 1e-6, batch_size=B)
            nb = max(2, (320 << 20) // (16 * B * D))
            pool = [tuple(torch.randn(B, D, generator=g, device=device).abs_() * (nv / 4.5) for _ in range(4))
                    for _ in range(nb)]
            for i in range(max(30, nb)):
                eng.step(pool[i % nb])
            torch.cuda.synchronize()
            ts = []
            for r in range(5):
                t0 = time.perf_counter()
                for i in range(100):
                    eng.step(pool[(r * 100 + i) % nb])
                torch.cuda.synchronize()
                ts.append((time.perf_counter() - t0) / 100)
            t = float(np.median(ts))
            H.profile_enable(True)
            for i in range(50):
                eng.step(pool[i % nb])
            torch.cuda.synchronize()
            H.profile_enable(False)
            prof = H.profile_read()
            # algorithmic flops per row: forward 2 D N + weight gradient 2 D N per side, N = head columns;
            # siamese: both sides through ONE head of L columns (16 D L), pcd: L + K L columns over the two sides (8 D L (K+1))
            flops_row = 16.0 * D * L if dist == 'siamese' else 8.0 * D * L * (K + 1)
            tf = flops_row * B / t / 1e12
            plan = H.plan_describe(eng.shape, B, 2, True, True)
            out[name] = {'us_per_step': round(t * 1e6, 2), 'rows_per_s': round(B / t, 1), 'batch_rows': B,
                         'launches_per_step': len(prof), 'hbm_frac_step': round(16.0 * D * B / t / 1e9 / HBM_PEAK_GBS, 4),
                         # fp32-equivalent matrix work of the step against the fp32-MFMA roof and against the ceiling of the
                         # bf16x3 arithmetic the kernels actually issue (2.5 PF bf16 / 6 partial products)
                         'mfma_tflops_step': round(tf, 2), 'mfma_frac_step': round(tf / FP32_MFMA_PEAK_TF, 4),
                         'bf16x3_frac_step': round(tf / BF16X3_PEAK_TF, 4),
                         'binding_roof': 'mfma' if tf / FP32_MFMA_PEAK_TF > 16.0 * D * B / t / 1e9 / HBM_PEAK_GBS else 'hbm',
                         'kernels': [plan['proj'], plan['mid'], plan['grad']],
                         'final_loss': round(eng.read_scalars()['total'], 6)}
            del pool, eng
            torch.cuda.empty_cache()
        except Exception as e:          # a side measurement must not take the headline line down
            out[name] = {'error': repr(e)}
    try:
        from cfl.models.mrcgan import GanPhase
        B, L, zd, shape = 100, 64, 20, (64, 64, 3)
        ph = GanPhase('srgan', shape, 'tanh', zd, L, B, device, np.random.RandomState(0), lambda_gp=0.5, lambda_dra=0.5,
                      m_enc=0.05, m_prj=0.2)
        N = int(np.prod(shape))
        batch = [torch.tanh(torch.randn(B, N, device=device, generator=g))] + \
                [0.3 * torch.randn(B, L, device=device, generator=g) for _ in range(4)] + \
                [torch.randn(B, zd, device=device, generator=g), torch.rand(B, 1, device=device, generator=g)]
        for _ in range(3):
            ph.step(*batch)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 10
        for _ in range(n):
            ph.step(*batch)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        from cfl import hipgan
        hipgan.flop_counter = [0]
        ph.step(*batch)
        torch.cuda.synchronize()
        flops, hipgan.flop_counter = float(hipgan.flop_counter[0]), None
        out['config5_mrcgan_64x64_b100'] = {
            'ms_per_step': round(dt * 1e3, 3), 'images_per_s': round(B / dt, 1),
            # analytic flops of every convolution product the step issues (forward / input gradient / weight gradient of
            # the batched G and D passes and the gradient-penalty double backward; counted by cfl.hipgan as it calls the
            # library) over the step time: the conv stacks are matrix-bound
            'roofline': {'bound': 'mfma', 'flops_per_step': flops, 'achieved': round(flops / dt / 1e12, 1), 'unit': 'TFLOP/s',
                         'peak_fp32_mfma': FP32_MFMA_PEAK_TF, 'frac_fp32_mfma': round(flops / dt / 1e12 / FP32_MFMA_PEAK_TF, 4),
                         'peak_bf16x3': round(BF16X3_PEAK_TF, 1), 'frac_bf16x3': round(flops / dt / 1e12 / BF16X3_PEAK_TF, 4)}}
        del ph
        torch.cuda.empty_cache()
    except Exception as e:
        out['config5_mrcgan_64x64_b100'] = {'error': repr(e)}
    if 'error' not in out.get('config4_polyvore_pcd_k5', {'error': 1}) and not getattr(args, 'no_cli_loop', False):
        try:
            cl = cfl_cli_loop()
            cl['vs_step'] = round(out['config4_polyvore_pcd_k5']['us_per_step'] / cl['us_per_iteration'], 4)
            out['config4_polyvore_pcd_k5']['cli_loop'] = cl
        except Exception as e:
            out['config4_polyvore_pcd_k5']['cli_loop'] = {'error': repr(e)}
    if 'error' not in out['config5_mrcgan_64x64_b100'] and not getattr(args, 'no_cli_loop', False):
        try:
            out['config5_mrcgan_64x64_b100']['cli_loop'] = gan_cli_loop()
        except Exception as e:
            out['config5_mrcgan_64x64_b100']['cli_loop'] = {'error': repr(e)}
    return out


def cfl_cli_loop(n_items=8192, n_pairs=204800):
    """The distance epochs of `cfl.bin.train` -- the CFL model class of configs 3 / 4 and of every dyadic experiment of the reference
    -- END TO END at the config-4 shape (2048-d vectors, pcd K = 5, L = 20, weight-norm heads, pos_weight 0.25, B = 1024) on a
    synthetic vector dataset in the reference's format: three epochs through the CLI, us per training iteration of the LAST two
    (the epoch loop itself: seeded index streams, windows of the device pair lists, the fused multi-iteration library call,
    read-backs every 50 iterations; per-epoch checkpoints are outside the timed part).  tools/double_epoch_probe.py is the
    image + latent form (profiles/r05_cfl_epoch_loop.txt)."""
    import contextlib
    import shutil
    import tempfile
    import torch
    from cfl.bin import train, train_dist
    from cfl.synthetic import make_dataset
    tmp = tempfile.mkdtemp(prefix='bench_cfl_')
    calls = []
    orig = train_dist.train_steps

    def timed(model, train_src, val_src, batch_size, shard, n_steps, *a, **k):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = orig(model, train_src, val_src, batch_size, shard, n_steps, *a, **k)
        torch.cuda.synchronize()
        calls.append((n_steps, time.perf_counter() - t0))
        return r
    try:
        train_dist.train_steps = timed
        with contextlib.redirect_stdout(sys.stderr):
            root = os.path.join(tmp, 'data')
            make_dataset(os.path.join(root, 'poly'), D=2048, n_items=n_items, n_pos=n_pairs, n_neg=n_pairs, k=5, latent=20, seed=7)
            train.main(['--data-name', 'poly', '--data-root', root, '--checkpoint-root', os.path.join(tmp, 'ck'), '--log-root',
                        os.path.join(tmp, 'logs'), '--model-type', 'linear', '--data-type', 'linear', '--data-norm', '58.388599',
                        '--input-shape', '2048', '--dist-type', 'pcd', '--use-threshold', '--pos-weight', '0.25',
                        '--num-components', '5', '--latent-size', '20', '--batch-size', '1024', '--seed', '3', '--disable-eval',
                        '--reset', '--epochs', '3'])
    finally:
        train_dist.train_steps = orig
        shutil.rmtree(tmp, ignore_errors=True)
    if len(calls) < 2:
        raise RuntimeError('cfl.bin.train did not take the fused epoch loop (%d calls)' % len(calls))
    n = sum(c[0] for c in calls[1:])
    return {'us_per_iteration': round(1e6 * sum(c[1] for c in calls[1:]) / n, 3), 'iterations': n,
            'first_epoch_us_per_iteration': round(1e6 * calls[0][1] / calls[0][0], 3),
            'what': 'cfl.bin.train (CFL model class) distance epochs through the fused multi-iteration loop on resident features: '
                    'epochs 2 and 3 of a 3-epoch run, the epoch loop alone (checkpoints excluded)'}


def gan_cli_loop(n_items=2000, n_pairs=3000):
    """The MrCGAN post-epoch loop END TO END through the CLI (experiments/dyadic/run_gen.sh of the reference in synthetic form:
    image + latent records, 64x64x3 PNGs + 1024-d latents, L = 64, K = 2, B = 100, srgan, lambda_gp 0.5): one distance epoch,
    then three post epochs; ms per iteration of the LAST one, including batch assembly (record table, pinned uploads) and the
    loop's read-backs (tools/gan_e2e_probe.py is the stand-alone form; profiles/r05_gan_e2e_loop.txt the ladder)."""
    import contextlib
    import shutil
    import tempfile
    import torch
    from cfl.bin import train
    from cfl.models import cfl as M
    from cfl.synthetic import make_double_dataset
    tmp = tempfile.mkdtemp(prefix='bench_gan_')
    acc = {'epoch': 0.0, 'n': 0}
    orig_epoch, orig_step = M.CFL._post_epoch, M.CFL.post_step

    def post_step(self, *a, **k):
        acc['n'] += 1
        return orig_step(self, *a, **k)

    def post_epoch(self, *a, **k):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n0 = acc['n']
        r = orig_epoch(self, *a, **k)
        torch.cuda.synchronize()
        # the LAST post epoch counts (the first one also decodes every record once and tunes the step's stream placement)
        acc['epoch'], acc['n_last'] = time.perf_counter() - t0, acc['n'] - n0
        return r
    try:
        with contextlib.redirect_stdout(sys.stderr):       # (stdout carries the JSON line only)
            root = os.path.join(tmp, 'data')
            make_double_dataset(os.path.join(root, 'dy'), image_shape=(64, 64, 3), latent_dim=1024, n_items=n_items,
                                n_pos=n_pairs, n_neg=n_pairs, k=2, seed=5)
            base = ['--data-name', 'dy', '--data-root', root, '--checkpoint-root', os.path.join(tmp, 'ck'), '--log-root',
                    os.path.join(tmp, 'logs'), '--model-type', 'linear', '--data-type', 'tanh', '--data-mean', '0.5',
                    '--data-norm', '0.5', '--data-directed', '--latent-norm', '31.9098', '--data-is-image', '--data-is-double',
                    '--raw-latent', '--latent-shape', '1024', '--input-shape', '64', '64', '3', '--dist-type', 'pcd',
                    '--lambda-m', '0.5', '--use-threshold', '--num-components', '2', '--latent-size', '64', '--batch-size', '100',
                    '--seed', '3']
            train.main(base + ['--epochs', '1', '--reset'])
            M.CFL._post_epoch, M.CFL.post_step = post_epoch, post_step
            gan = ['--m-prj', '0.2', '--m-enc', '0.05', '--d-lr', '0.0002', '--d-beta1', '0.5', '--g-lr', '0.0002', '--g-beta1',
                   '0.5', '--gan', '--gan-type', 'srgan', '--lambda-gp', '0.5']
            train.main(base + gan + ['--load-pre-weights', '--epochs', '1', '--post-epochs', '3', '--disable-eval'])
    finally:
        M.CFL._post_epoch, M.CFL.post_step = orig_epoch, orig_step
        shutil.rmtree(tmp, ignore_errors=True)
    n = max(acc.get('n_last', 0), 1)
    return {'ms_per_iteration': round(1e3 * acc['epoch'] / n, 3), 'iterations': n, 'images': n_items, 'pairs': 2 * n_pairs,
            'what': 'cfl.bin.train --gan --post-epochs 3 (the last epoch timed) on an image + latent dataset in the reference record format (batch '
                    'assembly from the decoded-record table, pinned uploads, the GPU step, read-backs every 20 iterations)'}


def cli_loop(args, device, B=None, K=None, L=None):
    """steps/s of the inner loop of `python -m cfl.bin.train_dist` (cfl.bin.train_dist.train_steps: the dataset's
    seeded index stream -> positions into the HBM-resident features.b -> one fused step, scalars every 25
    iterations) on a synthetic dataset in the reference's on-disk format written for this run.  B / K / L override the
    headline's batch size / prototypes / latent size (the reference's own scripts: B = 100, K = 4, L = 10)."""
    import copy
    args = copy.copy(args)
    if B is not None:
        args.batch_size, args.num_components, args.latent_size = B, K, L
    import shutil
    import tempfile
    import torch
    from cfl.bin.train_dist import train_steps
    from cfl.input_data import ResidentFeatures, load_data_sets
    from cfl.models.dist import construct_model
    from cfl.ops import normalizer, unnormalizer
    from cfl.synthetic import make_dataset
    from cfl.engine import quiet_host_threads
    B, D = args.batch_size, args.input_size
    root = tempfile.mkdtemp(prefix='cfl_bench_')
    try:
        quiet_host_threads()            # what the command-line entry point does first (engine.init_from_env)
        t0 = time.perf_counter()
        make_dataset(os.path.join(root, 'syn'), D=D, n_items=args.cli_items, n_pos=args.cli_pairs,
                     n_neg=args.cli_pairs, splits=(('train', 1.0), ('val', 0.1), ('test', 0.02)))
        data = load_data_sets(os.path.join(root, 'syn'), D, seed=633)
        model, aux = construct_model(input_shape=(D,), latent_size=args.latent_size, normalize_value=NORMALIZE_VALUE,
                                     lr=1e-3, beta1=0.9, beta2=0.999, num_components=args.num_components,
                                     batch_size=B, data=data, reg_const=0.0,
                                     data_normalizer=normalizer(NORMALIZE_VALUE, 0., None, None),
                                     data_unnormalizer=unnormalizer(NORMALIZE_VALUE, 0.), seed=633, device=device)
        train_src, val_src = ResidentFeatures(aux.train, device), ResidentFeatures(aux.val, device)
        prep_s = time.perf_counter() - t0
        seen = []
        on_scalars = lambda i, s, v: seen.append((i, s['total'], v))
        train_steps(model, train_src, val_src, B, None, 200, on_scalars)
        torch.cuda.synchronize()
        n = 3000
        t0 = time.perf_counter()
        train_steps(model, train_src, val_src, B, None, n, on_scalars)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        # the reference's cadence (--scalar-every 1): a validation batch scored and the scalars read back every iteration
        n1 = 1000
        train_steps(model, train_src, val_src, B, None, 50, on_scalars, scalar_every=1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        train_steps(model, train_src, val_src, B, None, n1, on_scalars, scalar_every=1)
        torch.cuda.synchronize()
        el1 = time.perf_counter() - t0
        # the same loop with the host side only (index stream + call set-up, no kernels awaited): is the loop
        # GPU-bound or host-bound?
        t0 = time.perf_counter()
        for _ in range(n):
            train_src.next_indexed(B)
        host_idx_s = (time.perf_counter() - t0) / n
        return {'steps_per_s': round(n / el, 1), 'us_per_step': round(1e6 * el / n, 3),
                'triplets_per_s': round(n * B / el, 1), 'steps': n,
                'every1': {'us_per_step': round(1e6 * el1 / n1, 3), 'triplets_per_s': round(n1 * B / el1, 1), 'steps': n1,
                           'what': '--scalar-every 1: the reference cadence, one validation fetch per iteration '
                                   '(cfl/bin/train_dist.py:79-86 of the reference)'},
                'table_mib': int(train_src.table.numel() * 4 >> 20), 'pairs': int(aux.train.pairs_pos.shape[0]),
                'host_index_stream_us_per_step': round(1e6 * host_idx_s, 3),
                'dataset_prep_s': round(prep_s, 1), 'final_loss': round(float(seen[-1][1]), 6),
                'loop': 'cfl.bin.train_dist.train_steps (next_indexed -> cfl_pair_train_step_idx; scalars + a '
                        'validation batch every 25 steps)'}
    finally:
        shutil.rmtree(root, ignore_errors=True)


def make_engine(args, device, B):
    from cfl import hipabi as H
    from cfl.engine import PairEngine
    D, K, L = args.input_size, args.num_components, args.latent_size
    return PairEngine(D, L, K, 'pcd', weight_norm=False, has_bias=True, norm=H.make_norm(1.0 / NORMALIZE_VALUE),
                      loss=H.make_loss(), lr=1e-3, device=device, params=init_params(D, L, K), batch_size=B)


def make_pool(args, device, B, rank, teacher, pool_mib=None):
    """this rank's row block of every global batch of a pool larger than the Infinity Cache"""
    batch_bytes = 4 * B * args.input_size * 4
    nb = max(2, ((pool_mib or args.pool_mib) * (1 << 20) + batch_bytes - 1) // batch_bytes)
    return [make_block(B, args.input_size, device, block_seed(j, rank), teacher) for j in range(nb)]


def snapshot(eng):
    """the training state of an engine (parameters, Adam slots, kept planes, Adam powers): what `restore` puts back before
    every timed repeat, so that the timed steps are those of a model IN training (197 k steps on a 12-batch
    pool had driven the loss to 0.0 -- saturated dL/dY is not the state any user trains in)"""
    return dict(theta=eng.theta.clone(), m=eng.m.clone(), v=eng.v.clone(), planes=eng.planes.buf.clone(),
                valid=int(eng.planes.c.valid), b1=eng.beta1_power, b2=eng.beta2_power, gs=eng.global_step)


def restore(eng, snap):
    eng.theta.copy_(snap['theta']); eng.m.copy_(snap['m']); eng.v.copy_(snap['v'])
    eng.planes.buf.copy_(snap['planes'])
    eng.planes.c.valid = snap['valid']
    eng.beta1_power, eng.beta2_power, eng.global_step = snap['b1'], snap['b2'], snap['gs']


def dp_leg(args, device, rank, world, B, exchange, teacher, seconds=None, pool_mib=None):
    """One data-parallel configuration, every rank: `exchange` = allreduce (RCCL's ncclAllReduce called by the library on the
    launch stream) | oneshot (reduce-scatter fused into the weight-gradient launch, sharded Adam, all-gather) | none (the
    same launches and the stand-alone Adam, NO collective: what a rank's step costs before a byte crosses a link).  Median of
    >= 10 repeats of args.steps steps, each bracketed by barrier + synchronize, MAX over ranks; restored to the early-training
    state before every repeat."""
    import numpy as np
    import torch
    import torch.distributed as dist
    seconds = float(os.environ.get('CFL_BENCH_LEG_SECONDS', '2.0')) if seconds is None else seconds
    os.environ['CFL_DP_EXCHANGE'] = 'oneshot' if exchange == 'oneshot' else 'allreduce'
    os.environ['CFL_DP_NO_COLLECTIVE'] = '1' if exchange == 'none' else '0'
    try:
        eng = make_engine(args, device, B)
        pool = make_pool(args, device, B, rank, teacher, pool_mib)
        nb = len(pool)

        def sync_all():
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
                torch.cuda.synchronize()
        for i in range(min(args.restore_steps, 300)):     # (side legs: a shorter run-in)
            eng.step(pool[i % nb])
        snap = snapshot(eng)
        for i in range(max(nb, 20)):
            eng.step(pool[i % nb])
        sync_all()
        t0 = time.perf_counter()
        for i in range(args.steps):
            eng.step(pool[i % nb])
        sync_all()
        est = max(time.perf_counter() - t0, 1e-6)
        rt = torch.tensor([int(min(2000, max(10, np.ceil(seconds / est))))], device=device, dtype=torch.int64)
        if world > 1:
            dist.broadcast(rt, 0)
        times, pos = [], 0
        for _ in range(int(rt.item())):
            restore(eng, snap)
            sync_all()
            t0 = time.perf_counter()
            for i in range(args.steps):
                eng.step(pool[(pos + i) % nb])
            sync_all()
            times.append(time.perf_counter() - t0)
            pos += args.steps
        tt = torch.tensor(times, device=device, dtype=torch.float64)
        if world > 1:
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        el = float(np.median(tt.cpu().numpy()))
        sc = eng.read_scalars()
        lost = int(eng._oneshot.lost.item()) if eng._oneshot is not None else 0
        native = eng.dp_native()
        res = {'us_per_step': round(1e6 * el / args.steps, 3), 'triplets_per_s': round(args.steps * B * world / el, 1),
               'rows_per_gpu': B, 'repeats': len(times), 'final_loss': round(sc['total'], 6),
               'driven_by': 'library (one call per step)' if native is not None else 'python (torch.distributed.all_reduce)'}
        if exchange == 'oneshot':
            res['lost_handoffs'] = lost
        del pool, eng
        torch.cuda.empty_cache()
        return res
    finally:
        os.environ['CFL_DP_EXCHANGE'] = 'allreduce'
        os.environ['CFL_DP_NO_COLLECTIVE'] = '0'


def gan_dp_leg(device, rank, world, n=10):
    """MrCGAN post-epoch step (config 5: 64x64x3, L = 64, z = 20, srgan) under data parallelism, weak scaling: every rank runs the
    G / D step on ITS 100 rows of a global batch of 100 x world, then ONE all-reduce of [d gradient | g gradient | scalars] in front
    of the two Adams (cfl.models.mrcgan.GanPhase.shard_over) -- and the same step without the collective beside it."""
    import numpy as np
    import torch
    import torch.distributed as dist
    from cfl import engine
    from cfl.models.mrcgan import GanPhase
    B, L, zd, shape = 100, 64, 20, (64, 64, 3)
    g = torch.Generator(device=device)
    g.manual_seed(77 + rank)
    ph = GanPhase('srgan', shape, 'tanh', zd, L, B, device, np.random.RandomState(0), lambda_gp=0.5, lambda_dra=0.5, m_enc=0.05,
                  m_prj=0.2)
    N = int(np.prod(shape))
    batch = [torch.tanh(torch.randn(B, N, device=device, generator=g))] + \
            [0.3 * torch.randn(B, L, device=device, generator=g) for _ in range(4)] + \
            [torch.randn(B, zd, device=device, generator=g), torch.rand(B, 1, device=device, generator=g)]

    def timed():
        def sync_all():
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
                torch.cuda.synchronize()
        for _ in range(3):
            ph.step(*batch)
        sync_all()
        t0 = time.perf_counter()
        for _ in range(n):
            ph.step(*batch)
        sync_all()
        tt = torch.tensor([time.perf_counter() - t0], device=device, dtype=torch.float64)
        if world > 1:
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return float(tt.item()) / n
    alone = timed()
    ph.shard_over(engine.reduce_gradients)
    dt = timed()
    sc = ph.read_scalars()
    res = {'ms_per_step': round(dt * 1e3, 3), 'images_per_s': round(B * world / dt, 1), 'rows_per_gpu': B,
           'ms_per_step_without_collective': round(alone * 1e3, 3), 'all_reduce_mb': round(ph._flat.numel() * 4 / 1e6, 2),
           'd_total_loss': round(sc['d_total_loss'], 5), 'direct_rccl': engine.hot_communicator() is not None}
    del ph
    torch.cuda.empty_cache()
    return res


def dp_form(args, eng, pool, device, fused_us):
    """The DATA-PARALLEL form of the step on this one GPU, through a ONE-RANK RCCL process group (CFL_FORCE_DP=1): the
    same three launches as the fused step (projection on the bf16 matrix cores from the kept planes, row math, weight
    gradient emitting the flat gradient), then `all_reduce([gradient | scalars])` on the launch stream and the stand-alone
    TF-Adam that re-writes the planes.  What a rank of an N-GPU job executes per step, minus the wire time: the one-GPU
    cost of data parallelism (extra launches + RCCL's fixed cost), measured every round on the hardware the driver has."""
    import numpy as np
    import torch
    import torch.distributed as dist
    from cfl import hipabi as H
    own_group = not dist.is_initialized()
    os.environ['CFL_FORCE_DP'] = '1'
    try:
        if own_group:
            with socket.socket() as s:
                s.bind(('127.0.0.1', 0))
                port = s.getsockname()[1]
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
            os.environ['MASTER_PORT'] = str(port)
            dist.init_process_group('nccl', rank=0, world_size=1, device_id=device)
        nb = len(pool)
        for i in range(50):
            eng.step(pool[i % nb])
        torch.cuda.synchronize()
        ts = []
        for r in range(20):
            t0 = time.perf_counter()
            for i in range(args.steps):
                eng.step(pool[(r * args.steps + i) % nb])
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) / args.steps)
        t = float(np.median(ts))
        from cfl import engine as E
        # the same step with the collective issued through torch.distributed (ProcessGroupNCCL's own stream + two events)
        os.environ['CFL_DP_ALLREDUCE'] = 'torch'
        try:
            for i in range(20):
                eng.step(pool[i % nb])
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(5 * args.steps):
                eng.step(pool[i % nb])
            torch.cuda.synchronize()
            t_torch = (time.perf_counter() - t0) / (5 * args.steps)
        finally:
            os.environ.pop('CFL_DP_ALLREDUCE', None)
        # without the collective: the same launches, no RCCL kernel between them
        eng_reduce = None
        eng_reduce, E.reduce_gradients = E.reduce_gradients, (lambda buf: 1.0)
        os.environ['CFL_DP_NO_COLLECTIVE'] = '1'      # (the library-driven step: same launches, no ncclAllReduce between them)
        try:
            for i in range(20):
                eng.step(pool[i % nb])
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(5 * args.steps):
                eng.step(pool[i % nb])
            torch.cuda.synchronize()
            t_nocoll = (time.perf_counter() - t0) / (5 * args.steps)
        finally:
            E.reduce_gradients = eng_reduce
            os.environ['CFL_DP_NO_COLLECTIVE'] = '0'
        H.profile_enable(True)
        for i in range(50):
            eng.step(pool[i % nb])
        torch.cuda.synchronize()
        H.profile_enable(False)
        prof = H.profile_read()
        # the ONE-SHOT exchange on the same one-rank group (round 6: the reduce-scatter rides in the weight-gradient launch; the
        # step is proj, mid, grad(+push), cfl_dp_rs_adam, cfl_dp_rs_gather_planes = five launches, no collective library)
        oneshot = None
        try:
            os.environ['CFL_DP_EXCHANGE'] = 'oneshot'
            eng1 = make_engine(args, device, args.batch_size)
            for i in range(50):
                eng1.step(pool[i % nb])
            torch.cuda.synchronize()
            t1 = []
            for r in range(10):
                t0 = time.perf_counter()
                for i in range(args.steps):
                    eng1.step(pool[(r * args.steps + i) % nb])
                torch.cuda.synchronize()
                t1.append((time.perf_counter() - t0) / args.steps)
            H.profile_enable(True)
            for i in range(50):
                eng1.step(pool[i % nb])
            torch.cuda.synchronize()
            H.profile_enable(False)
            p1 = H.profile_read()
            oneshot = {'us_per_step': round(float(np.median(t1)) * 1e6, 3),
                       'launches_per_step': int(sum(c for _, c in p1.values()) // 50),
                       'library_launches': {k: int(c) // 50 for k, (ms, c) in p1.items()},
                       'lost_handoffs': int(eng1._oneshot.lost.item()),
                       'what': 'CFL_DP_EXCHANGE=oneshot: proj_bx3 -> mid -> grad (finished entries pushed into the owner\'s slots, '
                               'arrival flags raised by its last workgroup) -> sharded Adam + all-gather + planes in one launch '
                               '(cfl_dp_adam_gather_kernel)'}
            del eng1
        except Exception as e:          # noqa: BLE001
            oneshot = {'error': repr(e)}
        finally:
            os.environ['CFL_DP_EXCHANGE'] = 'allreduce'
        return {'us_per_step': round(t * 1e6, 3), 'us_per_step_without_collective': round(t_nocoll * 1e6, 3),
                'us_per_step_torch_all_reduce': round(t_torch * 1e6, 3),
                'fused_us_per_step': round(fused_us, 3), 'vs_fused': round(fused_us / (t * 1e6), 4),
                'library_launches_per_step': sorted(prof), 'backend': dist.get_backend(), 'world': dist.get_world_size(),
                'launches_per_step': int(sum(c for _, c in prof.values()) // 50) + 1,
                'exchange': 'allreduce', 'oneshot': oneshot,
                'driven_by': 'library: one cfl_pair_dp_step_planes call per step (ABI 6)' if eng.dp_native() is not None else 'python',
                'what': 'PairEngine.step through its data-parallel branch on a one-rank RCCL group: ONE library call = proj_bx3 -> mid '
                        '-> grad -> ncclAllReduce([gradient | scalars], %d floats, called by the library on the launch stream: '
                        'cfl/rccl.py hands over the entry point) -> cfl_adam_tf_planes' % eng.gradbuf.numel()}
    finally:
        os.environ['CFL_FORCE_DP'] = '0'
        if own_group and dist.is_initialized():
            from cfl import rccl
            rccl.shutdown()
            dist.destroy_process_group()


def _trace(msg):
    if os.environ.get('CFL_BENCH_TRACE'):
        print('[bench rank %s] %s' % (os.environ.get('RANK', '0'), msg), file=sys.stderr, flush=True)


def main():
    args = parse()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        return launch_ranks(args)

    # ONE JSON line on stdout, whatever the libraries below print there (RCCL writes its version banner to stdout when a
    # communicator is created): keep the real stdout aside and point fd 1 at stderr for the rest of the run
    real_stdout = os.dup(1)
    sys.stdout.flush()
    os.dup2(2, 1)

    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus != world:
        print('bench.py: --gpus %d but WORLD_SIZE %d' % (args.gpus, world), file=sys.stderr)
        return 2
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
    ranks_seen = dist.get_world_size() if world > 1 else 1

    _trace('process group ready (backend %s, device %s)' % (backend if world > 1 else '-', device))
    from cfl import hipabi as H

    B, D, K, L = args.batch_size, args.input_size, args.num_components, args.latent_size
    teacher = teacher_of(D, device)
    if args.dp_leg:
        # child job of the main N-rank run (started by its rank 0): ONE risky leg, its own JSON line, nothing else
        if os.environ.get('CFL_BENCH_FAIL_DP_LEG') == '1':      # (tests: a child job that dies must cost an `error` entry, not the line)
            os._exit(3)
        if os.environ.get('CFL_BENCH_FAIL_DP_LEG') == 'hang':   # (tests: a child job that wedges -- the parent's watchdog prints the line)
            time.sleep(25)
            os._exit(3)
        res = {'dp_leg': args.dp_leg}
        if args.dp_leg == 'oneshot':
            for b in (B, 2048):
                try:
                    res['b%d' % b] = dp_leg(args, device, rank, world, b, 'oneshot', teacher)
                except Exception as e:          # noqa: BLE001
                    res['b%d' % b] = {'error': repr(e)}
        if rank == 0:
            sys.stdout.flush()
            os.write(real_stdout, (json.dumps(res) + '\n').encode())
        if world > 1:
            dist.barrier()
            from cfl import rccl
            rccl.shutdown()
            dist.destroy_process_group()
        return 0
    eng = make_engine(args, device, B)
    batch_bytes = 4 * B * D * 4
    # this rank's row block of every global batch of the pool
    pool = make_pool(args, device, B, rank, teacher)
    nb = len(pool)

    def run(nsteps, start):
        for i in range(nsteps):
            eng.step(pool[(start + i) % nb])

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    _trace('pool ready (%d batches)' % nb)
    # the state every timed repeat starts from: `--restore-steps` steps into training (restored OUTSIDE the timed bracket)
    run(args.restore_steps, 0)
    snap = snapshot(eng)
    loss_at_restore = eng.read_scalars()['total']
    warm = max(args.warmup, nb)          # every pool batch is touched before anything is timed
    run(warm, args.restore_steps)
    sync_all()
    _trace('warm-up done')
    # short calibration (untimed for the result): how many repeats make --timed-seconds (6 s) of timed work -- longer than the
    # 5 s period of an external utilisation sampler, so that it sees the GPU busy; the median repeat is what is reported
    t0 = time.perf_counter()
    run(args.steps, warm)
    sync_all()
    est = max(time.perf_counter() - t0, 1e-6)
    repeats = args.repeats if args.repeats > 0 else int(min(50000, max(50, np.ceil(args.timed_seconds / est))))
    if world > 1:
        rt = torch.tensor([repeats], device=device, dtype=torch.int64)
        dist.broadcast(rt, 0)
        repeats = int(rt.item())
    times = []
    pos = warm + args.steps
    for _ in range(repeats):
        restore(eng, snap)               # (not timed: before the bracket's first barrier + synchronize)
        sync_all()
        t0 = time.perf_counter()
        run(args.steps, pos)
        sync_all()
        times.append(time.perf_counter() - t0)
        pos += args.steps
    tt = torch.tensor(times, device=device, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)       # slowest rank of every repeat
    times = np.sort(tt.cpu().numpy())
    elapsed = float(np.median(times))
    _trace('timed region done: %d repeats, median %.6f s' % (repeats, elapsed))
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
            'repeats': repeats,
            'ms_per_step_min': round(1e3 * float(times[0]) / args.steps, 5),
            'ms_per_step_max': round(1e3 * float(times[-1]) / args.steps, 5),
            'timed_s': round(float(times.sum()), 4),
            'ranks_seen': ranks_seen,
            'config': {
                'workload': 'Monomer-style 4096-d image features, PCD K=%d L=%d, batch %d rows '
                            '(4x[%d,%d] f32) per GPU per step, Dist model (FC heads+bias, '
                            'thr-BCE), fwd+bwd+TF-Adam' % (K, L, B, B, D),
                'input_size': D, 'num_components': K, 'latent_size': L,
                'batch_rows_per_gpu': B, 'global_batch_rows': B * world,
                'pool_mib_per_gpu': int(nb * batch_bytes >> 20),
                'parallelism': 'dp%d' % world if world > 1 else 'single',
                'final_loss': round(scal['total'], 6),
                'loss_at_restore': round(loss_at_restore, 6),
                'training_state': 'every timed repeat starts from the state %d steps into training (restored before the '
                                  'bracket) and runs steps %d .. %d of that trajectory: final_loss is the loss of a model in '
                                  'training' % (args.restore_steps, args.restore_steps, args.restore_steps + args.steps),
                'timing': 'median of %d repeats of the %d-step region, each bracketed by barrier + synchronize and '
                          'reduced with MAX over ranks' % (repeats, args.steps),
                'arithmetic': 'fp32 storage and accumulation everywhere; both contractions (projection of the fused '
                              'single-GPU step, weight gradient) run on the bf16 matrix cores with every fp32 operand '
                              'split exactly in three bf16 values (round-to-nearest parts, six partial products, fp32 '
                              'accumulate; error vs fp64 equal to the fp32-MFMA kernels, tests/test_hip_parity.py, '
                              'tests/test_planes_and_errors_gpu.py); CFL_EXACT_FP32=1 selects k-ordered fp32 MFMA',
            },
        }

    # ---- roofline of the dominant kernel: HIP events on the launch stream, a
    # second pass of the same K steps (events perturb the step time slightly, so
    # they are kept out of the pass that produces `value`) ----------------------
    # With N > 1 every step contains a collective, so EVERY rank runs these extra steps (a rank-0-only
    # pass would dead-lock in the all-reduce); only rank 0 records events and reports.
    nprof = min(max(args.steps, 100), 500)
    if not args.no_kernel_profile and rank != 0 and world > 1:
        run(nprof, 0)
        torch.cuda.synchronize()
    if rank == 0 and not args.no_kernel_profile:
        H.profile_enable(True)
        run(nprof, 0)
        torch.cuda.synchronize()
        H.profile_enable(False)
        prof = H.profile_read()
        kern = {k: {'avg_us': round(1e3 * ms / n, 3), 'launches': int(n)} for k, (ms, n) in prof.items()}
        dom = max(('proj', 'grad'), key=lambda k: prof.get(k, (0, 1))[0])
        raw_s = prof[dom][0] / prof[dom][1] * 1e-3
        # The event pairs perturb the stream: the intervals of one step add up to more than the step takes in
        # the timed region above (no events).  `achieved` / `frac` use the RAW interval (conservative); the
        # figure with that excess split evenly over the step's launches and removed -- which is what agrees with
        # the rocprofv3 kernel durations committed under profiles/ -- is reported beside it.
        step_kernels = [k for k in ('colnorm', 'proj', 'mid', 'grad', 'finalize') if k in prof]
        sum_intervals_s = sum(prof[k][0] / prof[k][1] for k in step_kernels) * 1e-3
        excess_s = max(sum_intervals_s - elapsed / args.steps, 0.0) / len(step_kernels) if world == 1 else 0.0
        corr_s = max(raw_s - excess_s, 1e-9)
        alg_bytes = 16.0 * D * B                       # 4 fp32 vectors per row, read once
        alg_flops = 4.0 * D * L * (K + 1) * B          # one of fwd / dW: half of 8*D*L*(K+1)
        traffic, traffic_all = None, {}
        traffic_src = ('profiles/traffic.json: builder box, separate `rocprofv3 --pmc FETCH_SIZE` / `WRITE_SIZE` passes '
                       'of this command (2 * FETCH_SIZE + WRITE_SIZE, gfx950 half-count correction); NOT measured in this run')
        live = traffic_live(args) if (world == 1 and not args.no_live_traffic) else None
        if live:
            traffic_all, traffic = live, live.get(dom)
            traffic_src = ('MEASURED IN THIS RUN: two child runs of tools/kernel_probe.py (the same step, same shapes) under '
                           '`rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate passes), per launch = '
                           '(2 * FETCH_SIZE + WRITE_SIZE) KiB: gfx950 tallies 128-byte read requests at 64 bytes (MI355X_MICROARCH.md)')
        tp = os.path.join(ROOT, 'profiles', 'traffic.json')
        if not live and os.path.exists(tp):
            try:
                traffic_all = json.load(open(tp))
                traffic = traffic_all.get(dom)
            except Exception:
                traffic = None
        hbm_frac_dom = alg_bytes / raw_s / 1e9 / HBM_PEAK_GBS
        mfma_frac_dom = alg_flops / raw_s / 1e12 / FP32_MFMA_PEAK_TF
        step_traffic = sum(v for k, v in traffic_all.items() if k in ('proj', 'mid', 'grad', 'finalize') and
                           isinstance(v, (int, float))) or None
        P_params = D * L * (K + 1) + L * (K + 1) + 1
        # kernel names from the library's own planner (cfl_plan_describe): ONE dispatch truth
        plan = H.plan_describe(eng.shape, B, 2, True, True)
        kname = plan['grad'] if dom == 'grad' else plan['proj']
        rp_us = _rocprof_avg_us(kname)
        out['roofline'] = {
            'kernel': kname,
            'plan': plan,
            # north_star asks for >= 60 % of the HBM roofline: unreachable at this batch size by construction, and the line says so
            'note': 'B = %d rows = %.1f MB of input per step is %.1f us at the 8 TB/s peak; the step is three DEPENDENT launches '
                    '(projection -> row math -> weight gradient + Adam) whose fixed cost (dispatch ~2.6 us each, cold first '
                    'miss, latency chains) is ~21 us of it (B-sweep fit t = 20.7 us + B / 35.4 M rows/s, '
                    'profiles/r04_p_b_sweep.md): 0.60 of HBM is out of reach at this batch size, and B -> infinity reaches '
                    '0.29 (x is read twice, PMC traffic 1.1x algorithmic per launch).  What the fraction measures here is '
                    'launch latency, not wasted bandwidth; DESIGN.md section 4' % (B, alg_bytes / 1e6, alg_bytes / 8e12 * 1e6),
            # the committed rocprofv3 average of the same kernel (profiles/kernel_stats.csv, builder box) and the fraction it
            # gives: the event-timed `frac` below is lower because the event pairs perturb the stream (see `timing`)
            'rocprof_avg_us': rp_us,
            'rocprof_frac': round(alg_bytes / (rp_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4) if rp_us else None,
            # `achieved` / `frac` are on the HBM roof (BASELINE's metric: input bytes); `binding_roof` names the roof
            # the dominant kernel sits closer to -- at L=20 that is the fp32-equivalent matrix-core roof (`mfma_f32`)
            'bound': 'hbm',
            'binding_roof': 'mfma_f32' if mfma_frac_dom > hbm_frac_dom else 'hbm',
            'achieved': round(alg_bytes / raw_s / 1e9, 1),
            'peak': HBM_PEAK_GBS,
            'unit': 'GB/s',
            'frac': round(hbm_frac_dom, 4),
            'traffic': traffic,
            'traffic_all_kernels': {k: v for k, v in traffic_all.items() if isinstance(v, (int, float))},
            'traffic_source': traffic_src,
            'avg_launch_us': round(raw_s * 1e6, 3),
            'algorithmic_bytes_per_launch': alg_bytes,
            'event_corrected': {'avg_launch_us': round(corr_s * 1e6, 3),
                                'achieved': round(alg_bytes / corr_s / 1e9, 1),
                                'frac': round(alg_bytes / corr_s / 1e9 / HBM_PEAK_GBS, 4),
                                'excess_per_launch_us': round(excess_s * 1e6, 3)},
            'mfma_f32': {'achieved_tflops': round(alg_flops / raw_s / 1e12, 2),
                         'peak_tflops': FP32_MFMA_PEAK_TF,
                         'frac': round(alg_flops / raw_s / 1e12 / FP32_MFMA_PEAK_TF, 4)},
            # the compute roof of the arithmetic the kernels actually ISSUE: every fp32 product is six bf16 partial products on
            # v_mfma_f32_16x16x32_bf16, so the fp32-equivalent ceiling is the dense bf16 peak / 6 (2.5 PF / 6 = 416.7 TF)
            'bf16x3': {'achieved_tflops': round(alg_flops / raw_s / 1e12, 2), 'peak_tflops': round(BF16X3_PEAK_TF, 1),
                       'frac': round(alg_flops / raw_s / 1e12 / BF16X3_PEAK_TF, 4),
                       'step_achieved_tflops': round(2.0 * alg_flops / (elapsed / args.steps) / 1e12, 2),
                       'step_frac': round(2.0 * alg_flops / (elapsed / args.steps) / 1e12 / BF16X3_PEAK_TF, 4)},
            'step': {'hbm_frac': round(alg_bytes / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS, 4),
                     'launches_per_step': len(step_kernels),
                     # x is read twice by design (projection, weight gradient): HBM traffic of a step vs its
                     # algorithmic bytes 16*D*B + 32*P (inputs once + parameters / Adam slots / gradient)
                     'traffic_bytes': step_traffic,
                     'algorithmic_bytes': 16.0 * D * B + 32.0 * P_params,
                     'traffic_source': 'measured in this run' if live else 'profiles/traffic.json (builder box)'},
            'kernels': kern,
            'timing': 'hipEvent pairs around every launch on the launch stream, separate pass of %d steps (`kernels` '
                      'lists the raw intervals; `achieved` / `frac` use the raw interval of the dominant kernel).  '
                      'The intervals of a step sum to more than ms_per_step of the event-free timed region; '
                      '`event_corrected` removes that excess, split evenly over the launches.  Compare '
                      'profiles/*kernel_stats.csv' % nprof,
        }
    if rank == 0 and world == 1:
        if not args.no_kernel_profile:
            out['roofline_eval'] = roofline_eval(args, eng, pool, device)
        if not args.no_cpu_baseline:
            # the AUC half of the metric on a TRAINED model: the timed repeats were restored to an early state, so train on
            # (untimed) before scoring held-out pairs with the HIP path and with the fp64 oracle
            run(40000, 0)
            torch.cuda.synchronize()
            out['eval_auc'] = eval_auc(args, eng, device, teacher)
            out['eval_auc']['trained_steps'] = int(eng.global_step)
            out['eval_auc']['train_loss'] = round(eng.read_scalars()['total'], 6)
            out['cpu_baseline'] = cpu_baseline(args, args.cpu_seconds)
            try:
                out['cpu_baseline']['loader'] = cpu_loader_baseline(args)
            except Exception as e:          # noqa: BLE001
                out['cpu_baseline']['loader'] = {'error': repr(e)}
        else:
            out['cpu_baseline'] = None
        if not args.no_dp_form:
            try:
                out['dp_form'] = dp_form(args, eng, pool, device, 1e6 * elapsed / args.steps)
            except Exception as e:          # a side measurement must not take the headline line down
                out['dp_form'] = {'error': repr(e)}
        if not args.no_other_configs and not args.no_kernel_profile:
            pool.clear()
            torch.cuda.empty_cache()
            out['other_configs'] = other_configs(device, args)
        if not args.no_cli_loop:
            pool.clear()
            torch.cuda.empty_cache()
            cl = cli_loop(args, device)
            cl['vs_value'] = round(cl['triplets_per_s'] / out['value'], 4)
            cl['vs_value_every1'] = round(cl['every1']['triplets_per_s'] / out['value'], 4)
            out['cli_loop'] = cl
            # ... and at the shape every script of the reference runs (experiments/monomer/run.sh:3-53: --num-components 4
            # --latent-size 10, batch 100), in the reference's own cadence: one validation fetch per iteration
            try:
                cr = cli_loop(args, device, B=100, K=4, L=10)
                out['cli_loop_reference_shape_b100'] = {
                    'us_per_step': cr['us_per_step'], 'triplets_per_s': cr['triplets_per_s'], 'every1': cr['every1'],
                    'final_loss': cr['final_loss'], 'what': 'the same loop at B = 100, K = 4, L = 10 (experiments/monomer/run.sh of the '
                    'reference); `every1` = --scalar-every 1, the reference\'s loop (cfl/bin/train_dist.py:79-86)'}
            except Exception as e:          # noqa: BLE001
                out['cli_loop_reference_shape_b100'] = {'error': repr(e)}
    elif rank == 0:
        out['cpu_baseline'] = None
    if world > 1 and not args.no_dp_legs:
        # ---- N > 1: the attribution of the scaling number, in the same line ------------------------------
        # every rank runs the in-process legs (they contain collectives); the one-shot exchange -- peer memory mapped through
        # hipIpc, kernels that poll flags other GPUs raise: never run on more than one GPU before -- goes into a CHILD job that
        # rank 0 starts on the same GPUs while the ranks of this job wait on a CPU-side (gloo) barrier: if that path hangs or
        # takes a process down, the child is killed and the line above survives with an `error` entry.
        legs = {}
        headline_us = 1e6 * elapsed / args.steps
        # last resort: these legs are side measurements on paths no multi-GPU box has run.  If they wedge (a collective that
        # never returns cannot be interrupted from Python), every rank's timer fires, rank 0 prints the line it already has --
        # the headline was measured above -- and the processes leave without the teardown collectives.
        import threading

        def give_up():
            import signal
            for pid in _live_children:          # exactly the process groups this process started
                try:
                    os.killpg(pid, signal.SIGKILL)
                except OSError:
                    pass
            if rank == 0:
                out['dp_scaling'] = dict(legs, error='the data-parallel legs did not finish within %d s; the line carries the '
                                                     'headline only' % args.dp_legs_timeout)
                os.write(real_stdout, (json.dumps(out) + '\n').encode())
            os._exit(0)

        watchdog = threading.Timer(args.dp_legs_timeout, give_up)
        watchdog.daemon = True
        watchdog.start()
        try:
            ctl = dist.new_group(backend='gloo')
        except Exception:          # noqa: BLE001
            ctl = None
        pool.clear()
        del eng
        torch.cuda.empty_cache()
        for name, b, exch in (('b%d_without_collective' % B, B, 'none'), ('b2048_allreduce', 2048, 'allreduce'),
                              ('b2048_without_collective', 2048, 'none')):
            try:
                legs[name] = dp_leg(args, device, rank, world, b, exch, teacher)
            except Exception as e:          # noqa: BLE001 (a side measurement must not take the headline line down)
                legs[name] = {'error': repr(e)}
        try:
            legs['config5_mrcgan_b100_per_gpu'] = gan_dp_leg(device, rank, world)
        except Exception as e:          # noqa: BLE001
            legs['config5_mrcgan_b100_per_gpu'] = {'error': repr(e)}
        torch.cuda.synchronize()
        if ctl is not None:
            child = None
            if rank == 0:
                rc, line = spawn_ranks(world, ['--gpus', str(world), '--dp-leg', 'oneshot', '--steps', str(args.steps), '--warmup',
                                               '10', '--pool-mib', str(args.pool_mib), '--restore-steps', str(args.restore_steps)],
                                       key='"dp_leg"', timeout=360, extra_env={'CFL_DP_TIMEOUT_S': '20'})
                try:
                    child = json.loads(line) if line else {'error': 'the one-shot child job printed no line (rc %d)' % rc}
                except ValueError:
                    child = {'error': 'unparsable line from the one-shot child job (rc %d)' % rc}
            import datetime
            try:
                dist.monitored_barrier(group=ctl, timeout=datetime.timedelta(seconds=480))
            except Exception as e:          # noqa: BLE001
                print('bench.py: control barrier: %r' % (e,), file=sys.stderr)
            if rank == 0:
                legs['oneshot'] = child
        if rank == 0:
            legs['b%d_allreduce' % B] = {'us_per_step': round(headline_us, 3), 'triplets_per_s': out['value'], 'rows_per_gpu': B,
                                         'what': 'the headline of this line'}
            legs['what'] = ('weak scaling, rows per GPU fixed: `allreduce` = RCCL ncclAllReduce of [gradient | scalars] called by the '
                            'library on the launch stream between the weight-gradient launch and the Adam launch (default '
                            'exchange); `oneshot` = reduce-scatter fused into the weight-gradient launch + sharded Adam + '
                            'all-gather (csrc/cfl_dp.hip; measured by a child job on the same GPUs); `without_collective` = the '
                            'same launches with no exchange at all (what a rank\'s step costs before a byte crosses a link); '
                            '`config5_mrcgan_b100_per_gpu` = the MrCGAN post-epoch step on 100 rows per GPU of a global batch of '
                            '100 x N, one all-reduce of [d gradient | g gradient | scalars] per step')
            out['dp_scaling'] = legs
        watchdog.cancel()
    if rank == 0:
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + '\n').encode())
    if world > 1:
        dist.barrier()
        from cfl import rccl
        rccl.shutdown()
        dist.destroy_process_group()
    return 0


if __name__ == '__main__':
    sys.exit(main())
