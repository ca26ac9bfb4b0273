"""The command lines of the reference's scripts under `torch.distributed.run` with TWO (one test: three) ranks (the box has one GPU: the ranks
share it and talk over gloo -- the code the RCCL ranks of an N-GPU job run, end to end through the CLI):

* `cfl.bin.train_dist` / `predict_dist`: every rank trains its rows of each seeded global batch, evaluation is collective,
  rank 0 alone writes checkpoints / best-model files / predict files; the run lands where the one-process run of the same
  command line lands (same batches; the summation order of the gradient differs);
* `cfl.bin.train --gan ... --post-epochs`: the MrCGAN post epochs sharded over the ranks (round 6: one all-reduce of
  [d gradient | g gradient | scalars] per iteration) against the one-process run.
"""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(module, flags, ranks, port=None, timeout=900, **extra_env):
    env = dict(os.environ, CFL_DIST_BACKEND='gloo', CFL_DP_MAX_BLOCKS='64', CFL_GAN_TUNE_STREAMS='0',
               PYTHONPATH=os.pathsep.join([ROOT, os.path.join(ROOT, 'compatibility-family-learning_amd')]), **extra_env)
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT'):
        env.pop(k, None)
    if ranks == 1:
        cmd = [sys.executable, '-m', module] + flags
    else:
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(ranks), '--master-addr',
               '127.0.0.1', '--master-port', str(port), '-m', module] + flags
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    return r


def _flags(data_root, out, data_name):
    return ['--data-name', data_name, '--data-root', data_root, '--checkpoint-root', os.path.join(out, 'ck'),
            '--log-root', os.path.join(out, 'logs')]


def test_train_dist_and_predict_under_two_ranks(tmp_path):
    from cfl.synthetic import make_dataset
    root = str(tmp_path / 'data')
    make_dataset(os.path.join(root, 'syn', 'toy'), D=200, n_items=600, n_pos=3000, n_neg=3000, k=3, latent=8, seed=1, scale=4.0)
    port = 31000 + os.getpid() % 2000
    model = ['--input-shape', '200', '--num-components', '3', '--latent-size', '10', '--normalize-value', '16.0', '--seed', '0',
             '--batch-size', '100', '--lr', '0.01']
    name = 'linear_dist_ls_10_nc_3_reg_0.0_norm_16.0'
    got = {}
    for tag, ranks in (('one', 1), ('two', 2)):
        out = str(tmp_path / tag)
        fl = _flags(root, out, 'syn/toy') + model
        _run('cfl.bin.train_dist', fl + ['--epochs', '2', '--reset'], ranks, port)
        _run('cfl.bin.predict_dist', fl + ['--predict-root', os.path.join(out, 'pred')], ranks, port + 1)
        ck = os.path.join(out, 'ck', 'syn', 'toy', name)
        st = torch.load(os.path.join(ck, 'model-1.pt'), weights_only=False)
        epoch, acc, auc = open(os.path.join(ck, 'best_acc_model', 'best_accuracy')).read().split('\t')
        lines = open(os.path.join(out, 'pred', 'syn', 'toy', name, 'predict_acc.txt')).read().splitlines()
        # resume for one more epoch -- the two-rank run through the ONE-SHOT exchange (sharded Adam slots: the checkpoint it
        # resumes from holds complete slots, and the one it writes must again)
        _run('cfl.bin.train_dist', fl + ['--epochs', '3'], ranks, port + 2, CFL_DP_EXCHANGE='oneshot')
        st3 = torch.load(os.path.join(ck, 'model-2.pt'), weights_only=False)
        got[tag] = (st, float(acc), float(auc), lines, st3)
    (s1, acc1, auc1, l1, t1), (s2, acc2, auc2, l2, t2) = got['one'], got['two']
    assert t1['global_step'] == t2['global_step'] == 90
    for part in ('variables', 'adam_m', 'adam_v'):
        for k, v in t1[part].items():
            v, w = np.asarray(v), np.asarray(t2[part][k])
            assert np.abs(v - w).max() <= 3e-3 * max(1.0, np.abs(v).max()), (part, k)
            if part != 'variables' and v.size > 64:
                # a slot gathered from the ranks' slices has no stale (still zero) stretch where the one-process run moved
                assert np.mean((w == 0) & (v != 0)) < 1e-3, (part, k)
    assert s1['global_step'] == s2['global_step'] == 60
    # same batches, same initialisation: after 60 Adam steps (lr 0.01) the two runs differ by what the summation order of two partial
    # gradient sums does to a trajectory -- small against the weights' own scale -- and they evaluate alike
    for k, v in s1['variables'].items():
        v, w = np.asarray(v), np.asarray(s2['variables'][k])
        d = np.abs(v - w).max()
        assert d <= 2e-3 * max(1.0, np.abs(v).max()), (k, d)
    assert auc1 > 0.8 and abs(auc1 - auc2) < 5e-3 and abs(acc1 - acc2) < 2e-2, (acc1, auc1, acc2, auc2)
    # the predict file of the two-rank run: rank 0 wrote every pair once, in the same order, with scores of the same model
    assert len(l1) == len(l2) and [ln.split()[:3] for ln in l1] == [ln.split()[:3] for ln in l2]
    sc1, sc2 = np.array([float(ln.split()[3]) for ln in l1]), np.array([float(ln.split()[3]) for ln in l2])
    assert np.abs(sc1 - sc2).max() < 5e-2 * max(1.0, np.abs(sc1).max())
    assert np.mean(np.sign(sc1) == np.sign(sc2)) > 0.97


def test_gan_post_epochs_under_two_ranks(tmp_path):
    from cfl.synthetic import make_double_dataset
    root = str(tmp_path / 'data')
    make_double_dataset(os.path.join(root, 'dy'), image_shape=(16, 16, 3), latent_dim=64, n_items=120, n_pos=160, n_neg=160, k=2, seed=5)
    port = 33000 + os.getpid() % 2000
    base = ['--model-type', 'linear', '--data-type', 'tanh', '--data-mean', '0.5', '--data-norm', '0.5', '--data-directed',
            '--latent-norm', '31.9098', '--data-is-image', '--data-is-double', '--raw-latent', '--latent-shape', '64',
            '--input-shape', '16', '16', '3', '--dist-type', 'pcd', '--lambda-m', '0.5', '--use-threshold', '--num-components',
            '2', '--latent-size', '8', '--batch-size', '16', '--lr', '0.01', '--seed', '3']
    gan = ['--m-prj', '0.2', '--m-enc', '0.05', '--d-lr', '0.0002', '--d-beta1', '0.5', '--g-lr', '0.0002', '--g-beta1', '0.5',
           '--gan', '--gan-type', 'srgan', '--lambda-gp', '0.5', '--z-dim', '6']
    name = 'cfl_pcd_linear_tanh_ls_8_nc_2_ut_norm_0.5_lm_0.5'
    gname = name + '_gan_z_6_m_prj_0.2_m_enc_0.05_dra_0.5_0.5_srgan'
    rows = {}
    for tag, ranks in (('one', 1), ('two', 2)):
        out = str(tmp_path / tag)
        fl = _flags(root, out, 'dy') + base
        _run('cfl.bin.train', fl + ['--epochs', '2', '--reset'], ranks, port)
        _run('cfl.bin.train', fl + gan + ['--load-pre-weights', '--epochs', '2', '--post-epochs', '1', '--disable-eval'], ranks, port + 1)
        gck = os.path.join(out, 'ck', 'dy', gname)
        st = torch.load(os.path.join(gck, 'model-{}.pt'.format(3 * (160 // 16))), weights_only=False)
        assert all(np.isfinite(np.asarray(v)).all() for v in st['variables'].values())
        m = np.asarray(st['adam_m']['CFL/Generator/outputs/Conv/V'])
        assert np.isfinite(m).all() and np.abs(m).max() > 0
        tsv = open(os.path.join(out, 'logs', 'dy', gname, 'gan_scalars.tsv')).read().splitlines()
        head, first = tsv[0].split('\t'), [float(c) for c in tsv[1].split('\t')]
        assert all(np.isfinite(first)) and len(tsv) >= 2
        rows[tag] = (dict(zip(head, first)), st)
        # resume for a second post epoch under the same number of ranks, then cfl.bin.predict (rank 0 alone writes)
        _run('cfl.bin.train', fl + gan + ['--load-pre-weights', '--epochs', '2', '--post-epochs', '2', '--disable-eval'], ranks, port + 2)
        st2 = torch.load(os.path.join(gck, 'model-{}.pt'.format(4 * (160 // 16))), weights_only=False)
        assert all(np.isfinite(np.asarray(v)).all() for v in st2['variables'].values())
        k = 'CFL/Generator/outputs/Conv/V'
        assert not np.array_equal(np.asarray(st2['variables'][k]), np.asarray(st['variables'][k]))      # the generator kept training
        _run('cfl.bin.predict', fl + ['--predict-root', os.path.join(out, 'pred')], ranks, port + 3)
        pred = open(os.path.join(out, 'pred', 'dy', name, 'predict_acc.txt')).read().splitlines()
        rows[tag] += (pred,)
    (r1, s1, p1), (r2, s2, p2) = rows['one'], rows['two']
    assert len(p1) == len(p2) and [ln.split()[:3] for ln in p1] == [ln.split()[:3] for ln in p2]
    # first logged iteration of the post epoch: the generator / discriminator start from the same seeded initialisation, the encoder
    # from distance epochs that differ by the summation order only -- the global-batch means of the two-rank run sit beside the
    # one-process run's
    for k in ('d_total_loss', 'g_total_loss', 'd_loss_real', 'd_grad_loss'):
        assert abs(r1[k] - r2[k]) <= 2e-2 * max(1.0, abs(r1[k])), (k, r1[k], r2[k])
    k = 'CFL/Discriminator/conv2/Conv_4/V'
    d = np.abs(np.asarray(s1['variables'][k]) - np.asarray(s2['variables'][k]))
    assert np.mean(d <= 2e-3) > 0.99, float(np.mean(d <= 2e-3))      # 10 Adam steps of lr 2e-4


def test_conv_encoder_epochs_under_two_ranks(tmp_path):
    """--model-type conv (ConvPCD trunk in front of the heads): under two ranks the trunk's gradient is exchanged beside the heads'
    [gradient | scalars]; the run of experiments/fashion_30/run.sh's command line lands beside the one-process run."""
    from cfl.synthetic import make_dataset
    root = str(tmp_path / 'data')
    make_dataset(os.path.join(root, 'syn', 'img'), D=64, n_items=400, n_pos=1500, n_neg=1500, k=2, latent=6, seed=3, scale=0.25)
    port = 35000 + os.getpid() % 2000
    model = ['--model-type', 'conv', '--data-type', 'sigmoid', '--dist-type', 'pcd', '--use-threshold', '--reg-const', '5e-4',
             '--num-components', '2', '--latent-size', '8', '--input-shape', '8', '8', '1', '--lr', '0.005', '--seed', '10',
             '--batch-size', '100']
    name = 'cfl_pcd_conv_sigmoid_ls_8_nc_2_ut_reg_0.0005'
    got = {}
    for tag, ranks in (('one', 1), ('two', 2)):
        out = str(tmp_path / tag)
        _run('cfl.bin.train', _flags(root, out, 'syn/img') + model + ['--epochs', '2', '--reset'], ranks, port)
        ck = os.path.join(out, 'ck', 'syn', 'img', name)
        st = torch.load(os.path.join(ck, 'model-30.pt'), weights_only=False)
        auc = float(open(os.path.join(ck, 'best_model', 'best_accuracy')).read().split('\t')[2])
        got[tag] = (st, auc)
    (s1, auc1), (s2, auc2) = got['one'], got['two']
    assert s1['global_step'] == s2['global_step'] == 30
    assert auc1 > 0.6 and abs(auc1 - auc2) < 2e-2, (auc1, auc2)
    # 30 Adam steps of lr 0.005 behind a leaky-relu trunk: entries whose tiny gradients change sign with the summation order drift
    # by up to 2 lr per step; the bulk of every variable stays together
    for k in ('CFL/DistEncoder/conv1/Conv/V', 'CFL/DistEncoder/outputs/fully_connected/V'):
        v, w = np.asarray(s1['variables'][k]), np.asarray(s2['variables'][k])
        assert np.mean(np.abs(v - w) <= 5e-3 * max(1.0, np.abs(v).max())) > 0.98, k


def test_cgan_on_an_image_dataset_under_two_ranks(tmp_path):
    """experiments/mnist_30/run_cgan.sh in miniature under two ranks: the distance epochs read HOST batches (image dataset: every rank
    draws the global batch and trains on its rows), the --cgan post epoch shards each HALF of the batch (the interpolation pairs row i
    with row i + B/2).  Against the one-process run."""
    from cfl.synthetic import make_double_dataset
    root = str(tmp_path / 'data')
    make_double_dataset(os.path.join(root, 'im'), image_shape=(16, 16, 1), n_items=120, n_pos=160, n_neg=160, k=2, seed=6, double=False)
    port = 37000 + os.getpid() % 2000
    base = ['--model-type', 'linear', '--data-type', 'sigmoid', '--data-is-image', '--input-shape', '16', '16', '1', '--dist-type',
            'pcd', '--use-threshold', '--num-components', '2', '--latent-size', '8', '--batch-size', '16', '--lr', '0.01', '--seed', '4']
    gan = ['--gan', '--cgan', '--gan-type', 'conv', '--lambda-gp', '0.5', '--z-dim', '6', '--t-dim', '5']
    gname = 'cfl_pcd_linear_sigmoid_ls_8_nc_2_ut_cgan_z_6_t_5_dra_0.5_0.5'
    rows = {}
    for tag, ranks in (('one', 1), ('two', 2)):
        out = str(tmp_path / tag)
        fl = _flags(root, out, 'im') + base
        _run('cfl.bin.train', fl + ['--epochs', '2', '--reset', '--disable-eval'], ranks, port)
        _run('cfl.bin.train', fl + gan + ['--epochs', '0', '--post-epochs', '1', '--disable-eval'], ranks, port + 1)
        enc = torch.load(os.path.join(out, 'ck', 'im', 'cfl_pcd_linear_sigmoid_ls_8_nc_2_ut', 'model-20.pt'), weights_only=False)
        st = torch.load(os.path.join(out, 'ck', 'im', gname, 'model-10.pt'), weights_only=False)
        assert all(np.isfinite(np.asarray(v)).all() for v in st['variables'].values())
        tsv = open(os.path.join(out, 'logs', 'im', gname, 'gan_scalars.tsv')).read().splitlines()
        rows[tag] = (enc, st, dict(zip(tsv[0].split('\t'), [float(c) for c in tsv[1].split('\t')])))
    (e1, s1, r1), (e2, s2, r2) = rows['one'], rows['two']
    k = 'CFL/DistEncoder/outputs/fully_connected/V'
    v, w = np.asarray(e1['variables'][k]), np.asarray(e2['variables'][k])
    assert np.abs(v - w).max() <= 2e-3 * max(1.0, np.abs(v).max())           # 20 distance steps from host-batch shards
    # (this post epoch starts from a FRESH encoder in both runs -- no --load-pre-weights --, so its first iteration sees the same
    # weights everywhere: the two-rank global-batch means equal the one-process numbers up to the summation order)
    for key in ('d_total_loss', 'g_total_loss', 'd_loss_real', 'd_grad_loss', 'g_loss_int'):
        assert abs(r1[key] - r2[key]) <= 1e-4 * max(1.0, abs(r1[key])), (key, r1[key], r2[key])
    k = 'CFL/Discriminator/conv2/Conv/V'
    d = np.abs(np.asarray(s1['variables'][k]) - np.asarray(s2['variables'][k]))
    assert np.mean(d <= 2e-3) > 0.99


def test_streamed_features_under_two_ranks_equal_resident_features(tmp_path):
    """A feature table that does not fit HBM is memory-mapped and gathered per batch (`CFL_FEATURES=stream`, StreamedFeatures): under
    two ranks that is the per-iteration loop on the rank's rows of each host batch instead of the fused multi-iteration call on
    resident features -- the same batches, the same exchange, the same parameters."""
    from cfl.synthetic import make_dataset
    root = str(tmp_path / 'data')
    make_dataset(os.path.join(root, 'syn', 'toy'), D=200, n_items=600, n_pos=2000, n_neg=2000, k=3, latent=8, seed=1, scale=4.0)
    port = 39000 + os.getpid() % 2000
    model = ['--input-shape', '200', '--num-components', '3', '--latent-size', '10', '--normalize-value', '16.0', '--seed', '0',
             '--batch-size', '100', '--lr', '0.01']
    name = 'linear_dist_ls_10_nc_3_reg_0.0_norm_16.0'
    got = {}
    for tag in ('resident', 'stream'):
        out = str(tmp_path / tag)
        _run('cfl.bin.train_dist', _flags(root, out, 'syn/toy') + model + ['--epochs', '1', '--reset'], 2, port, CFL_FEATURES=tag)
        got[tag] = torch.load(os.path.join(out, 'ck', 'syn', 'toy', name, 'model-0.pt'), weights_only=False)
    a, b = got['resident'], got['stream']
    assert a['global_step'] == b['global_step'] == 20
    for k, v in a['variables'].items():
        v, w = np.asarray(v), np.asarray(b['variables'][k])
        assert np.abs(v - w).max() <= 2e-6 * max(1.0, np.abs(v).max()), (k, float(np.abs(v - w).max()))


def test_reference_cadence_under_two_ranks(tmp_path):
    """`--scalar-every 1` (one validation fetch and one scalar read-back per iteration, the reference's loop) under two ranks: the
    validation rows ride in every rank's launches, the logged scalars are global-batch means that travelled in the exchange.
    Row by row against the one-process run of the same command line (same batches; the gradient's summation order differs)."""
    from cfl.synthetic import make_dataset
    root = str(tmp_path / 'data')
    make_dataset(os.path.join(root, 'syn', 'toy'), D=200, n_items=600, n_pos=2000, n_neg=2000, k=3, latent=8, seed=1, scale=4.0)
    port = 41000 + os.getpid() % 2000
    model = ['--input-shape', '200', '--num-components', '3', '--latent-size', '10', '--normalize-value', '16.0', '--seed', '0',
             '--batch-size', '100', '--lr', '0.01', '--scalar-every', '1']
    name = 'linear_dist_ls_10_nc_3_reg_0.0_norm_16.0'
    logs = {}
    for tag, ranks, env in (('one', 1, {}), ('two', 2, {}), ('two_oneshot', 2, {'CFL_DP_EXCHANGE': 'oneshot'})):
        out = str(tmp_path / tag)
        _run('cfl.bin.train_dist', _flags(root, out, 'syn/toy') + model + ['--epochs', '1', '--reset'], ranks, port, **env)
        rows = open(os.path.join(out, 'logs', 'syn', 'toy', name, 'scalars.tsv')).read().splitlines()
        logs[tag] = np.array([[float(c) for c in r.split('\t')] for r in rows])
    one = logs['one']
    assert one.shape == (20, 4) and list(one[:, 0]) == list(range(20))          # a row per iteration
    for tag in ('two', 'two_oneshot'):
        two = logs[tag]
        assert two.shape == one.shape and np.array_equal(two[:, 0], one[:, 0])
        assert np.abs(two[0, 1:] - one[0, 1:]).max() <= 1e-5 * max(1.0, np.abs(one[0, 1:]).max())     # same weights at iteration 0
        assert np.abs(two[:, 1] - one[:, 1]).max() <= 2e-3 * max(1.0, np.abs(one[:, 1]).max())        # loss, 20 Adam steps of lr 0.01
        assert np.abs(two[:, 2] - one[:, 2]).max() <= 0.011                                             # accuracy: one row of 200 may flip
        assert np.abs(two[:, 3] - one[:, 3]).max() <= 2e-3 * max(1.0, np.abs(one[:, 3]).max())        # threshold


def test_three_ranks_through_the_one_shot_exchange(tmp_path):
    """Three ranks (slices of the exchange buffer padded: its length is no multiple of three) sharing the GPU, batch 102 = 3 x 34 rows,
    one epoch of train_dist through the one-shot exchange, against the one-process run."""
    from cfl.synthetic import make_dataset
    root = str(tmp_path / 'data')
    make_dataset(os.path.join(root, 'syn', 'toy'), D=200, n_items=600, n_pos=2040, n_neg=2040, k=3, latent=8, seed=1, scale=4.0)
    port = 43000 + os.getpid() % 2000
    model = ['--input-shape', '200', '--num-components', '3', '--latent-size', '10', '--normalize-value', '16.0', '--seed', '0',
             '--batch-size', '102', '--lr', '0.01']
    name = 'linear_dist_ls_10_nc_3_reg_0.0_norm_16.0'
    got = {}
    for tag, ranks, env in (('one', 1, {}), ('three', 3, {'CFL_DP_EXCHANGE': 'oneshot'})):
        out = str(tmp_path / tag)
        _run('cfl.bin.train_dist', _flags(root, out, 'syn/toy') + model + ['--epochs', '1', '--reset'], ranks, port, **env)
        ck = os.path.join(out, 'ck', 'syn', 'toy', name)
        got[tag] = (torch.load(os.path.join(ck, 'model-0.pt'), weights_only=False),
                    float(open(os.path.join(ck, 'best_acc_model', 'best_accuracy')).read().split('\t')[2]))
    (a, auc1), (b, auc3) = got['one'], got['three']
    assert a['global_step'] == b['global_step'] == 20 and abs(auc1 - auc3) < 5e-3
    for part in ('variables', 'adam_m', 'adam_v'):
        for k, v in a[part].items():
            v, w = np.asarray(v), np.asarray(b[part][k])
            assert np.abs(v - w).max() <= 2e-3 * max(1.0, np.abs(v).max()), (part, k)
