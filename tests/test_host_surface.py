"""CPU checks of the host-side mirror of the reference interface: flag surface,
validation rules, run-name strings (they name checkpoint / predict directories and
are hard-coded in the reference's experiments/*/eval.sh), normalisers, best-stats
and checkpoint-state files."""
import os

import numpy as np
import pytest

from cfl import ops, utils
from cfl.bin import predict, predict_dist, train, train_dist
from cfl.models.cfl import CFL
from cfl.models.dist import Dist


def test_monomer_parser_defaults_and_run_sh_flags():
    a = train_dist.parse_args([])
    assert (a.data_name, a.seed, a.batch_size, a.num_components, a.latent_size, a.epochs) == \
        ('monomer/Baby-also_viewed', 633, 100, 2, 20, 120)
    assert tuple(a.input_shape) == (4096,) and a.normalize_value == 1.0 and a.lr == 0.001
    # experiments/monomer/run.sh:3-53
    a = train_dist.parse_args('--data-name monomer/Baby-also_viewed --input-shape 4096 --num-components 4 '
                              '--latent-size 10 --normalize-value 58.388599 --seed 0 --epochs 200'.split())
    assert (a.num_components, a.latent_size, a.normalize_value, a.seed, a.epochs) == (4, 10, 58.388599, 0, 200)
    assert predict_dist.parse_args([]).predict_root == 'predicts'


def _name(cls, **kw):
    m = cls.__new__(cls)
    base = dict(dist_type='pcd', model_type='linear', directed=False, pos_weight=None, caffe_margin=None,
                data_type='sigmoid', latent_size=20, num_components=2, act_type=None, disable_double=False,
                use_threshold=False, reg_const=0.0, data_norm=None, lambda_m=0.0, gan=False, cgan=False,
                z_dim=20, t_dim=None, m_prj=None, m_enc=None, lambda_gp=None, lambda_dra=0.5,
                gan_type='conv', run_tag=None)
    base.update(kw)
    for k, v in base.items():
        setattr(m, k, v)
    return m


def test_run_names_match_eval_sh_strings():
    # experiments/dyadic/eval.sh:15, experiments/fashion_30/eval.sh:11, experiments/monomer/eval.sh:16
    assert _name(CFL, pos_weight=0.0625, data_type='linear', latent_size=64, num_components=3,
                 use_threshold=True, data_norm=(31.9098,)).get_name() == \
        'cfl_pcd_linear_pw_0.0625_linear_ls_64_nc_3_ut_norm_31.9098'
    assert _name(CFL, dist_type='siamese', model_type='conv', latent_size=60, use_threshold=True,
                 reg_const=0.0005).get_name() == 'cfl_siamese_conv_sigmoid_ls_60_ut_reg_0.0005'
    d = Dist.__new__(Dist)
    d.latent_size, d.num_components, d.reg_const, d.normalize_value, d.run_tag = 10, 4, 0.0, 58.388599, None
    assert d.get_name() == 'linear_dist_ls_10_nc_4_reg_0.0_norm_58.388599'
    d.run_tag = 'x'
    assert d.get_name().endswith('_run_x')
    g = _name(CFL, gan=True, m_prj=0.2, m_enc=0.05, lambda_gp=0.5, gan_type='srgan', lambda_m=0.5, directed=True)
    assert g.get_name() == 'cfl_pcd_linear_di_sigmoid_ls_20_nc_2_lm_0.5_gan_z_20_m_prj_0.2_m_enc_0.05_dra_0.5_0.5_srgan'
    assert g.get_name(no_gan=True) == 'cfl_pcd_linear_di_sigmoid_ls_20_nc_2_lm_0.5'
    assert _name(CFL, gan=True, cgan=True, t_dim=8).get_name() == 'cfl_pcd_linear_sigmoid_ls_20_nc_2_cgan_z_20_t_8'


def test_dist_parser_validation_rules():
    ok = '--model-type linear --dist-type pcd --use-threshold'.split()
    a = train.parse_args(ok)
    assert a.batch_size == 100 and a.post_epochs == 100 and tuple(a.input_shape) == (28, 28, 1)
    assert predict.parse_args(ok).batch_size == 500
    with pytest.raises(AssertionError):            # non-siamese needs --use-threshold
        train.parse_args('--dist-type pcd'.split())
    with pytest.raises(AssertionError):            # caffe margin is siamese-only
        train.parse_args('--dist-type pcd --use-threshold --caffe-margin 100'.split())
    with pytest.raises(AssertionError):            # lambda_m excludes caffe margin
        train.parse_args('--dist-type siamese --caffe-margin 100 --lambda-m 0.5'.split())
    with pytest.raises(AssertionError):            # --t-dim needs --cgan
        train.parse_args(ok + ['--t-dim', '4'])
    with pytest.raises(SystemExit):
        train.parse_args(['--dist-type', 'nope'])
    a = train.parse_args('--dist-type siamese --caffe-margin 100. --num-components 1 --latent-size 256 '
                         '--model-type linear --data-type linear --data-norm 31.9098 --input-shape 1024 '
                         '--pos-weight 0.0625'.split())
    assert a.caffe_margin == 100.0 and a.data_norm == (31.9098,)


def test_normalizers():
    x = np.array([[0., 29.1943, 58.388599]], np.float32)
    n = ops.normalizer(58.388599, 0.)
    assert np.allclose(n(x), x / np.float32(58.388599), rtol=1e-6)
    assert np.allclose(ops.unnormalizer(58.388599, 0.)(n(x)), x, rtol=1e-6)
    c = n.to_cfl_norm()
    assert abs(c.mul - 1 / 58.388599) < 1e-9 and c.add == 0 and not c.has_lo and not c.has_hi
    dn, du, an, au, ln = ops.dist_normalizer((4,), None, None, (0.5,), (0.5,), 31.9098, 'tanh')
    y = dn(np.array([[0., 0.25, 0.75, 5.0]], np.float32))
    assert np.allclose(y, [[-1., -0.5, 0.5, 1.0]])            # (x - .5) / .5 clipped to [-1, 1]
    assert an is dn and au is du
    assert np.allclose(ln(np.array([[31.9098]], np.float32)), 1.0)
    c = dn.to_cfl_norm()
    assert (c.mul, c.add, c.lo, c.hi, c.has_lo, c.has_hi) == (2.0, -1.0, -1.0, 1.0, 1, 1)
    assert ops.dist_normalizer((4,), None, None, None, None, None, 'relu')[0].to_cfl_norm().has_hi == 0
    with pytest.raises(ValueError):
        ops.normalizer_v2((4,), mean=(0.1, 0.2, 0.3))                 # 3 values for 4 "channels"
    pc = ops.normalizer_v2((2, 2, 3), scale=2.0, mean=(0.1, 0.2, 0.3), norm=(0.5, 1.0, 2.0), clip_value_min=-1.0)
    xs = np.arange(24, dtype=np.float32).reshape(2, 12) / 10
    want = np.clip((xs.reshape(-1, 3) * 2.0 - [0.1, 0.2, 0.3]) / [0.5, 1.0, 2.0], -1.0, None).reshape(2, 12)
    assert pc.per_channel and np.allclose(pc(xs), want, atol=1e-6)
    assert np.allclose(ops.unnormalizer_v2((2, 2, 3), 2.0, (0.1, 0.2, 0.3), (0.5, 1.0, 2.0))(
        ((xs.reshape(-1, 3) * 2.0 - [0.1, 0.2, 0.3]) / [0.5, 1.0, 2.0]).reshape(2, 12).astype(np.float32)).reshape(2, 12), xs, atol=1e-5)
    with pytest.raises(ValueError):
        pc.to_cfl_norm()
    assert np.allclose(ops.lrelu(np.array([-1., 2.])), [-0.2, 2.])


def test_best_stats_and_saver_state_files(tmp_path):
    p = tmp_path / 'best_accuracy'
    s = utils.load_best_stats(str(p))
    assert (s.best_epoch, s.best_accuracy, s.best_auc) == (None, 0.0, 0.0)
    utils.save_best_stats(str(p), 7, 0.91, 0.95)
    assert p.read_text() == '7\t0.91\t0.95'
    s = utils.load_best_stats(str(p))
    assert (s.best_epoch, s.best_accuracy, s.best_auc) == (7, 0.91, 0.95)
    p.write_text('3\t0.5')
    assert utils.load_best_stats(str(p)).best_auc == 0.0

    class Fake(object):
        def __init__(self):
            self.v = 0

        def checkpoint_state(self):
            return {'v': self.v}

        def load_checkpoint_state(self, st):
            self.v = st['v']
    m, sv = Fake(), utils.Saver(max_to_keep=2)
    d = tmp_path / 'ck'
    for step in (0, 1, 2):
        m.v = step * 10
        sv.save(m, str(d / 'model'), global_step=step)
    assert sorted(os.listdir(d)) == ['checkpoint', 'model-1.pt', 'model-2.pt']
    assert (d / 'checkpoint').read_text().splitlines()[0] == 'model_checkpoint_path: "model-2"'
    m2 = Fake()
    saver, start = utils.load_model(m2, str(d))
    assert m2.v == 20 and start == 3             # resume at (step in name) + 1, cfl/utils.py:476-477
    assert utils.load_model(Fake(), str(tmp_path / 'none'))[1] == 0
    with pytest.raises(Exception, match='must have best model'):
        utils.load_model(Fake(), str(tmp_path / 'none'), str(tmp_path / 'nogan'))


def test_incremental_average():
    a = utils.IncrementalAverage()
    for v in (1.0, 2.0, 6.0):
        a.add(v)
    assert a.average == pytest.approx(3.0) and a.count == 3


def test_checkpoint_converter_npz_round_trip(tmp_path):
    """cfl.bin.convert_checkpoint: .pt <-> .npz keyed by the TensorFlow variable names (Adam slots as
    <name>/Adam, <name>/Adam_1); a missing TensorFlow bundle is an error (the TF directions themselves:
    tests/test_tf_bundle.py)."""
    import torch
    from cfl.bin import convert_checkpoint as C
    rng = np.random.RandomState(0)
    names = ['CFL/DistEncoder/outputs/fully_connected/V', 'CFL/DistEncoder/outputs/fully_connected/g',
             'CFL/Thresholder/threshold/threshold']
    state = {'variables': {n: rng.randn(3, 2).astype(np.float32) for n in names},
             'adam_m': {n: rng.randn(3, 2).astype(np.float32) for n in names},
             'adam_v': {n: np.abs(rng.randn(3, 2)).astype(np.float32) for n in names},
             'beta1_power': 0.81, 'beta2_power': 0.998, 'global_step': 7, 'name': 'x'}
    pt, npz, back = tmp_path / 'model-7.pt', tmp_path / 'm.npz', tmp_path / 'back.pt'
    from cfl.utils import Saver
    torch.save(Saver._plain(state), str(pt))
    assert C.main(['--to-npz', str(pt), str(npz)]) == 0
    with np.load(str(npz)) as z:
        assert names[0] + '/Adam_1' in z.files and int(z['global_step']) == 7
    assert C.main(['--from-npz', str(npz), str(back)]) == 0
    got = torch.load(str(back), weights_only=False)
    for key in ('variables', 'adam_m', 'adam_v'):
        for n in names:
            assert np.array_equal(got[key][n], state[key][n])
    assert got['global_step'] == 7 and abs(got['beta1_power'] - 0.81) < 1e-6
    with pytest.raises(OSError):
        C.main(['--from-tf', str(tmp_path / 'nope'), str(back)])


def test_input_transformer_descriptors():
    """cfl.ops.dist_transformer / dist_ae_transformer (cfl/ops.py:262-299) and the NumPy restatement of the
    window / resize / mirror arithmetic the GPU kernel is checked against."""
    assert ops.dist_transformer(None, (28, 28, 1), False, False) == (None, None)
    tr, va = ops.dist_transformer((20, 20, 1), (16, 16, 1), True, True)
    assert (tr.kind, tr.mirror, va.kind, va.mirror) == ('random_crop', True, 'crop', False)
    tr, va = ops.dist_transformer((20, 20, 1), (16, 16, 1), False, False)
    assert tr.kind == 'crop' and va.kind == 'crop'
    tr, va = ops.dist_transformer((8, 8, 3), (16, 16, 3), False, True)
    assert tr.kind == 'resize' and tr.mirror and va.kind == 'resize' and not va.mirror
    tr, va = ops.dist_transformer(None, (8, 8, 3), False, True)
    assert tr.kind == 'reshape' and tr.mirror and va is None
    assert ops.dist_ae_transformer((28, 28, 1), None) is None
    assert ops.dist_ae_transformer((28, 28, 1), (32, 32, 1)).kind == 'resize'
    rng = np.random.RandomState(0)
    x = rng.rand(3, 6 * 5 * 2).astype(np.float32)
    img = x.reshape(3, 6, 5, 2)
    crop = ops.ImageTransform((6, 5, 2), (4, 3, 2), 'crop')
    assert np.array_equal(crop(x).reshape(3, 4, 3, 2), img[:, 1:5, 1:4])          # central window
    off = np.array([[0, 0], [2, 2], [1, 0]], np.int32)
    got = ops.ImageTransform((6, 5, 2), (4, 3, 2), 'random_crop')(x, off).reshape(3, 4, 3, 2)
    assert np.array_equal(got[1], img[1, 2:6, 2:5]) and np.array_equal(got[2], img[2, 1:5, 0:3])
    pad = ops.ImageTransform((6, 5, 2), (8, 7, 2), 'crop')(x).reshape(3, 8, 7, 2)
    assert np.array_equal(pad[:, 1:7, 1:6], img) and pad[:, 0].sum() == 0 and pad[:, :, 0].sum() == 0
    same = ops.ImageTransform((6, 5, 2), (6, 5, 2), 'resize')(x)
    assert np.allclose(same, x)
    up = ops.ImageTransform((6, 5, 2), (12, 10, 2), 'resize')(x).reshape(3, 12, 10, 2)
    assert np.allclose(up[:, ::2, ::2], img)                                       # src = dst / 2: even pixels exact
    flip = ops.ImageTransform((6, 5, 2), (6, 5, 2), 'reshape', True)
    once = flip(x, None, np.ones(3, np.int32))
    assert np.array_equal(once.reshape(3, 6, 5, 2), img[:, :, ::-1]) and np.array_equal(flip(once, None, np.ones(3, np.int32)), x)
    o, f = ops.ImageTransform((20, 20, 1), (16, 16, 1), 'random_crop', True).draw(1000, np.random.RandomState(1))
    assert o.min() == 0 and o.max() == 4 and 0.4 < f.mean() < 0.6


def test_cli_entry_points_cap_the_host_thread_pool(monkeypatch):
    """engine.quiet_host_threads (called by init_from_env of every CLI entry point): torch's intra-op pool is capped
    unless the user chose a size with OMP_NUM_THREADS (idle OpenMP threads spinning at a barrier
    exhaust a container's CPU quota and the kernel then freezes the launching thread)."""
    import torch
    from cfl import engine
    before = torch.get_num_threads()
    try:
        torch.set_num_threads(max(before, 6))
        monkeypatch.setenv('OMP_NUM_THREADS', '6')
        engine.quiet_host_threads()
        assert torch.get_num_threads() == max(before, 6)          # the user's choice stands
        monkeypatch.delenv('OMP_NUM_THREADS')
        engine.quiet_host_threads()
        assert torch.get_num_threads() <= 4
        engine.quiet_host_threads(2)
        assert torch.get_num_threads() <= 2
    finally:
        torch.set_num_threads(before)


def test_scalar_every_flag_and_read_back_cadence():
    """--scalar-every N (not a reference flag; default 25; 1 = the reference's per-iteration validation fetch,
    cfl/bin/train_dist.py:79-86 of the reference): the flag parses, and train_steps reads back at iterations 0, N, 2N, ...
    and the last one, taking exactly one validation batch per read-back -- checked with a stand-in engine (no GPU; the
    general loop: the form with the validation rows inside the step's launches is tests/test_dp_step_gpu.py)."""
    from argparse import Namespace
    assert train_dist.parse_args([]).scalar_every == 25
    assert train_dist.parse_args(['--scalar-every', '1']).scalar_every == 1

    class Src(object):
        def __init__(self):
            self.windows, self.singles = [], 0

        def next_windows(self, batch, nsteps, shard=None):
            self.windows.append(nsteps)
            return Namespace(nsteps=nsteps)

        def next_indexed(self, batch, shard=None):
            self.singles += 1
            return ('table', 'streams')

    class Engine(object):
        steps = 0

        def step_windows(self, win):
            self.steps += win.nsteps

    for every, n_steps in ((25, 60), (1, 7), (10, 10), (4, 1)):
        recorded, scored = [], []
        train, val, model = Src(), Src(), Namespace(engine=Engine())
        orig = train_dist.DeferredScalars

        class Rec(object):
            def __init__(self, model, on_scalars):
                pass

            def record_scores(self, val_batch):       # (the validation batch is scored BEFORE the read-back iteration's update)
                scored.append(model.engine.steps)
                return ('slot', 8)

            def record_scalars(self, step, host, n):
                recorded.append(step)

            def flush(self):
                pass
        train_dist.DeferredScalars = Rec
        try:
            train_dist.train_steps(model, train, val, 8, None, n_steps, on_scalars=lambda *a: None, scalar_every=every)
        finally:
            train_dist.DeferredScalars = orig
        want = sorted(set(list(range(0, n_steps, every)) + [n_steps - 1]))
        assert recorded == want, (every, n_steps, recorded)
        assert scored == want, (every, n_steps, scored)      # scored when exactly `step` iterations had been taken: pre-update
        assert model.engine.steps == n_steps and sum(train.windows) == n_steps
        assert val.singles == len(want)            # one validation batch per read-back: at N = 1 the reference's stream
