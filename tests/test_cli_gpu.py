"""GPU end-to-end: the drop-in command lines of experiments/monomer/{run,eval}.sh and
experiments/dyadic/{run,eval}.sh on a synthetic dataset in the reference's on-disk
format: train -> checkpoints -> predict files -> evaluate_total; resume; the model
objects against the oracle on the batches the data pipeline produces."""
import contextlib
import io
import os

import numpy as np
import pytest
import torch

from oracle import cfl_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dataset(tmp_path_factory):
    from cfl.synthetic import make_dataset
    root = tmp_path_factory.mktemp('data')
    make_dataset(str(root / 'syn' / 'toy'), D=200, n_items=600, n_pos=3000, n_neg=3000, k=3,
                 latent=8, seed=1, scale=4.0)
    return str(root)


def _common(dataset, tmp):
    return ['--data-name', 'syn/toy', '--data-root', dataset, '--checkpoint-root', str(tmp / 'ck'),
            '--log-root', str(tmp / 'logs')]


def test_train_dist_predict_evaluate(dataset, tmp_path):
    from cfl.bin import evaluate_total, predict_dist, train_dist
    flags = _common(dataset, tmp_path) + ['--input-shape', '200', '--num-components', '3',
                                          '--latent-size', '10', '--normalize-value', '16.0',
                                          '--seed', '0', '--batch-size', '100', '--lr', '0.01']
    train_dist.main(flags + ['--epochs', '3', '--reset'])
    name = 'linear_dist_ls_10_nc_3_reg_0.0_norm_16.0'
    ck = tmp_path / 'ck' / 'syn' / 'toy' / name
    assert (ck / 'checkpoint').exists() and (ck / 'model-2.pt').exists()
    best = ck / 'best_acc_model' / 'best_accuracy'
    epoch, acc, auc = best.read_text().split('\t')
    assert float(auc) > 0.8 and float(acc) > 0.6
    state = torch.load(str(ck / 'model-2.pt'), weights_only=False)
    v = state['variables']
    assert v['Dist/Encoder/latent_outputs/fully_connected/weights'].shape == (200, 10)
    assert v['Dist/Encoder/pcd_outputs/fully_connected/weights'].shape == (200, 30)
    assert v['Dist/Encoder/pcd_outputs/fully_connected/biases'].shape == (30,)
    assert 'Dist/Thresholder/threshold/threshold' in v and state['global_step'] == 90
    # resume: epochs 3 -> 4 starts at epoch 3 and adds exactly one epoch of steps
    train_dist.main(flags + ['--epochs', '4'])
    assert torch.load(str(ck / 'model-3.pt'), weights_only=False)['global_step'] == 120

    predict_dist.main(flags + ['--predict-root', str(tmp_path / 'pred')])
    pdir = tmp_path / 'pred' / 'syn' / 'toy' / name
    lines = (pdir / 'predict_acc.txt').read_text().splitlines()
    a, rel, b, score = lines[0].split()
    assert rel == 'match' and len(a) == len(b) == 10 and float(score) == float(np.float32(score))
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        evaluate_total.evaluate([os.path.join(dataset, 'syn/toy')], [str(pdir)], select_auc=False,
                                name='m', avg=False, auc_model=False, only_larger=None)
    cells = buf.getvalue().strip().split('\t')
    assert len(cells) == 7 and cells[-1] == 'm'
    assert float(cells[5].rstrip('%')) > 80.0            # test AUC


def test_cfl_train_predict_linear(dataset, tmp_path):
    from cfl.bin import predict, train
    flags = _common(dataset, tmp_path) + [
        '--model-type', 'linear', '--data-type', 'linear', '--data-norm', '16.0', '--input-shape', '200',
        '--pos-weight', '0.5', '--data-switch', '--use-threshold', '--num-components', '3',
        '--latent-size', '8', '--dist-type', 'monomer', '--lambda-m', '0.5', '--lr', '0.01', '--seed', '1']
    train.main(flags + ['--epochs', '2', '--reset'])
    name = 'cfl_monomer_linear_pw_0.5_linear_ls_8_nc_3_ut_norm_16.0_lm_0.5'
    ck = tmp_path / 'ck' / 'syn' / 'toy' / name
    assert (ck / 'best_model' / 'best_accuracy').exists()
    assert (ck / 'best_acc_model' / 'best_accuracy_by_th').exists()
    v = torch.load(str(ck / 'model-60.pt'), weights_only=False)['variables']
    assert v['CFL/DistEncoder/outputs/fully_connected/V'].shape == (200, 8)
    assert v['CFL/DistEncoder/prototype_outputs/fully_connected/g'].shape == (24,)
    assert v['CFL/DistEncoder/monomer_outputs/fully_connected/V'].shape == (8, 3)
    pflags = [f for f in flags if f not in ('--data-switch',)]
    predict.start(pflags + ['--predict-root', str(tmp_path / 'pred')])
    pdir = tmp_path / 'pred' / 'syn' / 'toy' / name
    for f in ('predict.txt', 'predict_val.txt', 'predict_train.txt', 'predict_acc.txt'):
        assert (pdir / f).exists()


@pytest.mark.parametrize('extra', [
    ['--dist-type', 'monomer', '--lambda-m', '0.5', '--data-switch'],
    ['--dist-type', 'siamese', '--caffe-margin', '5.0'],
    ['--dist-type', 'pcd'],
])
def test_cfl_epochs_through_the_fused_loop_equal_the_per_iteration_loop(dataset, tmp_path, monkeypatch, extra):
    """cfl.bin.train's distance epochs on resident features go through cfl.bin.train_dist.train_steps (windows of the device pair
    lists, non-stalling read-backs); CFL_FUSED_EPOCHS=0 is the one-Python-iteration-per-step loop.  Same seeded streams, same
    trajectory: every variable and Adam slot after two epochs, and the best-model bookkeeping, must be equal."""
    from cfl.bin import train
    runs = []
    for tag, env in (('f', '1'), ('p', '0')):
        monkeypatch.setenv('CFL_FUSED_EPOCHS', env)
        flags = ['--data-name', 'syn/toy', '--data-root', str(dataset), '--checkpoint-root', str(tmp_path / tag), '--log-root',
                 str(tmp_path / (tag + '_logs')), '--batch-size', '20', '--model-type', 'linear', '--data-type', 'linear',
                 '--data-norm', '16.0', '--input-shape', '200', '--pos-weight', '0.5', '--use-threshold', '--num-components', '3',
                 '--latent-size', '8', '--lr', '0.01', '--seed', '1'] + extra
        train.main(flags + ['--epochs', '2', '--reset'])
        cks = sorted((tmp_path / tag).rglob('model-300.pt'))      # 3000 pairs / 20 = 150 iterations per epoch
        assert len(cks) == 1
        runs.append((torch.load(str(cks[0]), weights_only=False), (cks[0].parent / 'best_model' / 'best_accuracy').read_text()))
    (a, ba), (b, bb) = runs
    for part in ('variables', 'adam_m', 'adam_v'):
        assert set(a[part]) == set(b[part])
        for k in a[part]:
            assert np.array_equal(a[part][k], b[part][k]), (part, k, float(np.abs(a[part][k] - b[part][k]).max()))
    assert ba == bb


def test_bad_flag_combinations_fail_loudly(dataset, tmp_path):
    from cfl.bin import train
    with pytest.raises(AssertionError):      # cfl/utils.py:70-71: the CD loss is for siamese only
        train.main(_common(dataset, tmp_path) + ['--model-type', 'linear', '--use-threshold', '--caffe-margin', '5',
                                                 '--input-shape', '200'])


def test_model_matches_oracle_on_pipeline_batches(dataset):
    """Dist (D=200 -> zero-padded to 256 on the device) vs the oracle, same seeded
    SemiDataSet batches, 15 Adam steps; then dist_eval AUC parity."""
    from cfl import input_data, ops, utils
    from cfl.models.dist import construct_model
    path = os.path.join(dataset, 'syn/toy')
    data = input_data.load_data_sets(path, 200, seed=5)
    data_o = input_data.load_data_sets(path, 200, seed=5)
    model, _ = construct_model(input_shape=(200,), latent_size=6, num_components=2, lr=2e-3, beta1=0.9,
                               beta2=0.999, batch_size=50, normalize_value=16.0, reg_const=1e-3,
                               data_normalizer=ops.normalizer(16.0, 0.), data=data, seed=3)
    p, _, thr = model.engine.named_variables()
    cfg = O.EncoderCfg(D=200, L=6, K=2)
    p64 = {k: (v[:200] if k.endswith('/W') else v).astype(np.float64) for k, v in p.items()}
    tr = O.OracleTrainer(cfg, O.LossCfg(reg_const=1e-3), lr=2e-3, dtype=np.float64, params=p64)
    for i in range(15):
        b = data.train.next_batch(50)
        bo = data_o.train.next_batch(50)
        assert all(np.array_equal(x, y) for x, y in zip(b, bo))
        model.train_step(b)
        sc = tr.step(tuple(x.astype(np.float64) / 16.0 for x in bo))
        got = model.scalars()['total']
        assert abs(got - sc['total']) <= 1e-5 * max(1.0, abs(sc['total'])), (i, got, sc['total'])
    ev = utils.dist_eval(None, model, 64, data.val)

    class _OM(object):
        def predict(self, s, d):
            return tr.scores(s.astype(np.float64) / 16.0, d.astype(np.float64) / 16.0).reshape(-1, 1)
    ev_o = utils.dist_eval(None, _OM(), 64, data_o.val)
    assert abs(ev.auc - ev_o.auc) <= 1e-4 and abs(ev.accuracy - ev_o.accuracy) <= 2.0 / 700


def test_resident_gather_equals_host_batches(dataset):
    from cfl import input_data
    path = os.path.join(dataset, 'syn/toy', 'train')
    a = input_data.SemiDataSet(path, input_size=200, data_switch=True, seed=9)
    b = input_data.SemiDataSet(path, input_size=200, data_switch=True, seed=9)
    res = input_data.ResidentFeatures(a)
    assert res.padded_size == 256
    for _ in range(5):
        dev = res.next_batch(37)
        host = b.next_batch(37)
        for x, y in zip(dev, host):
            assert np.array_equal(x.cpu().numpy()[:, :200], y)
            assert not x[:, 200:].any()
    lo_hi = (10, 20)
    sh = res.next_batch(40, lo_hi)
    full = b.next_batch(40)
    assert np.array_equal(sh[1].cpu().numpy()[:, :200], full[1][10:20])


def test_indexed_batches_train_like_dense_batches(dataset):
    """ResidentFeatures.next_indexed hands the step the POSITIONS of the batch (a window of the device copy of the
    shuffled pair list); the fused kernels read the rows in place.  Same index stream (epoch wraps, data_switch
    coin flips, the B > N oversampling branch), and bit-identical training to dense gathered batches."""
    from cfl import hipabi as H
    from cfl import input_data
    from cfl.engine import PairEngine
    path = os.path.join(dataset, 'syn/toy', 'train')
    a = input_data.SemiDataSet(path, input_size=200, data_switch=True, seed=9)
    b = input_data.SemiDataSet(path, input_size=200, data_switch=True, seed=9)
    ra, rb = input_data.ResidentFeatures(a), input_data.ResidentFeatures(b)
    rng = np.random.RandomState(0)
    mk = lambda: PairEngine(256, 6, 3, 'pcd', weight_norm=True, has_bias=True, norm=H.make_norm(1.0 / 16.0),
                            loss=H.make_loss(pos_weight=0.25, lambda_m=0.5), lr=1e-3, device='cuda',
                            params={'outputs/W': rng_w(0, 256, 6), 'outputs/b': np.zeros(6, np.float32),
                                    'outputs/g': np.ones(6, np.float32), 'proto/W': rng_w(1, 256, 18),
                                    'proto/b': np.zeros(18, np.float32), 'proto/g': np.ones(18, np.float32)})
    rng_w = lambda s, d, n: (np.random.RandomState(s).randn(d, n) * 0.05).astype(np.float32)
    ea, eb = mk(), mk()
    n_pairs = a.pairs_pos.shape[0]
    for B in [37] * 6 + [n_pairs // 2 + 1] * 4 + [37] * 2:          # the long batches force epoch wraps
        table, streams = ra.next_indexed(B)
        dense = rb.next_batch(B)
        ea.step((table, streams))
        eb.step(dense)
        assert torch.equal(ea.theta, eb.theta) and torch.equal(ea.grad, eb.grad)
        assert torch.equal(ea.scalars, eb.scalars)
        sa = ea.scores(table, streams.pair(1))
        sb = eb.scores(dense[2], dense[3])
        assert torch.equal(sa, sb)
    assert a.head_labeled_pos == b.head_labeled_pos and np.array_equal(a.pairs_pos, b.pairs_pos)
    # shards: rows [lo, hi) of the global batch
    t, st = ra.next_indexed(40, (10, 20))
    full = rb.next_batch(40)
    assert st.n == 10
    assert torch.equal(ea.scores(t, st.pair(0)), eb.scores(full[0][10:20].contiguous(), full[1][10:20].contiguous()))
    # B > number of pairs: the reference's RandomState.choice oversampling (cfl/input_data.py:558-566)
    t, st = ra.next_indexed(n_pairs + 5)
    full = rb.next_batch(n_pairs + 5)
    assert torch.equal(ea.scores(t, st.pair(0)), eb.scores(full[0], full[1]))
    # out-of-range positions are clamped to the last table row, never read out of bounds
    bad = torch.tensor([0, 10 ** 9, -1, 3], dtype=torch.int32, device='cuda')
    ok = torch.tensor([0, ra.table.shape[0] - 1, ra.table.shape[0] - 1, 3], dtype=torch.int32, device='cuda')
    s_bad = ea.scores(ra.table, H.IndexStreams.from_tensors([bad, bad]))
    s_ok = ea.scores(ra.table, H.IndexStreams.from_tensors([ok, ok]))
    assert torch.equal(s_bad, s_ok)


def test_cfl_train_conv_model(tmp_path):
    """--model-type conv end to end on an MNIST-shaped synthetic pair set (8x8x1 images so
    the trunk is one 5x5 stride-2 layer), the command line of experiments/fashion_30/run.sh."""
    from cfl.bin import predict, train
    from cfl.synthetic import make_dataset
    root = tmp_path / 'data'
    make_dataset(str(root / 'syn' / 'img'), D=64, n_items=400, n_pos=1500, n_neg=1500, k=2, latent=6,
                 seed=3, scale=0.25)
    flags = ['--data-name', 'syn/img', '--data-root', str(root), '--checkpoint-root', str(tmp_path / 'ck'),
             '--log-root', str(tmp_path / 'logs'), '--model-type', 'conv', '--data-type', 'sigmoid',
             '--dist-type', 'pcd', '--use-threshold', '--reg-const', '5e-4', '--num-components', '2',
             '--latent-size', '8', '--input-shape', '8', '8', '1', '--lr', '0.005', '--seed', '10']
    train.main(flags + ['--epochs', '3', '--batch-size', '100', '--reset'])
    name = 'cfl_pcd_conv_sigmoid_ls_8_nc_2_ut_reg_0.0005'
    ck = tmp_path / 'ck' / 'syn' / 'img' / name
    auc = float((ck / 'best_model' / 'best_accuracy').read_text().split('\t')[2])
    assert auc > 0.7
    v = torch.load(str(ck / 'model-45.pt'), weights_only=False)['variables']
    assert v['CFL/DistEncoder/conv1/Conv/V'].shape == (5, 5, 1, 64)
    predict.start(flags + ['--predict-root', str(tmp_path / 'pred')])
    assert (tmp_path / 'pred' / 'syn' / 'img' / name / 'predict_acc.txt').exists()


def test_cfl_double_data_then_gan_post_epochs(tmp_path):
    """experiments/dyadic/run_gen.sh in miniature: distance epochs on an image+latent dataset, then
    `--load-pre-weights --gan --gan-type srgan --lambda-gp ... --post-epochs` (MrCGAN), then resume."""
    from cfl.bin import predict, train
    from cfl.synthetic import make_double_dataset
    root = tmp_path / 'data'
    make_double_dataset(str(root / 'dy'), image_shape=(16, 16, 3), latent_dim=64, n_items=120, n_pos=160,
                        n_neg=160, k=2, seed=5)
    base = ['--data-name', 'dy', '--data-root', str(root), '--checkpoint-root', str(tmp_path / 'ck'),
            '--log-root', str(tmp_path / 'logs'), '--model-type', 'linear', '--data-type', 'tanh',
            '--data-mean', '0.5', '--data-norm', '0.5', '--data-directed', '--latent-norm', '31.9098',
            '--data-is-image', '--data-is-double', '--raw-latent', '--latent-shape', '64', '--input-shape', '16',
            '16', '3', '--dist-type', 'pcd', '--lambda-m', '0.5', '--use-threshold', '--num-components', '2',
            '--latent-size', '8', '--batch-size', '16', '--lr', '0.01', '--seed', '3']
    train.main(base + ['--epochs', '3', '--reset'])
    name = 'cfl_pcd_linear_tanh_ls_8_nc_2_ut_norm_0.5_lm_0.5'
    ck = tmp_path / 'ck' / 'dy' / name
    epoch, acc, auc = (ck / 'best_model' / 'best_accuracy').read_text().split('\t')
    assert float(auc) > 0.6
    gan = ['--m-prj', '0.2', '--m-enc', '0.05', '--d-lr', '0.0002', '--d-beta1', '0.5', '--g-lr', '0.0002',
           '--g-beta1', '0.5', '--gan', '--gan-type', 'srgan', '--lambda-gp', '0.5', '--z-dim', '6']
    train.main(base + gan + ['--load-pre-weights', '--epochs', '3', '--post-epochs', '1', '--disable-eval'])
    gname = name + '_gan_z_6_m_prj_0.2_m_enc_0.05_dra_0.5_0.5_srgan'
    gck = tmp_path / 'ck' / 'dy' / gname
    nb = 160 // 16
    st = torch.load(str(gck / 'model-{}.pt'.format(4 * nb)), weights_only=False)
    v = st['variables']
    assert v['CFL/Generator/fc1/fully_connected/V'].shape == (14, 1024)
    assert v['CFL/Generator/subpixel_block1/Conv/V'].shape == (3, 3, 64, 512)
    assert v['CFL/Discriminator/conv2/Conv_4/V'].shape == (4, 4, 64, 128)
    assert v['CFL/Discriminator/latent_outputs/fully_connected/V'].shape[1] == 8
    # the encoder was warm-started from the no-gan run's best model and stays frozen in the post epoch
    from cfl.utils import latest_checkpoint
    best = torch.load(latest_checkpoint(str(ck / 'best_model')) + '.pt', weights_only=False)['variables']
    k = 'CFL/DistEncoder/outputs/fully_connected/V'
    assert np.array_equal(v[k], best[k])
    # the generator moved away from its initialisation and every scalar is finite
    assert np.isfinite(list(st['gan_powers']['g'])).all()
    m = st['adam_m']['CFL/Generator/outputs/Conv/V']
    assert np.isfinite(m).all() and np.abs(m).max() > 0
    rows = (tmp_path / 'logs' / 'dy' / gname / 'gan_scalars.tsv').read_text().splitlines()
    assert rows[0].split('\t')[0] == 'step' and 'd_total_loss' in rows[0] and len(rows) >= 2
    assert all(np.isfinite(float(c)) for c in rows[1].split('\t'))
    # resume: one more post epoch
    train.main(base + gan + ['--load-pre-weights', '--epochs', '3', '--post-epochs', '2', '--disable-eval'])
    assert (gck / 'model-{}.pt'.format(5 * nb)).exists()
    predict.start(base + ['--predict-root', str(tmp_path / 'pred')])
    assert (tmp_path / 'pred' / 'dy' / name / 'predict_acc.txt').exists()
    # cfl.bin.sample: grids from the trained generator (experiments/dyadic/sample_gen.sh)
    from PIL import Image
    from cfl.bin import sample
    sb = [f for f in base if f not in ('16',)]    # batch size 16 -> 20 (grids are 10 columns wide)
    sb = base[:base.index('--batch-size')] + ['--batch-size', '20'] + base[base.index('--batch-size') + 2:]
    for st in ('project', 'near', 'project_disc'):
        sample.start(sb + gan + ['--sample-root', str(tmp_path / 'samples'), '--sample-type', st])
    sdir = tmp_path / 'samples' / 'dy' / gname
    g = Image.open(str(sdir / 'project' / 'test_0000000000.png'))
    assert g.size == (16 * (2 * 2 + 1), 16 * 10)          # transposed grid: (K * B/10 + 1) columns of 10 images
    g = Image.open(str(sdir / 'near' / 'test_0000000000.png'))
    assert g.size == (16 * 3, 16 * 10)
    pd = sorted(os.listdir(str(sdir / 'project_disc')))
    assert any(f.endswith('_src.png') for f in pd) and any(f.endswith('_dst.png') for f in pd)
    main_png = [f for f in pd if not f.endswith(('_src.png', '_dst.png'))][0]
    assert Image.open(str(sdir / 'project_disc' / main_png)).size == (160, 16 * 2 * 2)


def test_gan_post_epoch_loop_is_the_same_with_and_without_its_host_pipeline(tmp_path, monkeypatch):
    """The post-epoch loop's host pipeline -- records decoded once into a table, labeled batches without their (unused)
    images, asynchronous pinned uploads -- must not change a single draw or value: the scalars and the final weights equal
    those of the plain loop (per-record decoding, synchronous uploads) bit for bit."""
    import shutil
    from cfl.bin import train
    from cfl.synthetic import make_double_dataset
    root = tmp_path / 'data'
    make_double_dataset(str(root / 'dy'), image_shape=(16, 16, 3), latent_dim=64, n_items=120, n_pos=96, n_neg=96, k=2, seed=9)

    def base(ck):
        return ['--data-name', 'dy', '--data-root', str(root), '--checkpoint-root', str(ck), '--log-root', str(ck) + '_logs',
                '--model-type', 'linear', '--data-type', 'tanh', '--data-mean', '0.5', '--data-norm', '0.5', '--data-directed',
                '--latent-norm', '31.9098', '--data-is-image', '--data-is-double', '--raw-latent', '--latent-shape', '64',
                '--input-shape', '16', '16', '3', '--dist-type', 'pcd', '--lambda-m', '0.5', '--use-threshold',
                '--num-components', '2', '--latent-size', '8', '--batch-size', '16', '--lr', '0.01', '--seed', '3']
    gan = ['--m-prj', '0.2', '--m-enc', '0.05', '--d-lr', '0.0002', '--d-beta1', '0.5', '--g-lr', '0.0002', '--g-beta1', '0.5',
           '--gan', '--gan-type', 'srgan', '--lambda-gp', '0.5', '--z-dim', '6', '--load-pre-weights', '--epochs', '1',
           '--post-epochs', '2', '--disable-eval']
    train.main(base(tmp_path / 'a') + ['--epochs', '1', '--reset'])
    shutil.copytree(str(tmp_path / 'a'), str(tmp_path / 'b'))
    train.main(base(tmp_path / 'a') + gan)
    for k, v in (('CFL_IMAGE_TABLE_MB', '0'), ('CFL_SYNC_UPLOAD', '1')):
        monkeypatch.setenv(k, v)
    train.main(base(tmp_path / 'b') + gan)
    gname = 'cfl_pcd_linear_tanh_ls_8_nc_2_ut_norm_0.5_lm_0.5_gan_z_6_m_prj_0.2_m_enc_0.05_dra_0.5_0.5_srgan'
    rows = [(tmp_path / (x + '_logs') / 'dy' / gname / 'gan_scalars.tsv').read_text() for x in 'ab']
    assert rows[0] == rows[1] and len(rows[0].splitlines()) >= 3
    nb = 96 // 16
    va, vb = (torch.load(str(tmp_path / x / 'dy' / gname / 'model-{}.pt'.format(3 * nb)), weights_only=False)['variables']
              for x in 'ab')
    assert set(va) == set(vb) and all(np.array_equal(va[k], vb[k]) for k in va)


def test_distance_epochs_on_resident_latents_equal_the_host_batches(tmp_path, monkeypatch):
    """Image + latent dataset, encoder on the latents: by default the latents of all records live in HBM and the distance epochs
    take the indexed kernels (ResidentFeatures over DecodedRecords.all_latents, byte offsets translated to table rows);
    CFL_DOUBLE_RESIDENT=0 assembles host batches as the reference does.  Same seeded streams, same arithmetic: the checkpoints,
    the best-model bookkeeping and -- the dataset object keeps its stream state across the phases -- the scalars of a
    following MrCGAN post epoch must be equal."""
    from cfl.bin import train
    from cfl.synthetic import make_double_dataset
    root = tmp_path / 'data'
    make_double_dataset(str(root / 'dy'), image_shape=(16, 16, 3), latent_dim=64, n_items=120, n_pos=112, n_neg=96, k=2, seed=9)

    def base(ck):
        return ['--data-name', 'dy', '--data-root', str(root), '--checkpoint-root', str(ck), '--log-root', str(ck) + '_logs',
                '--model-type', 'linear', '--data-type', 'tanh', '--data-mean', '0.5', '--data-norm', '0.5', '--data-directed',
                '--latent-norm', '31.9098', '--data-is-image', '--data-is-double', '--raw-latent', '--latent-shape', '64',
                '--input-shape', '16', '16', '3', '--dist-type', 'pcd', '--lambda-m', '0.5', '--use-threshold',
                '--num-components', '2', '--latent-size', '8', '--batch-size', '16', '--lr', '0.01', '--seed', '3']
    gan = ['--m-prj', '0.2', '--m-enc', '0.05', '--d-lr', '0.0002', '--d-beta1', '0.5', '--g-lr', '0.0002', '--g-beta1', '0.5',
           '--gan', '--gan-type', 'srgan', '--lambda-gp', '0.5', '--z-dim', '6', '--load-pre-weights', '--epochs', '3',
           '--post-epochs', '1', '--disable-eval']
    for tag, env in (('a', '1'), ('b', '0')):
        monkeypatch.setenv('CFL_DOUBLE_RESIDENT', env)
        train.main(base(tmp_path / tag) + ['--epochs', '3', '--reset'])
        train.main(base(tmp_path / tag) + gan)
    name = 'cfl_pcd_linear_tanh_ls_8_nc_2_ut_norm_0.5_lm_0.5'
    nb = 112 // 16
    va, vb = (torch.load(str(tmp_path / x / 'dy' / name / 'model-{}.pt'.format(3 * nb)), weights_only=False) for x in 'ab')
    for part in ('variables', 'adam_m', 'adam_v'):
        assert set(va[part]) == set(vb[part])
        for k in va[part]:
            assert np.array_equal(va[part][k], vb[part][k]), (part, k, np.abs(va[part][k] - vb[part][k]).max())
    best = [(tmp_path / x / 'dy' / name / 'best_model' / 'best_accuracy').read_text() for x in 'ab']
    assert best[0] == best[1]
    gname = name + '_gan_z_6_m_prj_0.2_m_enc_0.05_dra_0.5_0.5_srgan'
    rows = [(tmp_path / (x + '_logs') / 'dy' / gname / 'gan_scalars.tsv').read_text() for x in 'ab']
    assert rows[0] == rows[1] and len(rows[0].splitlines()) >= 2


def test_cfl_cgan_on_image_dataset(tmp_path):
    """experiments/mnist_30/run_cgan.sh in miniature: image-only dataset, linear encoder on the pixels, then the
    conditional-GAN baseline (--cgan, conv GAN type, gradient penalty), with and without --t-dim."""
    from cfl.bin import train
    from cfl.synthetic import make_double_dataset
    root = tmp_path / 'data'
    make_double_dataset(str(root / 'im'), image_shape=(16, 16, 1), n_items=120, n_pos=160, n_neg=160, k=2, seed=6,
                        double=False)
    base = ['--data-name', 'im', '--data-root', str(root), '--checkpoint-root', str(tmp_path / 'ck'),
            '--log-root', str(tmp_path / 'logs'), '--model-type', 'linear', '--data-type', 'sigmoid',
            '--data-is-image', '--input-shape', '16', '16', '1', '--dist-type', 'pcd', '--use-threshold',
            '--num-components', '2', '--latent-size', '8', '--batch-size', '16', '--lr', '0.01', '--seed', '4']
    train.main(base + ['--epochs', '2', '--reset', '--disable-eval'])
    for extra, tag in (([], '_cgan_z_6_dra_0.5_0.5'), (['--t-dim', '5'], '_cgan_z_6_t_5_dra_0.5_0.5')):
        gan = ['--gan', '--cgan', '--gan-type', 'conv', '--lambda-gp', '0.5', '--z-dim', '6'] + extra
        train.main(base + gan + ['--epochs', '0', '--post-epochs', '1', '--disable-eval'])
        gck = tmp_path / 'ck' / 'im' / ('cfl_pcd_linear_sigmoid_ls_8_nc_2_ut' + tag)
        st = torch.load(str(gck / 'model-10.pt'), weights_only=False)
        v = st['variables']
        assert ('CFL/Generator/fc_t/fully_connected/V' in v) == bool(extra)
        assert ('CFL/Discriminator/conv1/fc_t/fully_connected/V' in v) == bool(extra)
        cin = 64 + (5 if extra else 8)
        assert v['CFL/Discriminator/conv2/Conv/V'].shape == (5, 5, cin, 128)
        m = st['adam_m']['CFL/Generator/outputs/Conv2d_transpose/V']
        assert np.isfinite(m).all() and np.abs(m).max() > 0


def test_cfl_conv_encoder_then_gan(tmp_path):
    """experiments/mnist_30/run_gen.sh in miniature: vector dataset of 16x16 "pixels", ConvPCD encoder, then
    --gan (default conv GAN type) with m_prj / m_enc."""
    from cfl.bin import train
    from cfl.synthetic import make_dataset
    root = tmp_path / 'data'
    make_dataset(str(root / 'px'), D=256, n_items=200, n_pos=200, n_neg=200, k=2, latent=6, seed=2, scale=0.3)
    base = ['--data-name', 'px', '--data-root', str(root), '--checkpoint-root', str(tmp_path / 'ck'),
            '--log-root', str(tmp_path / 'logs'), '--model-type', 'conv', '--data-type', 'sigmoid', '--input-shape',
            '16', '16', '1', '--dist-type', 'pcd', '--lambda-m', '0.5', '--use-threshold', '--num-components', '2',
            '--latent-size', '8', '--batch-size', '20', '--seed', '5']
    train.main(base + ['--epochs', '1', '--reset', '--disable-eval'])
    gan = ['--m-prj', '0.5', '--m-enc', '0.1', '--d-lr', '0.001', '--d-beta1', '0.9', '--g-lr', '0.001', '--g-beta1',
           '0.9', '--gan', '--z-dim', '6']
    train.main(base + ['--epochs', '1', '--reset'])       # with evaluation: leaves a best_model to warm-start from
    train.main(base + gan + ['--load-pre-weights', '--epochs', '1', '--post-epochs', '1', '--disable-eval'])
    gck = tmp_path / 'ck' / 'px' / 'cfl_pcd_conv_sigmoid_ls_8_nc_2_ut_lm_0.5_gan_z_6_m_prj_0.5_m_enc_0.1'
    st = torch.load(str(gck / 'model-20.pt'), weights_only=False)
    v = st['variables']
    assert v['CFL/DistEncoder/conv1/Conv/V'].shape == (5, 5, 1, 64)
    # the conv trunk was warm-started from the no-gan run's best model and stays frozen in the post epoch
    from cfl.utils import latest_checkpoint
    best = torch.load(latest_checkpoint(str(tmp_path / 'ck' / 'px' / 'cfl_pcd_conv_sigmoid_ls_8_nc_2_ut_lm_0.5' /
                                            'best_model')) + '.pt', weights_only=False)['variables']
    for k in ('CFL/DistEncoder/conv1/Conv/V', 'CFL/DistEncoder/conv2/Conv/g', 'CFL/DistEncoder/outputs/fully_connected/V'):
        assert np.array_equal(v[k], best[k]), k
    assert v['CFL/Generator/conv_t1/Conv2d_transpose/V'].shape == (5, 5, 128, 128)
    assert v['CFL/Discriminator/conv2/Conv/V'].shape == (5, 5, 64, 128)
    m = st['adam_m']['CFL/Generator/fc1/fully_connected/V']
    assert np.isfinite(m).all() and np.abs(m).max() > 0


def test_cfl_random_crop_mirror_transformers(tmp_path):
    """--source-shape / --input-shape / --data-random-crop / --data-mirror: 20x20 source pixels, random 16x16 crops
    (central crop for validation and prediction), ConvPCD encoder."""
    from cfl.bin import predict, train
    from cfl.synthetic import make_dataset
    root = tmp_path / 'data'
    make_dataset(str(root / 'px'), D=400, n_items=200, n_pos=200, n_neg=200, k=2, latent=6, seed=3, scale=0.3)
    base = ['--data-name', 'px', '--data-root', str(root), '--checkpoint-root', str(tmp_path / 'ck'),
            '--log-root', str(tmp_path / 'logs'), '--model-type', 'conv', '--data-type', 'sigmoid', '--source-shape',
            '20', '20', '1', '--input-shape', '16', '16', '1', '--data-random-crop', '--data-mirror', '--dist-type',
            'pcd', '--use-threshold', '--num-components', '2', '--latent-size', '8', '--batch-size', '20', '--seed', '6']
    train.main(base + ['--epochs', '2', '--reset'])
    ck = tmp_path / 'ck' / 'px' / 'cfl_pcd_conv_sigmoid_ls_8_nc_2_ut'
    assert (ck / 'best_model' / 'best_accuracy').exists()
    v = torch.load(str(ck / 'model-20.pt'), weights_only=False)['variables']
    assert v['CFL/DistEncoder/conv1/Conv/V'].shape == (5, 5, 1, 64)
    predict.start(base + ['--predict-root', str(tmp_path / 'pred')])
    assert (tmp_path / 'pred' / 'px' / 'cfl_pcd_conv_sigmoid_ls_8_nc_2_ut' / 'predict.txt').exists()


def test_cfl_per_channel_normaliser_linear_and_conv(tmp_path):
    """--data-mean / --data-norm with one value per channel on RGB images: explicit normalisation pass in front of
    the linear heads (the pair kernels fold only scalar maps) and inside the conv trunk."""
    from cfl.bin import predict, train
    from cfl.synthetic import make_double_dataset
    root = tmp_path / 'data'
    make_double_dataset(str(root / 'rgb'), image_shape=(8, 8, 3), n_items=120, n_pos=160, n_neg=160, k=2, seed=8,
                        double=False)
    for mt, name in (('linear', 'cfl_pcd_linear_tanh_ls_8_nc_2_ut_norm_0.25_0.3_0.35'),
                     ('conv', 'cfl_pcd_conv_tanh_ls_8_nc_2_ut_norm_0.25_0.3_0.35')):
        base = ['--data-name', 'rgb', '--data-root', str(root), '--checkpoint-root', str(tmp_path / 'ck'),
                '--log-root', str(tmp_path / 'logs'), '--model-type', mt, '--data-type', 'tanh', '--data-mean', '0.5',
                '0.45', '0.4', '--data-norm', '0.25', '0.3', '0.35', '--data-is-image', '--input-shape', '8', '8', '3',
                '--dist-type', 'pcd', '--use-threshold', '--num-components', '2', '--latent-size', '8', '--batch-size',
                '16', '--lr', '0.01', '--seed', '9']
        train.main(base + ['--epochs', '2', '--reset'])
        ck = tmp_path / 'ck' / 'rgb' / name
        epoch, acc, auc = (ck / 'best_model' / 'best_accuracy').read_text().split('\t')
        assert np.isfinite(float(auc)) and float(auc) > 0.4
        predict.start(base + ['--predict-root', str(tmp_path / 'pred')])
        assert (tmp_path / 'pred' / 'rgb' / name / 'predict.txt').exists()


def test_cfl_directed_conv_encoders(tmp_path):
    """--directed with --model-type conv: separate source / target trunks (DistEncoderSrc / DistEncoderDst)."""
    from cfl.bin import predict, train
    from cfl.synthetic import make_dataset
    root = tmp_path / 'data'
    make_dataset(str(root / 'px'), D=256, n_items=200, n_pos=200, n_neg=200, k=2, latent=6, seed=4, scale=0.3)
    base = ['--data-name', 'px', '--data-root', str(root), '--checkpoint-root', str(tmp_path / 'ck'),
            '--log-root', str(tmp_path / 'logs'), '--model-type', 'conv', '--data-type', 'sigmoid', '--input-shape',
            '16', '16', '1', '--dist-type', 'pcd', '--use-threshold', '--directed', '--num-components', '2',
            '--latent-size', '8', '--batch-size', '20', '--seed', '7']
    train.main(base + ['--epochs', '2', '--reset'])
    ck = tmp_path / 'ck' / 'px' / 'cfl_pcd_conv_di_sigmoid_ls_8_nc_2_ut'
    v = torch.load(str(ck / 'model-20.pt'), weights_only=False)['variables']
    a, b = v['CFL/DistEncoderSrc/conv1/Conv/V'], v['CFL/DistEncoderDst/conv1/Conv/V']
    assert a.shape == b.shape == (5, 5, 1, 64) and not np.array_equal(a, b)
    assert 'CFL/DistEncoderDst/outputs/fully_connected/V' in v
    predict.start(base + ['--predict-root', str(tmp_path / 'pred')])
    assert (tmp_path / 'pred' / 'px' / 'cfl_pcd_conv_di_sigmoid_ls_8_nc_2_ut' / 'predict_acc.txt').exists()


def test_streamed_features_train_and_evaluate_like_resident_features(tmp_path, monkeypatch):
    """StreamedFeatures (a features.b that does not fit HBM, SURVEY 8(f).1's second option: mmap + pinned gather + async copy
    per batch; forced here with CFL_FEATURES=stream) against ResidentFeatures on the same dataset and seeds: the same index
    stream (heads, reshuffles, data_switch flips), bit-identical parameters after training through train_steps (with the
    validation fetch), bit-identical logged scalars, and the same dist_eval AUC / accuracy."""
    from cfl import hipabi as H
    from cfl import input_data, utils
    from cfl.bin import train_dist as TD
    from cfl.engine import PairEngine
    from cfl.synthetic import make_dataset
    D, L, K, B, total = 200, 6, 2, 48, 70          # D % 64 != 0: the pad columns of the staging rows stay zero
    make_dataset(str(tmp_path / 'toy'), D=D, n_items=300, n_pos=500, n_neg=420, k=2, latent=6, seed=3, scale=4.0,
                 splits=(('train', 1.0), ('val', 0.4)))
    rng = np.random.RandomState(0)
    cfg = O.EncoderCfg(D=256, L=L, K=K)
    params = O.init_encoder_params(cfg, rng, np.float32)
    for k in list(params):
        if k.endswith('/W'):
            params[k][D:] = 0.0

    class Model(object):
        pass

    def run(mode):
        monkeypatch.setenv('CFL_FEATURES', mode)
        tr = input_data.SemiDataSet(str(tmp_path / 'toy' / 'train'), input_size=D, data_switch=True, seed=9)
        va = input_data.SemiDataSet(str(tmp_path / 'toy' / 'val'), input_size=D, data_switch=False, seed=4)
        st, sv = input_data.feature_source(tr), input_data.feature_source(va)
        assert type(st).__name__ == ('StreamedFeatures' if mode == 'stream' else 'ResidentFeatures')
        m = Model()
        m.engine = PairEngine(256, L, K, norm=H.make_norm(1 / 16.0, valid_cols=D), loss=H.make_loss(reg_const=1e-3), params=params,
                              lr=2e-3, batch_size=B)
        seen = []
        TD.train_steps(m, st, sv, B, None, total, lambda i, s, v: seen.append((i, s, v)), scalar_every=5)
        # scoring of every validation pair through the source's chunked iterator
        sc = [torch.cat([m.engine.scores(t, s).clone() for t, s in sv.whole_indexed(w, 64)]) for w in ('pos', 'neg')]
        torch.cuda.synchronize()
        return m.engine, tr, va, seen, sc
    a, b = run('resident'), run('stream')
    assert torch.equal(a[0].theta, b[0].theta) and torch.equal(a[0].m, b[0].m)
    assert [(i, s) for i, s, _ in a[3]] == [(i, s) for i, s, _ in b[3]]
    assert [v for _, _, v in a[3]] == [v for _, _, v in b[3]]
    for x, y in ((a[1], b[1]), (a[2], b[2])):
        assert x.head_labeled_pos == y.head_labeled_pos and np.array_equal(x.pairs_pos, y.pairs_pos)
        assert x._rng.rand() == y._rng.rand()
    for x, y in zip(a[4], b[4]):
        assert torch.equal(x, y)
