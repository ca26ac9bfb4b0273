"""Flag parsers, evaluation / prediction loops and checkpoint helpers.

Host-side mirror of the reference's cfl/utils.py for the pair-distance path:
same function names, flag surface and file grammars; the
TensorFlow session is replaced by the model objects of cfl.models (the ``sess``
positional argument is kept for call-site compatibility and ignored).
"""
import argparse
import logging
import os
import re
from argparse import Namespace

import numpy as np

logger = logging.getLogger(__name__)


def log_args(args):
    for name, value in sorted(vars(args).items()):
        logger.warning('%s = %r', name, value)


def reduce_product(xs):
    out = 1
    for x in xs:
        out *= x
    return out


class IncrementalAverage(object):
    """Running mean (cfl/utils.py:500-507)."""

    def __init__(self):
        self.average = 0.0
        self.count = 0

    def add(self, value):
        self.count += 1
        self.average += (value - self.average) / self.count


# ---------------------------------------------------------------------------
# best-stats file: "<epoch>\t<accuracy>\t<auc>" without newline (App. B)
# ---------------------------------------------------------------------------
def load_best_stats(path):
    stats = Namespace(best_epoch=None, best_accuracy=0.0, best_auc=0.0)
    if os.path.exists(path):
        with open(path) as infile:
            tokens = infile.read().split('\t')
        stats.best_epoch = int(tokens[0])
        stats.best_accuracy = float(tokens[1])
        if len(tokens) >= 3:
            stats.best_auc = float(tokens[2])
    return stats


def save_best_stats(path, epoch, accuracy, auc):
    with open(path, 'w') as outfile:
        outfile.write('{}\t{}\t{}'.format(epoch, accuracy, auc))


# ---------------------------------------------------------------------------
# flag surface
# ---------------------------------------------------------------------------
def monomer_parser(data_name='monomer/Baby-also_viewed', data_root='parsed_data',
                   checkpoint_root='checkpoints', log_root='logs', run_tag=None, seed=633,
                   input_shape=(4096,), batch_size=100, normalize_value=1.0, num_components=2,
                   latent_size=20, lr=0.001, beta1=0.9, beta2=0.999, reg_const=0.0):
    """cfl/utils.py:510-553."""
    p = argparse.ArgumentParser()
    for flag, default in (('--data-name', data_name), ('--data-root', data_root),
                          ('--checkpoint-root', checkpoint_root), ('--log-root', log_root),
                          ('--run-tag', run_tag)):
        p.add_argument(flag, default=default)
    p.add_argument('--seed', type=int, default=seed)
    p.add_argument('--normalize-value', type=float, default=normalize_value)
    p.add_argument('--input-shape', nargs='+', type=int, default=input_shape, help='shape of input')
    p.add_argument('--batch-size', type=int, default=batch_size)
    p.add_argument('--num-components', type=int, default=num_components)
    p.add_argument('--latent-size', type=int, default=latent_size)
    for flag, default in (('--lr', lr), ('--beta1', beta1), ('--beta2', beta2),
                          ('--reg-const', reg_const)):
        p.add_argument(flag, type=float, default=default)
    return p


_CHOICES = ('sigmoid', 'tanh', 'relu', 'linear')


def dist_parser(data_name='mnist', data_root='parsed_data', checkpoint_root='checkpoints',
                log_root='logs', run_tag=None, seed=633, source_shape=None, input_shape=(28, 28, 1),
                ae_shape=None, batch_size=100, data_scale=None, data_mean=None, data_norm=None,
                latent_norm=None, data_type='sigmoid', model_type='conv', dist_type='pcd',
                act_type=None, num_components=2, latent_size=20, pos_weight=None, lr=0.001,
                beta1=0.9, beta2=0.999, z_dim=20, z_stddev=1., g_dim=64, g_lr=0.0002, g_beta1=0.5,
                g_beta2=0.999, m_prj=None, m_enc=None, gan_type='conv', caffe_margin=None,
                d_dim=64, t_dim=None, d_lr=0.0002, d_beta1=0.5, d_beta2=0.999, lambda_dra=0.5,
                lambda_gp=None, lambda_m=0.0, reg_const=0.0):
    """cfl/utils.py:84-224 -- the whole flag set is accepted; flags that belong to
    the MrCGAN phase are parsed and carried, see cfl.models.cfl for what is built."""
    p = argparse.ArgumentParser()
    A = p.add_argument
    # run
    A('--data-name', default=data_name)
    A('--data-root', default=data_root)
    A('--checkpoint-root', default=checkpoint_root)
    A('--log-root', default=log_root)
    A('--run-tag', default=run_tag)
    A('--seed', type=int, default=seed)
    # data: (data * data_scale - mean) / normalize_value
    A('--source-shape', nargs='+', type=int, default=source_shape, help='shape of input source')
    A('--input-shape', nargs='+', type=int, default=input_shape, help='shape to crop for model')
    A('--ae-shape', nargs='+', type=int, default=ae_shape, help='shape for autoencoder / generator')
    A('--batch-size', type=int, default=batch_size)
    A('--data-scale', type=float, default=data_scale)
    A('--data-mean', nargs='+', type=float, default=data_mean)
    A('--data-norm', nargs='+', type=float, default=data_norm)
    A('--data-type', default=data_type, choices=_CHOICES, help='range of input data; enforce clip as well')
    for flag in ('--data-mirror', '--data-random-crop', '--data-is-image', '--data-is-double',
                 '--data-disable-double', '--raw-latent'):
        A(flag, action='store_true')
    A('--latent-shape', type=int, nargs='+')
    A('--latent-norm', type=float, default=latent_norm)
    # model
    A('--pos-weight', type=float, default=pos_weight)
    A('--model-type', default=model_type, choices=('linear', 'conv'))
    A('--num-components', type=int, default=num_components)
    A('--latent-size', type=int, default=latent_size)
    for flag, default in (('--lr', lr), ('--beta1', beta1), ('--beta2', beta2)):
        A(flag, type=float, default=default)
    A('--z-dim', type=int, default=z_dim)
    A('--z-stddev', type=float, default=z_stddev)
    A('--g-dim', type=int, default=g_dim)
    for flag, default in (('--g-lr', g_lr), ('--g-beta1', g_beta1), ('--g-beta2', g_beta2),
                          ('--m-prj', m_prj), ('--m-enc', m_enc)):
        A(flag, type=float, default=default)
    A('--d-dim', type=int, default=d_dim)
    for flag, default in (('--d-lr', d_lr), ('--d-beta1', d_beta1), ('--d-beta2', d_beta2),
                          ('--lambda-dra', lambda_dra), ('--lambda-gp', lambda_gp),
                          ('--lambda-m', lambda_m)):
        A(flag, type=float, default=default)
    A('--gan-type', default=gan_type)
    A('--gan', action='store_true', help='enable GAN')
    A('--cgan', action='store_true', help='enable CGAN')
    A('--t-dim', type=int, default=t_dim)
    A('--dist-type', help='distance type', choices=('monomer', 'pcd', 'siamese'), default=dist_type)
    A('--act-type', help='act type', choices=_CHOICES, default=act_type)
    A('--use-threshold', action='store_true')
    A('--caffe-margin', type=float, default=caffe_margin, help='margin for CD loss for siamese training')
    A('--directed', action='store_true')
    A('--data-directed', action='store_true')
    A('--reg-const', type=float, default=reg_const)
    return p


def dist_check_args(args):
    """Validation rules of cfl/utils.py:45-81 (same assertions)."""
    if args.data_is_double:
        assert args.data_is_image
        assert args.latent_shape is not None
        args.latent_shape = tuple(args.latent_shape)
    for name in ('input_shape', 'ae_shape', 'source_shape', 'data_mean', 'data_norm'):
        if getattr(args, name):
            setattr(args, name, tuple(getattr(args, name)))
    if args.caffe_margin and args.caffe_margin > 0:
        assert args.dist_type == 'siamese', 'only use cd loss in siamese'
    if args.caffe_margin and args.caffe_margin < 0:
        assert args.lambda_m > 0
    if args.lambda_m > 0:
        assert not args.caffe_margin
    if args.dist_type != 'siamese':
        assert args.use_threshold, 'must use entropy loss'
    if not args.cgan:
        assert not args.t_dim
    log_args(args)


# ---------------------------------------------------------------------------
# evaluation / prediction over whole splits
# ---------------------------------------------------------------------------
def _scores(model, data, batch_size, negative):
    """Concatenated scores [n] of all positive (or negative) pairs of a split, through
    the model's scoring entry (``val_s_pos_predicts.outputs`` in the reference)."""
    fast = _resident_scores(model, data, negative)
    if fast is not None:
        return fast
    it = data.whole_neg_batches(batch_size) if negative else data.whole_pos_batches(batch_size)
    # double (image + latent) data: the encoder reads the latents (cfl/utils.py:234-244)
    pick = getattr(model, 'select_pair', lambda b: (b[0], b[1]))
    out = [model.predict(*pick(b)).reshape(-1) for b in it]
    return np.concatenate(out) if out else np.zeros(0, np.float32)


RESIDENT_EVAL_ROWS = 32768   # pairs per scoring launch when the split's features live in HBM (measured on one MI355X:
#                              8192 pairs per call reach 0.40-0.42 of the HBM roof, 32768 reach 0.53-0.56: profiles/r03_d_eval_*,
#                              bench.py roofline_eval.dist_eval_call)


def _resident_ready(model, data):
    """The split's resident feature table when the in-place scoring path applies (vector dataset, linear encoder, no
    per-batch transformer, matching padded width), else None.  Cheap and side-effect free apart from the one-time
    upload, so that every rank of a data-parallel run can decide -- identically -- BEFORE the first collective."""
    if getattr(model, 'trunk', None) is not None or not hasattr(model, 'engine'):
        return None
    if getattr(data, 'is_image', True):
        # image + latent dataset read through its latents: the latents of all records are the resident table
        # (input_data.ResidentFeatures); the image transformers do not touch them (cfl.models.cfl: _prep)
        if not (getattr(data, 'is_double', False) and getattr(model, 'uses_latent', False)) or \
                os.environ.get('CFL_DOUBLE_RESIDENT', '1') in ('0', ''):
            return None
    elif getattr(model, 'val_data_transformer', None) is not None or getattr(model, '_explicit_norm', None) is not None:
        return None
    try:
        import torch
        from .input_data import feature_source
    except Exception:          # pragma: no cover
        return None
    if not torch.cuda.is_available():
        return None
    res = getattr(data, '_resident', None)
    if res is None or res.device != model.device:
        res = data._resident = feature_source(data, model.device)
    if res.padded_size != model.padded_size:
        return None
    return res


def _resident_scores(model, data, negative, device_result=False, rows=None):
    """Vector datasets with a linear encoder: the split's features.b is uploaded once (cached on the dataset
    object) and the scoring kernels read the pair rows in place by index (cfl_pair_scores_idx) in large chunks --
    the same scores as the per-batch path, without one host-to-device copy of 2 x [batch, D] floats per batch.
    Returns None when the fast path does not apply (image data, conv encoder, models without an engine)."""
    res = _resident_ready(model, data)
    if res is None:
        return None
    import torch
    out = [model.engine.scores(table, streams)
           for table, streams in res.whole_indexed('neg' if negative else 'pos', RESIDENT_EVAL_ROWS, rows)]
    if rows is not None and not out:
        out = [torch.empty(0, dtype=torch.float32, device=model.device)]
    if device_result:
        return torch.cat(out).contiguous() if out else None
    return torch.cat(out).cpu().numpy().astype(np.float32) if out else np.zeros(0, np.float32)


def _sharded_eval(model, data, batch_size):
    """Data-parallel evaluation (new functionality, the reference is single-device).  Collective: EVERY
    rank must call dist_eval.  Returns None when there is one rank.
      * resident fast path (vector dataset, linear encoder): every rank scores a contiguous shard of the positive and
        of the negative pair list, the scores are gathered on rank 0, which computes AUC / accuracy (cfl_auc);
      * otherwise (image data, conv encoder, per-batch transformers, padded-width mismatch): rank 0 evaluates the
        whole split the single-GPU way.
    Either way rank 0 broadcasts (auc, accuracy), so that every rank takes the same best-model decisions.  Which
    branch runs is decided from properties every rank shares (_resident_ready), before any collective."""
    from . import engine as dpar
    world = dpar.world_size()
    if world <= 1:
        return None
    import torch
    import torch.distributed as dist
    rank = dpar.rank()
    result = torch.zeros(2, dtype=torch.float64, device=model.device)
    if _resident_ready(model, data) is None:
        if rank == 0:
            r = _local_eval(model, batch_size, data)
            result[0], result[1] = r.auc, r.accuracy
    else:
        parts = []
        for negative, pairs in ((False, data.pairs_pos), (True, data.pairs_neg)):
            n = pairs.shape[0]
            per = (n + world - 1) // world
            lo, hi = min(rank * per, n), min((rank + 1) * per, n)
            mine = _resident_scores(model, data, negative, True, rows=(lo, hi))
            padded = torch.zeros(per, dtype=torch.float32, device=model.device)
            padded[:hi - lo] = mine
            gathered = [torch.empty_like(padded) for _ in range(world)] if rank == 0 else None
            dist.gather(padded, gathered, dst=0)
            if rank == 0:
                parts.append(torch.cat([g[:min((r + 1) * per, n) - min(r * per, n)] for r, g in enumerate(gathered)]))
        if rank == 0:
            from . import hipgan
            auc, acc = hipgan.auc(parts[0].contiguous(), parts[1].contiguous())
            result[0], result[1] = auc, acc
    dist.broadcast(result, 0)
    auc, acc = float(result[0]), float(result[1])
    return Namespace(error=1.0 - acc, accuracy=acc, auc=auc, roc=None)


def _local_eval(model, batch_size, data):
    """dist_eval of one process over the whole split."""
    dp, dn = (_resident_scores(model, data, False, True), _resident_scores(model, data, True, True))
    if dp is not None and dn is not None and dp.numel() and dn.numel():
        from . import hipgan
        auc, acc = hipgan.auc(dp, dn)
        return Namespace(error=1.0 - acc, accuracy=acc, auc=auc, roc=None)
    from sklearn.metrics import roc_auc_score, roc_curve
    pos = _scores(model, data, batch_size, False)
    neg = _scores(model, data, batch_size, True)
    total = pos.shape[0] + neg.shape[0]
    correct = int((pos > 0).sum()) + int((neg <= 0).sum())
    y_true = [1] * pos.shape[0] + [0] * neg.shape[0]
    y_score = np.concatenate([pos, neg])
    return Namespace(error=(total - correct) / total, accuracy=correct / total,
                     auc=roc_auc_score(y_true, y_score), roc=roc_curve(y_true, y_score))


def dist_eval(sess, model, batch_size, data):
    """cfl/utils.py:227-274: accuracy by the sign of the score, AUC, ROC.  With HBM-resident features the scores
    never leave the GPU: AUC and accuracy come from cfl_auc (sort-based, ties = half credit, equal to
    sklearn's roc_auc_score) and `roc` is not materialised (no caller reads it)."""
    sharded = _sharded_eval(model, data, batch_size)
    if sharded is not None:
        return sharded
    return _local_eval(model, batch_size, data)


def dist_predict(sess, model, data, batch_size, predict_dir, output_name):
    """cfl/utils.py:277-321: '<id1> match <id2> <score>' per pair, positives first;
    the score is printed as ``str(np.float32)``."""
    logger.warning('predict %s...', output_name)
    os.makedirs(predict_dir, exist_ok=True)
    with open(os.path.join(predict_dir, output_name), 'w') as outfile:
        for negative, pairs in ((False, data.pairs_pos), (True, data.pairs_neg)):
            scores = _scores(model, data, batch_size, negative).astype(np.float32)
            for s, (i1, i2) in zip(scores, pairs):
                outfile.write('{} match {} {}\n'.format(data.index_to_asins[i1],
                                                        data.index_to_asins[i2], s))


# ---------------------------------------------------------------------------
# sampling grids of the MrCGAN generator (cfl/utils.py:324-462, cfl/ops.py:245-259)
# ---------------------------------------------------------------------------
def np_arrange_grid(rows, num_rows, num_cols, image_shape, transpose=False):
    """cfl/ops.py:245-259: [num_rows*num_cols, H*W*C] -> one [1, rows*H, cols*W, C] image."""
    rows = np.reshape(rows, (num_rows, num_cols, image_shape[0], image_shape[1], image_shape[2]))
    if transpose:
        num_rows, num_cols = num_cols, num_rows
        rows = np.transpose(rows, (1, 0, 2, 3, 4))
    height, width = image_shape[0] * num_rows, image_shape[1] * num_cols
    rows = np.transpose(rows, (0, 1, 3, 2, 4))
    rows = np.reshape(rows, (num_rows, width, image_shape[0], image_shape[2]))
    rows = np.transpose(rows, (0, 2, 1, 3))
    return np.reshape(rows, (1, height, width, image_shape[2]))


def imsave(path, image):
    """scipy.misc.imsave semantics: the array's min..max is stretched to 0..255."""
    from PIL import Image
    a = np.asarray(image, np.float64)
    if a.ndim == 3 and a.shape[2] == 1:
        a = a[:, :, 0]
    lo, hi = a.min(), a.max()
    a = np.zeros_like(a) if hi == lo else (a - lo) * (255.0 / (hi - lo))
    Image.fromarray(np.clip(np.rint(a), 0, 255).astype(np.uint8)).save(path)


def _tile_parts(model, parts, repeats):
    """Encoder input of a 10-item chunk tiled to batch_size rows + the images (cfl/utils.py:331-340)."""
    o = 1 if (model.is_double and model.uses_latent) else 0
    return np.tile(parts[o], (repeats, 1)), parts[0]


def dist_sample(sess, model, data, batch_size, sample_dir, output_name):
    """'project': per chunk of 10 positive pairs, row 0 = the source images, then batch_size/10 rows per
    component of G(z, prototype_k(source)) (gen_grid_all, cfl/models/cfl.py:1265-1288)."""
    os.makedirs(sample_dir, exist_ok=True)
    logger.warning('sample %s...', output_name)
    repeats, K = batch_size // 10, model.num_components
    for i, batches in enumerate(data.whole_pos_batches(10)):
        src = batches[:2] if model.is_double else batches[:1]
        enc, images = _tile_parts(model, src, repeats)
        if enc.shape[0] != batch_size:
            continue
        protos, _ = model.generate_prototypes(enc)
        rows = [model.ae_normalizer(images[:10]) if model.ae_normalizer is not None else images[:10]]
        for r in range(repeats):
            for k in range(K):
                rows.append(protos[k][10 * r:10 * r + 10])
        rows = np.concatenate(rows, 0)
        if model.ae_unnormalizer is not None:
            rows = model.ae_unnormalizer(rows)
        grid = np_arrange_grid(rows, K * repeats + 1, 10, model.ae_shape, transpose=True)[0]
        imsave(os.path.join(sample_dir, '{}_{:010d}.png'.format(output_name, i)), grid)


def dist_sample_near(sess, model, data, batch_size, sample_dir, output_name):
    """'near': row 0 = the target images, then G(z, encoder(target)) (gen_grid_all_target)."""
    os.makedirs(sample_dir, exist_ok=True)
    logger.warning('sample %s...', output_name)
    repeats = batch_size // 10
    for i, batches in enumerate(data.whole_pos_batches(10)):
        dst = batches[2:4] if model.is_double else batches[1:2]
        enc, images = _tile_parts(model, dst, repeats)
        if enc.shape[0] != batch_size:
            continue
        gen = model.generate_target(enc)
        rows = np.concatenate([model.ae_normalizer(images[:10]) if model.ae_normalizer is not None else images[:10],
                               gen[:10 * repeats]], 0)
        if model.ae_unnormalizer is not None:
            rows = model.ae_unnormalizer(rows)
        grid = np_arrange_grid(rows, 1 + repeats, 10, model.ae_shape, transpose=True)[0]
        imsave(os.path.join(sample_dir, '{}_{:010d}.png'.format(output_name, i)), grid)


def dist_sample_project_disc(sess, model, data, batch_size, sample_dir, output_name):
    """'project_disc': for every positive pair, batch_size samples per component sorted by the
    discriminator's score, plus the pair's own images (cfl/utils.py:381-462)."""
    os.makedirs(sample_dir, exist_ok=True)
    logger.warning('sample %s...', output_name)
    K = model.num_components
    o = 1 if (model.is_double and model.uses_latent) else 0
    per = 2 if model.is_double else 1
    for batches in data.whole_pos_batches(batch_size, source_ids=True):
        for item in zip(*batches):
            enc = np.tile(np.asarray(item[o]).reshape(1, -1), (batch_size, 1))
            protos, preds = model.generate_prototypes(enc)
            images = np.concatenate(protos)
            if model.ae_unnormalizer is not None:
                images = model.ae_unnormalizer(images)
            images = np.reshape(images, (images.shape[0], -1))
            order = np.argsort(-np.concatenate(preds).flatten())
            grid = np_arrange_grid(images[order], batch_size // 10 * K, 10, model.ae_shape)[0]
            imsave(os.path.join(sample_dir, '{}_{}.png'.format(output_name, item[-1])), grid)
            imsave(os.path.join(sample_dir, '{}_{}_src.png'.format(output_name, item[-1])),
                   np.asarray(item[0]).reshape(model.ae_shape))
            imsave(os.path.join(sample_dir, '{}_{}_dst.png'.format(output_name, item[-1])),
                   np.asarray(item[per]).reshape(model.ae_shape))


class ScalarWriter(object):
    """Stand-in for tf.summary.FileWriter (cfl/bin/train.py:55): appends the scalar summaries of the
    reference's `summary` / `post_summary` ops (cfl/models/cfl.py:1107-1243) to <log_dir>/<name>.tsv, one
    header line per file, one row per logged step."""

    def __init__(self, log_dir):
        self.log_dir = log_dir
        self._files = {}

    def add_scalars(self, name, step, values):
        f = self._files.get(name)
        keys = sorted(values)
        if f is None:
            path = os.path.join(self.log_dir, name + '.tsv')
            new = not os.path.exists(path) or os.path.getsize(path) == 0
            f = self._files[name] = open(path, 'a')
            if new:
                f.write('step\t' + '\t'.join(keys) + '\n')
        f.write(str(step) + '\t' + '\t'.join(repr(float(values[k])) for k in keys) + '\n')
        f.flush()

    def close(self):
        for f in self._files.values():
            f.close()
        self._files = {}


# ---------------------------------------------------------------------------
# checkpoints: reference directory layout, own tensor container
# ---------------------------------------------------------------------------
class Saver(object):
    """Stand-in for tf.train.Saver: ``<dir>/checkpoint`` state file in TensorFlow's
    text grammar plus ``<prefix>-<step>.pt`` (torch.save of {TF variable name:
    array in the reference layout, Adam slots, beta powers}); keeps the last 5."""

    def __init__(self, max_to_keep=5):
        self.max_to_keep = max_to_keep
        self._kept = []

    @staticmethod
    def _plain(obj):
        """numpy arrays / scalars -> torch tensors / python numbers, so that the file holds nothing torch.load
        needs to unpickle (it is read back with weights_only=True)"""
        import torch
        if isinstance(obj, dict):
            return {k: Saver._plain(v) for k, v in obj.items()}
        if isinstance(obj, (list, tuple)):
            return [Saver._plain(v) for v in obj]
        if isinstance(obj, np.ndarray):
            return torch.from_numpy(np.ascontiguousarray(obj))
        if isinstance(obj, np.generic):
            return obj.item()
        return obj

    def _seed_kept(self, dirname):
        """a saver created for a resumed run continues the directory's `checkpoint` list (tf.train.Saver recovers
        it from the state file), so that old model-* files keep being pruned"""
        state = os.path.join(dirname, 'checkpoint')
        if self._kept or not os.path.exists(state):
            return
        with open(state) as f:
            names = re.findall(r'all_model_checkpoint_paths:\s*"([^"]+)"', f.read())
        self._kept = [os.path.join(dirname, os.path.basename(n)) for n in names
                      if os.path.exists(os.path.join(dirname, os.path.basename(n)) + '.pt')]

    def save(self, model, save_path, global_step):
        import torch
        path = '{}-{}'.format(save_path, global_step)
        os.makedirs(os.path.dirname(path), exist_ok=True)
        self._seed_kept(os.path.dirname(path))
        torch.save(self._plain(model.checkpoint_state()), path + '.pt')
        self._kept = [p for p in self._kept if p != path] + [path]
        while len(self._kept) > self.max_to_keep:
            old = self._kept.pop(0)
            if os.path.exists(old + '.pt'):
                os.remove(old + '.pt')
        names = [os.path.basename(p) for p in self._kept]
        with open(os.path.join(os.path.dirname(path), 'checkpoint'), 'w') as f:
            f.write('model_checkpoint_path: "{}"\n'.format(names[-1]))
            for n in names:
                f.write('all_model_checkpoint_paths: "{}"\n'.format(n))
        return path

    def restore(self, model, path):
        import torch
        model.load_checkpoint_state(load_checkpoint_file(path + '.pt'))


def load_checkpoint_file(path):
    """A checkpoint as {'variables': {TF name: array}, 'adam_m', 'adam_v', ...}: tensors only, no pickled code."""
    import pickle
    import torch
    try:
        return torch.load(path, map_location='cpu', weights_only=True)
    except pickle.UnpicklingError as e:
        # files written before the tensors-only format pickled numpy arrays: not loadable with weights_only=True
        raise RuntimeError(
            '%s was written by an older build of this package (it pickles numpy objects, which are no longer '
            'unpickled on load).  Convert it once with\n    python -c "import torch; from cfl.utils import Saver; '
            "d = torch.load(%r, map_location='cpu', weights_only=False); torch.save(Saver._plain(d), %r)\"\n"
            'after checking that the file comes from a source you trust.  (%s)' % (path, path, path, e))


def latest_checkpoint(checkpoint_dir):
    state = os.path.join(checkpoint_dir, 'checkpoint')
    if not os.path.exists(state):
        return None
    with open(state) as f:
        m = re.search(r'model_checkpoint_path:\s*"([^"]+)"', f.read())
    if not m:
        return None
    path = os.path.join(checkpoint_dir, os.path.basename(m.group(1)))
    return path if os.path.exists(path + '.pt') else None


def _step_of(path):
    return int(re.search(r'(\d+)', os.path.basename(path).split('-')[-1]).group(1)) + 1


def load_model(model, checkpoint_dir, load_pre_weights=None):
    """cfl/utils.py:465-497: restore the latest checkpoint of ``checkpoint_dir`` and
    resume at (its step + 1); otherwise optionally warm-start the trainable
    variables from ``<load_pre_weights>/best_model`` and take the start step from
    that run's latest checkpoint.  Returns (saver, start_step)."""
    saver = Saver()
    path = latest_checkpoint(checkpoint_dir)
    if path:
        saver.restore(model, path)
        logger.info('%s loaded', path)
        return saver, _step_of(path)
    start_step = 0
    if load_pre_weights:
        best = latest_checkpoint(os.path.join(load_pre_weights, 'best_model'))
        last = latest_checkpoint(load_pre_weights)
        if not (best and last):
            raise Exception('must have best model! %s' % os.path.join(load_pre_weights, 'best_model'))
        model.assign_trainable(load_checkpoint_file(best + '.pt'), ignore_missing=True)
        start_step = _step_of(last)
        logger.info('%s loaded', best)
    return saver, start_step
