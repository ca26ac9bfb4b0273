"""Generate golden vectors for the data / eval side by IMPORTING the reference.

Run in the build container only (the reference tree never travels):

    python tests/golden/make_data_goldens.py

It imports ``/root/reference/cfl/input_data.py`` (with inert stand-ins for the
``tensorflow`` and ``scipy.misc`` *imports* only -- nothing of them is executed on
the vector-dataset path) and ``/root/reference/cfl/bin/evaluate_total.py``
(imports unmodified), runs them on a small seeded toy dataset and stores inputs
and outputs in ``tests/golden/data_goldens.npz`` / ``eval_goldens.json``.
Only data (inputs + expected outputs) is stored; no reference source.
"""
import contextlib
import io
import json
import os
import shutil
import sys
import tempfile
import types

import numpy as np

REF = '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))


def _stub_modules():
    tf = types.ModuleType('tensorflow')
    ex = types.ModuleType('tensorflow.examples')
    tut = types.ModuleType('tensorflow.examples.tutorials')
    mn = types.ModuleType('tensorflow.examples.tutorials.mnist')
    tut.mnist = mn
    ex.tutorials = tut
    tf.examples = ex
    sys.modules.update({
        'tensorflow': tf, 'tensorflow.examples': ex,
        'tensorflow.examples.tutorials': tut,
        'tensorflow.examples.tutorials.mnist': mn})
    import scipy
    misc = types.ModuleType('scipy.misc')
    misc.imread = misc.imsave = misc.imresize = None
    sys.modules['scipy.misc'] = misc
    scipy.misc = misc


def make_toy_split(ref_input_data, path, n_items, D, n_pos, n_neg, rng,
                   directed=False):
    os.makedirs(path, exist_ok=True)
    ids = ['%010d' % (i * 7 + 3) for i in range(n_items)]
    feats = rng.randn(n_items, D).astype(np.float32)
    with open(os.path.join(path, 'features.b'), 'wb') as f:
        for i in range(n_items):
            f.write(ids[i].encode('ascii'))
            ref_input_data.dump_array(f, feats[i])     # cfl/input_data.py:23-31
    pos = rng.randint(0, n_items, size=(n_pos, 2))
    neg = rng.randint(0, n_items, size=(n_neg, 2))
    with open(os.path.join(path, 'pairs_pos.txt'), 'w') as f:
        for a, b in pos:
            f.write('{} match {}\n'.format(ids[a], ids[b]))
    with open(os.path.join(path, 'pairs_neg.txt'), 'w') as f:
        for a, b in neg:
            f.write('{} also_viewed {}\n'.format(ids[a], ids[b]))
    if directed:
        with open(os.path.join(path, 'source.txt'), 'w') as f:
            for i in range(0, n_items, 2):
                f.write(ids[i] + '\n')
        with open(os.path.join(path, 'target.txt'), 'w') as f:
            for i in range(1, n_items, 2):
                f.write(ids[i] + '\n')
    return ids, feats, pos, neg


def main():
    _stub_modules()
    sys.path.insert(0, REF)
    import cfl.input_data as ref_in
    import cfl.bin.evaluate_total as ref_ev

    out = {}
    meta = {}
    tmp = tempfile.mkdtemp()
    try:
        rng = np.random.RandomState(20261002)
        D = 5
        root = os.path.join(tmp, 'toy')
        spec = {'train': (23, 17, 11), 'val': (9, 7, 4), 'test': (12, 10, 13)}
        for split, (n, npos, nneg) in spec.items():
            ids, feats, pos, neg = make_toy_split(
                ref_in, os.path.join(root, split), n, D, npos, nneg, rng,
                directed=True)
            with open(os.path.join(root, split, 'features.b'), 'rb') as f:
                out[split + '/features_b'] = np.frombuffer(f.read(), np.uint8)
            out[split + '/feats'] = feats
            out[split + '/pos'] = pos
            out[split + '/neg'] = neg
            meta[split + '/ids'] = ids

        # G2 + G3 + G4 on the reference SemiDataSet
        for seed in (0, 633):
            for data_switch in (False, True):
                tag = 'seed{}_sw{}'.format(seed, int(data_switch))
                data = ref_in.load_data_sets(root, D, data_switch=data_switch,
                                             seed=seed)
                tr = data.train
                out[tag + '/train/pairs_pos0'] = tr.pairs_pos.copy()
                out[tag + '/train/pairs_neg0'] = tr.pairs_neg.copy()
                meta[tag + '/train/num_examples'] = int(tr.num_examples)
                meta[tag + '/train/asins_to_index'] = dict(tr.asins_to_index)
                # 12 consecutive batches of 4 (17 pos / 11 neg pairs -> several
                # epoch wraps with permutation), then B > N (choice) batches
                for i in range(12):
                    b = tr.next_batch(4)
                    for j, a in enumerate(b):
                        out['{}/train/nb4_{}_{}'.format(tag, i, j)] = a
                for i in range(3):
                    b = tr.next_batch(20)
                    for j, a in enumerate(b):
                        out['{}/train/nb20_{}_{}'.format(tag, i, j)] = a
                out[tag + '/train/pairs_pos_end'] = tr.pairs_pos.copy()
                out[tag + '/train/pairs_neg_end'] = tr.pairs_neg.copy()
                # unlabeled stream (cfl/input_data.py:591-619)
                for i in range(8):
                    b = tr.next_unlabeled_batch(5, source_ids=True)
                    out['{}/train/unl5_{}'.format(tag, i)] = b[0]
                    meta['{}/train/unl5_ids_{}'.format(tag, i)] = list(b[1])

        data = ref_in.load_data_sets(root, D, seed=633)
        with contextlib.redirect_stderr(io.StringIO()):
            for split in ('val', 'test'):
                ds = getattr(data, split)
                for i, b in enumerate(ds.whole_pos_batches(3)):
                    out['{}/wpos3_{}_0'.format(split, i)] = b[0]
                    out['{}/wpos3_{}_1'.format(split, i)] = b[1]
                meta[split + '/wpos3_n'] = i + 1
                for i, b in enumerate(ds.whole_neg_batches(3, source_ids=True)):
                    out['{}/wneg3_{}_0'.format(split, i)] = b[0]
                    out['{}/wneg3_{}_1'.format(split, i)] = b[1]
                    meta['{}/wneg3_ids_{}'.format(split, i)] = list(b[2])
                meta[split + '/wneg3_n'] = i + 1

        # directed source/target streams (cfl/input_data.py:621-690)
        dd = ref_in.SemiDataSet(os.path.join(root, 'train'), input_size=D,
                                directed=True, seed=633)
        out['directed/source_indices0'] = dd.source_indices.copy()
        out['directed/target_indices0'] = dd.target_indices.copy()
        for i in range(5):
            out['directed/src5_{}'.format(i)] = dd.next_source_batch(5)[0]
            out['directed/dst5_{}'.format(i)] = dd.next_target_batch(5)[0]
        out['directed/src30'] = dd.next_source_batch(30)[0]   # B > N: choice

        # readers
        out['readers/by_pos'] = ref_in.load_features_by_positions(
            os.path.join(root, 'test', 'features.b'), [3, 0, 11, 3], D)
        meta['readers/asins_by_pos'] = ref_in.load_asins_by_positions(
            os.path.join(root, 'test', 'features.b'), [3, 0, 11, 3], D)
        meta['readers/indices'] = ref_in.load_features_indices(
            os.path.join(root, 'test', 'features.b'), D)

        # G5 evaluate_total on synthetic predict files
        ev = {}
        ev['acc_by_th'] = ref_ev.evaluate_accuracy_by_th(
            [1, 1, 0, 0], [.5, -.1, .2, -3])
        pred_root = os.path.join(tmp, 'pred')
        prng = np.random.RandomState(99)
        pred_inputs = {}
        for m in range(3):
            pdir = os.path.join(pred_root, 'm%d' % m)
            os.makedirs(pdir)
            for split, fname in (('train', 'predict_train.txt'),
                                 ('val', 'predict_val.txt'),
                                 ('test', 'predict.txt'),
                                 ('train', 'predict_train_acc.txt'),
                                 ('val', 'predict_val_acc.txt'),
                                 ('test', 'predict_acc.txt')):
                if m == 2 and fname == 'predict_train_acc.txt':
                    continue        # exercises the "no train file" -1 branch
                ids = meta[split + '/ids']
                lines = []
                for arr, label in ((out[split + '/pos'], 1), (out[split + '/neg'], 0)):
                    for a, b in arr:
                        s = np.float32(prng.randn() + (0.8 if label else -0.8) * (m + 1) / 2)
                        lines.append('{} match {} {}\n'.format(ids[a], ids[b], s))
                with open(os.path.join(pdir, fname), 'w') as f:
                    f.writelines(lines)
                pred_inputs['m{}/{}'.format(m, fname)] = lines
        ev['predict_files'] = pred_inputs
        results = []
        data_pairs = ref_ev.load_data_pairs(root)
        for m in range(3):
            r = ref_ev.evaluate_data_set(data_pairs, os.path.join(pred_root, 'm%d' % m), False)
            results.append(r)
        ev['results'] = [{s: {k: r[s][k] for k in ('accuracy', 'error', 'auc')}
                          for s in ('train', 'val', 'test')} for r in results]
        results_auc = [ref_ev.evaluate_data_set(
            data_pairs, os.path.join(pred_root, 'm%d' % m), True) for m in range(3)]
        ev['results_auc_model'] = [{s: {k: r[s][k] for k in ('accuracy', 'error', 'auc')}
                                    for s in ('train', 'val', 'test')} for r in results_auc]
        best_acc = ref_ev.select_best_result(results, False)
        best_auc = ref_ev.select_best_result(results, True)
        ev['best_acc_index'] = results.index(best_acc)
        ev['best_auc_index'] = results.index(best_auc)
        avg = ref_ev.average_result(results)
        ev['avg'] = {s: {k: float(v) for k, v in avg[s].items()} for s in avg}
        lines = {}
        for name, (res, is_avg) in {'best': (best_acc, False), 'avg': (avg, True)}.items():
            buf = io.StringIO()
            with contextlib.redirect_stdout(buf):
                ref_ev.print_result(res, 'toy', is_avg)
            lines[name] = buf.getvalue()
        ev['print'] = lines
        # the whole CLI function
        for flags in ((False, False), (True, False), (False, True)):
            buf = io.StringIO()
            with contextlib.redirect_stdout(buf):
                ref_ev.evaluate([root], [os.path.join(pred_root, 'm%d' % m) for m in range(3)],
                                select_auc=flags[0], name='cli', avg=flags[1],
                                auc_model=False, only_larger=None)
            ev['cli_auc{}_avg{}'.format(int(flags[0]), int(flags[1]))] = buf.getvalue()
        only = ref_ev.load_data_pairs(root, only_larger=1)
        ev['only_larger_counts'] = {s: [len(only[s]['pos_pairs']), len(only[s]['neg_pairs'])]
                                    for s in only}
    finally:
        shutil.rmtree(tmp)

    np.savez_compressed(os.path.join(HERE, 'data_goldens.npz'), **out)
    with open(os.path.join(HERE, 'data_goldens_meta.json'), 'w') as f:
        json.dump(meta, f, indent=0, sort_keys=True)
    with open(os.path.join(HERE, 'eval_goldens.json'), 'w') as f:
        json.dump(ev, f, indent=0, sort_keys=True)
    print('wrote', len(out), 'arrays')


if __name__ == '__main__':
    main()
