"""cfl.bin.evaluate_total vs outputs captured by importing the reference module
(tests/golden/eval_goldens.json)."""
import contextlib
import io
import json
import os

import numpy as np
import pytest

from cfl.bin import evaluate_total as E

HERE = os.path.dirname(os.path.abspath(__file__))
EV = json.load(open(os.path.join(HERE, 'golden', 'eval_goldens.json')))
G = np.load(os.path.join(HERE, 'golden', 'data_goldens.npz'))
META = json.load(open(os.path.join(HERE, 'golden', 'data_goldens_meta.json')))


@pytest.fixture(scope='module')
def dirs(tmp_path_factory):
    root = tmp_path_factory.mktemp('evd')
    data = root / 'toy'
    for split in ('train', 'val', 'test'):
        d = data / split
        d.mkdir(parents=True)
        ids = META[split + '/ids']
        for name, key, rel in (('pairs_pos.txt', '/pos', 'match'), ('pairs_neg.txt', '/neg', 'also_viewed')):
            with open(d / name, 'w') as f:
                for a, b in G[split + key]:
                    f.write('{} {} {}\n'.format(ids[a], rel, ids[b]))
    pred = root / 'pred'
    for key, lines in EV['predict_files'].items():
        m, fname = key.split('/')
        (pred / m).mkdir(parents=True, exist_ok=True)
        (pred / m / fname).write_text(''.join(lines))
    return str(data), [str(pred / ('m%d' % i)) for i in range(3)]


def test_accuracy_by_threshold_kat():
    assert list(E.evaluate_accuracy_by_th([1, 1, 0, 0], [.5, -.1, .2, -3])) == EV['acc_by_th'] == [0.5, 0.5]
    assert E.evaluate_accuracy_by_th([1, 0], [0.0, 0.0]) == (0.5, 0.5)      # score == th is "negative"


@pytest.mark.parametrize('auc_model', [False, True])
def test_evaluate_data_set(dirs, auc_model):
    data, preds = dirs
    pairs = E.load_data_pairs(data)
    want = EV['results_auc_model' if auc_model else 'results']
    for p, w in zip(preds, want):
        got = E.evaluate_data_set(pairs, p, auc_model)
        for split in ('train', 'val', 'test'):
            for k in ('accuracy', 'error', 'auc'):
                assert got[split][k] == pytest.approx(w[split][k], abs=1e-12), (split, k)


def test_selection_average_and_printing(dirs):
    data, preds = dirs
    pairs = E.load_data_pairs(data)
    results = [E.evaluate_data_set(pairs, p, False) for p in preds]
    assert results.index(E.select_best_result(results, False)) == EV['best_acc_index']
    assert results.index(E.select_best_result(results, True)) == EV['best_auc_index']
    avg = E.average_result(results)
    for split in avg:
        for k, v in EV['avg'][split].items():
            assert float(avg[split][k]) == pytest.approx(v, abs=1e-12)
    for name, res, is_avg in (('best', E.select_best_result(results, False), False), ('avg', avg, True)):
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            E.print_result(res, 'toy', is_avg)
        assert buf.getvalue() == EV['print'][name]


@pytest.mark.parametrize('select_auc,avg', [(False, False), (True, False), (False, True)])
def test_cli_function_output(dirs, select_auc, avg):
    data, preds = dirs
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        E.evaluate([data], preds, select_auc=select_auc, name='cli', avg=avg, auc_model=False, only_larger=None)
    assert buf.getvalue() == EV['cli_auc{}_avg{}'.format(int(select_auc), int(avg))]


def test_only_larger(dirs):
    data, _ = dirs
    only = E.load_data_pairs(data, only_larger=1)
    assert {s: [len(only[s]['pos_pairs']), len(only[s]['neg_pairs'])] for s in only} == EV['only_larger_counts']
