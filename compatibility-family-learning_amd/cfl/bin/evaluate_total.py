"""python -m cfl.bin.evaluate_total -- accuracy@0 / AUC report over predict files.

Drop-in for the reference's cfl/bin/evaluate_total.py:16-202 (same flags, same
function names, same one-line TSV output); pure NumPy + sklearn, no GPU.  Pinned by
tests/golden/eval_goldens.json (captured by importing the reference module).
"""
import argparse
import os
from collections import Counter

import numpy as np
from sklearn.metrics import roc_auc_score

SPLITS = ('train', 'val', 'test')


def load_pairs(path):
    """[(id1, id2, value)] from '<id1> <rel> <id2> [value]' lines."""
    out = []
    with open(path) as infile:
        for line in infile:
            t = line.split()
            out.append((t[0], t[2], float(t[3]) if len(t) >= 4 else 0.0))
    return out


def load_data_pairs(path, only_larger=None):
    data_pairs = {}
    for split in SPLITS:
        pos = load_pairs(os.path.join(path, split, 'pairs_pos.txt'))
        neg = load_pairs(os.path.join(path, split, 'pairs_neg.txt'))
        if only_larger:
            # keep pairs whose source item has more than N positive pairs
            counts = Counter(a for a, _, _ in pos)
            pos = [p for p in pos if counts[p[0]] > only_larger]
            neg = [p for p in neg if counts[p[0]] > only_larger]
        data_pairs[split] = {'pos_pairs': pos, 'neg_pairs': neg}
    return data_pairs


def evaluate_accuracy_by_th(y_true, y_score, th=0.0):
    """(accuracy, error) with 'score > th' predicting a positive."""
    y_true = np.asarray(y_true)
    y_score = np.asarray(y_score, dtype=np.float64)
    correct = int(np.count_nonzero((y_true > 0) == (y_score > th)))
    n = y_true.shape[0]
    return correct / n, (n - correct) / n


def evaluate_accuracy(pos_pairs, neg_pairs, pred_pairs):
    y_true = [1] * len(pos_pairs) + [0] * len(neg_pairs)
    y_score = [pred_pairs[(x, y)] for x, y, _ in pos_pairs] + [pred_pairs[(x, y)] for x, y, _ in neg_pairs]
    accuracy, error = evaluate_accuracy_by_th(y_true, y_score)
    return {'accuracy': accuracy, 'error': error, 'auc': roc_auc_score(y_true, y_score),
            'y_true': y_true, 'y_score': y_score}


def evaluate_data_set(data_pairs, predict_path, auc_model):
    suffix = '' if auc_model else '_acc'
    files = {'train': 'predict_train%s.txt' % suffix, 'val': 'predict_val%s.txt' % suffix,
             'test': 'predict%s.txt' % suffix}
    results = {}
    for split in SPLITS:
        path = os.path.join(predict_path, files[split])
        if split == 'train' and not os.path.exists(path):
            results[split] = {'accuracy': -1., 'error': -1., 'auc': -1.}
            continue
        pred = {(x, y): v for x, y, v in load_pairs(path)}
        results[split] = evaluate_accuracy(data_pairs[split]['pos_pairs'],
                                           data_pairs[split]['neg_pairs'], pred)
    return results


def select_best_result(results, select_auc):
    """First result with the strictly largest validation AUC / accuracy."""
    key = 'auc' if select_auc else 'accuracy'
    best = None
    for r in results:
        if best is None or r['val'][key] > best['val'][key]:
            best = r
    return best


def average_result(results):
    avg = {}
    for split in SPLITS:
        err = [r[split]['error'] for r in results]
        auc = [r[split]['auc'] for r in results]
        avg[split] = {'error': np.mean(err), 'auc': np.mean(auc),
                      'error_std': np.std(err), 'auc_std': np.std(auc)}
    return avg


def print_result(result, name, avg):
    cells = []
    for key in ('error', 'auc'):
        for split in SPLITS:
            if avg:
                cells.append('{:.2%}+-{:.2%}'.format(result[split][key], result[split][key + '_std']))
            else:
                cells.append('{:.2%}'.format(result[split][key]))
    print('\t'.join(cells + [name]))


def evaluate(data_path, predict_paths, select_auc, name, avg, auc_model, only_larger):
    if len(data_path) == 1:
        data_pairs = load_data_pairs(data_path[0], only_larger)
        results = [evaluate_data_set(data_pairs, p, auc_model) for p in predict_paths]
    else:
        assert len(data_path) == len(predict_paths)
        results = [evaluate_data_set(load_data_pairs(d, only_larger), p, auc_model)
                   for d, p in zip(data_path, predict_paths)]
    print_result(average_result(results) if avg else select_best_result(results, select_auc), name, avg)


def main():
    parser = argparse.ArgumentParser()
    parser.add_argument('--data-path', nargs='+', required=True)
    parser.add_argument('--predict-paths', nargs='+', required=True)
    parser.add_argument('--select-auc', action='store_true')
    parser.add_argument('--avg', action='store_true')
    parser.add_argument('--auc-model', action='store_true')
    parser.add_argument('--only-larger', type=int)
    parser.add_argument('--name', default='model')
    evaluate(**vars(parser.parse_args()))


if __name__ == '__main__':
    main()
