"""Host data path vs vectors captured by importing the reference's
cfl/input_data.py (tests/golden/make_data_goldens.py).  Bit-exact: integer pair
indexing, RandomState call order, float32 feature bytes."""
import json
import os

import numpy as np
import pytest

from cfl import input_data as D

HERE = os.path.dirname(os.path.abspath(__file__))
G = np.load(os.path.join(HERE, 'golden', 'data_goldens.npz'))
META = json.load(open(os.path.join(HERE, 'golden', 'data_goldens_meta.json')))
DIM = 5


@pytest.fixture(scope='module')
def root(tmp_path_factory):
    """Rebuild the toy dataset from the golden *inputs* (ids, features, pairs)."""
    root = tmp_path_factory.mktemp('toy')
    for split in ('train', 'val', 'test'):
        d = root / split
        d.mkdir()
        ids = META[split + '/ids']
        feats = G[split + '/feats']
        with open(d / 'features.b', 'wb') as f:
            for i, a in enumerate(ids):
                f.write(a.encode('ascii'))
                D.dump_array(f, feats[i])
        # writer parity: our dump_array reproduces the reference's bytes
        assert np.array_equal(np.frombuffer((d / 'features.b').read_bytes(), np.uint8),
                              G[split + '/features_b'])
        with open(d / 'pairs_pos.txt', 'w') as f:
            for a, b in G[split + '/pos']:
                f.write('{} match {}\n'.format(ids[a], ids[b]))
        with open(d / 'pairs_neg.txt', 'w') as f:
            for a, b in G[split + '/neg']:
                f.write('{} also_viewed {}\n'.format(ids[a], ids[b]))
        with open(d / 'source.txt', 'w') as f:
            f.writelines(ids[i] + '\n' for i in range(0, len(ids), 2))
        with open(d / 'target.txt', 'w') as f:
            f.writelines(ids[i] + '\n' for i in range(1, len(ids), 2))
    return str(root)


def test_readers(root):
    p = os.path.join(root, 'test', 'features.b')
    assert np.array_equal(D.load_features_by_positions(p, [3, 0, 11, 3], DIM), G['readers/by_pos'])
    assert D.load_asins_by_positions(p, [3, 0, 11, 3], DIM) == META['readers/asins_by_pos']
    assert D.load_features_indices(p, DIM) == META['readers/indices']
    got = list(D.load_features(p, DIM))
    assert [a for a, _ in got] == META['test/ids']
    assert np.array_equal(np.array([v for _, v in got]), G['test/feats'])


@pytest.mark.parametrize('seed', [0, 633])
@pytest.mark.parametrize('switch', [False, True])
def test_labeled_stream_bit_exact(root, seed, switch):
    tag = 'seed{}_sw{}'.format(seed, int(switch))
    data = D.load_data_sets(root, DIM, data_switch=switch, seed=seed)
    tr = data.train
    assert np.array_equal(tr.pairs_pos, G[tag + '/train/pairs_pos0'])
    assert np.array_equal(tr.pairs_neg, G[tag + '/train/pairs_neg0'])
    assert tr.pairs_pos.dtype == np.int64
    assert tr.num_examples == META[tag + '/train/num_examples']       # the N-1 quirk
    assert tr.asins_to_index == META[tag + '/train/asins_to_index']
    for i in range(12):                      # epoch wraps -> permutation
        b = tr.next_batch(4)
        for j in range(4):
            assert np.array_equal(b[j], G['{}/train/nb4_{}_{}'.format(tag, i, j)]), (i, j)
    for i in range(3):                       # B > N -> choice with replacement
        b = tr.next_batch(20)
        for j in range(4):
            assert np.array_equal(b[j], G['{}/train/nb20_{}_{}'.format(tag, i, j)]), (i, j)
    assert np.array_equal(tr.pairs_pos, G[tag + '/train/pairs_pos_end'])
    assert np.array_equal(tr.pairs_neg, G[tag + '/train/pairs_neg_end'])
    for i in range(8):
        b = tr.next_unlabeled_batch(5, source_ids=True)
        assert np.array_equal(b[0], G['{}/train/unl5_{}'.format(tag, i)])
        assert list(b[1]) == META['{}/train/unl5_ids_{}'.format(tag, i)]


def test_next_batch_indices_matches_next_batch(root):
    a = D.load_data_sets(root, DIM, data_switch=True, seed=7).train
    b = D.load_data_sets(root, DIM, data_switch=True, seed=7).train
    feats = G['train/feats']
    for _ in range(9):
        pos, neg, sw = a.next_batch_indices(6)
        ref = b.next_batch(6)
        c = (1, 0) if sw else (0, 1)
        got = (feats[pos[:, c[0]]], feats[pos[:, c[1]]], feats[neg[:, c[0]]], feats[neg[:, c[1]]])
        for x, y in zip(got, ref):
            assert np.array_equal(x, y)


def test_whole_batches_ragged_tail(root):
    data = D.load_data_sets(root, DIM, seed=633)
    for split in ('val', 'test'):
        ds = getattr(data, split)
        got = list(ds.whole_pos_batches(3))
        assert len(got) == META[split + '/wpos3_n']
        for i, b in enumerate(got):
            assert np.array_equal(b[0], G['{}/wpos3_{}_0'.format(split, i)])
            assert np.array_equal(b[1], G['{}/wpos3_{}_1'.format(split, i)])
        got = list(ds.whole_neg_batches(3, source_ids=True))
        assert len(got) == META[split + '/wneg3_n']
        for i, b in enumerate(got):
            assert np.array_equal(b[0], G['{}/wneg3_{}_0'.format(split, i)])
            assert np.array_equal(b[1], G['{}/wneg3_{}_1'.format(split, i)])
            assert list(b[2]) == META['{}/wneg3_ids_{}'.format(split, i)]


def test_directed_streams(root):
    dd = D.SemiDataSet(os.path.join(root, 'train'), input_size=DIM, directed=True, seed=633)
    assert np.array_equal(dd.source_indices, G['directed/source_indices0'])
    assert np.array_equal(dd.target_indices, G['directed/target_indices0'])
    for i in range(5):
        assert np.array_equal(dd.next_source_batch(5)[0], G['directed/src5_{}'.format(i)])
        assert np.array_equal(dd.next_target_batch(5)[0], G['directed/dst5_{}'.format(i)])
    assert np.array_equal(dd.next_source_batch(30)[0], G['directed/src30'])


def test_meta_lines(tmp_path):
    p = tmp_path / 'meta.txt'
    p.write_text('A000000001 x\n cat1\n cat2\nA000000002 y\nA000000003 z\n cat3\n')
    got = list(D.load_meta_lines(str(p)))
    assert [g[0] for g in got] == ['A000000001', 'A000000002', 'A000000003']
    assert [len(g[1]) for g in got] == [3, 1, 2]


def test_bad_feature_file(tmp_path):
    (tmp_path / 'features.b').write_bytes(b'x' * 31)
    with pytest.raises(ValueError):
        D.FeatureFile(str(tmp_path / 'features.b'), DIM)


def test_prefetched_reshuffle_is_the_same_stream(tmp_path):
    """SemiDataSet.prefetch_reshuffle computes the next epoch wrap's `pairs[rng.permutation(n)]` ahead of time on
    a copy of the generator (worker thread) and adopts it only if nobody drew from the generator in between: the
    index stream, the shuffled lists and the generator state stay bit-identical to the plain stream -- with equal
    and unequal list lengths (both lists wrapping in the same batch), data_switch coin flips and foreign draws."""
    from cfl import input_data
    from cfl.synthetic import make_dataset
    for n_pos, n_neg in ((900, 700), (640, 640)):
        root = str(tmp_path / ('toy%d' % n_neg))
        make_dataset(root, D=64, n_items=300, n_pos=n_pos, n_neg=n_neg, k=2, latent=6, seed=1)
        path = os.path.join(root, 'train')
        for sw in (False, True):
            a = input_data.SemiDataSet(path, input_size=64, data_switch=sw, seed=9)
            b = input_data.SemiDataSet(path, input_size=64, data_switch=sw, seed=9)
            for it in range(200):
                B = [64, 33, 100][it % 3]
                pa = a.next_batch_indices(B)
                a.prefetch_reshuffle(B)
                pb = b.next_batch_indices(B)
                assert np.array_equal(pa[0], pb[0]) and np.array_equal(pa[1], pb[1]) and pa[2] == pb[2], (sw, it)
                if it % 7 == 0:                                  # a foreign draw invalidates the prefetch
                    assert a._rng.rand() == b._rng.rand()
            assert np.array_equal(a.pairs_pos, b.pairs_pos) and np.array_equal(a.pairs_neg, b.pairs_neg)
            assert a._rng.rand() == b._rng.rand()


def test_library_reshuffle_equals_numpy_legacy_stream():
    """cfl_mt19937_reshuffle (host-only C: MT19937 + Fisher-Yates with masked-rejection intervals) against
    numpy.random.RandomState.permutation, bit for bit, from mid-stream generator states; the generator state after the
    call equals numpy's (the next draw agrees)."""
    import __graft_entry__ as g
    g.build()
    from cfl import hipabi as H
    from numpy.random import RandomState
    for seed, n in ((0, 10), (633, 200000), (9, 1), (77, 65537), (11, 2), (5, 1000)):
        a, b = RandomState(seed), RandomState(seed)
        a.rand(seed % 7)
        b.rand(seed % 7)
        rows = np.arange(2 * n, dtype=np.int64).reshape(n, 2) * 3 + 1
        ref = rows[a.permutation(n)]
        out, st = H.mt19937_reshuffle(b.get_state(), rows)
        b.set_state(st)
        assert np.array_equal(out, ref), (seed, n)
        assert a.randint(0, 10 ** 9) == b.randint(0, 10 ** 9), (seed, n)
