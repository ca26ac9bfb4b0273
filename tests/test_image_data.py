"""Image / image+latent ("double") dataset streams against goldens captured from the reference
(tests/golden/make_image_goldens.py): record offsets, id look-ups, seeded permutations and the
positions of every batch are bit-exact; the decoded pixels / latents are checked against the
arrays the toy files were encoded from (PNG is lossless)."""
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
G = np.load(os.path.join(HERE, 'golden', 'image_goldens.npz'))
META = json.load(open(os.path.join(HERE, 'golden', 'image_goldens_meta.json')))
KINDS = {'double_raw': (True, True), 'double_npz': (True, False), 'image': (False, False)}


@pytest.fixture(scope='module')
def roots(tmp_path_factory):
    root = tmp_path_factory.mktemp('imgdata')
    ids = META['ids']
    for kind in KINDS:
        d = root / kind
        d.mkdir()
        (d / 'features.b').write_bytes(G[kind + '/features_b'].tobytes())
        with open(d / 'pairs_pos.txt', 'w') as f:
            f.writelines('{} match {}\n'.format(ids[a], ids[b]) for a, b in G['pos'])
        with open(d / 'pairs_neg.txt', 'w') as f:
            f.writelines('{} match {}\n'.format(ids[a], ids[b]) for a, b in G['neg'])
        with open(d / 'source.txt', 'w') as f:
            f.writelines(ids[i] + '\n' for i in range(0, len(ids), 2))
        with open(d / 'target.txt', 'w') as f:
            f.writelines(ids[i] + '\n' for i in range(1, len(ids), 2))
    return root


def _expect(kind, positions):
    """What loading `positions` (byte offsets) must return."""
    off2i = {v: META['ids'].index(k) for k, v in META[kind + '/offsets'].items()}
    idx = [off2i[int(p)] for p in positions]
    img = (G['images'][idx].astype(np.float32) / 255.).reshape(len(idx), -1)
    return img, G['latents'][idx]


@pytest.mark.parametrize('kind', list(KINDS))
def test_offsets_and_ids(roots, kind):
    from cfl import input_data as D
    double, _ = KINDS[kind]
    path = str(roots / kind / 'features.b')
    offs = (D.load_double_offsets if double else D.load_images_offsets)(path)
    assert offs == META[kind + '/offsets']
    some = [offs[META['ids'][i]] for i in (4, 0, 12, 4)]
    assert D.load_asins_by_offsets(path, some) == META[kind + '/asins_by_offsets']


@pytest.mark.parametrize('kind', list(KINDS))
@pytest.mark.parametrize('directed', [False, True])
def test_streams_match_reference(roots, kind, directed):
    from cfl import input_data as D
    double, raw = KINDS[kind]
    tag = '{}/dir{}'.format(kind, int(directed))
    ds = D.SemiDataSet(str(roots / kind), input_size=90, is_image=True, is_double=double, directed=directed,
                       data_switch=True, raw_latent=raw, seed=633)
    assert ds.num_examples == META[tag + '/num_examples']
    assert np.array_equal(ds.item_indices, G[tag + '/item_indices0'])
    assert np.array_equal(ds.pairs_pos, G[tag + '/pairs_pos0'])
    if directed:
        assert np.array_equal(ds.source_indices, G[tag + '/source_indices0'])
        assert np.array_equal(ds.target_indices, G[tag + '/target_indices0'])
    per = 2 if double else 1
    for i in range(6):
        b = ds.next_batch(4)
        assert len(b) == META[tag + '/nb_len'] == 4 * per
        for j in range(4):
            img, lat = _expect(kind, G['{}/nb4_{}_{}'.format(tag, i, j * per)])
            assert np.array_equal(b[j * per], img)
            if double:
                assert np.array_equal(b[j * per + 1], lat)
    for i in range(5):
        b = ds.next_unlabeled_batch(5)
        assert len(b) == META[tag + '/unl_len'] == per
        img, lat = _expect(kind, G['{}/unl5_{}'.format(tag, i)])
        assert np.array_equal(b[0], img)
        if double:
            assert np.array_equal(b[1], lat)
    for i in range(4):
        for name, fn in (('src4', ds.next_source_batch), ('dst4', ds.next_target_batch)):
            b = fn(4)
            img, lat = _expect(kind, G['{}/{}_{}'.format(tag, name, i)])
            assert np.array_equal(b[0], img)
            if double:
                assert np.array_equal(b[1], lat)


def test_whole_batches_and_dump_image(roots, tmp_path):
    from cfl import input_data as D
    ds = D.SemiDataSet(str(roots / 'double_raw'), input_size=90, is_image=True, is_double=True, raw_latent=True)
    chunks = list(ds.whole_pos_batches(4, source_ids=True))
    assert [c[0].shape[0] for c in chunks] == [4, 4, 1] and len(chunks[0]) == 5
    assert chunks[0][4] == [META['ids'][a] for a in G['pos'][:4, 0]]
    # dump_image round trip (png is lossless)
    p = tmp_path / 'img.b'
    with open(p, 'wb') as f:
        for i in range(3):
            f.write(META['ids'][i].encode('ascii'))
            D.dump_image(f, G['images'][i])
    got = list(D.load_images(str(p)))
    assert [a for a, _ in got] == META['ids'][:3]
    assert np.array_equal(got[1][1], (G['images'][1].astype(np.float32) / 255.).reshape(-1))


@pytest.mark.parametrize('kind', list(KINDS))
def test_decoded_record_table_equals_the_per_record_readers(roots, kind, monkeypatch):
    """DecodedRecords (records decoded once into a uint8 / latent table, batches gathered from it) must hand out exactly
    what the reference's per-record readers do -- bit for bit, for repeated, permuted and duplicate positions -- and step
    aside (per-record path) when the cap is 0, an offset is unknown or a record is no uint8 image."""
    from cfl import input_data as D
    double, raw = KINDS[kind]
    path = str(roots / kind / 'features.b')
    offs = (D.load_double_offsets if double else D.load_images_offsets)(path)
    ids = META['ids']
    pos = [offs[ids[i]] for i in (4, 0, 12, 4, 7, 7, 1)]
    direct = (D.load_double_images_by_offsets(path, pos, raw_latent=raw) if double else D.load_images_by_offsets(path, pos))
    ds = D.SemiDataSet(str(roots / kind), input_size=90, is_image=True, is_double=double, raw_latent=raw)
    for _ in range(2):          # second round: every record comes from the table
        got = ds._load_features_by_positions(np.array(pos))
        if double:
            assert got[0].dtype == np.float32 and np.array_equal(got[0], direct[0]) and np.array_equal(got[1], direct[1])
            assert got[1].dtype == np.float32
        else:
            assert got.dtype == np.float32 and np.array_equal(got, direct)
    tab = ds._records
    assert tab.usable and int(tab.have.sum()) == 5 and tab.images.dtype == np.uint8
    exp = _expect(kind, pos)
    assert np.array_equal(got[0] if double else got, exp[0])
    # a whole seeded stream through the table equals the stream of a dataset whose table is disabled
    a = D.SemiDataSet(str(roots / kind), input_size=90, is_image=True, is_double=double, raw_latent=raw, seed=7)
    a._load_features_by_positions(np.array(pos[:1]))      # (the table is built at the first load: before the cap is set to 0)
    monkeypatch.setenv('CFL_IMAGE_TABLE_MB', '0')
    b = D.SemiDataSet(str(roots / kind), input_size=90, is_image=True, is_double=double, raw_latent=raw, seed=7)
    for _ in range(6):
        xa, xb = a.next_batch(3), b.next_batch(3)
        assert len(xa) == len(xb) and all(np.array_equal(u, v) and u.dtype == v.dtype for u, v in zip(xa, xb))
        ua, ub = a.next_unlabeled_batch(4), b.next_unlabeled_batch(4)
        assert all(np.array_equal(u, v) for u, v in zip(ua, ub))
    assert b._records is not None and not b._records.usable and a._records.usable
    monkeypatch.delenv('CFL_IMAGE_TABLE_MB')
    # unknown offset -> the table steps aside
    assert tab.rows([pos[0] + 1]) is None


def test_labeled_batches_without_images(roots):
    """labeled_images = False: the image parts of a labeled double batch are zero placeholders of the right shape, the
    latents and the seeded stream are those of the full batch."""
    from cfl import input_data as D
    a = D.SemiDataSet(str(roots / 'double_raw'), input_size=90, is_image=True, is_double=True, raw_latent=True, seed=11)
    b = D.SemiDataSet(str(roots / 'double_raw'), input_size=90, is_image=True, is_double=True, raw_latent=True, seed=11)
    b.labeled_images = False
    for _ in range(5):
        xa, xb = a.next_batch(3), b.next_batch(3)
        assert len(xa) == len(xb) == 8
        for i in range(0, 8, 2):
            assert xb[i].shape == xa[i].shape and xb[i].dtype == np.float32 and not xb[i].any()
            assert np.array_equal(xa[i + 1], xb[i + 1])
        ua, ub = a.next_unlabeled_batch(4), b.next_unlabeled_batch(4)       # unlabeled batches keep their images
        assert all(np.array_equal(u, v) for u, v in zip(ua, ub))
