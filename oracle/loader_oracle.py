"""TEST / BENCH INFRASTRUCTURE -- a CPU restatement of the reference's batch ASSEMBLY (never imported by the product path).

The reference builds every labeled batch by reading each of its 4 * B vectors with one seek + one read on `features.b`
(`load_features_by_positions`, cfl/input_data.py:212-228 of the reference, called four times per batch by
`SemiDataSet.next_labeled_batch`, cfl/input_data.py:570-573).  Record p of a vector file starts at byte (4 D + 10) p: a
10-byte ASCII id, then D little-endian float32 (`dump_array`, cfl/input_data.py:23-31).

Used by bench.py's `cpu_baseline.loader` leg ("optionally also time the faithful per-row-seek data loader")
and pinned against the product's reader (which is itself pinned to goldens captured from the reference's own module,
tests/test_input_data.py) by tests/test_oracle.py.
"""
from array import array

import numpy as np

ID_BYTES = 10


def write_features(path, rng, n_items, D):
    """a synthetic vector `features.b`: ids "%010d", |N(0, 1)| features (cfl/input_data.py:23-31: id, then array('f'))"""
    with open(path, 'wb') as f:
        for i in range(n_items):
            f.write(('%010d' % i).encode('ascii'))
            array('f', np.abs(rng.randn(D)).astype(np.float32).tolist()).tofile(f)


def load_features_by_positions(path, positions, D):
    """cfl/input_data.py:212-228: one seek and one fromfile per position"""
    features = []
    with open(path, 'rb') as infile:
        for pos in positions:
            infile.seek(ID_BYTES + (D * 4 + ID_BYTES) * int(pos))
            feature = array('f')
            feature.fromfile(infile, D)
            features.append(feature)
    return np.array(features)


def labeled_batch_by_seek(path, positions_pos, positions_neg, D):
    """the four reads of SemiDataSet.next_labeled_batch (cfl/input_data.py:570-573): src / dst of the positive pairs, then of the
    negative pairs"""
    return (load_features_by_positions(path, positions_pos[:, 0], D), load_features_by_positions(path, positions_pos[:, 1], D),
            load_features_by_positions(path, positions_neg[:, 0], D), load_features_by_positions(path, positions_neg[:, 1], D))
