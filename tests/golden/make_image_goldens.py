"""Golden vectors for the image / image+latent ("double") dataset streams, captured by
IMPORTING the reference (build container only):

    python tests/golden/make_image_goldens.py

The reference decodes images with ``scipy.misc.imread``, which does not exist in this image,
so only the INTEGER side is captured: record offsets, id look-ups, the seeded permutations and
the byte-offset "positions" every batch call would load.  For that the instance method
``_load_features_by_positions`` of the reference's SemiDataSet objects is replaced by one that
returns the positions it was asked for (for double data: twice, as (image, latent)).  Everything
else -- offset scanning, RandomState call order, epoch wraps, directed source/target streams --
is the reference's own code.  The toy record files (inputs) are stored in the .npz as bytes.
"""
import io
import json
import os
import shutil
import struct
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_data_goldens import REF, _stub_modules  # noqa: E402


def png_bytes(arr):
    from PIL import Image
    buf = io.BytesIO()
    Image.fromarray(arr).save(buf, format='png')
    return buf.getvalue()


def write_split(path, ids, images, latents, double, raw_latent, pos, neg):
    os.makedirs(path, exist_ok=True)
    with open(os.path.join(path, 'features.b'), 'wb') as f:
        for i, asin in enumerate(ids):
            img = png_bytes(images[i])
            f.write(asin.encode('ascii'))
            if double:
                if raw_latent:
                    lat = latents[i].astype('<f4').tobytes()
                else:
                    b = io.BytesIO()
                    np.savez(b, data=latents[i])
                    lat = b.getvalue()
                f.write(struct.pack('<ii', len(img), len(lat)))
                f.write(img)
                f.write(lat)
            else:
                f.write(struct.pack('<i', len(img)))
                f.write(img)
    with open(os.path.join(path, 'pairs_pos.txt'), 'w') as f:
        for a, b in pos:
            f.write('{} match {}\n'.format(ids[a], ids[b]))
    with open(os.path.join(path, 'pairs_neg.txt'), 'w') as f:
        for a, b in neg:
            f.write('{} match {}\n'.format(ids[a], ids[b]))
    with open(os.path.join(path, 'source.txt'), 'w') as f:
        f.writelines(ids[i] + '\n' for i in range(0, len(ids), 2))
    with open(os.path.join(path, 'target.txt'), 'w') as f:
        f.writelines(ids[i] + '\n' for i in range(1, len(ids), 2))


def main():
    _stub_modules()
    sys.path.insert(0, REF)
    import cfl.input_data as ref_in

    out, meta = {}, {}
    tmp = tempfile.mkdtemp()
    try:
        rng = np.random.RandomState(20261003)
        n, H, W, C, Dl = 13, 6, 5, 3, 7
        ids = ['%010d' % (i * 11 + 5) for i in range(n)]
        images = rng.randint(0, 256, size=(n, H, W, C)).astype(np.uint8)
        latents = rng.randn(n, Dl).astype(np.float32)
        pos = rng.randint(0, n, size=(9, 2))
        neg = rng.randint(0, n, size=(7, 2))
        out['images'], out['latents'], out['pos'], out['neg'] = images, latents, pos, neg
        meta['ids'] = ids
        for kind, double, raw in (('double_raw', True, True), ('double_npz', True, False), ('image', False, False)):
            path = os.path.join(tmp, kind)
            write_split(path, ids, images, latents, double, raw, pos, neg)
            fpath = os.path.join(path, 'features.b')
            with open(fpath, 'rb') as f:
                out[kind + '/features_b'] = np.frombuffer(f.read(), np.uint8)
            offs = (ref_in.load_double_offsets if double else ref_in.load_images_offsets)(fpath)
            meta[kind + '/offsets'] = {k: int(v) for k, v in offs.items()}
            some = [offs[ids[i]] for i in (4, 0, 12, 4)]
            meta[kind + '/asins_by_offsets'] = ref_in.load_asins_by_offsets(fpath, some)
            for directed in (False, True):
                tag = '{}/dir{}'.format(kind, int(directed))
                ds = ref_in.SemiDataSet(path, input_size=H * W * C, is_image=True, is_double=double,
                                        directed=directed, data_switch=True, raw_latent=raw, seed=633)
                if double:
                    ds._load_features_by_positions = lambda p: (np.array(p), np.array(p))
                else:
                    ds._load_features_by_positions = lambda p: np.array(p)
                meta[tag + '/num_examples'] = int(ds.num_examples)
                out[tag + '/item_indices0'] = ds.item_indices.copy()
                out[tag + '/pairs_pos0'] = ds.pairs_pos.copy()
                if directed:
                    out[tag + '/source_indices0'] = ds.source_indices.copy()
                    out[tag + '/target_indices0'] = ds.target_indices.copy()
                for i in range(6):
                    b = ds.next_batch(4)
                    meta[tag + '/nb_len'] = len(b)
                    for j, a in enumerate(b):
                        out['{}/nb4_{}_{}'.format(tag, i, j)] = np.asarray(a)
                for i in range(5):
                    b = ds.next_unlabeled_batch(5)
                    meta[tag + '/unl_len'] = len(b)
                    out['{}/unl5_{}'.format(tag, i)] = np.asarray(b[0])
                for i in range(4):
                    out['{}/src4_{}'.format(tag, i)] = np.asarray(ds.next_source_batch(4)[0])
                    out['{}/dst4_{}'.format(tag, i)] = np.asarray(ds.next_target_batch(4)[0])
        np.savez_compressed(os.path.join(HERE, 'image_goldens.npz'), **out)
        with open(os.path.join(HERE, 'image_goldens_meta.json'), 'w') as f:
            json.dump(meta, f, indent=0, sort_keys=True)
        print('wrote', len(out), 'arrays')
    finally:
        shutil.rmtree(tmp)


if __name__ == '__main__':
    main()
