"""Synthetic datasets in the reference's on-disk format.

There is no network for the Amazon / Polyvore data, so tests, the CLI smoke runs and
the benchmark use a generator that writes exactly what the Monomer split tool emits
(experiments/monomer/monomer.patch:71-150): ``<split>/features.b`` (10-byte id +
D little-endian f32), ``pairs_pos.txt`` / ``pairs_neg.txt``.  Positives are planted by
a hidden K-prototype teacher (dst = proto_k(src) + noise) so AUC is non-trivial;
negatives are uniform random pairs."""
import os

import numpy as np

from .input_data import dump_array


def make_split(path, n_items, D, n_pos, n_neg, rng, teacher, scale, noise=0.15, relation='also_viewed'):
    os.makedirs(path, exist_ok=True)
    k_protos, latent = teacher
    half = n_items // 2
    z = rng.randn(half, latent.shape[0]).astype(np.float32)
    src = np.abs(z @ latent + 0.3 * rng.randn(half, D).astype(np.float32))
    which = rng.randint(0, len(k_protos), size=half)
    dst = np.empty_like(src)
    for k, P in enumerate(k_protos):
        m = which == k
        dst[m] = np.abs((z[m] @ P) @ latent + noise * rng.randn(int(m.sum()), D).astype(np.float32))
    feats = (np.concatenate([src, dst]) * scale).astype(np.float32)
    ids = ['%010d' % i for i in range(feats.shape[0])]
    with open(os.path.join(path, 'features.b'), 'wb') as f:
        for i, a in enumerate(ids):
            f.write(a.encode('ascii'))
            dump_array(f, feats[i])
    s = rng.randint(0, half, size=n_pos)
    pos = np.stack([s, s + half], 1)
    neg = np.stack([rng.randint(0, half, size=n_neg), half + rng.randint(0, half, size=n_neg)], 1)
    neg = neg[neg[:, 0] + half != neg[:, 1]]
    for name, pairs in (('pairs_pos.txt', pos), ('pairs_neg.txt', neg)):
        with open(os.path.join(path, name), 'w') as f:
            for a, b in pairs:
                f.write('{} {} {}\n'.format(ids[a], relation, ids[b]))
    # --directed / --data-directed read the two item roles (cfl/input_data.py:403-428)
    with open(os.path.join(path, 'source.txt'), 'w') as f:
        f.writelines(a + '\n' for a in ids[:half])
    with open(os.path.join(path, 'target.txt'), 'w') as f:
        f.writelines(a + '\n' for a in ids[half:])
    return feats, pos, neg


def make_dataset(root, D=4096, n_items=2000, n_pos=4000, n_neg=4000, k=3, latent=16, seed=633,
                 scale=58.388599 / 4.5, splits=(('train', 1.0), ('val', 0.25), ('test', 0.25))):
    """Write <root>/{train,val,test}; returns root."""
    rng = np.random.RandomState(seed)
    lat = (rng.randn(latent, D) / np.sqrt(latent)).astype(np.float32)
    protos = [np.linalg.qr(rng.randn(latent, latent))[0].astype(np.float32) for _ in range(k)]
    for name, frac in splits:
        make_split(os.path.join(root, name), max(8, int(n_items * frac)) // 2 * 2, D,
                   max(4, int(n_pos * frac)), max(4, int(n_neg * frac)), rng, (protos, lat), scale)
    return root


def make_double_dataset(root, image_shape=(16, 16, 3), latent_dim=64, n_items=120, n_pos=160, n_neg=160, k=2,
                        seed=633, latent_scale=31.9098 / 4.0, raw_latent=True,
                        splits=(('train', 1.0), ('val', 0.5), ('test', 0.5)), double=True):
    """Image + latent ("double") dataset in the reference's record format (cfl/input_data.py:107-170):
    id[10] | int32 size1 | int32 size2 | png | latent (double=False: image-only records id | int32 size | png,
    cfl/input_data.py:34-84), plus pairs_pos/neg.txt and source.txt / target.txt
    (for --data-directed).  Every item's image is a smooth colour pattern decoded from its latent, so the
    generator has something to learn; positives are planted by a K-prototype teacher on the latents."""
    import struct
    from io import BytesIO
    from .input_data import dump_image
    rng = np.random.RandomState(seed)
    H, W, C = image_shape
    hid = 8
    lat_basis = (rng.randn(hid, latent_dim) / np.sqrt(hid)).astype(np.float32)
    img_basis = rng.randn(hid, 4, 4, C).astype(np.float32)
    protos = [np.linalg.qr(rng.randn(hid, hid))[0].astype(np.float32) for _ in range(k)]

    def render(z):
        low = np.tensordot(z, img_basis, axes=(1, 0))                 # [n, 4, 4, C]
        up = low.repeat(H // 4, axis=1).repeat(W // 4, axis=2)
        return (255.0 / (1.0 + np.exp(-up))).astype(np.uint8)

    for name, frac in splits:
        path = os.path.join(root, name)
        os.makedirs(path, exist_ok=True)
        half = max(8, int(n_items * frac)) // 2
        z = rng.randn(half, hid).astype(np.float32)
        which = rng.randint(0, k, size=half)
        zd = np.stack([z[i] @ protos[which[i]] for i in range(half)]) + 0.1 * rng.randn(half, hid).astype(np.float32)
        zs = np.concatenate([z, zd]).astype(np.float32)
        latents = (np.abs(zs @ lat_basis) * latent_scale).astype(np.float32)
        images = render(zs)
        ids = ['%010d' % i for i in range(2 * half)]
        with open(os.path.join(path, 'features.b'), 'wb') as f:
            for i, a in enumerate(ids):
                img = BytesIO()
                dump_image(img, images[i])
                img = img.getvalue()[4:]                              # dump_image prefixes its own size
                if raw_latent:
                    lat = latents[i].astype('<f4').tobytes()
                else:
                    b = BytesIO()
                    np.savez(b, data=latents[i])
                    lat = b.getvalue()
                f.write(a.encode('ascii'))
                if double:
                    f.write(struct.pack('<ii', len(img), len(lat)))
                    f.write(img)
                    f.write(lat)
                else:
                    f.write(struct.pack('<i', len(img)))
                    f.write(img)
        npos, nneg = max(4, int(n_pos * frac)), max(4, int(n_neg * frac))
        s = rng.randint(0, half, size=npos)
        pos = np.stack([s, s + half], 1)
        neg = np.stack([rng.randint(0, half, size=nneg), half + rng.randint(0, half, size=nneg)], 1)
        neg = neg[neg[:, 0] + half != neg[:, 1]]
        for fname, pairs in (('pairs_pos.txt', pos), ('pairs_neg.txt', neg)):
            with open(os.path.join(path, fname), 'w') as f:
                for a, b in pairs:
                    f.write('{} match {}\n'.format(ids[a], ids[b]))
        with open(os.path.join(path, 'source.txt'), 'w') as f:
            f.writelines(a + '\n' for a in ids[:half])
        with open(os.path.join(path, 'target.txt'), 'w') as f:
            f.writelines(a + '\n' for a in ids[half:])
    return root
