"""Dataset readers and the pair-batch generator of the cfl hot path.

Mirrors the vector-dataset side of the reference's ``cfl/input_data.py`` (same
function / class / attribute names, same NumPy ``RandomState`` call sequence, so
that batch *indices* are bit-identical for a given seed -- pinned by
tests/golden/data_goldens.npz which was captured by importing the reference):

    dump_array, load_features, load_features_by_positions,
    load_asins_by_positions, load_features_indices, load_meta_lines,
    load_data_sets, SemiDataSet (next_batch / whole_pos_batches / ...)

Differences, all host-side mechanics rather than behaviour:
  * ``features.b`` is memory-mapped once as a structured array instead of one
    ``seek`` + ``fromfile`` per row (cfl/input_data.py:212-228 is the reference's
    real CPU bottleneck);
  * ``SemiDataSet.next_batch_indices`` exposes the index arrays so that a
    device-resident feature table can be gathered on the GPU
    (``ResidentFeatures``, include/cfl_hip.h cfl_gather_rows);
  * image / image+latent ("double") datasets (cfl/input_data.py:34-192, used by the
    MrCGAN phase) decode with PIL instead of the removed ``scipy.misc.imread``; item
    "positions" of such files are byte offsets, exactly as in the reference.
"""
import os
import struct
from argparse import Namespace
from array import array
from collections import defaultdict
from io import BytesIO

import numpy as np

import logging
logger = logging.getLogger(__name__)
from numpy.random import RandomState

ID_BYTES = 10  # every item id is exactly 10 ASCII bytes


def _record_dtype(input_size):
    return np.dtype([('id', 'S%d' % ID_BYTES), ('x', '<f4', (int(input_size),))])


def dump_array(outfile, floats):
    """Append one little-endian float32 vector (cfl/input_data.py:23-31)."""
    np.asarray(floats, dtype='<f4').tofile(outfile)


class FeatureFile(object):
    """Read-only memory map of a vector ``features.b``: records of
    ``10 + 4*D`` bytes, position p at byte ``(4*D + 10) * p``."""

    def __init__(self, path, input_size):
        self.path = path
        self.input_size = int(input_size)
        size = os.path.getsize(path)
        rec = ID_BYTES + 4 * self.input_size
        if size % rec:
            raise ValueError('%s: size %d is not a multiple of the %d-byte record' % (path, size, rec))
        self.n = size // rec
        self.records = (np.memmap(path, dtype=_record_dtype(input_size), mode='r', shape=(self.n,))
                        if self.n else np.zeros(0, _record_dtype(input_size)))

    def features(self, positions):
        return np.ascontiguousarray(self.records['x'][np.asarray(positions, dtype=np.int64)])

    def ids(self, positions=None):
        raw = self.records['id'] if positions is None else self.records['id'][np.asarray(positions, dtype=np.int64)]
        return [b.decode('ascii') for b in raw]

    def all_features(self):
        return np.ascontiguousarray(self.records['x'])


_FILES = {}


def _open(path, input_size):
    key = (os.path.abspath(path), int(input_size), os.path.getmtime(path), os.path.getsize(path))
    ff = _FILES.get(key)
    if ff is None:
        ff = _FILES[key] = FeatureFile(path, input_size)
    return ff


def load_features(path, input_size=28 * 28):
    """Yield (id, vector) for every record (cfl/input_data.py:195-209)."""
    ff = _open(path, input_size)
    for i, asin in enumerate(ff.ids()):
        yield asin, ff.records['x'][i]


def load_features_by_positions(path, positions, input_size=28 * 28):
    """[len(positions), input_size] float32 (cfl/input_data.py:212-228)."""
    return _open(path, input_size).features(positions)


def load_asins_by_positions(path, positions, input_size=28 * 28):
    """cfl/input_data.py:231-245."""
    return _open(path, input_size).ids(positions)


def load_features_indices(path, input_size=28 * 28):
    """{id: position}; a repeated id keeps its last position (cfl/input_data.py:248-265)."""
    return {asin: i for i, asin in enumerate(_open(path, input_size).ids())}


def load_meta_lines(path):
    """Group a meta.txt into (id, lines): a line starting with a space continues
    the current item (cfl/input_data.py:268-287)."""
    current, lines = None, []
    with open(path) as infile:
        for line in infile:
            if line.startswith(' '):
                assert lines and current, 'must have valid id'
                lines.append(line)
                continue
            if current:
                yield current, lines
            current, lines = line.split(' ', 1)[0].strip(), [line]
    if current:
        yield current, lines


# ---- image / double record files (cfl/input_data.py:34-192) -----------------------------------
#   image  record: id[10] | int32 size | encoded image
#   double record: id[10] | int32 size1 | int32 size2 | encoded image | latent
#                  (latent = raw little-endian f32 when raw_latent, else an .npz with key 'data')
def imread(file_or_bytes):
    """Decode to an ndarray (HxW or HxWxC uint8), the contract of scipy.misc.imread."""
    from PIL import Image
    with Image.open(file_or_bytes) as im:
        return np.asarray(im)


def dump_image(outfile, image, image_format='png'):
    """Append int32 size + the encoded image (cfl/input_data.py:34-47); `image` is uint8 or float in [0,1]."""
    from PIL import Image
    arr = np.asarray(image)
    if arr.dtype != np.uint8:
        arr = np.clip(np.rint(arr * 255.0), 0, 255).astype(np.uint8)
    if arr.ndim == 3 and arr.shape[2] == 1:
        arr = arr[:, :, 0]          # single-channel images are stored as greyscale
    buf = BytesIO()
    Image.fromarray(arr).save(buf, format=image_format)
    outfile.write(struct.pack('<i', buf.getbuffer().nbytes))
    outfile.write(buf.getvalue())


def _scan_offsets(path, n_sizes):
    offset, offsets = 0, {}
    with open(path, 'rb') as infile:
        while True:
            asin = infile.read(ID_BYTES).decode('ascii')
            if not asin:
                break
            offsets[asin] = offset
            sizes = struct.unpack('<' + 'i' * n_sizes, infile.read(4 * n_sizes))
            offset += ID_BYTES + 4 * n_sizes + sum(sizes)
            infile.seek(offset)
    return offsets


def load_images_offsets(path):
    """{id: byte offset} of an image file (cfl/input_data.py:173-192)."""
    return _scan_offsets(path, 1)


def load_double_offsets(path):
    """{id: byte offset} of an image+latent file (cfl/input_data.py:151-170)."""
    return _scan_offsets(path, 2)


def load_asins_by_offsets(path, offsets):
    """cfl/input_data.py:136-148."""
    asins = []
    with open(path, 'rb') as infile:
        for offset in offsets:
            infile.seek(int(offset))
            asins.append(infile.read(ID_BYTES).decode('ascii'))
    return asins


def load_images(path):
    """Yield (id, flattened float32 image / 255) (cfl/input_data.py:50-64)."""
    with open(path, 'rb') as infile:
        while True:
            asin = infile.read(ID_BYTES).decode('ascii')
            if not asin:
                break
            size = struct.unpack('<i', infile.read(4))[0]
            image = imread(BytesIO(infile.read(size))).astype(np.float32) / 255.
            yield asin, image.reshape((-1,))


def load_images_by_offsets(path, offsets):
    """[len(offsets), H*W*C] float32 in [0,1] (cfl/input_data.py:67-84)."""
    images = []
    with open(path, 'rb') as infile:
        for offset in offsets:
            infile.seek(int(offset) + ID_BYTES)
            size = struct.unpack('<i', infile.read(4))[0]
            images.append(imread(BytesIO(infile.read(size))).astype(np.float32) / 255.)
    return np.array(images).reshape((len(offsets), -1))


def load_double_images_by_offsets(path, offsets, raw_latent=False):
    """(images [n, H*W*C] in [0,1], latents [n, Dlat]) (cfl/input_data.py:107-133)."""
    images, latents = [], []
    with open(path, 'rb') as infile:
        for offset in offsets:
            infile.seek(int(offset) + ID_BYTES)
            size1, size2 = struct.unpack('<ii', infile.read(8))
            images.append(imread(BytesIO(infile.read(size1))).astype(np.float32) / 255.)
            if raw_latent:
                latent = array('f')
                latent.fromfile(infile, size2 // 4)
            else:
                latent = np.load(BytesIO(infile.read(size2)))['data']
            latents.append(latent)
    return (np.array(images).reshape((len(offsets), -1)),
            np.array(latents, dtype=np.float32).reshape((len(offsets), -1)))


class DecodedRecords(object):
    """Decoded image (+ latent) records of one image / double ``features.b``, kept in host memory (new functionality; SURVEY.md
    8(f).1).  The reference seeks, reads and DECODES every record of every batch, every iteration
    (cfl/input_data.py:67-133: ~600 PNG decodes per MrCGAN post-epoch iteration at batch 100 -- an order of magnitude more
    host time than the GPU step it feeds).  Here a record is decoded ONCE, the first time a batch asks for it: its pixels
    go into one uint8 table [n_records, H*W*C] (12 KiB per 64x64x3 image), its latent into one float32 table, and a batch
    is a fancy-index gather of table rows followed by the reference's own conversion ``uint8.astype(float32) / 255.`` --
    the same values bit for bit, applied to the gathered rows instead of image by image.

    Records that are not uint8 images of one common shape make the table give up (`usable` False): the caller reads such a
    file record by record as before.  CFL_IMAGE_TABLE_MB caps the tables (default: 32768 or a quarter of the host's memory,
    whichever is smaller); 0 disables them."""

    def __init__(self, path, offsets, is_double, raw_latent):
        self.path, self.is_double, self.raw_latent = path, is_double, raw_latent
        offs = np.sort(np.asarray(list(offsets), dtype=np.int64))
        self.row_of = {int(o): i for i, o in enumerate(offs)}
        self._off_of = offs       # row -> byte offset
        self.n = offs.shape[0]
        self.have = np.zeros(self.n, dtype=bool)
        self.images = None        # uint8 [n, H*W*C], allocated at the first decode
        self.latents = None       # float32 [n, Dl]
        self.usable = self.n > 0 and self.cap_bytes() > 0

    @staticmethod
    def cap_bytes():
        """CFL_IMAGE_TABLE_MB, or by default the smaller of 32 GiB and a quarter of the host's physical memory"""
        env = os.environ.get('CFL_IMAGE_TABLE_MB')
        if env is not None:
            try:
                return int(float(env) * (1 << 20))
            except ValueError:
                pass
        cap = 32768 << 20
        try:
            cap = min(cap, os.sysconf('SC_PAGE_SIZE') * os.sysconf('SC_PHYS_PAGES') // 4)
        except (ValueError, OSError, AttributeError):
            pass
        return cap

    def _decode(self, offsets):
        """[(uint8 pixels, latent or None)] of the records at `offsets`, or None when a record does not fit the tables"""
        out = []
        with open(self.path, 'rb') as infile:
            for offset in offsets:
                infile.seek(int(offset) + ID_BYTES)
                if self.is_double:
                    size1, size2 = struct.unpack('<ii', infile.read(8))
                else:
                    size1, size2 = struct.unpack('<i', infile.read(4))[0], 0
                img = imread(BytesIO(infile.read(size1)))
                if img.dtype != np.uint8:
                    return None
                lat = None
                if self.is_double:
                    if self.raw_latent:
                        lat = np.frombuffer(infile.read(size2), dtype='<f4')
                    else:
                        lat = np.asarray(np.load(BytesIO(infile.read(size2)))['data'], dtype=np.float32).reshape(-1)
                out.append((img.reshape(-1), lat))
        return out

    def rows(self, offsets):
        """table rows of the records at `offsets`, decoding the ones not seen yet; None when the table cannot serve them"""
        if not self.usable:
            return None
        try:
            rows = np.fromiter((self.row_of[int(o)] for o in offsets), dtype=np.int64, count=len(offsets))
        except KeyError:
            return None
        missing = np.unique(rows[~self.have[rows]])
        if missing.size:
            dec = self._decode(self._off_of[missing])
            if dec is None:
                self.usable = False
                return None
            if self.images is None:
                px = dec[0][0].shape[0]
                dl = dec[0][1].shape[0] if self.is_double else 0
                if self.n * (px + 4 * dl) > self.cap_bytes():
                    self.usable = False
                    return None
                self.images = np.empty((self.n, px), dtype=np.uint8)
                self.latents = np.empty((self.n, dl), dtype=np.float32) if self.is_double else None
            for r, (img, lat) in zip(missing, dec):
                if img.shape[0] != self.images.shape[1] or (self.is_double and lat.shape[0] != self.latents.shape[1]):
                    self.usable = False       # ragged records: the per-record readers handle (or reject) them as before
                    return None
                self.images[r] = img
                if self.is_double:
                    self.latents[r] = lat
            self.have[missing] = True
        return rows

    def all_latents(self):
        """(sorted byte offsets [n], latents [n, Dl]) of EVERY record of a double file -- only the latents are read, no image is
        decoded (a seek past the image bytes) -- or None when the records' latents are ragged.  What ResidentFeatures keeps in
        HBM for a model that trains on the latents of an image + latent dataset."""
        if not self.is_double or self.n == 0:
            return None
        rows = []
        with open(self.path, 'rb') as infile:
            for offset in self._off_of:
                infile.seek(int(offset) + ID_BYTES)
                size1, size2 = struct.unpack('<ii', infile.read(8))
                infile.seek(size1, 1)
                if self.raw_latent:
                    rows.append(np.frombuffer(infile.read(size2), dtype='<f4'))
                else:
                    rows.append(np.asarray(np.load(BytesIO(infile.read(size2)))['data'], dtype=np.float32).reshape(-1))
        if len({r.shape[0] for r in rows}) != 1:
            return None
        return self._off_of, np.stack(rows).astype(np.float32, copy=False)

    def load(self, offsets):
        """what load_images_by_offsets / load_double_images_by_offsets return for `offsets`, or None"""
        rows = self.rows(offsets)
        if rows is None:
            return None
        images = self.images[rows].astype(np.float32) / 255.
        return (images, self.latents[rows]) if self.is_double else images


def _read_pairs(path, index, reorder=False):
    pairs = []
    with open(path) as infile:
        rows = [line.strip().split() for line in infile]
    rows = [(r[0], r[2]) for r in rows if r]
    if reorder:
        # round-robin over first-character buckets, popping from the back
        # (cfl/input_data.py:434-447), used for ordered visualisation sets
        buckets = defaultdict(list)
        for a, b in rows:
            buckets[a[0]].append((a, b))
        rows = []
        while any(buckets.values()):
            for key in sorted(buckets):
                if buckets[key]:
                    rows.append(buckets[key].pop())
    for a, b in rows:
        pairs.append([index[a], index[b]])
    return np.array(pairs)


def _read_ids(path):
    with open(path) as infile:
        return [line.strip() for line in infile]


def load_data_sets(path, input_size, data_switch=False, raw_latent=False, is_image=False,
                   is_double=False, directed=False, reorder=False, seed=633):
    """train / val / test splits (cfl/input_data.py:290-341): ``data_switch`` only
    on train, ``reorder`` only on test, the same seed for all three."""
    common = dict(input_size=input_size, raw_latent=raw_latent, is_image=is_image,
                  is_double=is_double, directed=directed, seed=seed)
    return Namespace(
        train=SemiDataSet(os.path.join(path, 'train'), data_switch=data_switch, **common),
        val=SemiDataSet(os.path.join(path, 'val'), **common),
        test=SemiDataSet(os.path.join(path, 'test'), reorder=reorder, **common))


class SemiDataSet(object):
    """Positive / negative pair lists over one ``features.b`` plus the seeded batch
    streams of cfl/input_data.py:344-690 (vector, image and image+latent datasets)."""

    def __init__(self, path, input_size=28 * 28, data_switch=False, is_image=False,
                 is_double=False, directed=False, reorder=False, raw_latent=False, seed=633):
        self._rng = RandomState(seed)
        self.input_size = input_size
        self.feature_path = os.path.join(path, 'features.b')
        self.is_image, self.is_double = is_image, is_double
        self.directed = directed
        self.data_switch = data_switch
        self.raw_latent = raw_latent

        self._records = None      # DecodedRecords of an image / double file (built at the first batch)
        self.labeled_images = True   # False: labeled batches of a double dataset carry placeholder images (_load_labeled)
        if is_image:
            # positions are byte offsets; the unlabeled order is permuted once up front
            # (cfl/input_data.py:379-391)
            self._file = None
            self.asins_to_index = (load_double_offsets if is_double else load_images_offsets)(self.feature_path)
            self.index_to_asins = {i: a for a, i in self.asins_to_index.items()}
            self.num_examples = len(self.index_to_asins)
            self.item_indices = np.array(sorted(self.index_to_asins))
            self.item_indices = self.item_indices[self._rng.permutation(self.num_examples)]
        else:
            self._file = _open(self.feature_path, input_size)
            self.asins_to_index = load_features_indices(self.feature_path, input_size)
            self.index_to_asins = {i: a for a, i in self.asins_to_index.items()}
            # quirk kept on purpose: the LAST item is never drawn as unlabeled
            # (cfl/input_data.py:399)
            self.num_examples = max(self.index_to_asins)
            self.item_indices = np.arange(self.num_examples)
        self.head_unlabeled = 0

        if directed:
            for name in ('source', 'target'):
                idx = np.array(sorted(self.asins_to_index[a] for a in _read_ids(os.path.join(path, name + '.txt'))))
                idx = idx[self._rng.permutation(idx.shape[0])]
                setattr(self, name + '_indices', idx)
                setattr(self, 'num_' + name, idx.shape[0])
                setattr(self, 'head_' + name, 0)

        self.pairs_pos = _read_pairs(os.path.join(path, 'pairs_pos.txt'), self.asins_to_index, reorder)
        self.pairs_neg = _read_pairs(os.path.join(path, 'pairs_neg.txt'), self.asins_to_index)
        self.head_labeled_pos = self.head_labeled_neg = 0
        self.num_examples_labeled_pos = self.pairs_pos.shape[0]
        self.num_examples_labeled_neg = self.pairs_neg.shape[0]
        self._prefetch = None     # reshuffle of the next epoch wrap, computed ahead of time (prefetch_reshuffle)
        self._int32_of = {}       # 'pos' / 'neg' -> (pair list object, its int32 copy) when the worker made one

    # -- feature access ------------------------------------------------------
    def _load_features_by_positions(self, indices):
        if self.is_image:
            if self._records is None:
                self._records = DecodedRecords(self.feature_path, self.index_to_asins, self.is_double, self.raw_latent)
            got = self._records.load(indices)
            if got is not None:
                return got
            if self.is_double:
                return load_double_images_by_offsets(self.feature_path, indices, raw_latent=self.raw_latent)
            return load_images_by_offsets(self.feature_path, indices)
        return self._file.features(indices)

    def _load_asins_by_positions(self, indices):
        if self.is_image:
            return load_asins_by_offsets(self.feature_path, indices)
        return self._file.ids(indices)

    def _parts(self, loaded):
        """A loaded item batch as a list of arrays: [x] or, for double data, [image, latent]."""
        return list(loaded) if self.is_double else [loaded]

    # -- labeled stream --------------------------------------------------------
    def _draw(self, which, batch_size):
        pairs = getattr(self, 'pairs_' + which)
        n = pairs.shape[0]
        head = getattr(self, 'head_labeled_' + which)
        if head + batch_size > n:           # epoch wrap: reshuffle in place
            head = 0
            pairs = self._reshuffled(which, pairs, n)
            setattr(self, 'pairs_' + which, pairs)
        setattr(self, 'head_labeled_' + which, head)
        return pairs, head, n

    # -- the per-epoch reshuffle, ahead of time ---------------------------------------------------------------
    # `pairs[rng.permutation(n)]` is a serial MT19937 shuffle + a random gather: ~7 ms per list of 200 k pairs,
    # nothing next to the reference's seconds-long epochs, but as long as a whole epoch of fused steps here.  The
    # stream only draws from the generator at epoch wraps (and per batch with data_switch / oversampling), so the next
    # wrap's permutation can be computed EARLY, on a copy of the generator state in a worker thread, and adopted at
    # the wrap if -- and only if -- nobody drew from the generator in between (state compared); otherwise it is
    # thrown away and the wrap reshuffles as before.  Same stream, bit for bit.  The worker runs the library's
    # host-side restatement of the legacy shuffle (cfl_mt19937_reshuffle: numpy holds the interpreter lock through
    # permutation() and fancy indexing -- measured: 10 ms stalls of the launching thread -- a ctypes call does not).
    @staticmethod
    def _same_state(a, b):
        return a[2] == b[2] and a[3] == b[3] and a[4] == b[4] and np.array_equal(a[1], b[1])

    def _reshuffled(self, which, pairs, n):
        job, self._prefetch = self._prefetch, None
        if job is not None:
            job['thread'].join()
            step = job['steps'][0] if job['steps'] else None
            # ('after' is missing when the worker failed -- library error, out of memory: reshuffle synchronously then)
            if (step is not None and 'after' in step and step['which'] == which and step['pairs'] is pairs and
                    self._same_state(self._rng.get_state(), step['before'])):
                self._rng.set_state(step['after'])
                if len(job['steps']) > 1:          # the other list wraps in the same batch: keep its result
                    self._prefetch = dict(thread=job['thread'], steps=job['steps'][1:])
                self._int32_of[which] = (step['result'], step['result32'])    # for the device upload
                return step['result']
        return pairs[self._rng.permutation(n)]

    def prefetch_reshuffle(self, batch_size):
        """Start computing the reshuffle(s) of the next epoch wrap in the background (no-op when one is pending,
        when every batch draws from the generator anyway, or when the batch is larger than a list)."""
        if self._prefetch is not None or self.data_switch:
            return
        import threading
        left = {}
        for which in ('pos', 'neg'):
            n = getattr(self, 'pairs_' + which).shape[0]
            if batch_size > n:
                return
            left[which] = (n - getattr(self, 'head_labeled_' + which)) // batch_size   # batches before its wrap
        first = min(left.values())
        order = [w for w in ('pos', 'neg') if left[w] == first]        # _draw order within one batch: pos, then neg
        try:
            from . import hipabi
            hipabi.lib()
        except Exception:
            return                      # no library: every wrap reshuffles synchronously (the reference's way)
        state = self._rng.get_state()
        steps = [dict(which=w, pairs=getattr(self, 'pairs_' + w)) for w in order]

        def work():
            cur = state
            try:
                for st in steps:
                    st['before'] = cur
                    st['result'], cur, st['result32'] = hipabi.mt19937_reshuffle(cur, st['pairs'], want32=True)
                    st['after'] = cur
            except Exception as e:          # the wrap falls back to numpy (a step without 'after' is never adopted)
                steps[0].setdefault('error', repr(e))
        th = threading.Thread(target=work, daemon=True)
        th.start()
        self._prefetch = dict(thread=th, steps=steps)

    def next_batch_indices(self, batch_size):
        """The integer side of next_labeled_batch (cfl/input_data.py:542-580):
        returns (pos_pairs [B,2], neg_pairs [B,2], switched) and advances the
        stream with exactly the reference's RandomState call sequence."""
        pp, hp, npos = self._draw('pos', batch_size)
        pn, hn, nneg = self._draw('neg', batch_size)
        positions_pos = pp[hp:hp + batch_size]
        positions_neg = pn[hn:hn + batch_size]
        if batch_size > npos:
            positions_pos = pp[self._rng.choice(npos, batch_size)]
        if batch_size > nneg:
            positions_neg = pn[self._rng.choice(nneg, batch_size)]
        assert positions_pos.shape[0] == batch_size
        assert positions_neg.shape[0] == batch_size
        switched = bool(self.data_switch and self._rng.rand() > 0.5)
        self.head_labeled_pos += batch_size
        self.head_labeled_neg += batch_size
        return positions_pos, positions_neg, switched

    def next_labeled_batch(self, batch_size, return_labels=False):
        if return_labels:
            raise NotImplementedError()
        pos, neg, switched = self.next_batch_indices(batch_size)
        cols = (1, 0) if switched else (0, 1)
        out = []
        for pairs in (pos, neg):
            for c in cols:
                out.extend(self._parts(self._load_labeled(pairs[:, c])))
        return tuple(out)   # 4 arrays, or 8 (image, latent interleaved) for double data

    def _load_labeled(self, positions):
        """Items of a LABELED batch.  With `labeled_images` False (set by a consumer that reads only the latents of labeled
        double batches: the pair model on latents, the MrCGAN post epochs) the image parts are zero placeholders of the right
        shape (a read-only broadcast, no memory) and only the latents are gathered from the decoded-record table."""
        if self.is_double and not self.labeled_images:
            if self._records is None:
                self._records = DecodedRecords(self.feature_path, self.index_to_asins, self.is_double, self.raw_latent)
            rows = self._records.rows(positions)
            if rows is not None:
                tab = self._records
                return np.broadcast_to(np.zeros(1, np.float32), (rows.shape[0], tab.images.shape[1])), tab.latents[rows]
        return self._load_features_by_positions(positions)

    def next_batch(self, batch_size, return_labels=False):
        return self.next_labeled_batch(batch_size, return_labels)

    # -- whole-set iterators (dist_eval / dist_predict) --------------------------
    def _whole(self, pairs, batch_size, source_ids):
        for i in range(0, pairs.shape[0], batch_size):
            chunk = pairs[i:i + batch_size]
            out = tuple(self._parts(self._load_features_by_positions(chunk[:, 0])) +
                        self._parts(self._load_features_by_positions(chunk[:, 1])))
            if source_ids:
                out += (self._load_asins_by_positions(chunk[:, 0]),)
            yield out

    def whole_pos_batches(self, batch_size, source_ids=False):
        return self._whole(self.pairs_pos, batch_size, source_ids)

    def whole_neg_batches(self, batch_size, source_ids=False):
        return self._whole(self.pairs_neg, batch_size, source_ids)

    def whole_unlabeled_batches(self, batch_size, source_ids=False):
        for i in range(0, self.num_examples, batch_size):
            positions = self.item_indices[i:i + batch_size]
            data = self._parts(self._load_features_by_positions(positions))
            if source_ids:
                data.append(self._load_asins_by_positions(positions))
            yield data

    # -- unlabeled / directed streams (cfl/input_data.py:591-690) ------------------
    def _stream(self, attr, head_attr, n, batch_size, allow_choice):
        head = getattr(self, head_attr)
        idx = getattr(self, attr)
        if head + batch_size > n:
            head = 0
            idx = idx[self._rng.permutation(n)]
            setattr(self, attr, idx)
        positions = idx[head:head + batch_size]
        if allow_choice and batch_size > n:
            positions = idx[self._rng.choice(n, batch_size)]
            assert positions.shape[0] == batch_size
        setattr(self, head_attr, head + batch_size)
        return positions

    def _finish(self, positions, return_labels, source_ids):
        data = self._parts(self._load_features_by_positions(positions))
        if return_labels or source_ids:
            asins = self._load_asins_by_positions(positions)
            if return_labels:
                data.append(np.array([self.categories[a] for a in asins]))
            if source_ids:
                data.append(asins)
        return data

    def next_unlabeled_batch(self, batch_size, return_labels=False, source_ids=False):
        pos = self._stream('item_indices', 'head_unlabeled', self.num_examples, batch_size, False)
        return self._finish(pos, return_labels, source_ids)

    def next_source_batch(self, batch_size, return_labels=False, source_ids=False):
        if not self.directed:
            return self.next_unlabeled_batch(batch_size, return_labels, source_ids)
        pos = self._stream('source_indices', 'head_source', self.num_source, batch_size, True)
        return self._finish(pos, return_labels, source_ids)

    def next_target_batch(self, batch_size, return_labels=False, source_ids=False):
        if not self.directed:
            return self.next_unlabeled_batch(batch_size, return_labels, source_ids)
        # the reference applies no oversampling to the target stream
        pos = self._stream('target_indices', 'head_target', self.num_target, batch_size, False)
        return self._finish(pos, return_labels, source_ids)


class ResidentFeatures(object):
    """The whole ``features.b`` of a split kept in HBM, and the labeled pair lists beside it.

    Replaces the reference's batch assembly (cfl/input_data.py:542-589: index pairs -> cfl/input_data.py:212-228: one
    seek + read per vector) without materialising a batch at all: ``next_indexed`` advances the dataset's own seeded
    index stream (``SemiDataSet.next_batch_indices``: the reference's RandomState call sequence, bit-exact) and
    hands the training step the *positions* of the batch -- a window of the device copy of the (shuffled) pair
    list, walked in place with stride 2 -- and the fused pair kernels read each feature row where it lies in the
    table (cfl_pair_train_step_idx).  The pair lists are re-uploaded only when the dataset reshuffles them (once
    per epoch).  ``next_batch`` still assembles dense batches with cfl_gather_rows for consumers that want them.
    The feature dimension is zero-padded to a multiple of 64 for the kernels."""

    def __init__(self, dataset, device='cuda'):
        import torch
        from . import hipabi
        self._h = hipabi
        self._offsets = None       # image + latent datasets: sorted record offsets (positions are byte offsets there: _rows)
        if dataset.is_image:
            # An image + latent ("double") dataset whose consumer trains on the LATENTS (cfl.models.cfl: uses_latent): the
            # latents of all records are the feature table.  The dataset's positions stay byte offsets (the reference's, and
            # what the golden streams pin); they are translated to table rows where they cross to the device.
            if not dataset.is_double:
                raise ValueError('resident features of an image-only dataset: nothing to keep (the pixels are the input)')
            if dataset._records is None:
                dataset._records = DecodedRecords(dataset.feature_path, dataset.index_to_asins, True, dataset.raw_latent)
            got = dataset._records.all_latents()
            if got is None:
                raise ValueError('ragged latents in ' + dataset.feature_path)
            self._offsets, x = got
        else:
            x = dataset._file.all_features()
        D = x.shape[1]
        self.input_size = D
        self.padded_size = (D + 63) // 64 * 64
        t = torch.from_numpy(x)
        if self.padded_size != D:
            t = torch.nn.functional.pad(t, (0, self.padded_size - D))
        self.table = t.contiguous().to(device)
        self.device = self.table.device
        self.dataset = dataset
        if self.table.shape[0] >= 2 ** 31:
            raise ValueError('feature table too large for int32 row indices')
        self._pairs = {}      # 'pos' / 'neg' -> (host array object that was uploaded, device int32 [n, 2])
        self._staging = {}    # 'pos' / 'neg' -> [pinned int32 [n, 2], event of the last upload from it]

    def _rows(self, positions):
        """table rows of dataset positions (identity for vector datasets; byte offset -> rank among the sorted offsets for
        image + latent datasets)"""
        if self._offsets is None:
            return positions
        return np.searchsorted(self._offsets, np.asarray(positions, dtype=np.int64))

    def gather(self, positions, out=None):
        import torch
        idx = torch.as_tensor(np.ascontiguousarray(self._rows(positions), dtype=np.int64)).to(self.device, non_blocking=True)
        return self._h.gather_rows(self.table, idx, out)

    def next_batch(self, batch_size, shard=None):
        """Device tensors (pos_src, pos_dst, neg_src, neg_dst), same rows as
        ``SemiDataSet.next_batch`` would return for this call.  ``shard=(lo, hi)``
        keeps only that row range of the global batch (data parallelism: every rank
        advances the same seeded index stream and gathers its own slice)."""
        pos, neg, switched = self.dataset.next_batch_indices(batch_size)
        if shard is not None:
            pos, neg = pos[shard[0]:shard[1]], neg[shard[0]:shard[1]]
        c = (1, 0) if switched else (0, 1)
        return (self.gather(pos[:, c[0]]), self.gather(pos[:, c[1]]),
                self.gather(neg[:, c[0]]), self.gather(neg[:, c[1]]))

    def _device_pairs(self, which):
        """int32 device copy of dataset.pairs_<which> in its CURRENT order (the dataset replaces the array object
        when it reshuffles: cfl/input_data.py:543-551)."""
        import torch
        host = getattr(self.dataset, 'pairs_' + which)
        cached = self._pairs.get(which)
        if cached is None or cached[0] is not host:
            # pinned staging + stream-ordered copy: the upload neither waits for the queued training steps (which
            # still read the previous device copy -- kept alive by `cached` until this assignment, and by the
            # stream order afterwards) nor stalls the host
            # (hipHostMalloc costs ~10 ms: ONE pinned staging buffer per list, allocated once and reused; the event
            # says when the previous upload has left it)
            st = self._staging.get(which)
            if st is None or st[0].shape != host.shape:
                st = self._staging[which] = [torch.empty(host.shape, dtype=torch.int32).pin_memory(), None]
            if st[1] is not None:
                st[1].synchronize()
            ready = getattr(self.dataset, '_int32_of', {}).get(which)
            if self._offsets is not None:
                host32 = self._rows(host).astype(np.int32)
            else:
                host32 = ready[1] if (ready is not None and ready[0] is host) else host.astype(np.int32)
            # one memcpy into the pinned buffer.  NOT tensor.copy_: torch splits a 400 k-element host copy over
            # its whole OpenMP pool, whose threads then spin at the team barrier (measured on a 256-core box: 0.8
            # CPU-seconds per call, enough to run the container into its CPU quota and stall the launch thread)
            np.copyto(st[0].numpy(), host32, casting='no')
            dev = st[0].to(self.device, non_blocking=True)
            st[1] = torch.cuda.Event()
            st[1].record()
            cached = self._pairs[which] = (host, dev)
        return cached[1]

    def next_indexed(self, batch_size, shard=None):
        """(table, IndexStreams) of the next labeled batch for PairEngine.step: no row is copied.  Same rows, same
        order as ``next_batch``; ``shard`` as there."""
        ds = self.dataset
        pos, neg, switched = ds.next_batch_indices(batch_size)
        lo, hi = shard if shard is not None else (0, batch_size)
        c = (1, 0) if switched else (0, 1)
        ptrs, keep = [], []
        for which, rows in (('pos', pos), ('neg', neg)):
            host = getattr(ds, 'pairs_' + which)
            head = getattr(ds, 'head_labeled_' + which) - batch_size      # next_batch_indices has advanced it
            if np.may_share_memory(rows, host):
                dev = self._device_pairs(which)                         # window [head, head + B) of the pair list
                base = dev.data_ptr() + 8 * (head + lo)
            else:
                # B > number of pairs: the reference oversamples with RandomState.choice (cfl/input_data.py:558-566);
                # such a batch is not a window of the list -- upload its positions
                import torch
                dev = torch.from_numpy(np.ascontiguousarray(self._rows(rows[lo:hi]), dtype=np.int32)).to(self.device)
                base = dev.data_ptr()
            keep.append(dev)
            ptrs += [base + 4 * c[0], base + 4 * c[1]]
        ds.prefetch_reshuffle(batch_size)
        return self.table, self._h.IndexStreams(ptrs, 2, hi - lo, keep=keep)

    def available_windows(self, batch_size):
        """how many consecutive labeled batches the next next_windows() call could return (0: the next batch wraps an
        epoch or is larger than a list) -- without touching the dataset's stream"""
        ds = self.dataset
        npos, nneg = ds.pairs_pos.shape[0], ds.pairs_neg.shape[0]
        if batch_size > npos or batch_size > nneg:
            return 0
        return max(0, min((npos - ds.head_labeled_pos) // batch_size, (nneg - ds.head_labeled_neg) // batch_size))

    def next_window_any(self, batch_size, shard=None):
        """ONE labeled batch as a window descriptor (nsteps = 1) whatever the stream does next -- an epoch wrap reshuffles the
        list first, exactly as next_batch_indices does, and the batch is then the window at its head -- or None, with the
        stream untouched, when a batch is not a window at all (B larger than a pair list: the reference oversamples)."""
        from argparse import Namespace
        ds = self.dataset
        if batch_size > ds.pairs_pos.shape[0] or batch_size > ds.pairs_neg.shape[0]:
            return None
        _, _, switched = ds.next_batch_indices(batch_size)
        lo, hi = shard if shard is not None else (0, batch_size)
        win = Namespace(table=self.table, pos_pairs=self._device_pairs('pos'), neg_pairs=self._device_pairs('neg'),
                        pos_head=ds.head_labeled_pos - batch_size, neg_head=ds.head_labeled_neg - batch_size,
                        batch_rows=batch_size, shard_lo=lo, rows=hi - lo, nsteps=1,
                        switched=[switched] if ds.data_switch else None)
        ds.prefetch_reshuffle(batch_size)
        return win

    def next_windows(self, batch_size, max_steps, shard=None):
        """Up to `max_steps` consecutive labeled batches as ONE window descriptor for PairEngine.step_windows, or
        None when the very next batch is not a plain window of both pair lists (an epoch wrap reshuffles, or the
        batch is larger than a list) -- the caller then takes that batch through next_indexed.  The dataset's
        stream is advanced exactly as `nsteps` calls of next_batch_indices would: between reshuffles a batch only
        moves the two heads and, with data_switch, draws one coin flip (cfl/input_data.py:542-577)."""
        from argparse import Namespace
        ds = self.dataset
        npos, nneg = ds.pairs_pos.shape[0], ds.pairs_neg.shape[0]
        k = min(int(max_steps), (npos - ds.head_labeled_pos) // batch_size, (nneg - ds.head_labeled_neg) // batch_size)
        if k <= 0 or batch_size > npos or batch_size > nneg:
            return None
        lo, hi = shard if shard is not None else (0, batch_size)
        win = Namespace(table=self.table, pos_pairs=self._device_pairs('pos'), neg_pairs=self._device_pairs('neg'),
                        pos_head=ds.head_labeled_pos, neg_head=ds.head_labeled_neg, batch_rows=batch_size,
                        shard_lo=lo, rows=hi - lo, nsteps=k, switched=None)
        if ds.data_switch:
            win.switched = [bool(ds._rng.rand() > 0.5) for _ in range(k)]
        ds.head_labeled_pos += k * batch_size
        ds.head_labeled_neg += k * batch_size
        ds.prefetch_reshuffle(batch_size)
        return win

    def whole_indexed(self, which, batch_size, rows=None):
        """(table, IndexStreams of (src, dst)) chunks over the pairs of pairs_<which> in file order, for scoring
        (the whole_pos_batches / whole_neg_batches of cfl/utils.py:233-266).  rows=(lo, hi): only that range of
        the list (a rank's shard of a data-parallel evaluation)."""
        import torch
        host = getattr(self.dataset, 'pairs_' + which)
        lo, hi = rows if rows is not None else (0, host.shape[0])
        host = host[lo:hi]
        dev = torch.from_numpy(np.ascontiguousarray(self._rows(host), dtype=np.int32)).to(self.device)
        for i in range(0, host.shape[0], batch_size):
            n = min(batch_size, host.shape[0] - i)
            base = dev.data_ptr() + 8 * i
            yield self.table, self._h.IndexStreams([base, base + 4], 2, n, keep=[dev])


class PinnedUploader(object):
    """Host batches -> device tensors without stalling the host (new functionality).  `torch.from_numpy(x).to(device)` from
    pageable memory is a synchronous copy on the CURRENT stream: it waits for everything enqueued there -- the previous
    training step -- so host batch assembly and the GPU step take turns (measured on the MrCGAN post-epoch loop: 16.5 ms
    per iteration for a 12.4 ms step and 3 ms of host work).  Here the rows are copied into a rotating pinned buffer
    (one ring per shape) and uploaded asynchronously on a copy stream of the uploader's own; the consumer's stream waits
    for the upload's event, the host does not wait at all unless a ring wraps onto an upload that has not left its buffer."""

    NBUF = 8                   # pinned buffers per shape ...
    RING_BYTES = 256 << 20     # ... within this many bytes per shape (at least two)
    MAX_BYTES = 128 << 20      # larger arrays take the synchronous pageable copy

    def __init__(self, device):
        import torch
        self.device = torch.device(device)
        self.stream = torch.cuda.Stream(device=self.device)
        self._rings = {}       # shape -> [next, [[pinned, event], ...]]

    def upload(self, x):
        """contiguous float32 device tensor with the values of the array `x` (ready on the CURRENT stream)"""
        import torch
        a = np.ascontiguousarray(x, dtype=np.float32)
        if a.nbytes == 0 or a.nbytes > self.MAX_BYTES:      # (nothing to stage / not worth pinning: the plain copy)
            return torch.from_numpy(a).to(self.device)
        ring = self._rings.get(a.shape)
        if ring is None:
            nbuf = max(2, min(self.NBUF, self.RING_BYTES // a.nbytes))
            ring = self._rings[a.shape] = [0, [[torch.empty(a.shape, dtype=torch.float32).pin_memory(), None]
                                              for _ in range(nbuf)]]
        slot = ring[1][ring[0]]
        ring[0] = (ring[0] + 1) % len(ring[1])
        if slot[1] is not None:
            slot[1].synchronize()          # the previous upload from this pinned buffer has left it
        np.copyto(slot[0].numpy(), a)
        cur = torch.cuda.current_stream(self.device)
        with torch.cuda.stream(self.stream):
            dev = torch.empty(a.shape, dtype=torch.float32, device=self.device)
            dev.copy_(slot[0], non_blocking=True)
            slot[1] = torch.cuda.Event()
            slot[1].record(self.stream)
        cur.wait_event(slot[1])
        dev.record_stream(cur)             # allocated on the copy stream, used (and eventually freed) under `cur`
        return dev


class StreamedFeatures(object):
    """A split whose ``features.b`` does NOT fit in HBM (or ``CFL_FEATURES=stream``): the file stays memory-mapped on the
    host and every batch's rows are gathered into pinned staging memory and copied to the device asynchronously --
    SURVEY 8(f).1's second option, replacing the reference's one seek + read per vector (cfl/input_data.py:212-228) with
    one vectorised gather + one DMA per batch.  Same interface as ResidentFeatures: a batch is handed to the kernels as a
    (mini table, IndexStreams) pair -- the 4 x B gathered rows uploaded as a small table that the indexed entry points
    walk with the identity index streams -- so every consumer (training step, validation fetch, dist_eval / dist_predict)
    is unchanged and the arithmetic is the resident path's bit for bit.  The dataset's seeded index stream is advanced by
    the same calls (bit-exact).  Staging buffers rotate (NBUF): the copy of batch i+1 overlaps the step of batch i; a
    buffer is reused only after the event behind its last upload.  Windows of device pair lists do not exist here
    (next_windows / next_window_any return None: the loops take their per-iteration path)."""

    NBUF = 3

    def __init__(self, dataset, device='cuda'):
        import torch
        from . import hipabi
        self._h = hipabi
        self.dataset = dataset
        self.file = dataset._file
        self.input_size = int(self.file.input_size)
        self.padded_size = (self.input_size + 63) // 64 * 64
        self.device = torch.device(device)
        if self.device.type != 'cuda':
            raise hipabi.CflHipError('StreamedFeatures feeds a HIP device (no CPU fallback)')
        self._ring = {}        # rows per upload -> [next, [(pinned, device, event), ...]]
        self._arange = {}      # rows per stream -> int32 device vector 0 .. n-1 (the identity index stream)
        self.table = None      # (no resident table)

    def _buffers(self, rows):
        import torch
        ring = self._ring.get(rows)
        if ring is None:
            bufs = []
            for _ in range(self.NBUF):
                pinned = torch.zeros(rows, self.padded_size, dtype=torch.float32).pin_memory()     # (pad columns stay zero)
                bufs.append([pinned, torch.empty(rows, self.padded_size, dtype=torch.float32, device=self.device), None])
            ring = self._ring[rows] = [0, bufs]
        slot = ring[1][ring[0]]
        ring[0] = (ring[0] + 1) % self.NBUF
        if slot[2] is not None:
            slot[2].synchronize()          # the previous upload from this pinned buffer has left it
        return slot

    def _upload(self, position_lists):
        """rows file[positions] of every list, concatenated, as one device mini table [sum(len), padded D]"""
        import torch
        total = sum(len(p) for p in position_lists)
        slot = self._buffers(total)
        host = slot[0].numpy()
        o = 0
        x = self.file.records['x']
        for p in position_lists:
            n = len(p)
            host[o:o + n, :self.input_size] = x[np.asarray(p, dtype=np.int64)]
            o += n
        slot[1].copy_(slot[0], non_blocking=True)
        slot[2] = torch.cuda.Event()
        slot[2].record()
        return slot[1]

    def _identity(self, n, base):
        import torch
        key = (n, base)
        t = self._arange.get(key)
        if t is None:
            t = self._arange[key] = torch.arange(base, base + n, dtype=torch.int32, device=self.device)
        return t

    def _streams(self, n, groups):
        ts = [self._identity(n, k * n) for k in range(groups)]
        return self._h.IndexStreams([t.data_ptr() for t in ts], 1, n, keep=ts)

    def next_indexed(self, batch_size, shard=None):
        pos, neg, switched = self.dataset.next_batch_indices(batch_size)
        lo, hi = shard if shard is not None else (0, batch_size)
        c = (1, 0) if switched else (0, 1)
        table = self._upload([pos[lo:hi, c[0]], pos[lo:hi, c[1]], neg[lo:hi, c[0]], neg[lo:hi, c[1]]])
        return table, self._streams(hi - lo, 4)

    def next_batch(self, batch_size, shard=None):
        table, streams = self.next_indexed(batch_size, shard)
        n = streams.n
        return tuple(table[k * n:(k + 1) * n] for k in range(4))

    def available_windows(self, batch_size):
        return 0

    def next_windows(self, batch_size, max_steps, shard=None):
        return None

    def next_window_any(self, batch_size, shard=None):
        return None

    def whole_indexed(self, which, batch_size, rows=None):
        host = getattr(self.dataset, 'pairs_' + which)
        lo, hi = rows if rows is not None else (0, host.shape[0])
        host = host[lo:hi]
        step = min(int(batch_size), 4096)      # (2 x 4096 rows of 4 D bytes per staging buffer, not dist_eval's 32768-pair calls)
        for i in range(0, host.shape[0], step):
            chunk = host[i:i + step]
            table = self._upload([chunk[:, 0], chunk[:, 1]])
            yield table, self._streams(chunk.shape[0], 2)


def feature_source(dataset, device='cuda'):
    """ResidentFeatures when the split's feature table fits the device comfortably, StreamedFeatures otherwise.
    CFL_FEATURES=resident|stream forces the choice; CFL_RESIDENT_FRACTION (default 0.6) is the share of the device's FREE
    memory a table may take (the table is uploaded once and stays)."""
    import torch
    if dataset.is_image:
        return ResidentFeatures(dataset, device)      # (image + latent dataset: the latents of all records; no streamed form)
    mode = os.environ.get('CFL_FEATURES', 'auto')
    if mode == 'stream':
        return StreamedFeatures(dataset, device)
    if mode != 'resident' and torch.cuda.is_available():
        ff = dataset._file
        need = ff.n * ((int(ff.input_size) + 63) // 64 * 64) * 4
        dev = torch.device(device)
        free, _ = torch.cuda.mem_get_info(dev if dev.index is not None else torch.cuda.current_device())
        if need > float(os.environ.get('CFL_RESIDENT_FRACTION', '0.6')) * free:
            logger.warning('%s: %.1f GB of features do not fit %.1f GB of free device memory: streaming batches from the host',
                           ff.path, need / 1e9, free / 1e9)
            return StreamedFeatures(dataset, device)
    return ResidentFeatures(dataset, device)
