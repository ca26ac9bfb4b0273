"""cfl/tf_bundle.py: TensorFlow's checkpoint format V2 (tensor bundle + SSTable index) read and written without
TensorFlow -- SURVEY 8(f).4, the exchange format of tf.train.Saver at cfl/utils.py:465-497 of the reference.

No TensorFlow is installable in the build image, so the pins are the format's published known answers and a full
round trip: CRC-32C test vector, CRC masking, footer magic, block trailers, prefix-compressed keys across restart
points and several data blocks, and the naming the reference's graph gives its Adam slots (doubled scope prefix)."""
import os
import struct

import numpy as np
import pytest

from cfl import tf_bundle as T


def test_crc32c_known_answers():
    assert T.crc32c(b'') == 0
    assert T.crc32c(b'123456789') == 0xE3069283                 # the CRC-32C check value (RFC 3720, B.4)
    assert T.crc32c(b'\x00' * 32) == 0x8A9136AA                 # RFC 3720 B.4: 32 bytes of zeros
    assert T.crc32c(b'\xff' * 32) == 0x62A8AB43                 #               32 bytes of ones
    assert T.crc32c(bytes(range(32))) == 0x46DD794E             #               32 incrementing bytes
    a, b = b'hello ', b'world'
    assert T.crc32c(b, T.crc32c(a)) == T.crc32c(a + b)          # Extend
    big = bytes(range(256)) * 40                                # the library's host routine (>= 4 KiB) == the byte loop
    c = 0xffffffff
    for x in big:
        c = T._TABLE[(c ^ x) & 0xff] ^ (c >> 8)
    assert T.crc32c(big) == c ^ 0xffffffff
    for v in (0, 1, 0x12345678, 0xffffffff, 0xE3069283):
        assert T.unmask(T.mask(v)) == v
    assert T.mask(0xE3069283) == ((((0xE3069283 >> 15) | (0xE3069283 << 17)) + 0xa282ead8) & 0xffffffff)


def test_table_layout_and_round_trip(tmp_path):
    rng = np.random.RandomState(0)
    keys = sorted({('CFL/DistEncoder/layer%03d/%s' % (i // 3, 'Vgb'[i % 3])).encode() for i in range(200)} | {b''})
    items = [(k, bytes(rng.randint(0, 256, size=rng.randint(0, 90)).astype(np.uint8))) for k in keys]
    path = str(tmp_path / 't.index')
    T.write_table(path, items, block_size=700)                  # several data blocks, restarts inside them
    raw = open(path, 'rb').read()
    assert struct.unpack('<Q', raw[-8:])[0] == 0xdb4775248b80fb57
    got = T.read_table(path)
    assert list(got.items()) == items
    # the index block names every data block; each block's trailer is type 0 + the masked crc of (contents + type)
    footer = raw[-48:]
    pos = 0
    _, pos = T._read_varint(footer, pos)
    _, pos = T._read_varint(footer, pos)
    ioff, pos = T._read_varint(footer, pos)
    isize, pos = T._read_varint(footer, pos)
    handles = list(T._block_entries(raw[ioff:ioff + isize]))
    assert len(handles) > 3
    last = b''
    for sep, h in handles:
        off, p = T._read_varint(h, 0)
        size, _ = T._read_varint(h, p)
        block = raw[off:off + size]
        assert raw[off + size] == 0
        assert T.unmask(struct.unpack_from('<I', raw, off + size + 1)[0]) == T.crc32c(b'\x00', T.crc32c(block))
        ks = [k for k, _ in T._block_entries(block)]
        assert ks == sorted(ks) and (not last or ks[0] > last) and ks[-1] <= sep
        last = ks[-1]
    # a flipped byte is caught by the block checksum
    bad = bytearray(raw)
    bad[10] ^= 1
    open(path, 'wb').write(bytes(bad))
    with pytest.raises(ValueError, match='checksum'):
        T.read_table(path)


def test_bundle_round_trip_and_entry_fields(tmp_path):
    rng = np.random.RandomState(1)
    arrays = {'Dist/Encoder/latent_outputs/fully_connected/weights': rng.randn(64, 20).astype(np.float32),
              'Dist/Encoder/latent_outputs/fully_connected/biases': rng.randn(20).astype(np.float32),
              'Dist/Thresholder/threshold/threshold': np.float32(1e-6), 'Dist/beta1_power': np.float32(0.9),
              'global_step': np.int64(11), 'aux64': rng.randn(3, 2, 2), 'aux32i': np.arange(6, dtype=np.int32).reshape(2, 3)}
    prefix = str(tmp_path / 'model-11')
    T.write_bundle(prefix, arrays)
    assert sorted(os.listdir(str(tmp_path))) == ['checkpoint', 'model-11.data-00000-of-00001', 'model-11.index']
    assert T.latest_checkpoint(str(tmp_path)) == prefix
    back = T.read_bundle(prefix)
    assert set(back) == set(arrays)
    for k, v in arrays.items():
        v = np.asarray(v)
        assert back[k].dtype == v.dtype and back[k].shape == v.shape and np.array_equal(back[k], v), k
    # header: num_shards 1, little endian, version producer 1;  entries: dtype enum, dims, offsets tile the data file
    table = T.read_table(prefix + '.index')
    assert table[b''] == b'\x08\x01\x1a\x02\x08\x01'
    size = os.path.getsize(prefix + '.data-00000-of-00001')
    spans = []
    for k, v in table.items():
        if k:
            e = T._parse_entry(v)
            a = np.asarray(arrays[k.decode()])
            assert e['dtype'] == {np.dtype('float32'): 1, np.dtype('float64'): 2, np.dtype('int32'): 3, np.dtype('int64'): 9}[a.dtype]
            assert tuple(e['shape']) == a.shape and e['size'] == a.nbytes and e['shard_id'] == 0
            spans.append((e['offset'], e['offset'] + e['size']))
    spans.sort()
    assert spans[0][0] == 0 and spans[-1][1] == size and all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
    # a corrupted tensor byte is caught by the entry's crc
    raw = bytearray(open(prefix + '.data-00000-of-00001', 'rb').read())
    raw[5] ^= 0x40
    open(prefix + '.data-00000-of-00001', 'wb').write(bytes(raw))
    with pytest.raises(ValueError, match='checksum'):
        T.read_bundle(prefix)


def test_converter_tf_directions_round_trip_with_tf1_names(tmp_path):
    """convert_checkpoint --to-tf / --from-tf without TensorFlow: variables keep their names, Adam slots get the doubled scope
    prefix TF-1 gives slots created inside the model's variable scope, the power accumulators are numbered in the
    optimisers' creation order (cfl/models/cfl.py:1065-1096)."""
    import torch
    from cfl.bin import convert_checkpoint as C
    from cfl.utils import Saver
    rng = np.random.RandomState(2)
    names = ['CFL/DistEncoder/outputs/fully_connected/V', 'CFL/DistEncoder/outputs/fully_connected/g',
             'CFL/Thresholder/threshold/threshold', 'CFL/Generator/fc1/fully_connected/V', 'CFL/Discriminator/conv/Conv/V']
    shapes = [(64, 8), (8,), (), (26, 32), (4, 4, 3, 32)]
    var = {n: np.asarray(rng.randn(*s), np.float32) for n, s in zip(names, shapes)}
    var[names[2]] = float(var[names[2]])       # the threshold is a python float in a real checkpoint_state() (scalar shape [] in TF)
    state = {'variables': var, 'adam_m': {n: 0.1 * v for n, v in var.items()}, 'adam_v': {n: v * v for n, v in var.items()},
             'beta1_power': 0.9 ** 5, 'beta2_power': 0.999 ** 5, 'global_step': 40, 'name': 'cfl_pcd_linear_ls_8_nc_2_gan_x',
             'gan_powers': {'g': (0.5 ** 3, 0.999 ** 3), 'd': (0.5 ** 3, 0.999 ** 3)}}
    pt, prefix, back = str(tmp_path / 'model-40.pt'), str(tmp_path / 'tf' / 'model-40'), str(tmp_path / 'back.pt')
    os.makedirs(os.path.dirname(prefix))
    torch.save(Saver._plain(state), pt)
    assert C.main(['--to-tf', pt, prefix]) == 0
    from cfl import tf_bundle
    tensors = tf_bundle.read_bundle(prefix)
    assert tensors[names[2]].shape == () and tensors[names[4]].shape == (4, 4, 3, 32)
    for n in names:
        assert n in tensors and 'CFL/' + n + '/Adam' in tensors and 'CFL/' + n + '/Adam_1' in tensors
    # no '_ut' in the model name: the threshold has its own optimiser (created first), then s_optim, post_g, post_d
    assert {k for k in tensors if 'power' in k} == {'CFL/beta%d_power%s' % (b, s) for b in (1, 2) for s in ('', '_1', '_2', '_3')}
    assert np.float32(tensors['CFL/beta1_power_1']) == np.float32(0.9 ** 5) and np.float32(tensors['CFL/beta1_power_2']) == np.float32(0.125)
    assert C.main(['--from-tf', prefix, back, '--step', '40']) == 0
    got = C._load_pt(back)
    for key in ('variables', 'adam_m', 'adam_v'):
        assert set(got[key]) == set(names)
        for n in names:
            assert np.array_equal(np.asarray(got[key][n]).reshape(-1), np.asarray(state[key][n], np.float32).reshape(-1)), (key, n)
    assert abs(got['beta1_power'] - 0.9 ** 5) < 1e-7 and got['global_step'] == 40
    assert abs(got['gan_powers']['g'][0] - 0.125) < 1e-7 and abs(got['gan_powers']['d'][1] - 0.999 ** 3) < 1e-7
    # a reference-written checkpoint also carries ExponentialMovingAverage shadows: dropped on the way in
    tensors['CFL/Mean_3/ExponentialMovingAverage'] = np.float32(0.5)
    assert 'CFL/Mean_3/ExponentialMovingAverage' not in C.from_tf_names(tensors)


def test_reader_parses_a_hand_assembled_table(tmp_path):
    """An SSTable assembled byte by byte from the format description (LevelDB table_format.md / tensorflow/core/lib/io) WITHOUT
    the writer under test: two data blocks (the second with a shared key prefix), an empty metaindex block, an index block with
    one entry per data block, the 48-byte footer.  Pins the reader to the layout rather than to its own writer."""
    def vi(n):
        out = bytearray()
        while n >= 0x80:
            out.append((n & 0x7f) | 0x80)
            n >>= 7
        out.append(n)
        return bytes(out)

    def entry(shared, key_suffix, value):
        return vi(shared) + vi(len(key_suffix)) + vi(len(value)) + key_suffix + value

    def block(entries_bytes, restarts):
        body = entries_bytes + b''.join(struct.pack('<I', r) for r in restarts) + struct.pack('<I', len(restarts))
        return body, body + b'\x00' + struct.pack('<I', T.mask(T.crc32c(b'\x00', T.crc32c(body))))
    # data block 1: "" -> header, "a/b" -> "one"      data block 2: "a/bc" -> "two", "a/bd" -> "three" (shares "a/b")
    b1_body, b1 = block(entry(0, b'', b'\x08\x01') + entry(0, b'a/b', b'one'), [0])
    b2_body, b2 = block(entry(0, b'a/bc', b'two') + entry(3, b'd', b'three'), [0])
    meta_body, meta = block(b'', [0])
    off1, off2, offm = 0, len(b1), len(b1) + len(b2)
    idx_entries = entry(0, b'a/b', vi(off1) + vi(len(b1_body))) + entry(0, b'a/bd', vi(off2) + vi(len(b2_body)))
    r2 = len(entry(0, b'a/b', vi(off1) + vi(len(b1_body))))
    idx_body, idx = block(idx_entries, [0, r2])
    offi = offm + len(meta)
    footer = vi(offm) + vi(len(meta_body)) + vi(offi) + vi(len(idx_body))
    footer += b'\x00' * (40 - len(footer)) + struct.pack('<II', 0x8b80fb57, 0xdb477524)     # magic: low word first
    path = str(tmp_path / 'hand.index')
    open(path, 'wb').write(b1 + b2 + meta + idx + footer)
    got = T.read_table(path)
    assert list(got.items()) == [(b'', b'\x08\x01'), (b'a/b', b'one'), (b'a/bc', b'two'), (b'a/bd', b'three')]
