"""TensorFlow "tensor bundle" (checkpoint format V2) reader / writer in pure Python -- no TensorFlow needed.

What the reference's ``tf.train.Saver`` writes and reads at cfl/utils.py:465-497 (``saver.save`` / ``saver.restore`` /
``tf.train.get_checkpoint_state``): for a prefix ``model-<step>``

    model-<step>.index                  an SSTable (the LevelDB table format, tensorflow/core/lib/io/table*): key "" ->
                                        BundleHeaderProto, key <variable name> -> BundleEntryProto, keys sorted
    model-<step>.data-00000-of-00001    the tensors' bytes (little-endian, row-major), at the offsets the entries name
    checkpoint                          text proto naming the latest prefix (CheckpointState)

Restated from the published format (tensorflow/core/util/tensor_bundle/tensor_bundle.{h,cc}, tensorflow/core/protobuf/
tensor_bundle.proto, tensorflow/core/lib/io/{format,block_builder,table_builder}.cc = LevelDB's table_format.md):

  table     [data block]* [metaindex block] [index block] [footer 48 B]
  block     entries + uint32 restart offsets + uint32 count; then a 5-byte trailer: compression type (0 = none) and the
            MASKED crc32c of (block bytes + type byte), little-endian
  entry     varint32 shared-key-bytes, varint32 unshared-key-bytes, varint32 value-bytes, key suffix, value; a restart
            point (shared = 0) every 16 entries in data blocks, every entry in the index block
  index     one entry per data block: key >= the block's last key (here: the last key itself), value = BlockHandle
            (varint64 offset, varint64 size -- without the trailer)
  footer    metaindex handle, index handle, zero padding to 40 bytes, magic 0xdb4775248b80fb57 (little-endian)
  masked    crc -> ((crc >> 15 | crc << 17) + 0xa282ead8) mod 2^32   (crc32c = Castagnoli, reflected, init / xor-out ~0)

  BundleHeaderProto  1: num_shards (varint)   2: endianness (LITTLE = 0)   3: VersionDef { 1: producer }
  BundleEntryProto   1: dtype   2: TensorShapeProto { 2: repeated Dim { 1: size } }   3: shard_id   4: offset   5: size
                     6: crc32c (fixed32, MASKED crc32c of the tensor's bytes)   7: slices (partitioned variables: not
                     supported here -- the reference has none)

NOT VERIFIED AGAINST A REAL TENSORFLOW: none is installable in the build image.  The writer and the reader
are tested against each other and against the format's known answers (crc32c test vector, footer magic, block trailers);
a reference-side ``saver.restore`` of a file written here has to be tried wherever a TF-1 install exists.
"""
import os
import struct

import numpy as np

MAGIC = 0xdb4775248b80fb57
MASK_DELTA = 0xa282ead8
RESTART_INTERVAL = 16
BLOCK_SIZE = 262144          # table::Options default used by BundleWriter

# tensorflow/core/framework/types.proto
DT_FLOAT, DT_DOUBLE, DT_INT32, DT_INT64 = 1, 2, 3, 9
_NP_OF = {DT_FLOAT: np.dtype('<f4'), DT_DOUBLE: np.dtype('<f8'), DT_INT32: np.dtype('<i4'), DT_INT64: np.dtype('<i8')}
_DT_OF = {np.dtype('float32'): DT_FLOAT, np.dtype('float64'): DT_DOUBLE, np.dtype('int32'): DT_INT32,
          np.dtype('int64'): DT_INT64}


# ---- crc32c -------------------------------------------------------------------------------------------------------------
def _make_table():
    poly = 0x82f63b78
    tab = []
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ poly if c & 1 else c >> 1
        tab.append(c)
    return tab


_TABLE = _make_table()
_TABLE_NP = np.array(_TABLE, dtype=np.uint32)


def crc32c(data, crc=0):
    """CRC-32C (Castagnoli) of `data` (bytes-like), continuing from `crc`.  Small inputs: a byte loop; large ones: the
    library's host routine when it is loaded (cfl_crc32c), else the same loop (slow, correct)."""
    data = bytes(data)
    if len(data) >= 4096:
        fast = _native_crc()
        if fast is not None:
            return fast(data, len(data), crc)
    c = crc ^ 0xffffffff
    tab = _TABLE
    for b in data:
        c = tab[(c ^ b) & 0xff] ^ (c >> 8)
    return c ^ 0xffffffff


_native = False


def _native_crc():
    global _native
    if _native is False:
        _native = None
        try:
            import ctypes as C
            from . import hipabi
            L = hipabi.lib()
            fn = L.cfl_crc32c
            fn.argtypes = [C.c_char_p, C.c_size_t, C.c_uint32]
            fn.restype = C.c_uint32
            _native = fn
        except Exception:
            _native = None
    return _native


def mask(crc):
    return (((crc >> 15) | (crc << 17)) + MASK_DELTA) & 0xffffffff


def unmask(m):
    rot = (m - MASK_DELTA) & 0xffffffff
    return ((rot >> 17) | (rot << 15)) & 0xffffffff


# ---- varints / protobuf ----------------------------------------------------------------------------------------------
def _varint(n):
    n &= (1 << 64) - 1
    out = bytearray()
    while n >= 0x80:
        out.append((n & 0x7f) | 0x80)
        n >>= 7
    out.append(n)
    return bytes(out)


def _read_varint(buf, pos):
    shift = result = 0
    while True:
        b = buf[pos]
        pos += 1
        result |= (b & 0x7f) << shift
        if not b & 0x80:
            return result, pos
        shift += 7
        if shift > 70:
            raise ValueError('varint too long')


def _pb_fields(buf):
    """(field number, wire type, value) of a serialised protobuf message; length-delimited values as bytes"""
    pos, n = 0, len(buf)
    while pos < n:
        key, pos = _read_varint(buf, pos)
        f, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = _read_varint(buf, pos)
        elif wt == 1:
            v = struct.unpack_from('<Q', buf, pos)[0]
            pos += 8
        elif wt == 2:
            ln, pos = _read_varint(buf, pos)
            v = bytes(buf[pos:pos + ln])
            pos += ln
        elif wt == 5:
            v = struct.unpack_from('<I', buf, pos)[0]
            pos += 4
        else:
            raise ValueError('unsupported protobuf wire type %d' % wt)
        yield f, wt, v


def _pb_varint(field, value):
    return _varint(field << 3) + _varint(value)


def _pb_bytes(field, payload):
    return _varint((field << 3) | 2) + _varint(len(payload)) + payload


def _header_proto():
    version = _pb_varint(1, 1)                        # VersionDef.producer = kTensorBundleVersion (1)
    return _pb_varint(1, 1) + _pb_bytes(3, version)   # num_shards = 1; endianness LITTLE (0) is the default: omitted


def _entry_proto(dtype, shape, offset, size, crc_masked):
    dims = b''.join(_pb_bytes(2, _pb_varint(1, int(d))) for d in shape)
    out = _pb_varint(1, dtype) + _pb_bytes(2, dims)
    if offset:
        out += _pb_varint(4, offset)
    out += _pb_varint(5, size)
    out += _varint((6 << 3) | 5) + struct.pack('<I', crc_masked)
    return out


def _parse_entry(buf):
    e = dict(dtype=0, shape=[], shard_id=0, offset=0, size=0, crc32c=None, slices=0)
    for f, wt, v in _pb_fields(buf):
        if f == 1:
            e['dtype'] = v
        elif f == 2:
            for f2, _, v2 in _pb_fields(v):
                if f2 == 2:
                    size = 0
                    for f3, _, v3 in _pb_fields(v2):
                        if f3 == 1:
                            size = v3 - (1 << 64) if v3 >= (1 << 63) else v3
                    e['shape'].append(size)
                elif f2 == 3 and v2:
                    raise ValueError('tensor of unknown rank in a checkpoint')
        elif f == 3:
            e['shard_id'] = v
        elif f == 4:
            e['offset'] = v
        elif f == 5:
            e['size'] = v
        elif f == 6:
            e['crc32c'] = v
        elif f == 7:
            e['slices'] += 1
    return e


# ---- table (SSTable) -------------------------------------------------------------------------------------------------
class _BlockBuilder(object):
    def __init__(self, restart_interval):
        self.interval = restart_interval
        self.buf = bytearray()
        self.restarts = [0]
        self.counter = 0
        self.last_key = b''
        self.empty = True

    def add(self, key, value):
        shared = 0
        if self.counter < self.interval:
            m = min(len(self.last_key), len(key))
            while shared < m and self.last_key[shared] == key[shared]:
                shared += 1
        else:
            self.restarts.append(len(self.buf))
            self.counter = 0
        self.buf += _varint(shared) + _varint(len(key) - shared) + _varint(len(value)) + key[shared:] + value
        self.last_key = key
        self.counter += 1
        self.empty = False

    def size_estimate(self):
        return len(self.buf) + 4 * len(self.restarts) + 4

    def finish(self):
        return bytes(self.buf) + b''.join(struct.pack('<I', r) for r in self.restarts) + struct.pack('<I', len(self.restarts))


def _block_with_trailer(contents):
    typ = b'\x00'                                     # kNoCompression
    return contents + typ + struct.pack('<I', mask(crc32c(typ, crc32c(contents))))


def write_table(path, items, block_size=BLOCK_SIZE):
    """items: iterable of (key bytes, value bytes) in strictly increasing key order"""
    out = bytearray()
    index = _BlockBuilder(1)
    data = _BlockBuilder(RESTART_INTERVAL)
    prev = None

    def flush():
        nonlocal data
        if data.empty:
            return
        contents = data.finish()
        handle = _varint(len(out)) + _varint(len(contents))
        out.extend(_block_with_trailer(contents))
        index.add(data.last_key, handle)
        data = _BlockBuilder(RESTART_INTERVAL)
    for key, value in items:
        if prev is not None and not key > prev:
            raise ValueError('table keys must be strictly increasing')
        prev = key
        data.add(key, value)
        if data.size_estimate() >= block_size:
            flush()
    flush()
    meta = _BlockBuilder(RESTART_INTERVAL).finish()
    meta_handle = _varint(len(out)) + _varint(len(meta))
    out.extend(_block_with_trailer(meta))
    idx = index.finish()
    index_handle = _varint(len(out)) + _varint(len(idx))
    out.extend(_block_with_trailer(idx))
    footer = meta_handle + index_handle
    footer += b'\x00' * (40 - len(footer))
    footer += struct.pack('<Q', MAGIC)
    out.extend(footer)
    with open(path, 'wb') as f:
        f.write(bytes(out))


def _read_block(buf, offset, size, verify=True):
    contents = buf[offset:offset + size]
    typ = buf[offset + size:offset + size + 1]
    if len(contents) != size or len(typ) != 1:
        raise ValueError('truncated table block')
    if typ != b'\x00':
        raise ValueError('compressed table block (type %d): not supported' % typ[0])
    if verify:
        stored = struct.unpack_from('<I', buf, offset + size + 1)[0]
        if unmask(stored) != crc32c(typ, crc32c(contents)):
            raise ValueError('table block checksum mismatch')
    return contents


def _block_entries(contents):
    n_restarts = struct.unpack_from('<I', contents, len(contents) - 4)[0]
    limit = len(contents) - 4 - 4 * n_restarts
    pos, key = 0, b''
    while pos < limit:
        shared, pos = _read_varint(contents, pos)
        unshared, pos = _read_varint(contents, pos)
        vlen, pos = _read_varint(contents, pos)
        key = key[:shared] + bytes(contents[pos:pos + unshared])
        pos += unshared
        yield key, bytes(contents[pos:pos + vlen])
        pos += vlen


def read_table(path, verify=True):
    """{key bytes: value bytes} of an SSTable, in file order"""
    with open(path, 'rb') as f:
        buf = f.read()
    if len(buf) < 48 or struct.unpack_from('<Q', buf, len(buf) - 8)[0] != MAGIC:
        raise ValueError('%s is not an SSTable (bad magic)' % path)
    footer = buf[len(buf) - 48:]
    pos = 0
    _, pos = _read_varint(footer, pos)           # metaindex handle
    _, pos = _read_varint(footer, pos)
    ioff, pos = _read_varint(footer, pos)
    isize, pos = _read_varint(footer, pos)
    out = {}
    for _, handle in _block_entries(_read_block(buf, ioff, isize, verify)):
        off, p = _read_varint(handle, 0)
        size, p = _read_varint(handle, p)
        for k, v in _block_entries(_read_block(buf, off, size, verify)):
            out[k] = v
    return out


# ---- bundle ----------------------------------------------------------------------------------------------------------
def write_bundle(prefix, arrays, checkpoint_state=True):
    """Write {name: array} as the bundle `prefix`.index + `prefix`.data-00000-of-00001 (float32 / float64 / int32 /
    int64 tensors), plus -- checkpoint_state -- the `checkpoint` file of its directory naming it the latest checkpoint
    (what tf.train.get_checkpoint_state reads at cfl/utils.py:470)."""
    names = sorted(arrays, key=lambda s: s.encode())
    items = [(b'', _header_proto())]
    offset = 0
    with open(prefix + '.data-00000-of-00001', 'wb') as data:
        for name in names:
            a = np.asarray(arrays[name])
            if a.dtype not in _DT_OF:
                raise ValueError('%s: dtype %s cannot be written' % (name, a.dtype))
            raw = np.ascontiguousarray(a.astype(a.dtype.newbyteorder('<'), copy=False)).tobytes()
            data.write(raw)
            items.append((name.encode(), _entry_proto(_DT_OF[a.dtype], a.shape, offset, len(raw), mask(crc32c(raw)))))
            offset += len(raw)
    write_table(prefix + '.index', items)
    if checkpoint_state:
        base = os.path.basename(prefix)
        with open(os.path.join(os.path.dirname(prefix) or '.', 'checkpoint'), 'w') as f:
            f.write('model_checkpoint_path: "%s"\nall_model_checkpoint_paths: "%s"\n' % (base, base))


def read_bundle(prefix, verify=True):
    """{name: array} of the bundle `prefix` (as tf.train.load_checkpoint(prefix).get_tensor would return them)"""
    table = read_table(prefix + '.index', verify)
    if b'' not in table:
        raise ValueError('%s.index has no bundle header' % prefix)
    num_shards, endian = 1, 0
    for f, _, v in _pb_fields(table[b'']):
        if f == 1:
            num_shards = v
        elif f == 2:
            endian = v
    if endian != 0:
        raise ValueError('big-endian bundle: not supported')
    shards = {}
    out = {}
    for key, value in table.items():
        if key == b'':
            continue
        e = _parse_entry(value)
        if e['slices']:
            raise ValueError('%s: partitioned variable (tensor slices): not supported' % key.decode())
        if e['dtype'] not in _NP_OF:
            raise ValueError('%s: dtype enum %d: not supported' % (key.decode(), e['dtype']))
        sid = e['shard_id']
        if sid not in shards:
            with open('%s.data-%05d-of-%05d' % (prefix, sid, num_shards), 'rb') as f:
                shards[sid] = f.read()
        raw = shards[sid][e['offset']:e['offset'] + e['size']]
        if len(raw) != e['size']:
            raise ValueError('%s: data shard too short' % key.decode())
        if verify and e['crc32c'] is not None and unmask(e['crc32c']) != crc32c(raw):
            raise ValueError('%s: tensor checksum mismatch' % key.decode())
        out[key.decode()] = np.frombuffer(raw, dtype=_NP_OF[e['dtype']]).reshape(e['shape']).copy()
    return out


def latest_checkpoint(checkpoint_dir):
    """the prefix named by `checkpoint_dir`/checkpoint (tf.train.latest_checkpoint), or None"""
    try:
        with open(os.path.join(checkpoint_dir, 'checkpoint')) as f:
            for line in f:
                if line.startswith('model_checkpoint_path:'):
                    p = line.split(':', 1)[1].strip().strip('"')
                    return p if os.path.isabs(p) else os.path.join(checkpoint_dir, p)
    except OSError:
        pass
    return None
