"""Test-side WRITER of TensorFlow checkpoint files (V1 tensor-slice table and
V2 tensor bundle), restated from the published formats independently of the
reader in wavenet/tf_checkpoint.py: LevelDB-style table (data blocks with
prefix-compressed keys and restart points every 16 entries, 5-byte trailers
with a masked CRC-32C, index block, empty metaindex block, 48-byte footer),
`SavedTensorSlices` / `BundleEntryProto` messages written field by field.
Test infrastructure only: TensorFlow is absent here, so these files stand in
for the ones `tf.train.Saver` writes (train.py:104-114)."""
import struct

import numpy as np

MAGIC = 0xdb4775248b80fb57
DT = {np.dtype('float32'): 1, np.dtype('float64'): 2, np.dtype('int32'): 3,
      np.dtype('int64'): 9}


def varint(v):
    out = bytearray()
    v &= (1 << 64) - 1
    while True:
        b = v & 0x7f
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def field(no, wt, payload):
    if wt == 0:
        return varint((no << 3) | 0) + varint(payload)
    if wt == 2:
        return varint((no << 3) | 2) + varint(len(payload)) + payload
    if wt == 5:
        return varint((no << 3) | 5) + payload
    raise ValueError(wt)


def crc32c(data):
    tab = []
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ 0x82f63b78 if c & 1 else c >> 1
        tab.append(c)
    c = 0xffffffff
    for b in data:
        c = tab[(c ^ b) & 0xff] ^ (c >> 8)
    return c ^ 0xffffffff


def mask(crc):
    return (((crc >> 15) | (crc << 17)) + 0xa282ead8) & 0xffffffff


def snappy_literal(data):
    """A valid Snappy stream that stores `data` as literals only."""
    out = bytearray(varint(len(data)))
    pos = 0
    while pos < len(data):
        chunk = data[pos:pos + 60000]
        n = len(chunk) - 1
        if n < 60:
            out.append(n << 2)
        elif n < 256:
            out += bytes([60 << 2, n])
        else:
            out += bytes([61 << 2]) + struct.pack('<H', n)
        out += chunk
        pos += len(chunk)
    return bytes(out)


def build_block(entries, restart_interval=16):
    out, restarts, last = bytearray(), [], b''
    for i, (k, v) in enumerate(entries):
        shared = 0
        if i % restart_interval == 0:
            restarts.append(len(out))
        else:
            while shared < min(len(last), len(k)) and last[shared] == k[shared]:
                shared += 1
        out += varint(shared) + varint(len(k) - shared) + varint(len(v))
        out += k[shared:] + v
        last = k
    if not restarts:
        restarts = [0]
    for r in restarts:
        out += struct.pack('<I', r)
    out += struct.pack('<I', len(restarts))
    return bytes(out)


def write_table(path, entries, block_bytes=4096, snappy=False):
    """entries: [(key bytes, value bytes)], sorted by key."""
    assert [k for k, _ in entries] == sorted(k for k, _ in entries)
    f = bytearray()

    def emit(block):
        ctype = 0
        if snappy:
            block, ctype = snappy_literal(block), 1
        off = len(f)
        f.extend(block)
        f.append(ctype)
        f.extend(struct.pack('<I', mask(crc32c(bytes(block) + bytes([ctype])))))
        return varint(off) + varint(len(block))

    index, cur, size = [], [], 0
    for k, v in entries:
        cur.append((k, v))
        size += len(k) + len(v)
        if size >= block_bytes:
            index.append((cur[-1][0], emit(build_block(cur))))
            cur, size = [], 0
    if cur:
        index.append((cur[-1][0], emit(build_block(cur))))
    meta = emit(build_block([]))
    idx = emit(build_block(index, restart_interval=1))
    footer = meta + idx
    footer += b'\0' * (40 - len(footer)) + struct.pack('<Q', MAGIC)
    f.extend(footer)
    with open(path, 'wb') as out:
        out.write(bytes(f))


def shape_proto(shape):
    return b''.join(field(2, 2, field(1, 0, int(d))) for d in shape)


def write_v1(path, tensors, raw_tensors=None, **kw):
    """tensors: {name: ndarray}; one full slice per tensor, typed repeated
    values (float_val / double_val / int_val / int64_val), as TF <= 0.11.
    raw_tensors: {name: (dtype code, shape, value-field bytes)} written as
    given (a constant stored as ONE value, a string tensor ...)."""
    meta = b''
    entries = []
    for name, (dt, shape, vals) in sorted((raw_tensors or {}).items()):
        full = b''.join(field(1, 2, b'') for _ in shape)
        meta += field(1, 2, field(1, 2, name.encode()) + field(2, 2, shape_proto(shape)) +
                      field(3, 0, dt) + field(4, 2, full))
        tensor = field(1, 0, dt) + field(2, 2, shape_proto(shape)) + vals
        data = field(1, 2, name.encode()) + field(2, 2, full) + field(3, 2, tensor)
        entries.append((b'\x00' + name.encode() + b'\x00\x01', field(2, 2, data)))
    for name, a in sorted(tensors.items()):
        a = np.require(a, requirements='C')
        dt = DT[a.dtype]
        full = b''.join(field(1, 2, b'') for _ in a.shape)   # Extent{} = whole dimension
        meta += field(1, 2, field(1, 2, name.encode()) + field(2, 2, shape_proto(a.shape)) +
                      field(3, 0, dt) + field(4, 2, full))
        if dt == 1:
            vals = field(5, 2, a.astype('<f4').tobytes())
        elif dt == 2:
            vals = field(6, 2, a.astype('<f8').tobytes())
        elif dt == 3:
            vals = field(7, 2, b''.join(varint(int(x)) for x in a.reshape(-1)))
        else:
            vals = field(10, 2, b''.join(varint(int(x)) for x in a.reshape(-1)))
        tensor = field(1, 0, dt) + field(2, 2, shape_proto(a.shape)) + vals
        data = field(1, 2, name.encode()) + field(2, 2, full) + field(3, 2, tensor)
        # (real keys are an ordered-code of name + slice; any unique non-empty
        # sorted key serves a reader that takes the name from the value)
        entries.append((b'\x00' + name.encode() + b'\x00\x01', field(2, 2, data)))
    head = field(1, 2, meta + field(2, 2, field(1, 0, 7)))       # versions.producer
    write_table(path, [(b'', head)] + sorted(entries), **kw)


def write_v2(prefix, tensors, **kw):
    """prefix.index + prefix.data-00000-of-00001 (TF >= 0.12)."""
    data = bytearray()
    entries = [(b'', field(1, 0, 1) + field(3, 2, field(1, 0, 1)))]   # num_shards, version
    for name, a in sorted(tensors.items()):
        a = np.require(a, requirements='C')
        raw = a.astype(a.dtype.newbyteorder('<')).tobytes()
        e = field(1, 0, DT[a.dtype]) + field(2, 2, shape_proto(a.shape))
        if len(data):
            e += field(4, 0, len(data))
        e += field(5, 0, len(raw)) + field(6, 5, struct.pack('<I', mask(crc32c(raw))))
        entries.append((name.encode(), e))
        data += raw
    write_table(prefix + '.index', entries, **kw)
    with open(prefix + '.data-00000-of-00001', 'wb') as f:
        f.write(bytes(data))
