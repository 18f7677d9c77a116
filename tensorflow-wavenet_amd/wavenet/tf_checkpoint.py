"""Read the reference's TensorFlow checkpoints without TensorFlow.

The reference saves / restores its trainable variables with `tf.train.Saver`
(train.py:104-134, 265-269; generate.py:176-182).  Two on-disk formats exist:

  V1  (TF <= 0.11, the reference's TF 0.10): ONE file `model.ckpt-N`, a
      LevelDB-style sorted table (tensorflow/core/lib/io/table*, format.cc)
      whose first entry (key "") is a `SavedTensorSlices` message holding the
      meta data and whose other entries each hold one `SavedSlice` (name,
      slice extents, `TensorProto` with typed repeated values) --
      tensorflow/core/util/saved_tensor_slice.proto, tensor_slice_writer.cc;
  V2  (TF >= 0.12, "tensor bundle"): `prefix.index` -- the same table format,
      key "" = `BundleHeaderProto`, key <name> = `BundleEntryProto` (dtype,
      shape, shard, offset, size) -- plus raw little-endian bytes in
      `prefix.data-SSSSS-of-NNNNN` (tensorflow/core/util/tensor_bundle).

This module restates those published formats: table footer / blocks / prefix-
compressed entries, the five protobuf messages involved (a minimal wire-format
parser, no generated code) and Snappy block decompression.  STATUS: no
TensorFlow and no TensorFlow-written checkpoint is available in this
environment, so the reader is tested against files produced by
`tests/tf_ckpt_writer.py` (an independent restatement of the writer side of
the same specifications) -- self-consistent, NOT pinned against a real file.

Name mapping (`to_state_dict`): checkpoint keys are the variable names of
model.py:120-226.  Bias variables are named `Variable`, `Variable_1`, ... per
scope in the reference's own checkpoints, because model.py:28 passes the
intended name as `trainable`; they are assigned by creation order
(filter_bias, gate_bias, dense_bias, slip_bias per layer; postprocess1_bias,
postprocess2_bias).  Checkpoints written with the intended names load too.
"""
import os
import re
import struct

import numpy as np

TABLE_MAGIC = 0xdb4775248b80fb57

# tensorflow/core/framework/types.proto
_DTYPES = {1: np.float32, 2: np.float64, 3: np.int32, 4: np.uint8,
           5: np.int16, 6: np.int8, 9: np.int64, 10: np.bool_}


# ---------------------------------------------------------------- primitives
def _varint(buf, pos):
    out = shift = 0
    while True:
        b = buf[pos]
        pos += 1
        out |= (b & 0x7f) << shift
        if not b & 0x80:
            return out, pos
        shift += 7
        if shift > 70:
            raise ValueError('malformed varint')


def _fields(buf):
    """(field number, wire type, value) of a serialized protobuf message:
    wire type 0 -> int, 1 -> 8 raw bytes, 2 -> bytes, 5 -> 4 raw bytes."""
    buf = bytes(buf)
    pos, n = 0, len(buf)
    while pos < n:
        key, pos = _varint(buf, pos)
        no, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = _varint(buf, pos)
        elif wt == 1:
            v, pos = buf[pos:pos + 8], pos + 8
        elif wt == 2:
            ln, pos = _varint(buf, pos)
            v, pos = buf[pos:pos + ln], pos + ln
        elif wt == 5:
            v, pos = buf[pos:pos + 4], pos + 4
        else:
            raise ValueError('unsupported protobuf wire type %d' % wt)
        if pos > n:
            raise ValueError('truncated protobuf message')
        yield no, wt, v


def _signed64(v):
    return v - (1 << 64) if v >= (1 << 63) else v


_CRC_TABLE = None


def crc32c(data, crc=0):
    """CRC-32C (Castagnoli), as used by the table trailers."""
    global _CRC_TABLE
    if _CRC_TABLE is None:
        tab = []
        for i in range(256):
            c = i
            for _ in range(8):
                c = (c >> 1) ^ 0x82f63b78 if c & 1 else c >> 1
            tab.append(c)
        _CRC_TABLE = tab
    c = crc ^ 0xffffffff
    for b in bytes(data):
        c = _CRC_TABLE[(c ^ b) & 0xff] ^ (c >> 8)
    return c ^ 0xffffffff


def masked_crc(crc):
    return (((crc >> 15) | (crc << 17)) + 0xa282ead8) & 0xffffffff


def snappy_decompress(data):
    """Raw Snappy block format (length varint, then literal / copy tags)."""
    data = bytes(data)
    total, pos = _varint(data, 0)
    out = bytearray()
    n = len(data)
    while pos < n:
        tag = data[pos]
        pos += 1
        kind = tag & 3
        if kind == 0:                                   # literal
            ln = tag >> 2
            if ln >= 60:
                nb = ln - 59
                ln = int.from_bytes(data[pos:pos + nb], 'little')
                pos += nb
            ln += 1
            out += data[pos:pos + ln]
            pos += ln
            continue
        if kind == 1:
            ln = ((tag >> 2) & 7) + 4
            off = ((tag >> 5) << 8) | data[pos]
            pos += 1
        elif kind == 2:
            ln = (tag >> 2) + 1
            off = int.from_bytes(data[pos:pos + 2], 'little')
            pos += 2
        else:
            ln = (tag >> 2) + 1
            off = int.from_bytes(data[pos:pos + 4], 'little')
            pos += 4
        if off == 0 or off > len(out):
            raise ValueError('malformed snappy copy')
        for _ in range(ln):                             # may overlap itself
            out.append(out[-off])
    if len(out) != total:
        raise ValueError('snappy length mismatch')
    return bytes(out)


# --------------------------------------------------------------------- table
def _read_block(data, offset, size, verify):
    raw = data[offset:offset + size]
    if len(raw) != size or offset + size + 5 > len(data):
        raise ValueError('table block out of range')
    ctype = data[offset + size]
    if verify:
        want, = struct.unpack('<I', data[offset + size + 1:offset + size + 5])
        have = masked_crc(crc32c(data[offset:offset + size + 1]))
        if want != have:
            raise ValueError('table block checksum mismatch')
    if ctype == 0:
        return raw
    if ctype == 1:
        return snappy_decompress(raw)
    raise ValueError('unknown table block compression %d' % ctype)


def _block_entries(block):
    """(key, value) pairs of one block (prefix-compressed keys; the restart
    array at the end is only needed for seeking)."""
    if len(block) < 4:
        raise ValueError('table block too short')
    nrestart, = struct.unpack('<I', block[-4:])
    end = len(block) - 4 - 4 * nrestart
    if end < 0:
        raise ValueError('bad restart array')
    pos, key = 0, b''
    while pos < end:
        shared, pos = _varint(block, pos)
        non_shared, pos = _varint(block, pos)
        vlen, pos = _varint(block, pos)
        if shared > len(key):
            raise ValueError('bad shared key length')
        key = key[:shared] + block[pos:pos + non_shared]
        pos += non_shared
        yield key, block[pos:pos + vlen]
        pos += vlen


def read_table(path, verify_checksums=True):
    """All (key, value) entries of a table file, in key order."""
    with open(path, 'rb') as f:
        data = f.read()
    if len(data) < 48:
        raise ValueError('%s: too short for a table' % path)
    footer = data[-48:]
    magic, = struct.unpack('<Q', footer[40:])
    if magic != TABLE_MAGIC:
        raise ValueError('%s: not a TensorFlow table (bad magic)' % path)
    pos = 0
    _, pos = _varint(footer, pos)          # metaindex handle (unused)
    _, pos = _varint(footer, pos)
    ioff, pos = _varint(footer, pos)
    isize, pos = _varint(footer, pos)
    out = []
    for _, handle in _block_entries(_read_block(data, ioff, isize, verify_checksums)):
        boff, p = _varint(handle, 0)
        bsize, p = _varint(handle, p)
        out.extend(_block_entries(_read_block(data, boff, bsize, verify_checksums)))
    return out


def is_table(path):
    try:
        with open(path, 'rb') as f:
            f.seek(-8, os.SEEK_END)
            return struct.unpack('<Q', f.read(8))[0] == TABLE_MAGIC
    except (OSError, struct.error):
        return False


# ---------------------------------------------------------------- proto bits
def _shape(buf):
    """TensorShapeProto -> list of ints."""
    dims = []
    for no, wt, v in _fields(buf):
        if no == 2 and wt == 2:
            size = 0
            for n2, w2, v2 in _fields(v):
                if n2 == 1 and w2 == 0:
                    size = _signed64(v2)
            dims.append(size)
    return dims


def _extents(buf):
    """TensorSliceProto -> [(start, length or None)] per dimension."""
    out = []
    for no, wt, v in _fields(buf):
        if no == 1 and wt == 2:
            start, length = 0, None
            for n2, w2, v2 in _fields(v):
                if n2 == 1 and w2 == 0:
                    start = _signed64(v2)
                elif n2 == 2 and w2 == 0:
                    length = _signed64(v2)
            out.append((start, length))
    return out


def _tensor_proto(buf):
    """TensorProto -> (dtype code, shape, flat numpy array), or None for a
    dtype that cannot be a model weight (strings, resources ...: skipped, as
    read_v2 skips them)."""
    dtype, shape, content = 0, [], None
    vals = {5: [], 6: [], 7: [], 10: [], 11: []}   # float, double, int, int64, bool
    for no, wt, v in _fields(buf):
        if no == 1 and wt == 0:
            dtype = v
        elif no == 2 and wt == 2:
            shape = _shape(v)
        elif no == 4 and wt == 2:
            content = v
        elif no in vals:
            vals[no].append((wt, v))
    if dtype not in _DTYPES:
        return None
    np_dt = np.dtype(_DTYPES[dtype])
    if content is not None and len(content):
        return dtype, shape, np.frombuffer(content, dtype=np_dt.newbyteorder('<')).astype(np_dt)
    field = {1: 5, 2: 6, 3: 7, 4: 7, 5: 7, 6: 7, 9: 10, 10: 11}[dtype]
    parts = []
    for wt, v in vals[field]:
        if field == 5:
            parts.append(np.frombuffer(v, dtype='<f4'))           # packed or one fixed32
        elif field == 6:
            parts.append(np.frombuffer(v, dtype='<f8'))
        elif wt == 2:                                            # packed varints
            p, items = 0, []
            while p < len(v):
                x, p = _varint(v, p)
                items.append(_signed64(x))
            parts.append(np.asarray(items, dtype=np.int64))
        else:
            parts.append(np.asarray([_signed64(v)], dtype=np.int64))
    flat = np.concatenate(parts) if parts else np.zeros(0)
    return dtype, shape, flat.astype(np_dt)


# ------------------------------------------------------------------- V1 / V2
def read_v1(path, verify_checksums=True):
    """{variable name: ndarray} of a V1 (tensor slice) checkpoint file."""
    entries = read_table(path, verify_checksums)
    shapes, types = {}, {}
    pieces = {}
    for key, value in entries:
        for no, wt, v in _fields(value):
            if no == 1 and wt == 2 and key == b'':            # SavedTensorSliceMeta
                for n2, w2, v2 in _fields(v):
                    if n2 == 1 and w2 == 2:                   # SavedSliceMeta
                        name, shape, typ = None, [], 0
                        for n3, w3, v3 in _fields(v2):
                            if n3 == 1 and w3 == 2:
                                name = v3.decode('utf-8')
                            elif n3 == 2 and w3 == 2:
                                shape = _shape(v3)
                            elif n3 == 3 and w3 == 0:
                                typ = v3
                        if name is not None:
                            shapes[name], types[name] = shape, typ
            elif no == 2 and wt == 2:                         # SavedSlice
                name, ext, tensor = None, [], None
                for n2, w2, v2 in _fields(v):
                    if n2 == 1 and w2 == 2:
                        name = v2.decode('utf-8')
                    elif n2 == 2 and w2 == 2:
                        ext = _extents(v2)
                    elif n2 == 3 and w2 == 2:
                        tensor = _tensor_proto(v2)
                if name is not None and tensor is not None:   # (None: not a weight dtype)
                    pieces.setdefault(name, []).append((ext, tensor))
    out = {}
    for name, plist in pieces.items():
        shape = shapes.get(name)
        if shape is None:
            raise ValueError('checkpoint slice of %r has no meta entry' % name)
        np_dt = _DTYPES[plist[0][1][0]]
        full = np.zeros(shape, dtype=np_dt)
        for ext, (_, _, flat) in plist:
            idx, sub = [], []
            for d, size in enumerate(shape):
                start, length = ext[d] if d < len(ext) else (0, None)
                length = size - start if length is None else length
                idx.append(slice(start, start + length))
                sub.append(length)
            want = int(np.prod(sub, dtype=np.int64))
            if 0 < flat.size < want:
                # TensorProto's typed value lists may be shorter than the
                # tensor: the LAST value stands for the rest (a constant
                # tensor is stored as one value)
                flat = np.concatenate([flat, np.full(want - flat.size, flat[-1],
                                                     dtype=flat.dtype)])
            elif flat.size == 0 and want:
                flat = np.zeros(want, dtype=np_dt)     # (no value at all: zeros)
            if flat.size != want:
                raise ValueError('checkpoint slice of %r: %d values for extents %s'
                                 % (name, flat.size, sub))
            full[tuple(idx)] = flat.reshape(sub)
        out[name] = full
    return out


def read_v2(prefix, verify_checksums=True):
    """{variable name: ndarray} of a V2 (tensor bundle) checkpoint prefix."""
    entries = read_table(prefix + '.index', verify_checksums)
    num_shards = 1
    out, files = {}, {}
    for key, value in entries:
        if key == b'':
            for no, wt, v in _fields(value):
                if no == 1 and wt == 0:
                    num_shards = v
                elif no == 2 and wt == 0 and v != 0:
                    raise ValueError('big-endian tensor bundles are not supported')
            continue
        dtype, shape, shard, offset, size, crc, sliced = 0, [], 0, 0, 0, None, False
        for no, wt, v in _fields(value):
            if no == 1 and wt == 0:
                dtype = v
            elif no == 2 and wt == 2:
                shape = _shape(v)
            elif no == 3 and wt == 0:
                shard = v
            elif no == 4 and wt == 0:
                offset = v
            elif no == 5 and wt == 0:
                size = v
            elif no == 6 and wt == 5:
                crc, = struct.unpack('<I', v)
            elif no == 7:
                sliced = True
        if sliced:
            raise ValueError('partitioned variables are not supported (%r)' % key)
        if dtype not in _DTYPES:
            continue                                   # (strings etc.: not model weights)
        if shard not in files:
            with open('%s.data-%05d-of-%05d' % (prefix, shard, num_shards), 'rb') as f:
                files[shard] = f.read()
        raw = files[shard][offset:offset + size]
        if len(raw) != size:
            raise ValueError('tensor %r runs past its data shard' % key)
        if verify_checksums and crc is not None and masked_crc(crc32c(raw)) != crc:
            raise ValueError('tensor %r: checksum mismatch' % key)
        np_dt = np.dtype(_DTYPES[dtype])
        out[key.decode('utf-8')] = np.frombuffer(raw, dtype=np_dt.newbyteorder('<')) \
            .astype(np_dt).reshape(shape)
    return out


def checkpoint_format(path):
    """'v2' / 'v1' / None for a path as handed to Saver.restore."""
    if os.path.exists(path + '.index') and is_table(path + '.index'):
        return 'v2'
    if os.path.isfile(path) and is_table(path):
        return 'v1'
    return None


def read_checkpoint(path, verify_checksums=True):
    fmt = checkpoint_format(path)
    if fmt == 'v2':
        return read_v2(path, verify_checksums)
    if fmt == 'v1':
        return read_v1(path, verify_checksums)
    raise ValueError('%s is not a TensorFlow checkpoint' % path)


# -------------------------------------------------------------- name mapping
_BIAS_ORDER_LAYER = ['filter_bias', 'gate_bias', 'dense_bias', 'slip_bias']
_BIAS_ORDER_POST = ['postprocess1_bias', 'postprocess2_bias']


def to_state_dict(tensors, variable_names):
    """Map checkpoint tensors onto `variable_names` (the names of
    WaveNetModel.named_variables(), i.e. the reference's intended names).
    Returns ({name: ndarray}, [checkpoint keys that were not used]); raises
    KeyError listing what is missing."""
    tensors = {re.sub(r':0$', '', k): v for k, v in tensors.items()}
    out, used, missing = {}, set(), []
    for name in variable_names:
        src = name if name in tensors else None
        if src is None:
            scope, leaf = name.rsplit('/', 1)
            order = _BIAS_ORDER_LAYER if leaf in _BIAS_ORDER_LAYER else \
                _BIAS_ORDER_POST if leaf in _BIAS_ORDER_POST else None
            if order is not None:
                k = order.index(leaf)
                alias = scope + ('/Variable' if k == 0 else '/Variable_%d' % k)
                if alias in tensors:
                    src = alias
        if src is None:
            missing.append(name)
        else:
            out[name] = tensors[src]
            used.add(src)
    if missing:
        raise KeyError('checkpoint lacks %d variable(s): %s'
                       % (len(missing), ', '.join(missing[:6])))
    return out, sorted(set(tensors) - used)


def load_into(net, path, verify_checksums=True):
    """Restore a WaveNetModel from a TensorFlow checkpoint written by the
    reference (the counterpart of saver.restore, train.py:117-134,
    generate.py:176-182).  Returns the checkpoint keys it did not use."""
    tensors = read_checkpoint(path, verify_checksums)
    sd, unused = to_state_dict(tensors, [n for n, _ in net.named_variables()])
    for n, v in net.named_variables():
        if tuple(sd[n].shape) != tuple(v.shape):
            raise ValueError('%s: checkpoint shape %s, model shape %s'
                             % (n, tuple(sd[n].shape), tuple(v.shape)))
    net.load_state_dict(sd)
    return unused
