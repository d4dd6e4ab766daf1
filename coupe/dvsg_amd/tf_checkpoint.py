"""TensorFlow checkpoint files without TensorFlow: what `StabNet.init_vars` (model.py:125-154) needs.

The reference's training entry point initialises localizationNet from the slim ImageNet checkpoint
`./pretrained/resnet_v1_50.ckpt` through `slim.assign_from_checkpoint_fn(..., ignore_missing_vars=False)`, restoring every
slim model variable of `stabNet/localizationNet/` EXCEPT those under `.../resnet_v1_50/conv1` (the checkpoint's root conv
has 3 input channels, the model's 21; model.py:126,131-137) under the name with the `stabNet/localizationNet/` prefix
removed (:140).  `init_from_slim_checkpoint` does the same on a dict of arrays in the reference's naming
(`weights.make_synthetic_weights` / a `.npz` of ckpt_manager.py), so an evaluation-only user can start from the ImageNet
trunk exactly as the trainer would.

Both on-disk formats TF 1.x reads are implemented, from their published definitions (TensorFlow is not installed here and
the reference ships no checkpoint: PARITY UNPINNED -- the only tests are round trips through the writer below and
hand-checked byte layouts):

* V2 "tensor bundle" (`<prefix>.index` + `<prefix>.data-00000-of-0000N`; tensorflow/core/util/tensor_bundle): the index is
  an SSTable (tensorflow/core/lib/io/table: LevelDB's table format -- prefix-compressed key/value blocks with restart
  arrays, each followed by a 1-byte compression type + masked CRC32C, a metaindex block, an index block and a 48-byte
  footer ending in the magic 0xdb4775248b80fb57) whose key "" holds a BundleHeaderProto and whose other keys are variable
  names with BundleEntryProto values (dtype, shape, shard_id, offset, size, crc32c); the data files hold the raw
  little-endian tensor bytes.
* V1 (a single file such as the 2016 slim model zoo's `resnet_v1_50.ckpt`; tensorflow/core/util/tensor_slice_writer +
  saved_tensor_slice.proto): the same SSTable container; key "" holds SavedTensorSlices{meta}, every other key (an
  ordered-code encoding of the tensor name and slice) holds SavedTensorSlices{data{name, slice, TensorProto}} with the
  values inline.

Blocks may be Snappy-compressed (TF's table builder compresses when it saves >= 12.5 %): a decompressor is included.
Only what inference needs is supported: float32 / float16 / float64 / int32 / int64 tensors, full (unsliced) variables.
"""
import os
import struct

import numpy as np

TABLE_MAGIC = 0xdb4775248b80fb57
_DTYPES = {1: np.float32, 2: np.float64, 3: np.int32, 9: np.int64, 19: np.float16}   # tensorflow/core/framework/types.proto
_DT_OF = {np.dtype(v): k for k, v in _DTYPES.items()}


# ------------------------------------------------------------------------------------------------ small codecs
def _varint(buf, pos):
    shift = result = 0
    while True:
        b = buf[pos]
        pos += 1
        result |= (b & 0x7F) << shift
        if not b & 0x80:
            return result, pos
        shift += 7
        if shift > 63:
            raise ValueError("varint too long")


def _put_varint(v):
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _crc32c_table():
    t = []
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
        t.append(c)
    return t


_CRC_T = _crc32c_table()


def crc32c(data, crc=0):
    """CRC-32C (Castagnoli), the checksum of TF's tables and bundle entries (byte-wise: fine for index blocks; tensor
    payloads are only checked up to `_VERIFY_DATA_LIMIT` bytes unless verify="all")."""
    crc ^= 0xFFFFFFFF
    for b in bytes(data):
        crc = _CRC_T[(crc ^ b) & 0xFF] ^ (crc >> 8)
    return crc ^ 0xFFFFFFFF


_VERIFY_DATA_LIMIT = 1 << 20


def masked_crc(data):
    """crc32c::Mask: rotate right by 15 and add a constant (stored after every table block)."""
    c = crc32c(data)
    return (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF


def snappy_decompress(buf):
    """Raw Snappy block format (format_description.txt): varint uncompressed length, then literal / copy elements."""
    n, pos = _varint(buf, 0)
    out = bytearray()
    while pos < len(buf):
        tag = buf[pos]
        pos += 1
        kind = tag & 3
        if kind == 0:                                   # literal
            ln = tag >> 2
            if ln >= 60:
                nb = ln - 59
                ln = int.from_bytes(buf[pos:pos + nb], "little")
                pos += nb
            ln += 1
            out += buf[pos:pos + ln]
            pos += ln
            continue
        if kind == 1:                                   # copy, 1-byte offset
            ln = ((tag >> 2) & 7) + 4
            off = ((tag >> 5) << 8) | buf[pos]
            pos += 1
        elif kind == 2:                                 # copy, 2-byte offset
            ln = (tag >> 2) + 1
            off = int.from_bytes(buf[pos:pos + 2], "little")
            pos += 2
        else:                                           # copy, 4-byte offset
            ln = (tag >> 2) + 1
            off = int.from_bytes(buf[pos:pos + 4], "little")
            pos += 4
        if off == 0 or off > len(out):
            raise ValueError("snappy: bad copy offset")
        for _ in range(ln):                             # copies may overlap their own output
            out.append(out[-off])
    if len(out) != n:
        raise ValueError("snappy: length mismatch (%d != %d)" % (len(out), n))
    return bytes(out)


def _proto_fields(buf):
    """Yield (field number, wire type, value) of one protobuf message; length-delimited values as bytes."""
    pos = 0
    while pos < len(buf):
        key, pos = _varint(buf, pos)
        field, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = _varint(buf, pos)
        elif wt == 1:
            v = buf[pos:pos + 8]
            pos += 8
        elif wt == 2:
            ln, pos = _varint(buf, pos)
            v = buf[pos:pos + ln]
            pos += ln
        elif wt == 5:
            v = buf[pos:pos + 4]
            pos += 4
        else:
            raise ValueError("protobuf wire type %d not supported" % wt)
        yield field, wt, v


def _shape_of(buf):
    """TensorShapeProto: repeated Dim dim = 2 { int64 size = 1 }."""
    dims = []
    for f, _, v in _proto_fields(buf):
        if f == 2:
            size = 0
            for g, _, w in _proto_fields(v):
                if g == 1:
                    size = w
            dims.append(size)
    return tuple(dims)


def _shape_proto(shape):
    out = b""
    for d in shape:
        dim = b"\x08" + _put_varint(int(d))
        out += b"\x12" + _put_varint(len(dim)) + dim
    return out


# ------------------------------------------------------------------------------------------------ SSTable
def _read_block(data, offset, size, verify=True):
    raw = data[offset:offset + size]
    ctype = data[offset + size]
    # the checksum is a byte-wise pure-Python CRC-32C: blocks beyond _VERIFY_DATA_LIMIT (the data blocks of a V1 file,
    # which hold the tensors themselves -- ~100 MB for resnet_v1_50) are only checked with verify="all"
    if verify and (verify == "all" or size <= _VERIFY_DATA_LIMIT):
        want = struct.unpack("<I", data[offset + size + 1:offset + size + 5])[0]
        if masked_crc(data[offset:offset + size + 1]) != want:
            raise ValueError("table block at %d: checksum mismatch" % offset)
    if ctype == 0:
        return raw
    if ctype == 1:
        return snappy_decompress(raw)
    raise ValueError("table block at %d: unknown compression type %d" % (offset, ctype))


def _block_entries(block):
    n_restarts = struct.unpack("<I", block[-4:])[0]
    end = len(block) - 4 - 4 * n_restarts
    pos, key = 0, b""
    while pos < end:
        shared, pos = _varint(block, pos)
        non_shared, pos = _varint(block, pos)
        vlen, pos = _varint(block, pos)
        key = key[:shared] + block[pos:pos + non_shared]
        pos += non_shared
        yield key, block[pos:pos + vlen]
        pos += vlen


def read_table(path, verify=True):
    """All (key, value) pairs of a TF / LevelDB table file, in key order."""
    with open(path, "rb") as f:
        data = f.read()
    if len(data) < 48 or struct.unpack("<Q", data[-8:])[0] != TABLE_MAGIC:
        raise ValueError("%s is not a TensorFlow table file (bad magic)" % path)
    footer = data[-48:]
    _, pos = _varint(footer, 0)            # metaindex handle: offset
    _, pos = _varint(footer, pos)          #                   size
    ioff, pos = _varint(footer, pos)
    isize, pos = _varint(footer, pos)
    out = []
    for _, handle in _block_entries(_read_block(data, ioff, isize, verify)):
        boff, p = _varint(handle, 0)
        bsize, _ = _varint(handle, p)
        out.extend(_block_entries(_read_block(data, boff, bsize, verify)))
    return out


def write_table(path, items, block_size=4096, restart_interval=16):
    """Minimal table writer (uncompressed blocks) for the round-trip tests and for producing fixtures."""
    items = sorted(items)
    chunks, index = [], []

    def emit(block_items):
        buf, restarts, last = bytearray(), [], b""
        for i, (k, v) in enumerate(block_items):
            shared = 0
            if i % restart_interval == 0:
                restarts.append(len(buf))
            else:
                while shared < min(len(k), len(last)) and k[shared] == last[shared]:
                    shared += 1
            buf += _put_varint(shared) + _put_varint(len(k) - shared) + _put_varint(len(v)) + k[shared:] + v
            last = k
        for r in restarts:
            buf += struct.pack("<I", r)
        buf += struct.pack("<I", len(restarts))
        return bytes(buf)

    offset = 0

    def add(block):
        nonlocal offset
        trailer = b"\x00" + struct.pack("<I", masked_crc(block + b"\x00"))
        chunks.append(block + trailer)
        handle = _put_varint(offset) + _put_varint(len(block))
        offset += len(block) + 5
        return handle

    cur, cur_bytes = [], 0
    for k, v in items:
        cur.append((k, v))
        cur_bytes += len(k) + len(v)
        if cur_bytes >= block_size:
            index.append((cur[-1][0], add(emit(cur))))
            cur, cur_bytes = [], 0
    if cur:
        index.append((cur[-1][0], add(emit(cur))))
    meta = add(emit([]))
    idx = add(emit(index))
    footer = (meta + idx).ljust(40, b"\x00") + struct.pack("<Q", TABLE_MAGIC)
    with open(path, "wb") as f:
        for c in chunks:
            f.write(c)
        f.write(footer)


# ------------------------------------------------------------------------------------------------ V2 bundles
def _read_bundle(prefix, verify):
    entries = read_table(prefix + ".index", verify)
    if not entries or entries[0][0] != b"":
        raise ValueError("%s.index has no bundle header" % prefix)
    num_shards, little = 1, True
    for f, _, v in _proto_fields(entries[0][1]):         # BundleHeaderProto: num_shards = 1, endianness = 2, version = 3
        if f == 1:
            num_shards = v
        elif f == 2:
            little = v == 0
    if not little:
        raise ValueError("big-endian bundles are not supported")
    shards = {}
    out = {}
    for key, val in entries[1:]:
        dtype = shard = offset = size = 0
        shape, crc, sliced = (), None, False
        for f, _, v in _proto_fields(val):               # BundleEntryProto
            if f == 1:
                dtype = v
            elif f == 2:
                shape = _shape_of(v)
            elif f == 3:
                shard = v
            elif f == 4:
                offset = v
            elif f == 5:
                size = v
            elif f == 6:
                crc = struct.unpack("<I", v)[0]
            elif f == 7:
                sliced = True
        name = key.decode("utf-8")
        if sliced:
            raise ValueError("%s: partitioned (sliced) variables are not supported" % name)
        if dtype not in _DTYPES:
            continue                                     # strings / resources: nothing the model restores
        if shard not in shards:
            shards[shard] = np.memmap("%s.data-%05d-of-%05d" % (prefix, shard, num_shards), dtype=np.uint8, mode="r")
        raw = bytes(shards[shard][offset:offset + size])
        if verify and crc is not None and (verify == "all" or len(raw) <= _VERIFY_DATA_LIMIT):
            c = crc32c(raw)
            if (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF != crc:
                raise ValueError("%s: data checksum mismatch" % name)
        arr = np.frombuffer(raw, dtype=np.dtype(_DTYPES[dtype]).newbyteorder("<")).reshape(shape)
        out[name] = arr.astype(_DTYPES[dtype])
    return out


def write_bundle(prefix, arrays):
    """Write {name: ndarray} as a one-shard V2 checkpoint (`prefix.index`, `prefix.data-00000-of-00001`)."""
    items = [(b"", b"\x08\x01" + b"\x1a\x02\x08\x01")]   # num_shards = 1, version { producer = 1 }
    offset = 0
    with open(prefix + ".data-00000-of-00001", "wb") as f:
        for name in sorted(arrays):
            a = np.asarray(arrays[name]).copy(order="C")       # (ascontiguousarray would turn a scalar into shape (1,))
            raw = a.astype(a.dtype.newbyteorder("<")).tobytes()
            f.write(raw)
            shape = _shape_proto(a.shape)
            entry = (b"\x08" + _put_varint(_DT_OF[a.dtype]) + b"\x12" + _put_varint(len(shape)) + shape +
                     (b"\x20" + _put_varint(offset) if offset else b"") + b"\x28" + _put_varint(len(raw)))
            if len(raw) <= _VERIFY_DATA_LIMIT:   # (this fixture writer leaves the checksum of big tensors out: pure-Python CRC)
                c = crc32c(raw)
                entry += b"\x35" + struct.pack("<I", (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF)
            items.append((name.encode("utf-8"), entry))
            offset += len(raw)
    write_table(prefix + ".index", items)


# ------------------------------------------------------------------------------------------------ V1 single file
def _tensor_proto(buf):
    """TensorProto -> ndarray: dtype = 1, tensor_shape = 2, tensor_content = 4, float_val = 5, double_val = 6,
    int_val = 7, int64_val = 10, half_val = 13 (packed or repeated)."""
    dtype, shape, content = 0, (), None
    vals = {5: [], 6: [], 7: [], 10: [], 13: []}
    packed = {5: [], 6: []}        # packed float_val / double_val runs stay NumPy arrays (one Python object per RUN, not per weight)
    for f, wt, v in _proto_fields(buf):
        if f == 1:
            dtype = v
        elif f == 2:
            shape = _shape_of(v)
        elif f == 4:
            content = v
        elif f in vals:
            if wt == 2:                                   # packed
                if f == 5:
                    packed[5].append(np.frombuffer(v, "<f4"))
                elif f == 6:
                    packed[6].append(np.frombuffer(v, "<f8"))
                else:
                    p = 0
                    while p < len(v):
                        x, p = _varint(v, p)
                        vals[f].append(x)
            elif wt == 5:
                vals[f].append(struct.unpack("<f", v)[0])
            elif wt == 1:
                vals[f].append(struct.unpack("<d", v)[0])
            else:
                vals[f].append(v)
    if dtype not in _DTYPES:
        return None
    np_dt = np.dtype(_DTYPES[dtype])
    n = int(np.prod(shape)) if shape else 1
    if content is not None and len(content):
        return np.frombuffer(content, np_dt.newbyteorder("<")).reshape(shape).astype(np_dt)
    for f in (5, 6):               # repeated (unpacked) values first, then the packed runs, as they were appended
        if packed[f]:
            vals[f] = np.concatenate([np.asarray(vals[f], dtype=packed[f][0].dtype)] + packed[f])
    src = {1: vals[5], 2: vals[6], 3: vals[7], 9: vals[10], 19: vals[13]}[dtype]
    if dtype == 19:
        a = np.array(src, dtype=np.uint16).view(np.float16)
    else:
        a = np.array(src, dtype=np_dt)
    if a.size == 1 and n > 1:
        a = np.full(n, a[0], dtype=np_dt)                 # TensorProto's "one value repeated" shorthand
    return a.reshape(shape)


def _read_v1(path, verify):
    out = {}
    for key, val in read_table(path, verify):
        if key == b"":
            continue                                      # SavedTensorSlices { meta }
        for f, _, v in _proto_fields(val):                # SavedTensorSlices: meta = 1, data = 2
            if f != 2:
                continue
            name, tensor, extents = None, None, []
            for g, _, w in _proto_fields(v):              # SavedSlice: name = 1, slice = 2, data = 3
                if g == 1:
                    name = w.decode("utf-8")
                elif g == 2:
                    for h, _, x in _proto_fields(w):      # TensorSliceProto: repeated Extent extent = 1
                        if h == 1:
                            extents.append(list(_proto_fields(x)))
                elif g == 3:
                    tensor = _tensor_proto(w)
            if name is None or tensor is None:
                continue
            if any(e for e in extents):                   # a non-empty extent = a partial slice
                raise ValueError("%s: partitioned (sliced) variables are not supported" % name)
            out[name] = tensor
    return out


def write_v1(path, arrays):
    """Write {name: ndarray} as a V1 single-file checkpoint (full slices, tensor_content)."""
    items = [(b"", b"\x0a\x00")]
    for name in sorted(arrays):
        a = np.asarray(arrays[name]).copy(order="C")
        shape = _shape_proto(a.shape)
        raw = a.astype(a.dtype.newbyteorder("<")).tobytes()
        tp = b"\x08" + _put_varint(_DT_OF[a.dtype]) + b"\x12" + _put_varint(len(shape)) + shape + b"\x22" + _put_varint(len(raw)) + raw
        extents = b"".join(b"\x0a\x00" for _ in a.shape)              # one empty Extent (= the full range) per dimension
        nb = name.encode("utf-8")
        ss = (b"\x0a" + _put_varint(len(nb)) + nb + b"\x12" + _put_varint(len(extents)) + extents +
              b"\x1a" + _put_varint(len(tp)) + tp)
        # the real key is an ordered-code encoding of (name, slice); readers only need it unique and sorted
        items.append((b"\x00" + nb + b"\x00\x01", b"\x12" + _put_varint(len(ss)) + ss))
    write_table(path, items)


# ------------------------------------------------------------------------------------------------ public
def load_checkpoint(path, verify=True):
    """{variable name: ndarray} from a TF checkpoint: `path` is a V2 prefix (`path.index` exists) or a V1 file.
    verify: True checks the CRC-32C of every table block and tensor payload of up to 1 MiB (the checksum is a byte-wise
    pure-Python loop; in a V1 file the tensors sit INSIDE table blocks, so its large data blocks -- ~100 MB for
    resnet_v1_50 -- go unchecked, like a bundle's large payloads); "all" checks everything (minutes on such a file);
    False checks nothing."""
    if os.path.isfile(path + ".index"):
        return _read_bundle(path, verify)
    if os.path.isfile(path):
        return _read_v1(path, verify)
    raise FileNotFoundError("no TensorFlow checkpoint at %s (neither %s.index nor the file itself)" % (path, path))


def init_from_slim_checkpoint(weights, ckpt_path, scope="stabNet/localizationNet/", model="resnet_v1_50",
                              exclude=("conv1",), verify=True):
    """model.py:125-154 `init_vars` on arrays: every `<scope><model>/...` conv / BatchNorm array of `weights` -- except
    those under `<scope><model>/<exclude>` (the root conv: 21 input channels here, 3 in the ImageNet checkpoint) -- is
    replaced by the checkpoint's `<model>/...` variable.  As with `ignore_missing_vars=False` (:149), a variable the
    checkpoint lacks, or one of another shape, is an error.  Dense layers (`df/dense*`: tensorlayer variables, not slim
    model variables) are left alone.  Returns a new dict in `weights`' own key style (with or without ':0')."""
    ckpt = load_checkpoint(ckpt_path, verify)
    out = dict(weights)
    restored = 0
    for key in weights:
        name = key[:-2] if key.endswith(":0") else key
        if not name.startswith(scope + model + "/"):
            continue
        if any(name.startswith(scope + model + "/" + e) for e in exclude):     # :131-137
            continue
        src = name[len(scope):]                                                # :140
        if src not in ckpt:
            raise KeyError("checkpoint %s has no variable %s (ignore_missing_vars=False, model.py:149)" % (ckpt_path, src))
        a = np.asarray(ckpt[src], dtype=np.float32)
        if a.shape != np.shape(weights[key]):
            raise ValueError("%s: checkpoint shape %s, model shape %s" % (src, a.shape, np.shape(weights[key])))
        out[key] = a
        restored += 1
    if restored == 0:
        raise ValueError("nothing to restore: no key of `weights` starts with %s%s/" % (scope, model))
    return out
