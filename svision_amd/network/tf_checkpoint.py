"""Reader (and writer, for tests/benchmarks) of TF1 ``Saver`` V2 checkpoints -- the ``-m`` boundary.

The reference restores its CNN with ``tf.train.Saver().restore(sess, model_path)``
(src/network/predict.py:181-184) from a checkpoint *prefix* (README: ``svision-cnn-model.ckpt``
-> ``.index``, ``.data-00000-of-00001``, ``.meta``).  TensorFlow is not available here, so the
"tensor bundle" layout is decoded directly:

* ``{prefix}.index`` -- a LevelDB-style SSTable: blocks of prefix-compressed entries
  (varint32 shared, varint32 non_shared, varint32 value_len, key delta, value) followed by a
  uint32 restart array + count, each block trailed by 1 compression byte and a 4-byte masked
  crc32c; a 48-byte footer holds the metaindex and index BlockHandles and the magic
  0xdb4775248b80fb57.  Key "" maps to BundleHeaderProto, every other key (variable name) to a
  BundleEntryProto {dtype=1, shape=2, shard_id=3, offset=4, size=5, crc32c=6}.
* ``{prefix}.data-XXXXX-of-YYYYY`` -- raw little-endian tensor bytes.

Variables the graph does not name (optimizer slots, global_step) are ignored, as
restore-by-name does.  The real SVision weights are not available offline: the reader is
validated by round trip through :func:`write_checkpoint` (SURVEY 8(a'): unpinned).
"""
import os
import struct

import numpy as np

_MAGIC = 0xdb4775248b80fb57
_DT_FLOAT = 1
_DTYPES = {1: np.float32, 2: np.float64, 3: np.int32, 9: np.int64}


# ---------------------------------------------------------------- varints / protobuf
def _get_varint(buf, p):
    r = s = 0
    while True:
        b = buf[p]
        p += 1
        r |= (b & 0x7F) << s
        if b < 0x80:
            return r, p
        s += 7


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


def _parse_proto(buf):
    """-> {field: [values]} with varints as int, length-delimited as bytes, fixed32/64 as int."""
    out = {}
    p = 0
    while p < len(buf):
        tag, p = _get_varint(buf, p)
        f, wt = tag >> 3, tag & 7
        if wt == 0:
            v, p = _get_varint(buf, p)
        elif wt == 2:
            n, p = _get_varint(buf, p)
            v = bytes(buf[p:p + n])
            p += n
        elif wt == 5:
            v = struct.unpack_from("<I", buf, p)[0]
            p += 4
        elif wt == 1:
            v = struct.unpack_from("<Q", buf, p)[0]
            p += 8
        else:
            raise ValueError("unsupported protobuf wire type %d" % wt)
        out.setdefault(f, []).append(v)
    return out


def _parse_shape(buf):
    dims = []
    for d in _parse_proto(buf).get(2, []):
        dims.append(_parse_proto(d).get(1, [0])[0])
    return tuple(dims)


# ---------------------------------------------------------------- crc32c (Castagnoli), masked as LevelDB does
_CRC_TABLE = None


def _crc32c(data):
    global _CRC_TABLE
    if _CRC_TABLE is None:
        tbl = []
        for i in range(256):
            c = i
            for _ in range(8):
                c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
            tbl.append(c)
        _CRC_TABLE = tbl
    c = 0xFFFFFFFF
    tbl = _CRC_TABLE
    for b in data:
        c = tbl[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def _mask_crc(c):
    return (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF


def _crc32c_fast(buf):
    """crc32c through libsvx.so (slicing-by-8) when it is built, the Python loop otherwise (index blocks are tiny)."""
    from .. import _lib
    try:
        lib = _lib.load()
    except (OSError, _lib.SvxMissing) as exc:                 # library not built (CPU-only tooling); an ABI mismatch still raises
        import logging
        logging.warning("libsvx.so not loadable (%s): crc32c of %d bytes in the Python byte loop", exc, len(buf))
        return _crc32c(bytes(buf))
    a = np.ascontiguousarray(np.frombuffer(buf, np.uint8))
    return int(lib.svx_crc32c(a.ctypes.data, a.size))


# ---------------------------------------------------------------- SSTable
def _read_block(data, off, size):
    if len(data) < off + size + 5:                            # block + type byte + masked crc32c
        raise ValueError("truncated SSTable block")
    if data[off + size] != 0:
        raise ValueError("compressed SSTable blocks are not supported (type %d)" % data[off + size])
    blk = data[off:off + size]
    stored = struct.unpack_from("<I", data, off + size + 1)[0]
    if stored != _mask_crc(_crc32c(data[off:off + size + 1])):
        raise ValueError("corrupt checkpoint index: block checksum mismatch at offset %d" % off)
    n_restarts = struct.unpack_from("<I", blk, size - 4)[0]
    end = size - 4 - 4 * n_restarts
    entries, key, p = [], b"", 0
    while p < end:
        shared, p = _get_varint(blk, p)
        non_shared, p = _get_varint(blk, p)
        vlen, p = _get_varint(blk, p)
        key = key[:shared] + bytes(blk[p:p + non_shared])
        p += non_shared
        entries.append((key, bytes(blk[p:p + vlen])))
        p += vlen
    return entries


def read_index(prefix):
    """-> {variable name: dict(dtype, shape, shard_id, offset, size)} plus the header under ''."""
    with open(prefix + ".index", "rb") as f:
        data = f.read()
    if len(data) < 48 or struct.unpack_from("<Q", data, len(data) - 8)[0] != _MAGIC:
        raise ValueError("%s.index is not a TF V2 checkpoint index (bad magic)" % prefix)
    footer = data[-48:]
    p = 0
    _mo, p = _get_varint(footer, p)
    _ms, p = _get_varint(footer, p)
    io, p = _get_varint(footer, p)
    isz, p = _get_varint(footer, p)
    out = {}
    for _sep, handle in _read_block(data, io, isz):
        bo, q = _get_varint(handle, 0)
        bs, q = _get_varint(handle, q)
        for key, val in _read_block(data, bo, bs):
            msg = _parse_proto(val)
            if key == b"":
                out[""] = {"num_shards": msg.get(1, [1])[0], "endianness": msg.get(2, [0])[0]}
            else:
                out[key.decode()] = {"dtype": msg.get(1, [0])[0], "shape": _parse_shape(msg.get(2, [b""])[0]),
                                     "shard_id": msg.get(3, [0])[0], "offset": msg.get(4, [0])[0],
                                     "size": msg.get(5, [0])[0], "crc32c": msg.get(6, [0])[0]}
    return out


def read_checkpoint(prefix, names=None):
    """-> {name: float32 ndarray} for the variables in ``names`` (default: every float tensor)."""
    if not os.path.exists(prefix + ".index"):
        raise FileNotFoundError("checkpoint prefix %r: %s.index not found (pass the prefix, e.g. "
                                "svision-cnn-model.ckpt, not one of the three files)" % (prefix, prefix))
    index = read_index(prefix)
    header = index.pop("", {"num_shards": 1, "endianness": 0})
    if header.get("endianness", 0) != 0:
        raise ValueError("big-endian checkpoints are not supported")
    shards = {}
    out = {}
    for name, e in index.items():
        if names is not None and name not in names:
            continue
        if e["dtype"] not in _DTYPES:
            continue
        sid = e["shard_id"]
        if sid not in shards:
            path = "%s.data-%05d-of-%05d" % (prefix, sid, header["num_shards"])
            shards[sid] = np.memmap(path, dtype=np.uint8, mode="r")
        if e["offset"] + e["size"] > shards[sid].size:
            raise ValueError("checkpoint shard %d is truncated: tensor %s needs bytes %d..%d of %d"
                             % (sid, name, e["offset"], e["offset"] + e["size"], shards[sid].size))
        raw = shards[sid][e["offset"]:e["offset"] + e["size"]].tobytes()
        if e.get("crc32c") and e["crc32c"] != _mask_crc(_crc32c_fast(raw)):      # 0 = not recorded (own writer without crc_tensors)
            raise ValueError("corrupt checkpoint: checksum mismatch in tensor %s" % name)
        arr = np.frombuffer(raw, dtype=_DTYPES[e["dtype"]]).reshape(e["shape"])
        out[name] = arr
    if names is not None:
        missing = [n for n in names if n not in out]
        if missing:
            raise KeyError("checkpoint %s lacks tensors %s" % (prefix, missing))   # TF: NotFoundError
    return out


def _proto_field(field, wt, payload):
    return _put_varint((field << 3) | wt) + payload


def write_checkpoint(prefix, tensors, crc_tensors=False):
    """Write float32 tensors as a single-shard V2 checkpoint readable by :func:`read_checkpoint`
    (same layout TensorFlow's BundleWriter emits: sorted keys, uncompressed blocks)."""
    names = sorted(tensors)
    entries = []
    header = _proto_field(1, 0, _put_varint(1)) + _proto_field(3, 2, _put_varint(2) + _proto_field(1, 0, _put_varint(1)))
    entries.append((b"", header))
    off = 0
    with open("%s.data-00000-of-00001" % prefix, "wb") as f:
        for name in names:
            arr = np.asarray(tensors[name], dtype=np.float32, order="C")
            raw = arr.tobytes()
            f.write(raw)
            shape = b"".join(_proto_field(2, 2, (lambda d: _put_varint(len(d)) + d)(_proto_field(1, 0, _put_varint(int(s)))))
                             for s in arr.shape)
            crc = _mask_crc(_crc32c(raw)) if crc_tensors else 0
            msg = (_proto_field(1, 0, _put_varint(_DT_FLOAT)) + _proto_field(2, 2, _put_varint(len(shape)) + shape)
                   + (_proto_field(4, 0, _put_varint(off)) if off else b"") + _proto_field(5, 0, _put_varint(len(raw)))
                   + _proto_field(6, 5, struct.pack("<I", crc)))
            entries.append((name.encode(), msg))
            off += len(raw)

    def build_block(items):
        body = bytearray()
        restarts = []
        prev = b""
        for i, (k, v) in enumerate(items):
            if i % 16 == 0:
                restarts.append(len(body))
                shared = 0
            else:
                shared = 0
                while shared < min(len(prev), len(k)) and prev[shared] == k[shared]:
                    shared += 1
            body += _put_varint(shared) + _put_varint(len(k) - shared) + _put_varint(len(v)) + k[shared:] + v
            prev = k
        if not restarts:
            restarts = [0]
        for r in restarts:
            body += struct.pack("<I", r)
        body += struct.pack("<I", len(restarts))
        return bytes(body)

    out = bytearray()

    def emit(block):
        o = len(out)
        out.extend(block)
        out.extend(b"\x00" + struct.pack("<I", _mask_crc(_crc32c(block + b"\x00"))))
        return _put_varint(o) + _put_varint(len(block))

    # a few data blocks to exercise the index block
    handles = []
    for i in range(0, len(entries), 6):
        chunk = entries[i:i + 6]
        handles.append((chunk[-1][0] + b"\x00", emit(build_block(chunk))))
    meta = emit(build_block([]))
    index = emit(build_block(handles))
    footer = meta + index
    footer += b"\x00" * (40 - len(footer)) + struct.pack("<Q", _MAGIC)
    out.extend(footer)
    with open(prefix + ".index", "wb") as f:
        f.write(bytes(out))
