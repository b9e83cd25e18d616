"""Reader (and minimal writer) for TensorFlow checkpoint-V2 "tensor bundles", without TensorFlow.

The reference restores its weights with `tf.train.Saver.restore` from
`models/hmr_noS5.ckpt-642561` and `models/hmmr_model.ckpt-1119816`
(src/evaluation/tester.py:92-116, src/config.py:27-35).  A checkpoint `<prefix>` is

    <prefix>.index                   an SSTable (LevelDB table format) mapping
                                     "" -> BundleHeaderProto, <variable name> -> BundleEntryProto
    <prefix>.data-0000k-of-0000n     raw little-endian tensor bytes

This module parses that format directly (varints, prefix-compressed blocks, optional Snappy
block compression, the two protobuf messages) so `Tester(config)` can load the reference's own
checkpoints:  read_checkpoint(prefix) -> {variable name: ndarray}.

`write_checkpoint` emits the same format (one data shard, uncompressed blocks); it exists so
the reader can be round-trip tested here, where TensorFlow is not installable.  Format sources:
tensorflow/core/util/tensor_bundle/{tensor_bundle.cc,tensor_bundle.proto},
tensorflow/core/lib/io/{format.cc,block.cc,table_builder.cc} (restated from the published
format, no TF code is used).
"""
from __future__ import annotations

import os
import struct

import numpy as np

TABLE_MAGIC = 0xDB4775248B80FB57
_DTYPES = {1: np.float32, 2: np.float64, 3: np.int32, 4: np.uint8, 5: np.int16, 6: np.int8,
           9: np.int64, 10: np.bool_, 17: np.uint16, 19: np.float16, 22: np.uint32, 23: np.uint64}
_DTYPE_CODES = {np.dtype(v): k for k, v in _DTYPES.items()}

# --------------------------------------------------------------------------- crc32c (Castagnoli)
_CRC_TABLE = []
for _i in range(256):
    _c = _i
    for _ in range(8):
        _c = (_c >> 1) ^ 0x82F63B78 if _c & 1 else _c >> 1
    _CRC_TABLE.append(_c)


def crc32c(data, crc=0):
    crc ^= 0xFFFFFFFF
    for b in bytes(data):
        crc = _CRC_TABLE[(crc ^ b) & 0xFF] ^ (crc >> 8)
    return crc ^ 0xFFFFFFFF


def _mask(crc):
    return ((((crc >> 15) | (crc << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


# --------------------------------------------------------------------------- varints / protobuf
def _varint(buf, pos):
    out, shift = 0, 0
    while True:
        b = buf[pos]
        pos += 1
        out |= (b & 0x7F) << shift
        if not b & 0x80:
            return out, pos
        shift += 7


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


def _proto_fields(buf):
    """Yield (field number, wire type, value) of one protobuf message."""
    pos, n = 0, len(buf)
    while pos < n:
        key, pos = _varint(buf, pos)
        fno, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = _varint(buf, pos)
        elif wt == 1:
            v = buf[pos:pos + 8]; pos += 8
        elif wt == 2:
            ln, pos = _varint(buf, pos)
            v = buf[pos:pos + ln]; pos += ln
        elif wt == 5:
            v = buf[pos:pos + 4]; pos += 4
        else:
            raise ValueError("unsupported protobuf wire type %d" % wt)
        yield fno, wt, v


def _signed64(v):
    return v - (1 << 64) if v >= (1 << 63) else v


def _parse_entry(buf):
    """BundleEntryProto -> dict(dtype, shape, shard_id, offset, size, crc32c, sliced)."""
    e = {"dtype": 0, "shape": [], "shard_id": 0, "offset": 0, "size": 0, "crc32c": None, "sliced": False}
    for fno, wt, v in _proto_fields(buf):
        if fno == 1:
            e["dtype"] = v
        elif fno == 2:                       # TensorShapeProto { repeated Dim dim = 2 { int64 size = 1 } }
            for f2, _, v2 in _proto_fields(v):
                if f2 == 2:
                    size = 0
                    for f3, _, v3 in _proto_fields(v2):
                        if f3 == 1:
                            size = _signed64(v3)
                    e["shape"].append(size)
        elif fno == 3:
            e["shard_id"] = v
        elif fno == 4:
            e["offset"] = v
        elif fno == 5:
            e["size"] = v
        elif fno == 6:
            e["crc32c"] = struct.unpack("<I", v)[0]
        elif fno == 7:
            e["sliced"] = True
    return e


# --------------------------------------------------------------------------- snappy (block type 1)
def snappy_decompress(src):
    n, pos = _varint(src, 0)
    out = bytearray()
    while pos < len(src):
        tag = src[pos]; pos += 1
        kind = tag & 3
        if kind == 0:
            ln = tag >> 2
            if ln >= 60:
                nb = ln - 59
                ln = int.from_bytes(src[pos:pos + nb], "little"); pos += nb
            ln += 1
            out += src[pos:pos + ln]; pos += ln
            continue
        if kind == 1:
            ln = ((tag >> 2) & 7) + 4
            off = ((tag >> 5) << 8) | src[pos]; pos += 1
        elif kind == 2:
            ln = (tag >> 2) + 1
            off = src[pos] | (src[pos + 1] << 8); pos += 2
        else:
            ln = (tag >> 2) + 1
            off = int.from_bytes(src[pos:pos + 4], "little"); pos += 4
        if off == 0 or off > len(out):
            raise ValueError("corrupt snappy stream")
        for _ in range(ln):                  # may overlap its own output
            out.append(out[-off])
    if len(out) != n:
        raise ValueError("snappy length mismatch")
    return bytes(out)


# --------------------------------------------------------------------------- SSTable
def _read_block(data, offset, size, verify=True):
    contents = data[offset:offset + size]
    btype = data[offset + size]
    if verify:
        stored = struct.unpack("<I", data[offset + size + 1:offset + size + 5])[0]
        if _mask(crc32c(data[offset:offset + size + 1])) != stored:
            raise ValueError("checkpoint index: block checksum mismatch at offset %d" % offset)
    if btype == 1:
        contents = snappy_decompress(contents)
    elif btype != 0:
        raise ValueError("checkpoint index: unknown block compression %d" % btype)
    return contents


def _block_entries(block):
    """(key, value) pairs of one prefix-compressed block."""
    num_restarts = struct.unpack("<I", block[-4:])[0]
    limit = len(block) - 4 - 4 * num_restarts
    pos, key = 0, b""
    while pos < limit:
        shared, pos = _varint(block, pos)
        non_shared, pos = _varint(block, pos)
        vlen, pos = _varint(block, pos)
        key = key[:shared] + bytes(block[pos:pos + non_shared]); pos += non_shared
        yield key, bytes(block[pos:pos + vlen]); pos += vlen


def read_index(path, verify=True):
    """<prefix>.index -> (header dict, {name: entry dict})."""
    data = open(path, "rb").read()
    if len(data) < 48 or struct.unpack("<Q", data[-8:])[0] != TABLE_MAGIC:
        raise ValueError("%s is not a TensorFlow checkpoint-V2 index (bad table magic)" % path)
    footer = data[-48:]
    pos = 0
    _, pos = _varint(footer, pos); _, pos = _varint(footer, pos)          # metaindex handle (unused)
    ioff, pos = _varint(footer, pos); isize, pos = _varint(footer, pos)   # index handle
    entries, header = {}, {}
    for _, handle in _block_entries(_read_block(data, ioff, isize, verify)):
        boff, p2 = _varint(handle, 0)
        bsize, _ = _varint(handle, p2)
        for key, value in _block_entries(_read_block(data, boff, bsize, verify)):
            if key == b"":
                for fno, _, v in _proto_fields(value):    # BundleHeaderProto
                    if fno == 1:
                        header["num_shards"] = v
                    elif fno == 2:
                        header["endianness"] = v
                continue
            entries[key.decode("utf-8")] = _parse_entry(value)
    header.setdefault("num_shards", 1)
    if header.get("endianness", 0) != 0:
        raise NotImplementedError("big-endian checkpoints are not supported")
    return header, entries


def is_checkpoint(prefix):
    return os.path.exists(str(prefix) + ".index")


def read_checkpoint(prefix, names=None, verify_data=False):
    """{variable name: ndarray} of checkpoint `<prefix>` (what Saver.restore would assign)."""
    header, entries = read_index(str(prefix) + ".index")
    n = header["num_shards"]
    shards = {}
    out = {}
    for name, e in entries.items():
        if names is not None and name not in names:
            continue
        if e["sliced"]:
            raise NotImplementedError("partitioned variable %s" % name)
        if e["dtype"] not in _DTYPES:
            continue                          # e.g. string tensors (not part of the model)
        sid = e["shard_id"]
        if sid not in shards:
            shards[sid] = np.memmap("%s.data-%05d-of-%05d" % (prefix, sid, n), dtype=np.uint8, mode="r")
        raw = shards[sid][e["offset"]:e["offset"] + e["size"]]
        if verify_data and e["crc32c"] is not None and _mask(crc32c(raw.tobytes())) != e["crc32c"]:
            raise ValueError("checksum mismatch in tensor %s" % name)
        dt = np.dtype(_DTYPES[e["dtype"]])
        arr = np.frombuffer(raw.tobytes(), dtype=dt)
        if int(np.prod(e["shape"], dtype=np.int64)) != arr.size:
            raise ValueError("tensor %s: %d elements on disk, shape %s" % (name, arr.size, e["shape"]))
        out[name] = arr.reshape(e["shape"])
    return out


# --------------------------------------------------------------------------- writer (tests / conversion)
def _block(entries, restart_interval=16):
    out, restarts, last = bytearray(), [], b""
    for i, (k, v) in enumerate(entries):
        if i % restart_interval == 0:
            restarts.append(len(out)); shared = 0
        else:
            shared = 0
            while shared < min(len(k), len(last)) and k[shared] == last[shared]:
                shared += 1
        out += _put_varint(shared) + _put_varint(len(k) - shared) + _put_varint(len(v)) + k[shared:] + v
        last = k
    if not restarts:
        restarts = [0]
    for r in restarts:
        out += struct.pack("<I", r)
    out += struct.pack("<I", len(restarts))
    return bytes(out)


def _pb(fno, wt, payload):
    key = _put_varint((fno << 3) | wt)
    if wt == 0:
        return key + _put_varint(payload)
    if wt == 2:
        return key + _put_varint(len(payload)) + payload
    return key + payload


def write_checkpoint(prefix, tensors, block_entries=8):
    """Write {name: ndarray} as a one-shard checkpoint-V2 bundle (uncompressed index blocks)."""
    names = sorted(tensors)
    data, recs = bytearray(), []
    for name in names:
        a = np.asarray(tensors[name])                      # (ascontiguousarray would promote 0-d to 1-d)
        raw = np.ascontiguousarray(a).astype(a.dtype.newbyteorder("<"), copy=False).tobytes()
        shape = b"".join(_pb(2, 2, _pb(1, 0, int(d))) for d in a.shape)
        entry = (_pb(1, 0, _DTYPE_CODES[a.dtype]) + _pb(2, 2, shape) + _pb(4, 0, len(data)) +
                 _pb(5, 0, len(raw)) + _pb(6, 5, struct.pack("<I", _mask(crc32c(raw)))))
        recs.append((name.encode("utf-8"), entry))
        data += raw
    with open("%s.data-00000-of-00001" % prefix, "wb") as f:
        f.write(bytes(data))
    header = _pb(1, 0, 1) + _pb(3, 2, _pb(1, 0, 1))           # num_shards = 1, version.producer = 1
    recs = [(b"", header)] + recs                             # "" sorts first
    out, index = bytearray(), []

    def emit(contents):
        off = len(out)
        out.extend(contents + b"\x00")
        out.extend(struct.pack("<I", _mask(crc32c(contents + b"\x00"))))
        return _put_varint(off) + _put_varint(len(contents))

    for i in range(0, len(recs), block_entries):
        chunk = recs[i:i + block_entries]
        index.append((chunk[-1][0], emit(_block(chunk))))
    meta = emit(_block([]))
    idx = emit(_block(index, restart_interval=1))
    footer = meta + idx
    footer += b"\x00" * (40 - len(footer)) + struct.pack("<Q", TABLE_MAGIC)
    out.extend(footer)
    with open("%s.index" % prefix, "wb") as f:
        f.write(bytes(out))
