"""Independent assemblers of the two on-disk formats the asset importers read, written from the
PUBLISHED layouts and sharing no code with `human_dynamics_amd/tf_checkpoint.py` (neither its writer nor its helpers)
or with Python 3's pickler:

* TensorFlow checkpoint-V2 "tensor bundle": a LevelDB-format table (`<prefix>.index`) + `.data-0000k-of-0000n` shards.
  Layout sources: LevelDB `doc/table_format.md` (block = entries + restart array + restart count; entry = varint32
  shared / non_shared / value_length + key delta + value; block trailer = 1 type byte + masked crc32c(contents + type);
  footer = metaindex handle + index handle, zero-padded to 40 bytes, + magic 0xdb4775248b80fb57 little-endian) and
  TensorFlow's `tensor_bundle.proto` / `tensor_shape.proto` / `versions.proto` (field numbers below).  The protobuf
  messages are serialised by the REAL `google.protobuf` runtime from descriptors built here, Snappy blocks are
  compressed by the REAL Snappy inside pyarrow, and the checksum is a bit-at-a-time CRC-32C (no table).
  What a TF-1.8 `Saver` writes and this reproduces: header entry under the empty key, keys sorted bytewise, restart
  interval 16, Snappy block compression when it saves >= 12.5 %, shortened separator keys in the index block, an empty
  metaindex block, tensors spread over several data shards.

* SMPL model pickles: Python-2 protocol-2 streams holding `chumpy.ch.Ch` objects (NEWOBJ + BUILD with the state dict
  chumpy's `__getstate__` returns), numpy arrays reduced the Python-2 way (`numpy.core.multiarray._reconstruct`, raw
  bytes as BINSTRING -- readable under Python 3 only with encoding='latin1') and `scipy.sparse.csc.csc_matrix`
  regressors (`copy_reg._reconstructor` + BUILD of the instance `__dict__`).  `Py2Pickle` emits exactly those opcodes.

Test infrastructure only.
"""
from __future__ import annotations

import struct

import numpy as np

TABLE_MAGIC = 0xDB4775248B80FB57
# tensorflow/core/framework/types.proto
DT_FLOAT, DT_DOUBLE, DT_INT32, DT_INT64 = 1, 2, 3, 9
_DT = {np.dtype(np.float32): DT_FLOAT, np.dtype(np.float64): DT_DOUBLE, np.dtype(np.int32): DT_INT32,
       np.dtype(np.int64): DT_INT64}


# --------------------------------------------------------------------------------------------------------- checksums
def crc32c_bitwise(data, crc=0):
    """CRC-32C (Castagnoli, reflected polynomial 0x82F63B78), one bit at a time."""
    crc ^= 0xFFFFFFFF
    for byte in bytes(data):
        crc ^= byte
        for _ in range(8):
            crc = (crc >> 1) ^ (0x82F63B78 & -(crc & 1))
    return crc ^ 0xFFFFFFFF


def masked_crc(data):
    """LevelDB / TF `crc32c::Mask`: rotate right by 15 and add a constant."""
    c = crc32c_bitwise(data)
    return (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF


def varint(v):
    out = bytearray()
    while v >= 0x80:
        out.append((v & 0x7F) | 0x80)
        v >>= 7
    out.append(v)
    return bytes(out)


# --------------------------------------------------------------------------------------------------------- protobuf
def bundle_messages():
    """(BundleHeaderProto, BundleEntryProto) classes built with google.protobuf from the published field numbers."""
    from google.protobuf import descriptor_pb2, descriptor_pool, message_factory
    F = descriptor_pb2.FieldDescriptorProto
    fd = descriptor_pb2.FileDescriptorProto()
    fd.name, fd.package, fd.syntax = "hmmr_test_bundle.proto", "hmmr_test_tf", "proto3"

    def add(msg, name, number, ftype, label=F.LABEL_OPTIONAL, type_name=None):
        f = msg.field.add()
        f.name, f.number, f.type, f.label = name, number, ftype, label
        if type_name:
            f.type_name = type_name

    shape = fd.message_type.add(); shape.name = "TensorShapeProto"
    dim = shape.nested_type.add(); dim.name = "Dim"
    add(dim, "size", 1, F.TYPE_INT64); add(dim, "name", 2, F.TYPE_STRING)
    add(shape, "dim", 2, F.TYPE_MESSAGE, F.LABEL_REPEATED, ".hmmr_test_tf.TensorShapeProto.Dim")
    add(shape, "unknown_rank", 3, F.TYPE_BOOL)
    ver = fd.message_type.add(); ver.name = "VersionDef"
    add(ver, "producer", 1, F.TYPE_INT32); add(ver, "min_consumer", 2, F.TYPE_INT32)
    add(ver, "bad_consumers", 3, F.TYPE_INT32, F.LABEL_REPEATED)
    hdr = fd.message_type.add(); hdr.name = "BundleHeaderProto"
    add(hdr, "num_shards", 1, F.TYPE_INT32); add(hdr, "endianness", 2, F.TYPE_INT32)      # enum LITTLE = 0, BIG = 1
    add(hdr, "version", 3, F.TYPE_MESSAGE, type_name=".hmmr_test_tf.VersionDef")
    ent = fd.message_type.add(); ent.name = "BundleEntryProto"
    add(ent, "dtype", 1, F.TYPE_INT32)                                                     # enum DataType
    add(ent, "shape", 2, F.TYPE_MESSAGE, type_name=".hmmr_test_tf.TensorShapeProto")
    add(ent, "shard_id", 3, F.TYPE_INT32); add(ent, "offset", 4, F.TYPE_INT64); add(ent, "size", 5, F.TYPE_INT64)
    add(ent, "crc32c", 6, F.TYPE_FIXED32)
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    get = getattr(message_factory, "GetMessageClass", None)
    if get is None:                                                                        # older protobuf runtimes
        fac = message_factory.MessageFactory(pool)
        get = fac.GetPrototype
    return (get(pool.FindMessageTypeByName("hmmr_test_tf.BundleHeaderProto")),
            get(pool.FindMessageTypeByName("hmmr_test_tf.BundleEntryProto")))


# --------------------------------------------------------------------------------------------------------- the table
def snappy_real(raw):
    import pyarrow as pa
    return pa.Codec("snappy").compress(raw, asbytes=True)


class TableBuilder(object):
    """LevelDB table builder as TF's `table::TableBuilder` drives it (block_restart_interval 16)."""

    def __init__(self, block_size=4096, compress=True, restart_interval=16):
        self.out = bytearray()
        self.block_size, self.compress, self.ri = block_size, compress, restart_interval
        self.index = []                       # (last key of the block, next block's first key or None, handle bytes)
        self._reset()
        self.compressed_blocks = 0

    def _reset(self):
        self.buf, self.restarts, self.count, self.last = bytearray(), [0], 0, b""

    def add(self, key, value):
        if self.buf and len(self.buf) + len(key) + len(value) > self.block_size:
            self._flush(next_key=key)
        shared = 0
        if self.count and self.count % self.ri == 0:
            self.restarts.append(len(self.buf))
        elif self.count:
            while shared < min(len(key), len(self.last)) and key[shared] == self.last[shared]:
                shared += 1
        self.buf += varint(shared) + varint(len(key) - shared) + varint(len(value)) + key[shared:] + value
        self.last, self.count = key, self.count + 1

    def _write_raw_block(self, contents, allow_compress):
        btype = 0
        if allow_compress and self.compress:
            z = snappy_real(bytes(contents))
            if len(z) < len(contents) - len(contents) // 8:       # table_builder.cc: keep it only if it saves >= 12.5 %
                contents, btype = z, 1
                self.compressed_blocks += 1
        off = len(self.out)
        self.out += contents
        self.out += bytes([btype]) + struct.pack("<I", masked_crc(bytes(contents) + bytes([btype])))
        return varint(off) + varint(len(contents))

    @staticmethod
    def _finish_block(buf, restarts):
        return bytes(buf) + b"".join(struct.pack("<I", r) for r in restarts) + struct.pack("<I", len(restarts))

    def _flush(self, next_key=None):
        handle = self._write_raw_block(self._finish_block(self.buf, self.restarts), True)
        self.index.append((self.last, next_key, handle))
        self._reset()

    @staticmethod
    def _separator(a, b):
        """BytewiseComparator::FindShortestSeparator: a short key k with a <= k < b."""
        n = 0
        while n < min(len(a), len(b)) and a[n] == b[n]:
            n += 1
        if n < min(len(a), len(b)) and a[n] < 0xFF and a[n] + 1 < b[n]:
            return a[:n] + bytes([a[n] + 1])
        return a

    @staticmethod
    def _successor(a):
        for i, c in enumerate(a):
            if c != 0xFF:
                return a[:i] + bytes([c + 1])
        return a

    def finish(self):
        if self.buf:
            self._flush()
        meta = self._write_raw_block(self._finish_block(b"", [0]), False)
        ib, ir, last = bytearray(), [], b""
        for i, (a, b, handle) in enumerate(self.index):            # index block: restart interval 1
            key = self._separator(a, b) if b is not None else self._successor(a)
            ir.append(len(ib))
            ib += varint(0) + varint(len(key)) + varint(len(handle)) + key + handle
        idx = self._write_raw_block(self._finish_block(ib, ir or [0]), True)
        footer = meta + idx
        footer += b"\x00" * (40 - len(footer)) + struct.pack("<Q", TABLE_MAGIC)
        self.out += footer
        return bytes(self.out)


def write_bundle(prefix, tensors, num_shards=2, block_size=4096, compress=True):
    """{name: ndarray} -> `<prefix>.index` + `<prefix>.data-0000k-of-0000n`; returns the number of Snappy blocks."""
    Header, Entry = bundle_messages()
    shards = [bytearray() for _ in range(num_shards)]
    tb = TableBuilder(block_size=block_size, compress=compress)
    h = Header()
    h.num_shards, h.endianness = num_shards, 0
    h.version.producer, h.version.min_consumer = 1, 0
    tb.add(b"", h.SerializeToString())
    for i, name in enumerate(sorted(tensors, key=lambda s: s.encode("utf-8"))):
        a = np.asarray(tensors[name])
        raw = a.astype(a.dtype.newbyteorder("<")).tobytes(order="C")
        sid = i % num_shards
        e = Entry()
        e.dtype = _DT[a.dtype]
        for d in a.shape:
            e.shape.dim.add().size = int(d)
        e.shard_id, e.offset, e.size, e.crc32c = sid, len(shards[sid]), len(raw), masked_crc(raw)
        shards[sid] += raw
        tb.add(name.encode("utf-8"), e.SerializeToString())
    with open(prefix + ".index", "wb") as f:
        f.write(tb.finish())
    for k, s in enumerate(shards):
        with open("%s.data-%05d-of-%05d" % (prefix, k, num_shards), "wb") as f:
            f.write(bytes(s))
    return tb.compressed_blocks


# --------------------------------------------------------------------------------------------------------- py2 pickles
class Py2Pickle(object):
    """Emits a Python-2 protocol-2 pickle stream (opcodes per Lib/pickletools.py): str -> SHORT_BINSTRING / BINSTRING
    (bytes-as-str, what Python 3 reads back only with encoding='latin1'), new-style instances -> NEWOBJ + BUILD,
    classic reconstructions -> GLOBAL copy_reg._reconstructor + REDUCE + BUILD, numpy arrays the way numpy's py2
    `__reduce__` does."""

    def __init__(self):
        self.b = bytearray(b"\x80\x02")               # PROTO 2

    def done(self):
        return bytes(self.b) + b"."                   # STOP

    # -- atoms
    def s(self, text):
        raw = text if isinstance(text, bytes) else text.encode("latin1")
        self.b += (b"U" + bytes([len(raw)]) if len(raw) < 256 else b"T" + struct.pack("<i", len(raw))) + raw

    def i(self, v):
        if 0 <= v < 256:
            self.b += b"K" + bytes([v])
        elif 0 <= v < 65536:
            self.b += b"M" + struct.pack("<H", v)
        else:
            self.b += b"J" + struct.pack("<i", v)

    def none(self):
        self.b += b"N"

    def boolean(self, v):
        self.b += b"\x88" if v else b"\x89"

    def glob(self, module, name):
        self.b += b"c" + module.encode() + b"\n" + name.encode() + b"\n"

    def tup(self, items):
        self.b += b"("                                # MARK
        for it in items:
            self.any(it)
        self.b += b"t"                                # TUPLE

    def dict_(self, d):
        self.b += b"}"                                # EMPTY_DICT
        if d:
            self.b += b"("
            for k, v in d.items():
                self.any(k); self.any(v)
            self.b += b"u"                            # SETITEMS

    # -- numpy, the Python-2 way
    def ndarray(self, a):
        a = np.ascontiguousarray(a)
        self.glob("numpy.core.multiarray", "_reconstruct")
        self.b += b"("
        self.glob("numpy", "ndarray"); self.tup((0,)); self.s("b")
        self.b += b"t" + b"R"                         # TUPLE, REDUCE -> empty array
        self.b += b"("                                # state: (version, shape, dtype, is_fortran, rawdata)
        self.i(1)
        self.tup(tuple(int(d) for d in a.shape))
        self.glob("numpy", "dtype")
        self.b += b"("
        self.s(a.dtype.str[1:]); self.i(0); self.i(1)
        self.b += b"t" + b"R"
        self.b += b"("                                # dtype state
        self.i(3); self.s("<"); self.none(); self.none(); self.none(); self.i(-1); self.i(-1); self.i(0)
        self.b += b"t" + b"b"
        self.boolean(False)
        self.s(a.astype(a.dtype.newbyteorder("<")).tobytes())
        self.b += b"t" + b"b"                         # TUPLE, BUILD

    def chumpy(self, x):
        """chumpy.ch.Ch leaf: `Ch.__new__(Ch)` + `__setstate__(dict)` with the keys chumpy's __getstate__ leaves in."""
        self.glob("chumpy.ch", "Ch")
        self.b += b")" + b"\x81"                      # EMPTY_TUPLE, NEWOBJ
        self.dict_({"_dirty_vars": Set(), "_itr": None, "x": np.asarray(x), "_depends_on_deps": {},
                    "_status": "new"})
        self.b += b"b"

    def csc(self, m):
        """scipy.sparse.csc.csc_matrix as py2 pickles hold it: copy_reg._reconstructor(cls, object, None) + __dict__."""
        self.glob("copy_reg", "_reconstructor")
        self.b += b"("
        self.glob("scipy.sparse.csc", "csc_matrix"); self.glob("__builtin__", "object"); self.none()
        self.b += b"t" + b"R"
        self.dict_({"format": "csc", "_shape": tuple(int(d) for d in m.shape), "indptr": np.asarray(m.indptr, np.int32),
                    "indices": np.asarray(m.indices, np.int32), "maxprint": 50, "data": np.asarray(m.data, np.float64)})
        self.b += b"b"

    def any(self, v):
        if v is None:
            self.none()
        elif isinstance(v, bool):
            self.boolean(v)
        elif isinstance(v, (int, np.integer)):
            self.i(int(v))
        elif isinstance(v, (str, bytes)):
            self.s(v)
        elif isinstance(v, tuple):
            self.tup(v)
        elif isinstance(v, dict):
            self.dict_(v)
        elif isinstance(v, Set):
            self.glob("__builtin__", "set"); self.b += b"(" + b"]" + b"t" + b"R"      # set([])
        elif isinstance(v, Chumpy):
            self.chumpy(v.x)
        elif isinstance(v, np.ndarray):
            self.ndarray(v)
        elif hasattr(v, "indptr"):
            self.csc(v)
        else:
            raise TypeError(type(v))


class Set(object):
    """marker: an empty py2 set"""


class Chumpy(object):
    """marker: wrap the array in a chumpy.ch.Ch record"""
    def __init__(self, x):
        self.x = np.asarray(x)
