"""A small HDF5 reader for `neutral_smpl_meanwjoints.h5` (ref src/evaluation/tester.py:118-123).

The reference initialises its `mean_param` variable from a file that `deepdish.io.save` wrote: PyTables on the HDF5 1.8 file format --
superblock version 0, old-style groups (a symbol table: v1 B-tree + local heap), version-1 object headers, one chunked `CArray` per
ndarray behind PyTables' default filter pipeline, shuffle-less Blosc (registered HDF5 filter 32001, compressor blosclz, Blosc's own
byte shuffle inside the frame).  Neither h5py / PyTables nor the blosc library exist in this image, so this module reads the published
formats directly:

* HDF5 File Format Specification 2.0 (superblock 0-3, object headers 1 and 2, symbol-table and link-message groups, dataspace /
  datatype / layout (compact, contiguous, chunked v3) / filter-pipeline / continuation messages, v1 B-trees of groups and of chunks);
* filters: deflate (1), shuffle (2), fletcher32 (3) and Blosc (32001): the Blosc 1 frame (16-byte header, block starts, per-block
  splits, byte shuffle / memcpy flags) with the blosclz (FastLZ level 1 / 2) and zlib codecs.

Only what a file of small numeric arrays needs: fixed-point and IEEE floating-point datasets of any rank, read whole.
`load(path)` returns {name: ndarray} for the datasets of the root group (nested groups as nested dicts), like `dd.io.load`.
"""
from __future__ import annotations

import struct
import zlib

import numpy as np

SIGNATURE = b"\x89HDF\r\n\x1a\n"
UNDEF = 0xFFFFFFFFFFFFFFFF


class Hdf5Error(ValueError):
    pass


# --------------------------------------------------------------------------- #
# Blosc 1 frames
# --------------------------------------------------------------------------- #
def blosclz_decompress(src, maxout):
    """blosclz (c-blosc 1.x blosclz.c; FastLZ's format): a control byte < 32 starts a run of ctrl + 1 literals; otherwise a match of
    length (ctrl >> 5) + 2 -- a length field of 7 is extended by the following bytes, each 255 continuing -- at distance
    ((ctrl & 31) << 8) + next byte + 1; a distance field of all ones (31, 255) is followed by a 16-bit distance beyond 8191.  The first
    byte is a literal run's control (its top three bits are masked off)."""
    src = memoryview(src)
    n = len(src)
    out = bytearray()
    if n == 0:
        return bytes(out)
    ip = 0
    ctrl = src[ip] & 31
    ip += 1
    loop = True
    while loop:
        if ctrl >= 32:
            length = (ctrl >> 5) - 1
            ofs = (ctrl & 31) << 8
            if length == 7 - 1:
                while True:
                    code = src[ip]; ip += 1
                    length += code
                    if code != 255:
                        break
            code = src[ip]; ip += 1
            ref = len(out) - ofs - code
            if code == 255 and ofs == (31 << 8):
                ofs = (src[ip] << 8) + src[ip + 1]; ip += 2
                ref = len(out) - ofs - 8191
            ref -= 1
            if ref < 0 or len(out) + length + 3 > maxout:
                raise Hdf5Error("blosclz: corrupt stream")
            if ip < n:
                ctrl = src[ip]; ip += 1
            else:
                loop = False
            for _ in range(length + 3):              # (byte by byte: a match may overlap its own output)
                out.append(out[ref]); ref += 1
        else:
            ctrl += 1
            if len(out) + ctrl > maxout or ip + ctrl > n:
                raise Hdf5Error("blosclz: corrupt stream")
            out += src[ip:ip + ctrl]
            ip += ctrl
            loop = ip < n
            if loop:
                ctrl = src[ip]; ip += 1
    return bytes(out)


def _unshuffle(buf, typesize):
    n = len(buf) // typesize
    if typesize <= 1 or n == 0:
        return buf
    body = np.frombuffer(buf[:n * typesize], np.uint8).reshape(typesize, n).T.tobytes()
    return body + buf[n * typesize:]                 # (the leftover bytes of a block are copied unshuffled)


def blosc_decompress(frame):
    """One Blosc 1 frame -> bytes (c-blosc README_HEADER.rst / blosc.c)."""
    frame = bytes(frame)
    if len(frame) < 16:
        raise Hdf5Error("blosc: frame shorter than its header")
    version, versionlz, flags, typesize = frame[0], frame[1], frame[2], frame[3]
    nbytes, blocksize, cbytes = struct.unpack_from("<III", frame, 4)
    if cbytes > len(frame) or version < 1:
        raise Hdf5Error("blosc: bad header")
    if flags & 0x4:
        raise Hdf5Error("blosc: bit-shuffled frames are not supported")
    if flags & 0x2:                                  # memcpyed: the data follows the header as it is
        return frame[16:16 + nbytes]
    codec = flags >> 5                               # 0 blosclz, 1 lz4, 2 snappy, 3 zlib, 4 zstd
    if codec not in (0, 3):
        raise Hdf5Error("blosc: codec %d is not supported (blosclz and zlib are)" % codec)
    dont_split = bool(flags & 0x10)
    nblocks = (nbytes + blocksize - 1) // blocksize if blocksize else 0
    bstarts = struct.unpack_from("<%di" % nblocks, frame, 16)
    out = []
    for b in range(nblocks):
        bsize = blocksize if (b < nblocks - 1 or nbytes % blocksize == 0) else nbytes % blocksize
        leftover = b == nblocks - 1 and nbytes % blocksize != 0
        # a block is split into `typesize` streams (one per byte position) unless the flag forbids it or the block is a leftover
        nsplits = typesize if (not dont_split and 1 < typesize <= 16 and bsize // typesize >= 128 and not leftover) else 1
        neblock = bsize // nsplits
        pos = bstarts[b]
        parts = []
        for _ in range(nsplits):
            csize, = struct.unpack_from("<i", frame, pos)
            pos += 4
            if csize == neblock:
                parts.append(frame[pos:pos + csize])
            elif codec == 0:
                parts.append(blosclz_decompress(frame[pos:pos + csize], neblock))
            else:
                parts.append(zlib.decompress(frame[pos:pos + csize]))
            if len(parts[-1]) != neblock:
                raise Hdf5Error("blosc: a split decompressed to %d bytes, expected %d" % (len(parts[-1]), neblock))
            pos += csize
        block = b"".join(parts)
        if flags & 0x1:
            block = _unshuffle(block, typesize)
        out.append(block)
    data = b"".join(out)
    if len(data) != nbytes:
        raise Hdf5Error("blosc: %d bytes out, header says %d" % (len(data), nbytes))
    return data


# --------------------------------------------------------------------------- #
# HDF5
# --------------------------------------------------------------------------- #
class _File(object):
    def __init__(self, buf):
        self.b = buf
        pos = 0
        while buf[pos:pos + 8] != SIGNATURE:         # the superblock sits at 0, 512, 1024, ...
            pos = 512 if pos == 0 else pos * 2
            if pos + 8 > len(buf):
                raise Hdf5Error("not an HDF5 file")
        self.sb = pos
        ver = buf[pos + 8]
        if ver in (0, 1):
            self.so, self.sl = buf[pos + 13], buf[pos + 14]
            p = pos + 24 + (4 if ver == 1 else 0)
            self.base = self.u(p, self.so)
            p += 4 * self.so                          # base, free-space, end-of-file, driver-info addresses
            # root group symbol table entry: link name offset, object header address, cache type, reserved, scratch
            self.root_oh = self.u(p + self.so, self.so)
        elif ver in (2, 3):
            self.so, self.sl = buf[pos + 9], buf[pos + 10]
            p = pos + 12
            self.base = self.u(p, self.so)
            self.root_oh = self.u(p + 3 * self.so, self.so)
        else:
            raise Hdf5Error("superblock version %d" % ver)

    def u(self, pos, size):
        return int.from_bytes(self.b[pos:pos + size], "little")

    def addr(self, pos):
        return self.u(pos, self.so)

    # ---- object headers -> [(type, bytes)]
    def messages(self, oh):
        b = self.b
        oh += self.base
        out = []
        if b[oh:oh + 4] == b"OHDR":                  # version 2
            flags = b[oh + 5]
            p = oh + 6
            if flags & 0x20:
                p += 16
            if flags & 0x10:
                p += 4
            csz = 1 << (flags & 3)
            size0 = self.u(p, csz)
            p += csz
            chunks = [(p, size0)]
            track = bool(flags & 0x4)
            while chunks:
                p, size = chunks.pop(0)
                end = p + size
                while p + 4 + (2 if track else 0) <= end:
                    mtype, msize, mflags = b[p], self.u(p + 1, 2), b[p + 3]
                    p += 4 + (2 if track else 0)
                    body = bytes(b[p:p + msize])
                    p += msize
                    if mtype == 0x10:                 # continuation: an OCHK block (signature, messages, checksum)
                        a, ln = self.addr_from(body, 0), int.from_bytes(body[self.so:self.so + self.sl], "little")
                        chunks.append((a + self.base + 4, ln - 8))
                    elif mtype != 0:
                        out.append((mtype, body))
            return out
        if b[oh] != 1:
            raise Hdf5Error("object header version %d at %d" % (b[oh], oh))
        nmsg, size = self.u(oh + 2, 2), self.u(oh + 8, 4)
        chunks = [(oh + 16, size)]
        while chunks and nmsg > 0:
            p, size = chunks.pop(0)
            end = p + size
            while p + 8 <= end and nmsg > 0:
                mtype, msize = self.u(p, 2), self.u(p + 2, 2)
                body = bytes(b[p + 8:p + 8 + msize])
                p += 8 + msize
                nmsg -= 1
                if mtype == 0x10:
                    a, ln = self.addr_from(body, 0), int.from_bytes(body[self.so:self.so + self.sl], "little")
                    chunks.append((a + self.base, ln))
                elif mtype != 0:
                    out.append((mtype, body))
        return out

    def addr_from(self, body, pos):
        return int.from_bytes(body[pos:pos + self.so], "little")

    # ---- groups
    def group_links(self, oh):
        """{name: object header address} of a group (old style: symbol table; new style: hard link messages)"""
        links = {}
        for mtype, body in self.messages(oh):
            if mtype == 0x11:                         # symbol table message: B-tree address, local heap address
                btree, heap = self.addr_from(body, 0), self.addr_from(body, self.so)
                hp = heap + self.base
                if self.b[hp:hp + 4] != b"HEAP":
                    raise Hdf5Error("local heap signature")
                data = self.addr(hp + 8 + 2 * self.sl) + self.base
                self._walk_group_tree(btree, data, links)
            elif mtype == 6:                          # link message
                ver, flags = body[0], body[1]
                p = 2
                ltype = 0
                if flags & 0x8:
                    ltype = body[p]; p += 1
                if flags & 0x4:
                    p += 8
                if flags & 0x10:
                    p += 1
                ln_size = 1 << (flags & 3)
                nlen = int.from_bytes(body[p:p + ln_size], "little"); p += ln_size
                name = body[p:p + nlen].decode("utf-8"); p += nlen
                if ltype == 0:
                    links[name] = self.addr_from(body, p)
        return links

    def _walk_group_tree(self, node, heap_data, links):
        b = self.b
        p = node + self.base
        if b[p:p + 4] != b"TREE" or b[p + 4] != 0:
            raise Hdf5Error("group B-tree node")
        level, used = b[p + 5], self.u(p + 6, 2)
        q = p + 8 + 2 * self.so
        for i in range(used):
            q += self.sl                              # key i
            child = self.addr(q)
            q += self.so
            if level > 0:
                self._walk_group_tree(child, heap_data, links)
                continue
            s = child + self.base
            if b[s:s + 4] != b"SNOD":
                raise Hdf5Error("symbol table node")
            nsym = self.u(s + 6, 2)
            e = s + 8
            for _ in range(nsym):
                name_off, oh = self.addr(e), self.addr(e + self.so)
                end = b.index(b"\0", heap_data + name_off)
                links[bytes(b[heap_data + name_off:end]).decode("utf-8")] = oh
                e += 2 * self.so + 4 + 4 + 16

    # ---- datasets
    def read_dataset(self, msgs):
        shape = dtype = layout = None
        filters = []
        for mtype, body in msgs:
            if mtype == 1:                            # dataspace
                ver, rank = body[0], body[1]
                p = 8 if ver == 1 else 4
                shape = tuple(int.from_bytes(body[p + i * self.sl:p + (i + 1) * self.sl], "little") for i in range(rank))
            elif mtype == 3:                          # datatype
                cls, bits0 = body[0] & 0xF, body[1]
                size = int.from_bytes(body[4:8], "little")
                order = ">" if (bits0 & 1) else "<"
                if cls == 0:
                    dtype = np.dtype("%s%s%d" % (order, "i" if (bits0 & 0x8) else "u", size))
                elif cls == 1:
                    dtype = np.dtype("%sf%d" % (order, size))
                else:
                    raise Hdf5Error("datatype class %d is not supported (fixed-point and floating-point are)" % cls)
            elif mtype == 8:                          # data layout
                layout = body
            elif mtype == 0xB:                        # filter pipeline
                ver, nf = body[0], body[1]
                p = 8 if ver == 1 else 2
                for _ in range(nf):
                    fid = int.from_bytes(body[p:p + 2], "little")
                    if ver == 1 or fid >= 256:
                        nlen = int.from_bytes(body[p + 2:p + 4], "little")
                    else:
                        nlen = 0
                    p += 2 + (2 if (ver == 1 or fid >= 256) else 0)
                    fflags, ncd = int.from_bytes(body[p:p + 2], "little"), int.from_bytes(body[p + 2:p + 4], "little")
                    p += 4
                    p += (nlen + 7) // 8 * 8 if ver == 1 else nlen
                    cd = [int.from_bytes(body[p + 4 * i:p + 4 * i + 4], "little") for i in range(ncd)]
                    p += 4 * ncd
                    if ver == 1 and ncd % 2:
                        p += 4
                    filters.append((fid, fflags, cd))
        if shape is None or dtype is None or layout is None:
            raise Hdf5Error("not a dataset (dataspace, datatype and layout messages expected)")
        nbytes = int(np.prod(shape, dtype=np.int64)) * dtype.itemsize
        if layout[0] != 3:
            raise Hdf5Error("data layout message version %d (3 expected)" % layout[0])
        cls = layout[1]
        if cls == 0:                                  # compact
            size = int.from_bytes(layout[2:4], "little")
            raw = layout[4:4 + size]
        elif cls == 1:                                # contiguous
            a = self.addr_from(layout, 2)
            raw = b"\0" * nbytes if a == UNDEF & ((1 << (8 * self.so)) - 1) else bytes(self.b[a + self.base:a + self.base + nbytes])
        elif cls == 2:                                # chunked: v1 B-tree of chunks
            rank1 = layout[2]
            btree = self.addr_from(layout, 3)
            p = 3 + self.so
            cdims = [int.from_bytes(layout[p + 4 * i:p + 4 * i + 4], "little") for i in range(rank1)]
            chunk_shape, elem = tuple(cdims[:-1]), cdims[-1]
            if elem != dtype.itemsize or len(chunk_shape) != len(shape):
                raise Hdf5Error("chunk geometry")
            out = np.zeros(shape, dtype)
            if btree != UNDEF & ((1 << (8 * self.so)) - 1):
                self._walk_chunk_tree(btree, rank1, chunk_shape, dtype, filters, out)
            return out
        else:
            raise Hdf5Error("layout class %d" % cls)
        return np.frombuffer(raw[:nbytes], dtype).reshape(shape).copy()

    def _walk_chunk_tree(self, node, rank1, chunk_shape, dtype, filters, out):
        b = self.b
        p = node + self.base
        if b[p:p + 4] != b"TREE" or b[p + 4] != 1:
            raise Hdf5Error("chunk B-tree node")
        level, used = b[p + 5], self.u(p + 6, 2)
        q = p + 8 + 2 * self.so
        for _ in range(used):
            csize, fmask = self.u(q, 4), self.u(q + 4, 4)
            offs = [self.u(q + 8 + 8 * i, 8) for i in range(rank1)]
            q += 8 + 8 * rank1
            child = self.addr(q)
            q += self.so
            if level > 0:
                self._walk_chunk_tree(child, rank1, chunk_shape, dtype, filters, out)
                continue
            raw = bytes(b[child + self.base:child + self.base + csize])
            for i in reversed(range(len(filters))):   # the pipeline is undone back to front
                if fmask & (1 << i):
                    continue
                fid, _, cd = filters[i]
                if fid == 1:
                    raw = zlib.decompress(raw)
                elif fid == 2:
                    raw = _unshuffle(raw, cd[0] if cd else dtype.itemsize)
                elif fid == 3:
                    raw = raw[:-4]
                elif fid == 32001:
                    raw = blosc_decompress(raw)
                else:
                    raise Hdf5Error("HDF5 filter %d is not supported (deflate, shuffle, fletcher32 and blosc are)" % fid)
            chunk = np.frombuffer(raw[:int(np.prod(chunk_shape, dtype=np.int64)) * dtype.itemsize], dtype).reshape(chunk_shape)
            sl = tuple(slice(o, min(o + c, s)) for o, c, s in zip(offs[:-1], chunk_shape, out.shape))
            out[sl] = chunk[tuple(slice(0, s.stop - s.start) for s in sl)]

    def load_group(self, oh):
        res = {}
        for name, child in sorted(self.group_links(oh).items()):
            msgs = self.messages(child)
            types = {t for t, _ in msgs}
            if 8 in types and 1 in types:
                res[name] = self.read_dataset(msgs)
            elif 0x11 in types or 6 in types or 2 in types or 0xA in types:
                res[name] = self.load_group(child)
        return res


def load(path):
    """{name: ndarray} (nested dicts for nested groups) of an HDF5 file's root group: what `deepdish.io.load` returns for a file that
    `deepdish.io.save` wrote from a dict of ndarrays (ref src/evaluation/tester.py:122)."""
    with open(path, "rb") as fh:
        buf = fh.read()
    f = _File(buf)                                   # (bytes: slices, .index and integer indexing are all that is used)
    return f.load_group(f.root_oh)
