"""An HDF5 + Blosc WRITER for the tests of human_dynamics_amd/hdf5_lite.py, written from the published formats and sharing nothing with the
reader: the HDF5 File Format Specification (superblock version 0, version-1 object headers with a continuation block, an old-style root
group = symbol-table message -> v1 B-tree -> symbol-table node + local heap, chunked datasets = layout message version 3 -> v1 B-tree of
chunks, filter-pipeline message version 1) and c-blosc's README_HEADER.rst / blosclz.c (the 16-byte frame header, block starts, splits,
byte shuffle; a greedy FastLZ-format encoder).  It lays a file out the way PyTables / deepdish do for a dict of small ndarrays
(`dd.io.save`: one CArray per array behind the Blosc filter, registered id 32001)."""
import struct

import numpy as np

UNDEF = 0xFFFFFFFFFFFFFFFF


# ---- blosclz (FastLZ format) encoder: greedy, hash-free (quadratic: test sizes only)
def blosclz_compress(data, min_match=3, max_dist=8190):
    data = bytes(data)
    out, lits, i, n = bytearray(), bytearray(), 0, len(data)

    def flush():
        for k in range(0, len(lits), 32):
            run = lits[k:k + 32]
            out.append(len(run) - 1)
            out.extend(run)
        lits.clear()
    if n:
        lits.append(data[0]); i = 1                   # (the stream opens with a literal run)
    while i < n:
        best, bd = 0, 0
        for d in range(1, min(i, max_dist) + 1):
            L = 0
            while i + L < n and data[i + L - d] == data[i + L] and L < 600:
                L += 1
            if L > best:
                best, bd = L, d
        if best >= min_match:
            flush()
            field = min(best - 2, 7)
            out.append((field << 5) | ((bd - 1) >> 8))
            if field == 7:
                ext = best - 9
                while ext >= 255:
                    out.append(255); ext -= 255
                out.append(ext)
            out.append((bd - 1) & 255)
            i += best
        else:
            lits.append(data[i]); i += 1
    flush()
    return bytes(out)


def shuffle(buf, typesize):
    n = len(buf) // typesize
    return np.frombuffer(buf[:n * typesize], np.uint8).reshape(n, typesize).T.tobytes() + buf[n * typesize:]


def blosc_frame(raw, typesize, mode="blosclz", blocksize=None, do_shuffle=True):
    """one Blosc 1 frame of `raw`; mode: 'memcpy' | 'blosclz' | 'zlib'"""
    raw = bytes(raw)
    nbytes = len(raw)
    blocksize = blocksize or nbytes
    if mode == "memcpy":
        return struct.pack("<BBBBIII", 2, 1, 0x2 | (0x1 if do_shuffle else 0), typesize, nbytes, blocksize, 16 + nbytes) + raw
    import zlib
    codec = {"blosclz": 0, "zlib": 3}[mode]
    nblocks = (nbytes + blocksize - 1) // blocksize
    body, bstarts = bytearray(), []
    for b in range(nblocks):
        blk = raw[b * blocksize:(b + 1) * blocksize]
        leftover = len(blk) < blocksize
        if do_shuffle:
            blk = shuffle(blk, typesize)
        nsplits = typesize if (1 < typesize <= 16 and len(blk) // typesize >= 128 and not leftover) else 1
        ne = len(blk) // nsplits
        bstarts.append(16 + 4 * nblocks + len(body))
        for s_ in range(nsplits):
            piece = blk[s_ * ne:(s_ + 1) * ne]
            c = blosclz_compress(piece) if codec == 0 else zlib.compress(piece)
            if len(c) >= len(piece):
                c = piece                             # stored: csize == the split's size
            body += struct.pack("<i", len(c)) + c
    flags = (0x1 if do_shuffle else 0) | (codec << 5)
    cbytes = 16 + 4 * nblocks + len(body)
    return struct.pack("<BBBBIII", 2, 1, flags, typesize, nbytes, blocksize, cbytes) + struct.pack("<%di" % nblocks, *bstarts) + bytes(body)


# ---- HDF5
class Writer(object):
    def __init__(self):
        self.buf = bytearray(b"\0" * 96)               # superblock v0 (56 bytes + the root symbol table entry of 40)

    def alloc(self, data, align=8):
        while len(self.buf) % align:
            self.buf.append(0)
        at = len(self.buf)
        self.buf += data
        return at

    @staticmethod
    def msg(mtype, body):
        body = bytes(body) + b"\0" * (-len(body) % 8)
        return struct.pack("<HHB3x", mtype, len(body), 0) + body

    def object_header(self, msgs, split_after=None):
        """version-1 object header; split_after: put the messages from that index on into a continuation block"""
        if split_after is None:
            body = b"".join(msgs)
            return self.alloc(struct.pack("<BxHII4x", 1, len(msgs), 1, len(body)) + body)
        tail = b"".join(msgs[split_after:])
        tail_at = self.alloc(tail)
        head = b"".join(msgs[:split_after]) + self.msg(0x10, struct.pack("<QQ", tail_at, len(tail)))
        return self.alloc(struct.pack("<BxHII4x", 1, len(msgs) + 1, 1, len(head)) + head)

    def dataset(self, arr, chunk_rows=None, mode="blosclz", blosc=True, extra_filters=()):
        arr = np.ascontiguousarray(arr)
        rank = arr.ndim
        dspace = struct.pack("<BBB5x", 1, rank, 0) + b"".join(struct.pack("<Q", d) for d in arr.shape)
        if arr.dtype.kind == "f":                      # IEEE little-endian: class 1; properties: sign location, exponent, mantissa
            bits = arr.dtype.itemsize * 8
            ex, mant = {32: (8, 23), 64: (11, 52)}[bits]
            dtype = struct.pack("<BBBBI", 0x11, 0x20, bits - 1, 0, arr.dtype.itemsize) + struct.pack("<HHBBBBI", 0, bits, mant, ex, 0, mant, (1 << (ex - 1)) - 1)
        else:                                          # fixed-point little-endian, signed when kind == 'i'
            dtype = struct.pack("<BBBBI", 0x10, 0x08 if arr.dtype.kind == "i" else 0, 0, 0, arr.dtype.itemsize) + struct.pack("<HH", 0, arr.dtype.itemsize * 8)
        chunk_rows = chunk_rows or max(1, arr.shape[0])
        cshape = (chunk_rows,) + arr.shape[1:]
        entries = []
        for r0 in range(0, arr.shape[0], chunk_rows):
            chunk = np.zeros(cshape, arr.dtype)
            part = arr[r0:r0 + chunk_rows]
            chunk[:part.shape[0]] = part
            raw = chunk.tobytes()
            for f in extra_filters:                    # (applied in pipeline order)
                if f == "shuffle":
                    raw = shuffle(raw, arr.dtype.itemsize)
                elif f == "deflate":
                    import zlib
                    raw = zlib.compress(raw)
            if blosc:
                raw = blosc_frame(raw, arr.dtype.itemsize, mode)
            at = self.alloc(raw)
            entries.append((len(raw), (r0,) + (0,) * (rank - 1), at))
        # v1 B-tree of chunks, one leaf node: key_i, child_i, ..., final key
        node = bytearray(b"TREE" + struct.pack("<BBHQQ", 1, 0, len(entries), UNDEF, UNDEF))
        for size, offs, at in entries:
            node += struct.pack("<II", size, 0) + b"".join(struct.pack("<Q", o) for o in offs) + struct.pack("<Q", 0) + struct.pack("<Q", at)
        node += struct.pack("<II", 0, 0) + struct.pack("<Q", arr.shape[0]) + b"\0" * (8 * rank)
        btree = self.alloc(bytes(node))
        layout = struct.pack("<BBB", 3, 2, rank + 1) + struct.pack("<Q", btree) + b"".join(struct.pack("<I", c) for c in cshape) + struct.pack("<I", arr.dtype.itemsize)
        filt = []
        for f in extra_filters:
            fid, name, cd = {"shuffle": (2, b"shuffle\0", [arr.dtype.itemsize]), "deflate": (1, b"deflate\0", [6])}[f]
            filt.append(struct.pack("<HHHH", fid, len(name), 1, len(cd)) + name + b"".join(struct.pack("<I", c) for c in cd) + (b"\0" * 4 if len(cd) % 2 else b""))
        if blosc:
            cd = [2, 2, arr.dtype.itemsize, int(np.prod(cshape)) * arr.dtype.itemsize, 9, 1, 0]
            filt.append(struct.pack("<HHHH", 32001, 8, 1, len(cd)) + b"blosc\0\0\0" + b"".join(struct.pack("<I", c) for c in cd) + b"\0" * 4)
        msgs = [self.msg(1, dspace), self.msg(3, dtype)]
        if filt:
            msgs.append(self.msg(0xB, struct.pack("<BB6x", 1, len(filt)) + b"".join(filt)))
        msgs += [self.msg(0xC, b"\x01\x00" + b"junk-attribute-bytes\0\0"), self.msg(8, layout)]      # (an attribute message in between: skipped by type)
        return self.object_header(msgs, split_after=3)

    def group(self, links):
        """old-style group: local heap with the names, one symbol-table node, a one-leaf B-tree; returns the object header address"""
        names = sorted(links)
        heap = bytearray(b"\0" * 8)                    # (offset 0: the empty name)
        offs = {}
        for nm in names:
            offs[nm] = len(heap)
            heap += nm.encode() + b"\0"
            heap += b"\0" * (-len(heap) % 8)
        data_at = self.alloc(bytes(heap))
        heap_at = self.alloc(b"HEAP" + struct.pack("<B3xQQQ", 0, len(heap), UNDEF, data_at))
        snod = bytearray(b"SNOD" + struct.pack("<BxH", 1, len(names)))
        for nm in names:
            snod += struct.pack("<QQII16x", offs[nm], links[nm], 0, 0)
        snod_at = self.alloc(bytes(snod))
        node = b"TREE" + struct.pack("<BBHQQ", 0, 0, 1, UNDEF, UNDEF) + struct.pack("<QQQ", 0, snod_at, offs[names[-1]])
        btree = self.alloc(node)
        return self.object_header([self.msg(0x11, struct.pack("<QQ", btree, heap_at))])

    def finish(self, root_oh):
        sb = (b"\x89HDF\r\n\x1a\n" + struct.pack("<BBBBBBBBHHI", 0, 0, 0, 0, 0, 8, 8, 0, 4, 16, 0) +
              struct.pack("<QQQQ", 0, UNDEF, len(self.buf), UNDEF) + struct.pack("<QQII16x", 0, root_oh, 0, 0))
        assert len(sb) == 96
        self.buf[:96] = sb
        return bytes(self.buf)


def write_dict(path, arrays, **kw):
    w = Writer()
    links = {}
    for name, a in arrays.items():
        if isinstance(a, dict):
            sub = {k: w.dataset(v, **kw) for k, v in a.items()}
            links[name] = w.group(sub)
        else:
            links[name] = w.dataset(a, **kw)
    open(path, "wb").write(w.finish(w.group(links)))
