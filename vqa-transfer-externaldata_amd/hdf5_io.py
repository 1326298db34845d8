"""Dependency-free reader / writer for the HDF5 files of the reference (no h5py, no TensorFlow).

The reference exchanges four kinds of HDF5 file, all written through h5py with default settings (library
format "earliest": superblock version 0, version-1 object headers, symbol-table groups, CONTIGUOUS un-chunked
un-compressed datasets):
  * region features   data/tools/vqa_v2/process_bottom_up_attention_36.py:47-53,95-100 and
                      vqa/vfeat_extractor_tf_record_memft.py:82-145: /image_features [N,R,D] f32, /normal_boxes
                      [N,R,4] f32, /spatial_features [N,R,6] f32, /num_boxes [N] i32, group /data_info with scalar
                      datasets vfeat_dim, max_box_num (+ a variable-length string pretrained_param_path);
                      read by vqa/model_vlmap_answer.py:59-70
  * weights.hdf5      vlmap_memft/export_word_weights.py:60-73 (v_word, l_word, l_answer_word, class_weights,
                      class_biases), read by modules.WordWeightAnswer (vlmap/modules.py:598-601)
  * data_info.hdf5    data/tools/vqa_v2/generator_tf_record_memft_genome.py:81-107 (/data_info/num_answers ...),
                      read by vqa/datasets/input_ops_vqa_tf_record_memft.py:13-15
  * results.hdf5      vqa/evaler.py:181-186 (heavy outputs)

`File` walks exactly that subset of the HDF5 file format (Format Specification 2.0: superblock v0/v1, object
header v1 with continuation blocks, symbol-table message -> group B-tree v1 -> symbol-table nodes + local heap,
dataspace v1/v2, datatype classes fixed-point / floating-point / string / variable-length string, data layout v3
contiguous / compact / chunked (B-tree v1, optional shuffle + deflate filters), global heap for variable-length
strings) and returns datasets as NumPy arrays -- contiguous little-endian data as a zero-copy np.memmap, so the
36 GB feature table is paged in while it is uploaded rather than read twice.  `write` produces files in the same
format (what libhdf5 / h5py / h5dump read back; tests/test_hdf5_io.py checks both directions against the real
libhdf5 when the image has one).  Anything outside the subset raises Hdf5FormatError naming the construct.
"""
from __future__ import annotations

import mmap
import os
import struct
import zlib

import numpy as np

SIGNATURE = b"\x89HDF\r\n\x1a\n"
UNDEF = 0xFFFFFFFFFFFFFFFF

MSG_NIL, MSG_DATASPACE, MSG_LINK_INFO, MSG_DATATYPE, MSG_FILL_OLD, MSG_FILL, MSG_LINK = 0x0, 0x1, 0x2, 0x3, 0x4, 0x5, 0x6
MSG_LAYOUT, MSG_FILTER, MSG_ATTRIBUTE, MSG_CONTINUATION, MSG_SYMBOL_TABLE, MSG_MTIME = 0x8, 0xB, 0xC, 0x10, 0x11, 0x12


class Hdf5FormatError(ValueError):
    pass


def is_hdf5(path):
    """True when the file starts with the HDF5 signature (the format's own dispatch: superblock at offset 0)."""
    try:
        with open(path, "rb") as f:
            return f.read(8) == SIGNATURE
    except OSError:
        return False


# ======================================================================================= reader
class _Datatype:
    def __init__(self, buf, off):
        b0 = buf[off]
        self.cls, self.version = b0 & 0x0F, b0 >> 4
        self.bits = (buf[off + 1], buf[off + 2], buf[off + 3])
        self.size = struct.unpack_from("<I", buf, off + 4)[0]
        self.base = None
        self.is_vlen_str = False
        p = off + 8
        order = ">" if (self.bits[0] & 1) else "<"
        if self.cls == 0:                                   # fixed-point
            signed = bool(self.bits[0] & 0x08)
            self.dtype = np.dtype("%s%s%d" % (order, "i" if signed else "u", self.size))
        elif self.cls == 1:                                 # floating-point (IEEE layouts only)
            if self.size not in (2, 4, 8):
                raise Hdf5FormatError("floating-point datatype of %d bytes is not supported" % self.size)
            self.dtype = np.dtype("%sf%d" % (order, self.size))
        elif self.cls == 3:                                 # fixed-length string
            self.dtype = np.dtype("S%d" % self.size)
        elif self.cls == 9:                                 # variable-length (strings only)
            self.is_vlen_str = (self.bits[0] & 0x0F) == 1
            self.base = _Datatype(buf, p)
            if not self.is_vlen_str:
                raise Hdf5FormatError("variable-length sequences are not supported (only variable-length strings)")
            self.dtype = np.dtype(object)
        else:
            raise Hdf5FormatError("datatype class %d is not supported" % self.cls)


class Dataset:
    def __init__(self, f, name, shape, dt, layout):
        self._f, self.name, self.shape, self._dt, self._layout = f, name, tuple(shape), dt, layout
        self.dtype = dt.dtype

    @property
    def size(self):
        n = 1
        for s in self.shape:
            n *= s
        return n

    def __len__(self):
        if not self.shape:
            raise TypeError("scalar dataset has no len()")
        return self.shape[0]

    def _raw_array(self):
        f, lay, dt = self._f, self._layout, self._dt
        n = self.size
        esz = dt.size
        if dt.is_vlen_str:
            raw = self._bytes(n * 16)
            out = np.empty(n, dtype=object)
            for i in range(n):
                ln, addr, idx = struct.unpack_from("<IQI", raw, 16 * i)
                out[i] = f._global_heap_object(addr, idx)[:ln].decode("utf-8", "replace") if ln else ""
            return out.reshape(self.shape)
        if lay["cls"] == 1:                                  # contiguous
            if lay["addr"] == UNDEF or n == 0:
                return np.zeros(self.shape, dt.dtype)         # never written: fill value (default zero)
            if lay["addr"] + n * esz > f._size:
                raise Hdf5FormatError("dataset %s extends past the end of the file" % self.name)
            return np.memmap(f.path, dtype=dt.dtype, mode="r", offset=f._base + lay["addr"], shape=self.shape)
        if lay["cls"] == 0:                                  # compact
            return np.frombuffer(lay["data"], dtype=dt.dtype, count=n).reshape(self.shape).copy()
        return self._read_chunked()

    def _bytes(self, nbytes):
        lay, f = self._layout, self._f
        if lay["cls"] == 0:
            return bytes(lay["data"][:nbytes])
        if lay["cls"] == 1:
            if lay["addr"] == UNDEF:
                return b"\0" * nbytes
            return bytes(f._mm[f._base + lay["addr"]:f._base + lay["addr"] + nbytes])
        return self._read_chunked().tobytes()

    def _read_chunked(self):
        f, lay, dt = self._f, self._layout, self._dt
        cdims = lay["chunk"]                                 # chunk dims (dataset rank) -- element size stripped
        rank = len(self.shape)
        out = np.zeros(self.shape, dt.dtype)
        if lay["addr"] == UNDEF:
            return out
        filters = lay.get("filters") or []
        for offs, addr, nbytes, mask in f._chunk_btree(lay["addr"], rank):
            raw = bytes(f._mm[f._base + addr:f._base + addr + nbytes])
            for k, (fid, cvals) in enumerate(reversed(filters)):
                if mask & (1 << (len(filters) - 1 - k)):
                    continue
                if fid == 1:
                    raw = zlib.decompress(raw)
                elif fid == 2:                               # shuffle
                    esz = cvals[0] if cvals else dt.size
                    a = np.frombuffer(raw, np.uint8)
                    raw = a.reshape(esz, -1).T.tobytes() if len(a) % esz == 0 else raw
                elif fid == 3:                               # fletcher32: checksum trails the data
                    raw = raw[:-4]
                else:
                    raise Hdf5FormatError("filter id %d is not supported" % fid)
            chunk = np.frombuffer(raw, dt.dtype, count=int(np.prod(cdims))).reshape(cdims)
            sl_out = tuple(slice(o, min(o + c, s)) for o, c, s in zip(offs, cdims, self.shape))
            sl_in = tuple(slice(0, s.stop - s.start) for s in sl_out)
            out[sl_out] = chunk[sl_in]
        return out

    def read(self):
        """The whole dataset as an ndarray (np.memmap for contiguous storage; native byte order is kept as stored)."""
        return self._raw_array()

    def __array__(self, dtype=None, copy=None):
        a = np.asarray(self._raw_array())
        return a.astype(dtype) if dtype is not None else a

    def __getitem__(self, key):
        a = self._raw_array()
        if key is Ellipsis or (isinstance(key, tuple) and len(key) == 0):
            if not self.shape:
                v = a.reshape(-1)[0]                              # NumPy scalar (str for a variable-length string)
                return v.decode("utf-8", "replace") if isinstance(v, bytes) else v
            return np.asarray(a)
        return a[key]

    @property
    def value(self):
        """h5py < 3 accessor the reference uses (`f['data_info']['max_box_num'].value`)."""
        return self[()]


class Group:
    def __init__(self, f, name, entries):
        self._f, self.name, self._entries = f, name, entries     # entries: name -> object header address

    def keys(self):
        return list(self._entries)

    def __iter__(self):
        return iter(self._entries)

    def __len__(self):
        return len(self._entries)

    def __contains__(self, name):
        try:
            self[name]
            return True
        except KeyError:
            return False

    def items(self):
        return [(k, self[k]) for k in self._entries]

    def get(self, name, default=None):
        try:
            return self[name]
        except KeyError:
            return default

    def __getitem__(self, name):
        node = self
        parts = [p for p in name.split("/") if p]
        if name.startswith("/"):
            node = self._f.root
        for i, p in enumerate(parts):
            if not isinstance(node, Group) or p not in node._entries:
                raise KeyError("%s (no object %r in %s)" % (name, p, node.name))
            node = node._f._open_object(node._entries[p], (node.name.rstrip("/") + "/" + p))
        return node


class File(Group):
    """Read-only HDF5 file: `with File(path) as f: np.asarray(f['image_features'])`, `f['data_info']['vfeat_dim'][()]`."""

    def __init__(self, path, mode="r"):
        if mode != "r":
            raise ValueError("hdf5_io.File is read-only; use hdf5_io.write() to create files")
        self.path = path
        self._fh = open(path, "rb")
        self._size = os.fstat(self._fh.fileno()).st_size
        if self._size < 96:
            raise Hdf5FormatError("%s: too short for an HDF5 file" % path)
        self._mm = mmap.mmap(self._fh.fileno(), 0, access=mmap.ACCESS_READ)
        self._cache = {}
        self._parse_superblock()
        Group.__init__(self, self, "/", self._group_entries(self._root_header))
        self.root = self

    # context manager / cleanup
    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def close(self):
        if self._mm is not None:
            try:
                self._mm.close()
            except (BufferError, ValueError):
                pass                                         # arrays handed out still reference the map
            self._mm = None
            self._fh.close()

    # ----------------------------------------------------------------- low level
    def _parse_superblock(self):
        mm = self._mm
        base = None
        off = 0
        while off + 8 <= self._size:                         # the superblock may sit at 0, 512, 1024, ...
            if mm[off:off + 8] == SIGNATURE:
                base = off
                break
            off = 512 if off == 0 else off * 2
        if base is None:
            raise Hdf5FormatError("%s is not an HDF5 file (signature not found)" % self.path)
        ver = mm[base + 8]
        if ver not in (0, 1):
            raise Hdf5FormatError("superblock version %d is not supported (files written with libver='latest'; "
                                  "the reference's h5py default writes version 0)" % ver)
        so, sl = mm[base + 13], mm[base + 14]
        if so != 8 or sl != 8:
            raise Hdf5FormatError("only 8-byte offsets / lengths are supported (got %d / %d)" % (so, sl))
        self.superblock = {"version": ver, "free_space_version": mm[base + 9], "root_entry_version": mm[base + 10],
                           "shared_header_version": mm[base + 12], "size_of_offsets": so, "size_of_lengths": sl,
                           "group_leaf_k": struct.unpack_from("<H", mm, base + 16)[0],
                           "group_internal_k": struct.unpack_from("<H", mm, base + 18)[0]}
        p = base + 24 + (4 if ver == 1 else 0)
        base_addr, _fs, eof, _drv = struct.unpack_from("<QQQQ", mm, p)
        self.superblock.update(base_address=base_addr, eof_address=eof)
        self._base = base + base_addr if base_addr != UNDEF else base
        # root group symbol table entry
        _name_off, hdr, _ctype, _res = struct.unpack_from("<QQII", mm, p + 32)
        self._root_header = hdr

    def _messages(self, addr):
        """[(type, flags, data-bytes)] of a version-1 object header, continuation blocks included."""
        key = ("msgs", addr)
        if key in self._cache:
            return self._cache[key]
        mm, b = self._mm, self._base
        if mm[b + addr:b + addr + 4] == b"OHDR":
            raise Hdf5FormatError("version-2 object headers are not supported (file written with libver='latest')")
        ver, _r, nmsg, _ref, hsize = struct.unpack_from("<BBHII", mm, b + addr)
        if ver != 1:
            raise Hdf5FormatError("object header version %d at %d is not supported" % (ver, addr))
        blocks = [(addr + 16, hsize)]
        out = []
        while blocks and len(out) < nmsg:
            p, n = blocks.pop(0)
            end = p + n
            while p + 8 <= end and len(out) < nmsg:
                mtype, msize, mflags = struct.unpack_from("<HHB", mm, b + p)
                data = bytes(mm[b + p + 8:b + p + 8 + msize])
                p += 8 + msize
                if mtype == MSG_CONTINUATION:
                    caddr, clen = struct.unpack("<QQ", data[:16])
                    blocks.append((caddr, clen))
                out.append((mtype, mflags, data))
        self._cache[key] = out
        return out

    def _local_heap_name(self, heap_addr, off):
        mm, b = self._mm, self._base
        if mm[b + heap_addr:b + heap_addr + 4] != b"HEAP":
            raise Hdf5FormatError("local heap signature missing at %d" % heap_addr)
        _sz, _free, data = struct.unpack_from("<QQQ", mm, b + heap_addr + 8)
        s = b + data + off
        e = mm.find(b"\0", s)
        return bytes(mm[s:e]).decode("utf-8")

    def _walk_group_btree(self, addr, heap, out):
        mm, b = self._mm, self._base
        sig = bytes(mm[b + addr:b + addr + 4])
        if sig == b"SNOD":
            nsym = struct.unpack_from("<H", mm, b + addr + 6)[0]
            for i in range(nsym):
                name_off, hdr = struct.unpack_from("<QQ", mm, b + addr + 8 + 40 * i)
                out[self._local_heap_name(heap, name_off)] = hdr
            return
        if sig != b"TREE":
            raise Hdf5FormatError("expected a group B-tree / symbol node at %d" % addr)
        ntype, _level, used = struct.unpack_from("<BBH", mm, b + addr + 4)
        if ntype != 0:
            raise Hdf5FormatError("B-tree node type %d where a group node was expected" % ntype)
        p = b + addr + 24
        for i in range(used):
            child = struct.unpack_from("<Q", mm, p + 8 + 16 * i)[0]     # key_i, child_i, key_i+1 ...
            self._walk_group_btree(child, heap, out)

    def _group_entries(self, hdr_addr):
        entries = {}
        for mtype, _fl, data in self._messages(hdr_addr):
            if mtype == MSG_SYMBOL_TABLE:
                btree, heap = struct.unpack("<QQ", data[:16])
                self._walk_group_btree(btree, heap, entries)
                return entries
            if mtype in (MSG_LINK_INFO, MSG_LINK):
                raise Hdf5FormatError("new-style (link message) groups are not supported")
        return None                                          # not a group

    def _global_heap_object(self, addr, index):
        mm, b = self._mm, self._base
        if mm[b + addr:b + addr + 4] != b"GCOL":
            raise Hdf5FormatError("global heap collection signature missing at %d" % addr)
        csize = struct.unpack_from("<Q", mm, b + addr + 8)[0]
        p, end = b + addr + 16, b + addr + csize
        while p + 16 <= end:
            idx, _ref, _r, osize = struct.unpack_from("<HHIQ", mm, p)
            if idx == index:
                return bytes(mm[p + 16:p + 16 + osize])
            if idx == 0:
                break
            p += 16 + ((osize + 7) // 8) * 8
        raise Hdf5FormatError("global heap object %d not found in collection at %d" % (index, addr))

    def _chunk_btree(self, addr, rank):
        """Yields (chunk offsets [rank], address, stored bytes, filter mask) of a chunked dataset's B-tree."""
        mm, b = self._mm, self._base
        if mm[b + addr:b + addr + 4] != b"TREE":
            raise Hdf5FormatError("chunk B-tree signature missing at %d" % addr)
        ntype, level, used = struct.unpack_from("<BBH", mm, b + addr + 4)
        if ntype != 1:
            raise Hdf5FormatError("B-tree node type %d where a chunk node was expected" % ntype)
        ksz = 8 + 8 * (rank + 1)
        p = b + addr + 24
        for i in range(used):
            kp = p + i * (ksz + 8)
            nbytes, mask = struct.unpack_from("<II", mm, kp)
            offs = struct.unpack_from("<%dQ" % (rank + 1), mm, kp + 8)[:rank]
            child = struct.unpack_from("<Q", mm, kp + ksz)[0]
            if level == 0:
                yield offs, child, nbytes, mask
            else:
                for x in self._chunk_btree(child, rank):
                    yield x

    def _open_object(self, hdr_addr, name):
        key = ("obj", hdr_addr)
        if key in self._cache:
            return self._cache[key]
        ent = self._group_entries(hdr_addr)
        if ent is not None:
            obj = Group(self, name, ent)
        else:
            shape = dt = layout = None
            filters = None
            for mtype, _fl, d in self._messages(hdr_addr):
                if mtype == MSG_DATASPACE:
                    ver, rank, flags = d[0], d[1], d[2]
                    if ver == 1:
                        q = 8
                    elif ver == 2:
                        q = 4
                        if d[3] == 2:
                            raise Hdf5FormatError("null dataspace (dataset %s)" % name)
                    else:
                        raise Hdf5FormatError("dataspace version %d" % ver)
                    shape = struct.unpack_from("<%dQ" % rank, d, q) if rank else ()
                elif mtype == MSG_DATATYPE:
                    dt = _Datatype(d, 0)
                elif mtype == MSG_FILTER:
                    filters = _parse_filters(d)
                elif mtype == MSG_LAYOUT:
                    ver = d[0]
                    if ver == 3:
                        cls = d[1]
                        if cls == 0:
                            n = struct.unpack_from("<H", d, 2)[0]
                            layout = {"cls": 0, "data": d[4:4 + n]}
                        elif cls == 1:
                            a, n = struct.unpack_from("<QQ", d, 2)
                            layout = {"cls": 1, "addr": a, "size": n}
                        elif cls == 2:
                            nd = d[2]
                            a = struct.unpack_from("<Q", d, 3)[0]
                            cd = struct.unpack_from("<%dI" % nd, d, 11)
                            layout = {"cls": 2, "addr": a, "chunk": tuple(cd[:-1])}
                        else:
                            raise Hdf5FormatError("data layout class %d" % cls)
                    elif ver in (1, 2):
                        nd, cls = d[1], d[2]
                        q = 8
                        a = UNDEF
                        if cls != 0:
                            a = struct.unpack_from("<Q", d, q)[0]
                            q += 8
                        dims = struct.unpack_from("<%dI" % nd, d, q)
                        q += 4 * nd
                        if cls == 1:
                            layout = {"cls": 1, "addr": a, "size": 0}
                        elif cls == 2:
                            layout = {"cls": 2, "addr": a, "chunk": tuple(dims[:-1])}
                        else:
                            n = struct.unpack_from("<I", d, q)[0]
                            layout = {"cls": 0, "data": d[q + 4:q + 4 + n]}
                    else:
                        raise Hdf5FormatError("data layout message version %d is not supported" % ver)
            if shape is None or dt is None or layout is None:
                raise Hdf5FormatError("object %s is neither a symbol-table group nor a complete dataset" % name)
            if layout["cls"] == 2:
                layout["filters"] = filters
            obj = Dataset(self, name, shape, dt, layout)
        self._cache[key] = obj
        return obj


def _parse_filters(d):
    """Filter pipeline message (v1 / v2) -> [(filter id, client values)] in application order."""
    ver, n = d[0], d[1]
    if ver not in (1, 2):
        raise Hdf5FormatError("filter pipeline message version %d" % ver)
    out = []
    p = 8 if ver == 1 else 2
    for _ in range(n):
        fid = struct.unpack_from("<H", d, p)[0]
        p += 2
        nlen = 0
        if ver == 1 or fid >= 256:
            nlen = struct.unpack_from("<H", d, p)[0]
            p += 2
        _flags, ncv = struct.unpack_from("<HH", d, p)
        p += 4
        p += (nlen + 7) // 8 * 8 if ver == 1 else nlen
        cvals = struct.unpack_from("<%dI" % ncv, d, p)
        p += 4 * ncv
        if ver == 1 and ncv % 2:
            p += 4
        out.append((fid, cvals))
    return out


def load_tree(path):
    """Whole file as nested dicts of ndarrays (small files: weights.hdf5, data_info.hdf5)."""
    def conv(g):
        out = {}
        for k, v in g.items():
            out[k] = conv(v) if isinstance(v, Group) else (np.array(v.read()) if v.shape else v[()])
        return out
    with File(path) as f:
        return conv(f)


# ======================================================================================= writer
def _pad8(n):
    return (n + 7) // 8 * 8


def _dtype_message(dt):
    dt = np.dtype(dt)
    if dt.kind in "iu":
        bits0 = 0x08 if dt.kind == "i" else 0x00
        return struct.pack("<BBBBIHH", 0x10 | 0, bits0, 0, 0, dt.itemsize, 0, 8 * dt.itemsize)
    if dt.kind == "f":
        if dt.itemsize == 4:
            sign, exp_loc, exp_sz, man_sz, bias = 31, 23, 8, 23, 127
        elif dt.itemsize == 8:
            sign, exp_loc, exp_sz, man_sz, bias = 63, 52, 11, 52, 1023
        elif dt.itemsize == 2:
            sign, exp_loc, exp_sz, man_sz, bias = 15, 10, 5, 10, 15
        else:
            raise Hdf5FormatError("cannot write float%d" % (8 * dt.itemsize))
        return struct.pack("<BBBBIHHBBBBI", 0x10 | 1, 0x20, sign, 0, dt.itemsize, 0, 8 * dt.itemsize, exp_loc, exp_sz,
                           0, man_sz, bias)
    if dt.kind == "S":
        return struct.pack("<BBBBI", 0x10 | 3, 0x00, 0, 0, dt.itemsize)       # null-terminated, ASCII
    raise Hdf5FormatError("cannot write dtype %s" % dt)


def _message(mtype, data, flags=0):
    data = data + b"\0" * (_pad8(len(data)) - len(data))
    return struct.pack("<HHBBBB", mtype, len(data), flags, 0, 0, 0) + data


def _object_header(msgs):
    body = b"".join(msgs)
    return struct.pack("<BBHII", 1, 0, len(msgs), 1, len(body)) + b"\0\0\0\0" + body


class Empty:
    """Placeholder for `create`: a contiguous dataset of this shape / dtype whose data region is allocated (zeros, a hole
    in the file) but not written -- the caller fills it through the np.memmap `create` hands back."""

    def __init__(self, shape, dtype=np.float32):
        self.shape, self.dtype = tuple(int(s) for s in shape), np.dtype(dtype).newbyteorder("<")


class _Writer:
    LEAF_K, INTERNAL_K = 4, 16

    def __init__(self, fh):
        self.fh = fh
        self.pos = 0
        self.regions = {}            # "/group/name" -> (address, shape, dtype) of every Empty placeholder

    def alloc(self, n, align=8):
        self.pos = (self.pos + align - 1) // align * align
        a = self.pos
        self.pos += n
        return a

    def put(self, addr, data):
        self.fh.seek(addr)
        self.fh.write(data)

    def dataset(self, value, name=None):
        if isinstance(value, Empty):
            nbytes = int(np.prod(value.shape, dtype=np.int64)) * value.dtype.itemsize
            daddr = self.alloc(nbytes, 8) if nbytes else UNDEF
            self.regions[name] = (daddr, value.shape, value.dtype)
            space = struct.pack("<BBBBI", 1, len(value.shape), 0, 0, 0) + b"".join(struct.pack("<Q", d) for d in value.shape)
            hdr = _object_header([_message(MSG_DATASPACE, space),
                                  _message(MSG_DATATYPE, _dtype_message(value.dtype), flags=1),
                                  _message(MSG_FILL, struct.pack("<BBBBI", 2, 2, 2, 1, 0)),
                                  _message(MSG_LAYOUT, struct.pack("<BBQQ", 3, 1, daddr, nbytes))])
            a = self.alloc(len(hdr))
            self.put(a, hdr)
            return a
        if isinstance(value, str):
            value = value.encode("utf-8")
        if isinstance(value, bytes):
            arr = np.array(value + b"\0", dtype="S%d" % (len(value) + 1))
        else:
            arr = np.asarray(value)
            if arr.dtype == np.bool_:
                arr = arr.astype(np.int8)
            if arr.dtype.kind == "U":
                arr = np.char.encode(arr, "utf-8")
            if arr.dtype.kind not in "iufS":
                raise Hdf5FormatError("cannot write an array of dtype %s" % arr.dtype)
            shape = arr.shape                                     # ascontiguousarray promotes 0-d to 1-d
            arr = np.ascontiguousarray(arr.astype(arr.dtype.newbyteorder("<")) if arr.dtype.byteorder == ">" else arr)
            arr = arr.reshape(shape)
        nbytes = arr.size * arr.dtype.itemsize
        daddr = UNDEF
        if nbytes:
            daddr = self.alloc(nbytes, 8)
            self.fh.seek(daddr)
            flat = arr.reshape(-1)
            step = max(1, (64 << 20) // max(arr.dtype.itemsize, 1))
            for i in range(0, flat.size, step):                  # stream large tables
                self.fh.write(flat[i:i + step].tobytes())
        rank = arr.ndim
        space = struct.pack("<BBBBI", 1, rank, 0, 0, 0) + b"".join(struct.pack("<Q", s) for s in arr.shape)
        msgs = [_message(MSG_DATASPACE, space), _message(MSG_DATATYPE, _dtype_message(arr.dtype), flags=1),
                _message(MSG_FILL, struct.pack("<BBBBI", 2, 2, 2, 1, 0)),
                _message(MSG_LAYOUT, struct.pack("<BBQQ", 3, 1, daddr, nbytes))]
        hdr = _object_header(msgs)
        a = self.alloc(len(hdr))
        self.put(a, hdr)
        return a

    def group(self, tree, prefix=""):
        """Writes the children first, then heap + symbol nodes + B-tree + header; returns (header, btree, heap)."""
        if len(tree) > 2 * self.LEAF_K * 2 * self.INTERNAL_K:
            raise Hdf5FormatError("too many entries in one group (%d)" % len(tree))
        children = []
        for name in sorted(tree, key=lambda s: s.encode("utf-8")):
            if "/" in name or not name:
                raise Hdf5FormatError("bad object name %r" % name)
            v = tree[name]
            if isinstance(v, dict):
                h, bt, hp = self.group(v, prefix + "/" + name)
                children.append((name, h, 1, bt, hp))
            else:
                children.append((name, self.dataset(v, prefix + "/" + name), 0, 0, 0))
        # local heap data: "" at offset 0, then the names, 8-byte aligned each
        heap_data = bytearray(8)
        offs = []
        for name, *_ in children:
            offs.append(len(heap_data))
            nb = name.encode("utf-8") + b"\0"
            heap_data += nb + b"\0" * (_pad8(len(nb)) - len(nb))
        # one free block at the end (libhdf5 keeps the heap larger than its contents)
        free_off = len(heap_data)
        heap_data += struct.pack("<QQ", 1, 16 + 64) + b"\0" * 64
        data_addr = self.alloc(len(heap_data))
        self.put(data_addr, bytes(heap_data))
        heap_addr = self.alloc(32)
        self.put(heap_addr, b"HEAP" + struct.pack("<BBBBQQQ", 0, 0, 0, 0, len(heap_data), free_off, data_addr))
        # symbol table nodes of <= 2K entries
        cap = 2 * self.LEAF_K
        snods = []
        for lo in range(0, len(children), cap):
            part = list(zip(children[lo:lo + cap], offs[lo:lo + cap]))
            body = b"SNOD" + struct.pack("<BBH", 1, 0, len(part))
            for (name, hdr, ctype, bt, hp), noff in part:
                scratch = struct.pack("<QQ", bt, hp) if ctype == 1 else b"\0" * 16
                body += struct.pack("<QQII", noff, hdr, ctype, 0) + scratch
            body += b"\0" * (8 + 40 * cap - len(body))
            a = self.alloc(len(body))
            self.put(a, body)
            snods.append((a, part[-1][1] if part else 0))
        # B-tree root, level 0: key0 = 0 (""), key_i = heap offset of the largest name in child i-1
        node = b"TREE" + struct.pack("<BBHQQ", 0, 0, len(snods), UNDEF, UNDEF) + struct.pack("<Q", 0)
        for a, last in snods:
            node += struct.pack("<QQ", a, last)
        node += b"\0" * (24 + 8 * (2 * self.INTERNAL_K + 1) + 8 * 2 * self.INTERNAL_K - len(node))
        bt_addr = self.alloc(len(node))
        self.put(bt_addr, node)
        hdr = _object_header([_message(MSG_SYMBOL_TABLE, struct.pack("<QQ", bt_addr, heap_addr)),
                              _message(MSG_NIL, b"\0" * 8)])
        h_addr = self.alloc(len(hdr))
        self.put(h_addr, hdr)
        return h_addr, bt_addr, heap_addr


def _finish_file(w, fh, root, bt, hp):
    eof = w.alloc(0, 8)
    sb = SIGNATURE + struct.pack("<BBBBBBBBHHI", 0, 0, 0, 0, 0, 8, 8, 0, w.LEAF_K, w.INTERNAL_K, 0)
    sb += struct.pack("<QQQQ", 0, UNDEF, eof, UNDEF)
    sb += struct.pack("<QQII", 0, root, 1, 0) + struct.pack("<QQ", bt, hp)
    assert len(sb) == 96
    w.put(0, sb)
    fh.seek(0, 2)
    if fh.tell() < eof:
        fh.truncate(eof)


def create(path, tree):
    """Like `write`, for tables that are filled row by row: `Empty(shape, dtype)` values become contiguous datasets whose
    data region is allocated but left as a hole; returns {"/name": np.memmap opened r+} for them.  The file is complete
    and readable (zeros where nothing was written yet) from the moment this returns -- the streaming counterpart of
    h5py's `f.create_dataset(...)` followed by row assignments (vqa/vfeat_extractor_tf_record_memft.py:118-139)."""
    with open(path, "wb") as fh:
        w = _Writer(fh)
        w.alloc(96)
        root, bt, hp = w.group(tree)
        _finish_file(w, fh, root, bt, hp)
        regions = dict(w.regions)
    return {name: (np.memmap(path, dtype=dt, mode="r+", offset=addr, shape=shape) if addr != UNDEF
                   else np.zeros(shape, dt))
            for name, (addr, shape, dt) in regions.items()}


def write(path, tree):
    """Creates `path` from a nested dict: dict -> group, ndarray / scalar / str -> contiguous dataset.
    Format: superblock v0, v1 object headers, symbol-table groups -- what h5py's defaults produce."""
    tmp = path + ".tmp%d" % os.getpid()
    with open(tmp, "wb") as fh:
        w = _Writer(fh)
        w.alloc(96)                                          # superblock
        root, bt, hp = w.group(tree)
        _finish_file(w, fh, root, bt, hp)
    os.replace(tmp, path)
    return path
