"""Minimal, dependency-free HDF5 writer for the slide-grid output format.

The reference packs the per-slide `.npy` code grids into one HDF5 file with `h5py`
(scripts/convert_npy_embeddings_to_hdf5/convert.py:27-32: one group per sub-directory -- `images`,
`masks` -- one contiguous dataset per slide, named by the file stem) which its downstream dataset reads
back (datamodules/camelyon16.py:226-235: `hdf5['images'][key]`, `hdf5['masks'][key + '_mask']`).
h5py is not installed for the interpreter this project runs under, so this module writes that layout
itself: HDF5 file-format version 0 superblock, "old-style" groups (v1 B-tree + local heap + symbol-table
nodes), version-1 object headers, contiguous little-endian datasets.  numpy bool arrays are stored the
way h5py stores them (enum {FALSE=0, TRUE=1} over int8) so `np.asarray(ds)` gives bool back.

Datasets are appended as they arrive (raw data + object header are written immediately: "streamed");
the group metadata and the superblock are written by `close()`.
Files are validated against libhdf5 (`h5dump`, h5py) in tests/test_hdf5.py when those tools exist.
"""
import struct
from pathlib import Path

import numpy as np

UNDEF = 0xFFFFFFFFFFFFFFFF
LEAF_K = 512            # group leaf node K: a symbol-table node holds up to 2*LEAF_K = 1024 entries
INTERNAL_K = 16         # group internal node K: a B-tree node holds up to 32 children


def _pad8(b: bytes) -> bytes:
    return b + b"\x00" * (-len(b) % 8)


def _datatype_message(dtype: np.dtype) -> bytes:
    """Datatype message body (version 1)."""
    dtype = np.dtype(dtype)
    if dtype == np.bool_:
        # enum over int8 with members FALSE = 0, TRUE = 1 (what h5py writes for numpy bool)
        base = _datatype_message(np.dtype("int8"))
        names = _pad8(b"FALSE\x00") + _pad8(b"TRUE\x00")
        values = struct.pack("<bb", 0, 1)
        head = struct.pack("<BBBBI", (1 << 4) | 8, 2, 0, 0, 1)          # class 8 (enum), 2 members, size 1
        return head + base + names + values
    if dtype.kind in "ui":
        bits0 = 0x08 if dtype.kind == "i" else 0x00                     # bit 3: signed; bit 0: little endian
        head = struct.pack("<BBBBI", (1 << 4) | 0, bits0, 0, 0, dtype.itemsize)
        return head + struct.pack("<HH", 0, 8 * dtype.itemsize)         # bit offset, precision
    if dtype.kind == "f" and dtype.itemsize in (4, 8):
        if dtype.itemsize == 4:
            bits = (0x20, 31, 0)        # LE, mantissa normalisation: implied msb; sign bit 31
            props = struct.pack("<HHBBBBI", 0, 32, 23, 8, 0, 23, 127)
        else:
            bits = (0x20, 63, 0)
            props = struct.pack("<HHBBBBI", 0, 64, 52, 11, 0, 52, 1023)
        head = struct.pack("<BBBBI", (1 << 4) | 1, bits[0], bits[1], bits[2], dtype.itemsize)
        return head + props
    raise TypeError(f"hdf5 writer: unsupported dtype {dtype}")


def _message(mtype: int, body: bytes, flags: int = 0) -> bytes:
    body = _pad8(body)
    return struct.pack("<HHBBBB", mtype, len(body), flags, 0, 0, 0) + body


def _object_header(messages) -> bytes:
    data = b"".join(messages)
    # version 1 prefix: version, reserved, #messages, reference count, header data size, 4 bytes alignment pad
    return struct.pack("<BBHII", 1, 0, len(messages), 1, len(data)) + b"\x00" * 4 + data


class H5Writer:
    """`with H5Writer(path) as f: f.create_dataset('images', 'normal_001', array)`."""

    def __init__(self, path):
        self.path = str(path)
        self.f = open(self.path, "wb")
        self.f.write(b"\x00" * 96)                   # superblock + root symbol-table entry, filled by close()
        self.groups = {}                             # group name -> {dataset name: object header address}
        self.closed = False

    # ---- low level ----------------------------------------------------------------------------------
    def _align(self):
        pos = self.f.tell()
        if pos % 8:
            self.f.write(b"\x00" * (8 - pos % 8))
        return self.f.tell()

    def _write(self, blob: bytes) -> int:
        addr = self._align()
        self.f.write(blob)
        return addr

    # ---- public -------------------------------------------------------------------------------------
    def create_group(self, name: str):
        if "/" in name or not name:
            raise ValueError("hdf5 writer: only single-level group names are supported")
        self.groups.setdefault(name, {})

    def create_dataset(self, group: str, name: str, data):
        """Append one contiguous dataset `/<group>/<name>` (written to disk immediately)."""
        if self.closed:
            raise ValueError("hdf5 writer: file is closed")
        self.create_group(group)
        if name in self.groups[group]:
            raise ValueError(f"hdf5 writer: dataset {group}/{name} exists")
        arr = np.ascontiguousarray(data)
        if arr.dtype.byteorder == ">":
            arr = arr.astype(arr.dtype.newbyteorder("<"))
        raw = arr.view(np.uint8) if arr.dtype != np.bool_ else arr.astype(np.int8).view(np.uint8)
        data_addr = self._align()
        self.f.write(raw.tobytes())
        nbytes = raw.size
        dims = arr.shape if arr.ndim else (1,)
        dataspace = struct.pack("<BBBBI", 1, len(dims), 0, 0, 0) + b"".join(struct.pack("<Q", d) for d in dims)
        msgs = [
            _message(0x0001, dataspace),
            _message(0x0003, _datatype_message(arr.dtype), flags=1),           # constant message
            _message(0x0005, struct.pack("<BBBB", 2, 2, 2, 0)),                # fill value v2: late alloc, never write, undefined
            _message(0x0008, struct.pack("<BBQQ", 3, 1, data_addr if nbytes else UNDEF, nbytes)),   # contiguous layout v3
        ]
        self.groups[group][name] = self._write(_object_header(msgs))

    def _write_group(self, entries):
        """entries: [(name, object header address, (btree, heap) or None)] -> (header addr, btree addr, heap addr)."""
        entries = sorted(entries, key=lambda e: e[0].encode())
        # local heap data segment: offset 0 = empty string, then the names, each padded to 8 bytes
        seg = bytearray(b"\x00" * 8)
        offs = []
        for name, _, _ in entries:
            offs.append(len(seg))
            seg += _pad8(name.encode() + b"\x00")
        seg_addr = self._write(bytes(seg))
        heap_addr = self._write(b"HEAP" + struct.pack("<BBBBQQQ", 0, 0, 0, 0, len(seg), 1, seg_addr))   # free-list head 1 = "no free block"
        # symbol-table nodes (leaves), up to 2*LEAF_K entries each
        per = 2 * LEAF_K
        chunks = [list(range(i, min(i + per, len(entries)))) for i in range(0, len(entries), per)] or [[]]
        if len(chunks) > 2 * INTERNAL_K:
            raise ValueError("hdf5 writer: more than 32768 objects in one group")
        snods = []
        for ch in chunks:
            body = b"SNOD" + struct.pack("<BBH", 1, 0, len(ch))
            for i in ch:
                name, ohdr, sub = entries[i]
                if sub is None:
                    body += struct.pack("<QQII", offs[i], ohdr, 0, 0) + b"\x00" * 16
                else:
                    body += struct.pack("<QQIIQQ", offs[i], ohdr, 1, 0, sub[0], sub[1])
            body += b"\x00" * (8 + per * 40 - len(body))
            snods.append(self._write(body))
        # one B-tree node, level 0, pointing at the symbol-table nodes
        node = b"TREE" + struct.pack("<BBHQQ", 0, 0, len(snods) if entries else 0, UNDEF, UNDEF)
        keys = [0] + [offs[ch[-1]] if ch else 0 for ch in chunks]
        for i in range(len(snods)):
            node += struct.pack("<QQ", keys[i], snods[i])
        node += struct.pack("<Q", keys[len(snods)])
        node += b"\x00" * (24 + (2 * INTERNAL_K + 1) * 8 + 2 * INTERNAL_K * 8 - len(node))
        btree_addr = self._write(node)
        hdr_addr = self._write(_object_header([_message(0x0011, struct.pack("<QQ", btree_addr, heap_addr))]))
        return hdr_addr, btree_addr, heap_addr

    def close(self):
        if self.closed:
            return
        root_entries = []
        for gname, dsets in self.groups.items():
            hdr, bt, hp = self._write_group([(n, a, None) for n, a in dsets.items()])
            root_entries.append((gname, hdr, (bt, hp)))
        root_hdr, root_bt, root_hp = self._write_group(root_entries)
        eof = self._align()
        sb = b"\x89HDF\r\n\x1a\n" + struct.pack("<BBBBBBBBHHI", 0, 0, 0, 0, 0, 8, 8, 0, LEAF_K, INTERNAL_K, 0)
        sb += struct.pack("<QQQQ", 0, UNDEF, eof, UNDEF)
        sb += struct.pack("<QQIIQQ", 0, root_hdr, 1, 0, root_bt, root_hp)      # root group symbol-table entry
        assert len(sb) == 96
        self.f.seek(0)
        self.f.write(sb)
        self.f.close()
        self.closed = True

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def write_hdf5(path, groups):
    """groups: {group name: {dataset name: ndarray}}."""
    with H5Writer(path) as f:
        for g, dsets in groups.items():
            f.create_group(g)
            for n, a in dsets.items():
                f.create_dataset(g, n, a)
    return str(path)


def convert_npy_to_hdf5(encodings_root, out_path=None):
    """Mirror of scripts/convert_npy_embeddings_to_hdf5/convert.py:27-32: every sub-directory of
    `encodings_root` (`images`, `masks`) becomes a group, every `<stem>.npy` a dataset `<stem>`."""
    root = Path(encodings_root)
    out_path = Path(out_path) if out_path is not None else root.with_suffix(".hdf5")
    with H5Writer(out_path) as f:
        for sub in sorted(p for p in root.iterdir() if p.is_dir()):
            f.create_group(sub.name)
            for npy in sorted(sub.glob("*.npy")):
                f.create_dataset(sub.name, npy.stem, np.load(str(npy), allow_pickle=False))
    return str(out_path)


# ---------------------------------------------------------------------------------------------------
# reader: the subset above plus what h5py's default (libver='earliest') writer emits for the same
# content -- object-header continuation blocks, multi-level group B-trees, compact layout.
# ---------------------------------------------------------------------------------------------------
class H5Reader:
    """`H5Reader(path)['images']['normal_001']` -> ndarray; groups behave like read-only dicts
    (the access pattern of datamodules/camelyon16.py:226-235)."""

    def __init__(self, path):
        with open(path, "rb") as fh:
            self.buf = fh.read()
        b = self.buf
        if b[:8] != b"\x89HDF\r\n\x1a\n":
            raise ValueError("hdf5 reader: bad signature")
        if b[8] != 0 or b[13] != 8 or b[14] != 8:
            raise ValueError("hdf5 reader: only superblock version 0 with 8-byte offsets is supported")
        self.base = struct.unpack_from("<Q", b, 24)[0]
        _, root_hdr, cache, _, bt, hp = struct.unpack_from("<QQIIQQ", b, 56)
        self.root = self._group(root_hdr)

    def _messages(self, addr):
        b = self.buf
        ver, _, nmsg, _, size = struct.unpack_from("<BBHII", b, addr)
        if ver != 1:
            raise ValueError("hdf5 reader: only version-1 object headers are supported")
        blocks = [(addr + 16, size)]
        out = []
        while blocks and len(out) < nmsg:
            pos, left = blocks.pop(0)
            end = pos + left
            while pos + 8 <= end and len(out) < nmsg:
                mtype, msize, flags = struct.unpack_from("<HHB", b, pos)
                body = b[pos + 8: pos + 8 + msize]
                pos += 8 + msize
                if mtype == 0x0010:                                   # continuation
                    off, ln = struct.unpack_from("<QQ", body, 0)
                    blocks.append((self.base + off, ln))
                out.append((mtype, body))
        return out

    def _heap_name(self, heap_addr, off):
        b = self.buf
        if b[heap_addr:heap_addr + 4] != b"HEAP":
            raise ValueError("hdf5 reader: bad local heap")
        seg = self.base + struct.unpack_from("<Q", b, heap_addr + 24)[0]
        end = b.index(b"\x00", seg + off)
        return b[seg + off:end].decode()

    def _btree_entries(self, node_addr, heap_addr, out):
        b = self.buf
        sig = b[node_addr:node_addr + 4]
        if sig == b"TREE":
            ntype, level, used = struct.unpack_from("<BBH", b, node_addr + 4)
            for i in range(used):
                child = struct.unpack_from("<Q", b, node_addr + 24 + 8 + 16 * i)[0]
                self._btree_entries(self.base + child, heap_addr, out)
        elif sig == b"SNOD":
            n = struct.unpack_from("<H", b, node_addr + 6)[0]
            for i in range(n):
                name_off, ohdr = struct.unpack_from("<QQ", b, node_addr + 8 + 40 * i)
                out[self._heap_name(heap_addr, name_off)] = self.base + ohdr
        else:
            raise ValueError("hdf5 reader: bad group node")

    def _group(self, hdr_addr):
        for mtype, body in self._messages(hdr_addr):
            if mtype == 0x0011:
                bt, hp = struct.unpack_from("<QQ", body, 0)
                entries = {}
                self._btree_entries(self.base + bt, self.base + hp, entries)
                return _H5Group(self, entries)
        return None

    @staticmethod
    def _dtype(body):
        cls, ver = body[0] & 0x0F, body[0] >> 4
        size = struct.unpack_from("<I", body, 4)[0]
        if cls == 0:
            return np.dtype(("<i" if body[1] & 0x08 else "<u") + str(size)), 12
        if cls == 1:
            return np.dtype("<f" + str(size)), 20
        if cls == 8:
            base, used = H5Reader._dtype(body[8:])
            nmemb = body[1] | (body[2] << 8)
            pos, names = 8 + used, []
            for _ in range(nmemb):
                end = body.index(b"\x00", pos)
                names.append(body[pos:end])
                pos = end + 1 if ver >= 3 else pos + (end - pos) // 8 * 8 + 8
            vals = np.frombuffer(body, base, nmemb, pos)
            if sorted(zip(names, vals.tolist())) == [(b"FALSE", 0), (b"TRUE", 1)] and size == 1:
                return np.dtype(np.bool_), pos + nmemb * size
            return base, pos + nmemb * size
        raise TypeError(f"hdf5 reader: unsupported datatype class {cls}")

    def _object(self, hdr_addr):
        grp = self._group(hdr_addr)
        if grp is not None:
            return grp
        shape = dtype = data = None
        for mtype, body in self._messages(hdr_addr):
            if mtype == 0x0001:
                rank = body[1]
                off = 8 if body[0] == 1 else 4
                shape = struct.unpack_from("<" + "Q" * rank, body, off)
            elif mtype == 0x0003:
                dtype, _ = self._dtype(body)
            elif mtype == 0x0008:
                if body[0] != 3:
                    raise ValueError("hdf5 reader: only version-3 data layouts are supported")
                if body[1] == 1:
                    addr, nbytes = struct.unpack_from("<QQ", body, 2)
                    data = b"" if addr == UNDEF else self.buf[self.base + addr: self.base + addr + nbytes]
                elif body[1] == 0:
                    nbytes = struct.unpack_from("<H", body, 2)[0]
                    data = body[4:4 + nbytes]
                else:
                    raise ValueError("hdf5 reader: chunked datasets are not supported")
        if shape is None or dtype is None or data is None:
            raise ValueError("hdf5 reader: incomplete dataset header")
        n = int(np.prod(shape, dtype=np.int64))
        if len(data) < n * dtype.itemsize:                            # storage never allocated
            return np.zeros(shape, dtype)
        return np.frombuffer(data, dtype, n).reshape(shape).copy()

    def __getitem__(self, key):
        obj = self.root
        for part in key.strip("/").split("/"):
            obj = obj[part]
        return obj

    def keys(self):
        return self.root.keys()


class _H5Group:
    def __init__(self, reader, entries):
        self._r, self._e = reader, entries

    def keys(self):
        return list(self._e.keys())

    def __contains__(self, k):
        return k in self._e

    def __len__(self):
        return len(self._e)

    def __getitem__(self, k):
        return self._r._object(self._e[k])


def read_hdf5(path):
    """-> {group: {name: ndarray}} for a two-level file."""
    r = H5Reader(path)
    return {g: {n: r[g][n] for n in r[g].keys()} for g in r.keys()}
