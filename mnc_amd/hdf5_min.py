"""Minimal pure-Python reader for the HDF5 files Caffe writes (`Net::ToHDF5`, used by the reference for its snapshots
because of the shared parameters: lib/caffeWrapper/SolverWrapper.py:105-114; tools/demo.py:46-48 loads
`mnc_model.caffemodel.h5`).  h5py is not a dependency of this package.

Supported subset = what libhdf5 (1.8 / 1.10, default property lists) produces for Caffe's calls
(`H5Gcreate2`, `H5LTmake_dataset_float/double/int`, `H5Lcreate_soft`):
  superblock version 0/1; old-style groups (symbol-table message -> v1 B-tree + local heap + SNOD nodes), including soft
  links; version-1 object headers with continuation blocks; simple dataspaces (v1/v2); little-endian float32/float64 and
  integer datatypes; contiguous and compact data layouts (layout message v3).  Anything else raises Hdf5Error naming it.

    read_tree(path) -> {"data/conv1_1/0": ndarray, ...}      every dataset, soft links resolved, keyed by its path
"""

import numpy as np

_SIG = b"\x89HDF\r\n\x1a\n"
_UNDEF = 0xFFFFFFFFFFFFFFFF


class Hdf5Error(ValueError):
    pass


class _File(object):
    def __init__(self, buf):
        self.b = buf
        base = -1
        for off in [0] + [512 << i for i in range(16)]:          # the superblock may sit at 0, 512, 1024, ...
            if buf[off:off + 8] == _SIG:
                base = off
                break
        if base < 0:
            raise Hdf5Error("not an HDF5 file (no signature)")
        ver = buf[base + 8]
        if ver not in (0, 1):
            raise Hdf5Error("superblock version %d is not supported (only 0/1, what libhdf5 writes by default)" % ver)
        self.O, self.L = buf[base + 13], buf[base + 14]
        if self.O not in (4, 8) or self.L not in (4, 8):
            raise Hdf5Error("unsupported offset/length sizes %d/%d" % (self.O, self.L))
        p = base + 24 + (4 if ver == 1 else 0)
        self.base = self.uint(p, self.O)
        p += 4 * self.O                                         # base, free-space info, end of file, driver info
        # root group symbol table entry
        self.root_header = self.uint(p + self.O, self.O)

    def uint(self, off, n):
        return int.from_bytes(self.b[off:off + n], "little")

    def addr(self, off):
        return self.uint(off, self.O) + self.base

    # ---- object headers ----------------------------------------------------------------------------------------
    def messages(self, header_addr):
        """[(type, data_offset, size)] of a version-1 object header, following continuation messages."""
        b = self.b
        if b[header_addr] != 1:
            if b[header_addr:header_addr + 4] == b"OHDR":
                raise Hdf5Error("version-2 object headers (libver 'latest' files) are not supported")
            raise Hdf5Error("object header version %d is not supported" % b[header_addr])
        nmsg = self.uint(header_addr + 2, 2)
        size = self.uint(header_addr + 8, 4)
        blocks = [(header_addr + 16, size)]
        out = []
        while blocks and len(out) < nmsg:
            p, left = blocks.pop(0)
            end = p + left
            while p + 8 <= end and len(out) < nmsg:
                mtype, msize = self.uint(p, 2), self.uint(p + 2, 2)
                data = p + 8
                out.append((mtype, data, msize))
                if mtype == 0x0010:
                    blocks.append((self.addr(data), self.uint(data + self.O, self.L)))
                p = data + msize
        return out

    # ---- groups ------------------------------------------------------------------------------------------------
    def heap_string(self, heap_addr, offset):
        if self.b[heap_addr:heap_addr + 4] != b"HEAP":
            raise Hdf5Error("bad local heap signature")
        seg = self.addr(heap_addr + 8 + 2 * self.L)
        start = seg + offset
        end = self.b.index(b"\0", start)
        return self.b[start:end].decode("utf-8")

    def group_entries(self, btree_addr, heap_addr):
        """[(name, header_addr | None, soft_link_target | None)] of an old-style group."""
        b = self.b
        if b[btree_addr:btree_addr + 4] != b"TREE":
            raise Hdf5Error("bad B-tree signature")
        if b[btree_addr + 4] != 0:
            raise Hdf5Error("B-tree node type %d where a group node was expected" % b[btree_addr + 4])
        level, used = b[btree_addr + 5], self.uint(btree_addr + 6, 2)
        p = btree_addr + 8 + 2 * self.O
        out = []
        for i in range(used):
            child = self.addr(p + self.L + i * (self.L + self.O))
            if level > 0:
                out.extend(self.group_entries(child, heap_addr))
                continue
            if b[child:child + 4] != b"SNOD":
                raise Hdf5Error("bad symbol table node signature")
            nsym = self.uint(child + 6, 2)
            e = child + 8
            for _ in range(nsym):
                name = self.heap_string(heap_addr, self.uint(e, self.O))
                header = self.uint(e + self.O, self.O)
                cache = self.uint(e + 2 * self.O, 4)
                scratch = e + 2 * self.O + 8
                if cache == 2:                                      # soft link: scratch = heap offset of the target path
                    out.append((name, None, self.heap_string(heap_addr, self.uint(scratch, 4))))
                else:
                    out.append((name, header + self.base, None))
                e += 2 * self.O + 24
        return out

    # ---- datasets ----------------------------------------------------------------------------------------------
    def dataset(self, msgs, path):
        b = self.b
        shape = dtype = None
        data = None
        for mtype, p, size in msgs:
            if mtype == 0x0001:                                     # dataspace
                ver, rank = b[p], b[p + 1]
                if ver == 1:
                    q = p + 8
                elif ver == 2:
                    if b[p + 3] == 2:
                        raise Hdf5Error("%s: null dataspace" % path)
                    q = p + 4
                else:
                    raise Hdf5Error("%s: dataspace message version %d" % (path, ver))
                shape = tuple(self.uint(q + i * self.L, self.L) for i in range(rank))
            elif mtype == 0x0003:                                   # datatype
                cls, bits0 = b[p] & 0x0F, b[p + 1]
                nbytes = self.uint(p + 4, 4)
                if bits0 & 1:
                    raise Hdf5Error("%s: big-endian data is not supported" % path)
                if cls == 1 and nbytes in (4, 8):
                    dtype = np.dtype("<f%d" % nbytes)
                elif cls == 0 and nbytes in (1, 2, 4, 8):
                    dtype = np.dtype("<%s%d" % ("i" if bits0 & 8 else "u", nbytes))
                else:
                    raise Hdf5Error("%s: datatype class %d size %d is not supported" % (path, cls, nbytes))
            elif mtype == 0x0008:                                   # data layout
                ver, lcls = b[p], b[p + 1]
                if ver != 3:
                    raise Hdf5Error("%s: data layout message version %d is not supported" % (path, ver))
                if lcls == 1:
                    data = ("contiguous", self.uint(p + 2, self.O), self.uint(p + 2 + self.O, self.L))
                elif lcls == 0:
                    n = self.uint(p + 2, 2)
                    data = ("compact", p + 4, n)
                else:
                    raise Hdf5Error("%s: chunked datasets are not supported (Caffe writes contiguous ones)" % path)
        if shape is None or dtype is None or data is None:
            raise Hdf5Error("%s: incomplete dataset header" % path)
        count = int(np.prod(shape)) if shape else 1
        kind, off, n = data
        if kind == "contiguous":
            if off == _UNDEF or off == (1 << (8 * self.O)) - 1:
                return np.zeros(shape, dtype)                       # never written
            off += self.base
        if n < count * dtype.itemsize:
            raise Hdf5Error("%s: %d data bytes for shape %r" % (path, n, shape))
        return np.frombuffer(self.b, dtype, count, off).reshape(shape).copy()

    def walk(self, header_addr, prefix, out, links):
        msgs = self.messages(header_addr)
        sym = [m for m in msgs if m[0] == 0x0011]
        if sym:
            p = sym[0][1]
            for name, child, target in self.group_entries(self.addr(p), self.addr(p + self.O)):
                path = prefix + name
                if target is not None:
                    links[path] = target
                else:
                    self.walk(child, path + "/", out, links)
            return
        if any(m[0] in (0x0002, 0x0006) for m in msgs):
            raise Hdf5Error("%s: new-style (link message) groups are not supported" % prefix)
        if any(m[0] == 0x0008 for m in msgs):
            out[prefix[:-1]] = self.dataset(msgs, prefix[:-1])


def read_tree(path):
    """Every dataset of the file as {"group/.../name": ndarray}; soft links resolve to (copies of) their targets."""
    with open(path, "rb") as f:
        buf = f.read()
    h = _File(buf)
    out, links = {}, {}
    h.walk(h.root_header + h.base, "", out, links)
    for _ in range(8):                                              # links to links
        pending = {}
        for name, target in links.items():
            t = target.lstrip("/")
            hits = {k: v for k, v in out.items() if k == t or k.startswith(t + "/")}
            if not hits and t in links:
                pending[name] = links[t]
                continue
            if not hits:
                raise Hdf5Error("soft link %s -> %s does not resolve" % (name, target))
            for k, v in hits.items():
                out[name + k[len(t):]] = v
        links = pending
        if not links:
            break
    return out


def read_caffe_weights(path):
    """`Net::ToHDF5` layout /data/<layer>/<index> -> {"<layer>/<index>": ndarray}.  Layer names may contain '/', which Caffe
    writes as nested groups; the parameter index is the last path component."""
    tree = read_tree(path)
    out = {}
    for k, v in tree.items():
        if k.startswith("data/"):
            out[k[len("data/"):]] = v
    if not out:
        raise Hdf5Error("%s has no /data group: not a Caffe HDF5 weight file" % path)
    return out
