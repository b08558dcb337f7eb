"""Weight containers at the input edge of the path.

The reference loads `caffe.Net(prototxt, caffemodel, caffe.TEST)` with either container Caffe can write
(tools/demo.py:46-48,129; lib/caffeWrapper/SolverWrapper.py:100-114):
  * `*.caffemodel`     binary protobuf `NetParameter` (`net.save`), one copy of the blobs per layer NAME;
  * `*.caffemodel.h5`  HDF5 (`net.save_to_hdf5`), /data/<layer>/<index>, shared parameters as soft links -- the format the
                       reference uses for MNC because of its shared parameters.
Both are read here without protobuf / h5py: a 60-line wire-format walker for the few `caffe.proto` fields that carry
weights, and mnc_amd/hdf5_min.py.  `load_weights` returns the engine's weight dict {"<layer>": [W, b, ...]}.

Field numbers (public BVLC caffe.proto, unchanged in the caffe-mnc fork):
  NetParameter   : layers = 2 (V1LayerParameter, legacy), layer = 100 (LayerParameter)
  LayerParameter : name = 1, type = 2, blobs = 7          V1LayerParameter : name = 4, blobs = 6
  BlobProto      : num/channels/height/width = 1..4 (legacy shape), data = 5 (packed float), shape = 7 (BlobShape),
                   double_data = 8 (packed double)        BlobShape : dim = 1 (packed int64)
"""
import numpy as np


class CaffemodelError(ValueError):
    pass


def _varint(buf, p):
    r = s = 0
    while True:
        if p >= len(buf):
            raise CaffemodelError("truncated varint")
        c = buf[p]
        p += 1
        r |= (c & 0x7F) << s
        if c < 0x80:
            return r, p
        s += 7
        if s > 70:
            raise CaffemodelError("varint too long")


def _fields(buf):
    """Yield (field_number, wire_type, value) of one message; length-delimited values are memoryviews."""
    p, n = 0, len(buf)
    while p < n:
        key, p = _varint(buf, p)
        num, wt = key >> 3, key & 7
        if wt == 0:
            v, p = _varint(buf, p)
        elif wt == 1:
            v, p = bytes(buf[p:p + 8]), p + 8
        elif wt == 2:
            ln, p = _varint(buf, p)
            if p + ln > n:
                raise CaffemodelError("length-delimited field runs past the end of its message")
            v, p = buf[p:p + ln], p + ln
        elif wt == 5:
            v, p = bytes(buf[p:p + 4]), p + 4
        else:
            raise CaffemodelError("unsupported protobuf wire type %d" % wt)
        yield num, wt, v


def _packed_varints(v):
    out, p = [], 0
    while p < len(v):
        x, p = _varint(v, p)
        out.append(x)
    return out


def _blob(buf):
    legacy = {}
    shape = None
    chunks, dchunks = [], []
    for num, wt, v in _fields(buf):
        if num in (1, 2, 3, 4) and wt == 0:
            legacy[num] = v
        elif num == 7 and wt == 2:
            dims = []
            for n2, w2, v2 in _fields(v):
                if n2 == 1:
                    dims.extend(_packed_varints(v2) if w2 == 2 else [v2])
            shape = tuple(dims)
        elif num == 5:
            chunks.append(np.frombuffer(v, "<f4") if wt == 2 else np.frombuffer(v, "<f4", 1))
        elif num == 8:
            dchunks.append(np.frombuffer(v, "<f8") if wt == 2 else np.frombuffer(v, "<f8", 1))
    if chunks:
        data = np.concatenate(chunks) if len(chunks) > 1 else chunks[0]
    elif dchunks:
        data = (np.concatenate(dchunks) if len(dchunks) > 1 else dchunks[0]).astype(np.float32)
    else:
        data = np.zeros(0, np.float32)
    if shape is None:
        if legacy:
            # legacy 4-D shape (num, channels, height, width): InnerProduct weights are [1,1,N,K] and every bias [1,1,1,N];
            # drop those leading singleton axes (Caffe's `Blob::ShapeEquals` tolerates them when such a file is loaded)
            shape = tuple(legacy.get(i, 1) for i in (1, 2, 3, 4))
            if shape[0] == 1 and shape[1] == 1:
                shape = shape[2:] if shape[2] != 1 else shape[3:]
        else:
            shape = (data.size,)
    if int(np.prod(shape)) != data.size:
        raise CaffemodelError("blob shape %r does not hold %d values" % (shape, data.size))
    return np.array(data, dtype=np.float32).reshape(shape)


def read_caffemodel(path):
    """Binary NetParameter -> {"<layer>": [blob0, blob1, ...]} for every layer that carries blobs (insertion-ordered)."""
    with open(path, "rb") as f:
        buf = memoryview(f.read())
    out = {}
    for num, wt, v in _fields(buf):
        if wt != 2 or num not in (2, 100):
            continue
        name_field, blob_field = (1, 7) if num == 100 else (4, 6)
        name, blobs = None, []
        for n2, w2, v2 in _fields(v):
            if n2 == name_field and w2 == 2:
                name = bytes(v2).decode("utf-8")
            elif n2 == blob_field and w2 == 2:
                blobs.append(_blob(v2))
        if blobs:
            if name is None:
                raise CaffemodelError("a layer with blobs has no name")
            out[name] = blobs
    if not out:
        raise CaffemodelError("%s holds no layer blobs: not a .caffemodel?" % path)
    return out


def load_weights(path):
    """Any supported container -> {"<layer>": [W, b]} (float32, Caffe layouts)."""
    path = str(path)
    if path.endswith(".npz"):
        src = dict(np.load(path))
    elif path.endswith((".h5", ".hdf5")):
        from . import hdf5_min
        src = hdf5_min.read_caffe_weights(path)
    elif path.endswith(".caffemodel"):
        return read_caffemodel(path)
    else:
        raise ValueError("unsupported weights container %r (.npz, .caffemodel, .caffemodel.h5 / .h5)" % path)
    out = {}
    for k, v in src.items():
        lname, idx = k.rsplit("/", 1)
        lst = out.setdefault(lname, [])
        while len(lst) <= int(idx):
            lst.append(None)
        lst[int(idx)] = np.asarray(v, dtype=np.float32)
    return out


def save_npz(weights, path):
    """{"<layer>": [W, b]} -> the .npz container ('<layer>/<index>' keys)."""
    flat = {}
    for lname, blobs in weights.items():
        for i, b in enumerate(blobs):
            if b is not None:
                flat["%s/%d" % (lname, i)] = np.asarray(b, dtype=np.float32)
    np.savez(path, **flat)


def save_flat(weights, path):
    """{"<layer>": [W, b]} -> the flat little-endian container mnc_net_load_file reads (include/mnc_hip.h): "MNCW0001",
    uint32 n, then n x {uint16 name_len, name, uint8 blob index, uint8 ndim, uint32 dims[ndim], float32 data}.  For hosts that
    load weights without Python (a C / Go / Java caller of mnc_forward_image)."""
    import struct
    entries = [(lname, i, np.ascontiguousarray(b, dtype="<f4")) for lname, blobs in weights.items()
               for i, b in enumerate(blobs) if b is not None]
    with open(path, "wb") as f:
        f.write(b"MNCW0001" + struct.pack("<I", len(entries)))
        for lname, i, a in entries:
            nm = lname.encode()
            f.write(struct.pack("<H", len(nm)) + nm + struct.pack("<BB", i, a.ndim) + struct.pack("<%dI" % a.ndim, *a.shape))
            f.write(a.tobytes())


def load_flat(path):
    """The flat MNCW0001 container (save_flat) -> {"<layer>": [W, b]} as read-only memory maps of the file: nothing is copied,
    ranks that map the same file share its pages (bench.py: one rank synthesises / converts the weights, the others map them)."""
    import struct
    weights = {}
    with open(path, "rb") as f:
        head = f.read(12)
        if head[:8] != b"MNCW0001":
            raise ValueError("%s is not an MNCW0001 file" % path)
        (n,) = struct.unpack("<I", head[8:])
        for _ in range(n):
            (ln,) = struct.unpack("<H", f.read(2))
            name = f.read(ln).decode()
            idx, nd = struct.unpack("<BB", f.read(2))
            dims = struct.unpack("<%dI" % nd, f.read(4 * nd))
            count = int(np.prod(dims)) if nd else 1
            off = f.tell()
            a = np.memmap(path, dtype="<f4", mode="r", offset=off, shape=tuple(dims)) if count else np.zeros(dims, "<f4")
            f.seek(off + 4 * count)
            blobs = weights.setdefault(name, [])
            while len(blobs) <= idx:
                blobs.append(None)
            blobs[idx] = a
    return weights
