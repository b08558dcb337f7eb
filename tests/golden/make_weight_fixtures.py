#!/usr/bin/env python3
"""Generates the weight-container fixtures of tests/test_weight_import.py with INDEPENDENT producers:

  tiny_weights.caffemodel.h5   written by the real libhdf5 (1.10.6, /opt/conda/lib, through ctypes) with exactly the calls
                               Caffe's Net::ToHDF5 makes: H5Gcreate2 per group, H5LTmake_dataset_float per blob,
                               H5Lcreate_soft for a shared parameter;
  tiny_weights.caffemodel      serialised by google.protobuf from a descriptor holding the public caffe.proto field
                               numbers (NetParameter.layer = 100 / .layers = 2, LayerParameter.blobs = 7, BlobProto.data = 5,
                               .shape = 7, legacy num/channels/height/width = 1..4);
  tiny_weights.npz             the arrays themselves (the expected values).

Run here (needs /opt/conda/lib/libhdf5 and google.protobuf); the outputs are committed, the tests only read them."""
import ctypes
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def arrays():
    rng = np.random.default_rng(2024)
    A = {
        "conv1_1": [rng.normal(size=(32, 3, 3, 3)).astype(np.float32), rng.normal(size=(32,)).astype(np.float32)],
        "conv1_2": [rng.normal(size=(32, 32, 3, 3)).astype(np.float32), rng.normal(size=(32,)).astype(np.float32)],
        "rpn/cls_score": [rng.normal(size=(18, 32, 1, 1)).astype(np.float32), rng.normal(size=(18,)).astype(np.float32)],
        "fc6": [rng.normal(size=(24, 392)).astype(np.float32), rng.normal(size=(24,)).astype(np.float32)],
    }
    for i in range(70):      # enough siblings under /data for libhdf5 to split symbol-table nodes and grow the group B-tree
        A["extra_%02d" % i] = [rng.normal(size=(2, 3)).astype(np.float32), rng.normal(size=(2,)).astype(np.float32)]
    A["seg_cls_score"] = [rng.normal(size=(21, 80)).astype(np.float32), np.zeros((21,), np.float32)]
    return A


SHARED = {"fc6_mask": "fc6", "seg_cls_score_ext": "seg_cls_score"}     # soft-linked parameters (MNC's stage sharing)


def write_hdf5(path, A):
    h5 = ctypes.CDLL("/opt/conda/lib/libhdf5.so.103", mode=ctypes.RTLD_GLOBAL)
    hl = ctypes.CDLL("/opt/conda/lib/libhdf5_hl.so.100")
    hid = ctypes.c_int64
    h5.H5open()
    h5.H5Fcreate.restype = hid
    h5.H5Fcreate.argtypes = [ctypes.c_char_p, ctypes.c_uint, hid, hid]
    h5.H5Gcreate2.restype = hid
    h5.H5Gcreate2.argtypes = [hid, ctypes.c_char_p, hid, hid, hid]
    hl.H5LTmake_dataset_float.argtypes = [hid, ctypes.c_char_p, ctypes.c_int, ctypes.POINTER(ctypes.c_uint64), ctypes.c_void_p]
    h5.H5Lcreate_soft.argtypes = [ctypes.c_char_p, hid, ctypes.c_char_p, hid, hid]
    h5.H5Gclose.argtypes = [hid]
    h5.H5Fclose.argtypes = [hid]
    f = h5.H5Fcreate(path.encode(), 2, 0, 0)                               # H5F_ACC_TRUNC, default property lists
    assert f > 0
    data = h5.H5Gcreate2(f, b"data", 0, 0, 0)
    diff = h5.H5Gcreate2(f, b"diff", 0, 0, 0)                              # Net::ToHDF5 always creates /diff too
    open_groups = [data, diff]

    def layer_group(name):
        g = data
        for part in name.split("/"):                                        # '/' in a layer name nests groups
            g = h5.H5Gcreate2(g, part.encode(), 0, 0, 0)
            assert g > 0
            open_groups.append(g)
        return g

    for name, blobs in A.items():
        g = layer_group(name)
        for i, b in enumerate(blobs):
            b = np.ascontiguousarray(b)
            dims = (ctypes.c_uint64 * b.ndim)(*b.shape)
            assert hl.H5LTmake_dataset_float(g, str(i).encode(), b.ndim, dims, b.ctypes.data) >= 0
    for name, owner in SHARED.items():
        g = layer_group(name)
        for i in range(len(A[owner])):
            assert h5.H5Lcreate_soft(("/data/%s/%d" % (owner, i)).encode(), g, str(i).encode(), 0, 0) >= 0
    for g in reversed(open_groups):
        h5.H5Gclose(g)
    h5.H5Fclose(f)


def write_caffemodel(path, A):
    from google.protobuf import descriptor_pb2, descriptor_pool, message_factory
    fd = descriptor_pb2.FileDescriptorProto(name="caffe_subset.proto", package="caffe", syntax="proto2")
    T = descriptor_pb2.FieldDescriptorProto

    def msg(name, fields):
        m = fd.message_type.add(name=name)
        for fname, num, typ, label, tname, packed in fields:
            f = m.field.add(name=fname, number=num, type=typ, label=label)
            if tname:
                f.type_name = ".caffe." + tname
            if packed:
                f.options.packed = True
    OPT, REP = T.LABEL_OPTIONAL, T.LABEL_REPEATED
    msg("BlobShape", [("dim", 1, T.TYPE_INT64, REP, None, True)])
    msg("BlobProto", [("shape", 7, T.TYPE_MESSAGE, OPT, "BlobShape", False), ("data", 5, T.TYPE_FLOAT, REP, None, True),
                      ("double_data", 8, T.TYPE_DOUBLE, REP, None, True), ("num", 1, T.TYPE_INT32, OPT, None, False),
                      ("channels", 2, T.TYPE_INT32, OPT, None, False), ("height", 3, T.TYPE_INT32, OPT, None, False),
                      ("width", 4, T.TYPE_INT32, OPT, None, False)])
    msg("LayerParameter", [("name", 1, T.TYPE_STRING, OPT, None, False), ("type", 2, T.TYPE_STRING, OPT, None, False),
                           ("bottom", 3, T.TYPE_STRING, REP, None, False), ("top", 4, T.TYPE_STRING, REP, None, False),
                           ("blobs", 7, T.TYPE_MESSAGE, REP, "BlobProto", False)])
    msg("V1LayerParameter", [("name", 4, T.TYPE_STRING, OPT, None, False), ("type", 5, T.TYPE_INT32, OPT, None, False),
                             ("blobs", 6, T.TYPE_MESSAGE, REP, "BlobProto", False)])
    msg("NetParameter", [("name", 1, T.TYPE_STRING, OPT, None, False),
                         ("layers", 2, T.TYPE_MESSAGE, REP, "V1LayerParameter", False),
                         ("input", 3, T.TYPE_STRING, REP, None, False),
                         ("layer", 100, T.TYPE_MESSAGE, REP, "LayerParameter", False)])
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    Net = message_factory.GetMessageClass(pool.FindMessageTypeByName("caffe.NetParameter"))
    net = Net(name="tiny")
    net.input.append("data")
    names = list(A)
    for k, name in enumerate(names):
        blobs = A[name]
        if k == len(names) - 1:                  # one legacy V1 layer with 4-D (num, channels, height, width) blob shapes
            L = net.layers.add(name=name, type=14)
            w = L.blobs.add(num=1, channels=1, height=blobs[0].shape[0], width=blobs[0].shape[1])
            w.data.extend(blobs[0].ravel().tolist())
            b = L.blobs.add(num=1, channels=1, height=1, width=blobs[1].shape[0])
            b.double_data.extend(blobs[1].astype(np.float64).ravel().tolist())      # and a double_data blob
            continue
        L = net.layer.add(name=name, type="Convolution" if blobs[0].ndim == 4 else "InnerProduct")
        L.bottom.append("x")
        L.top.append(name)
        for arr in blobs:
            bp = L.blobs.add()
            bp.shape.dim.extend(arr.shape)
            bp.data.extend(arr.ravel().tolist())
    relu = net.layer.add(name="relu1_1", type="ReLU")                               # a layer without blobs
    relu.bottom.append("conv1_1")
    with open(path, "wb") as f:
        f.write(net.SerializeToString())


if __name__ == "__main__":
    A = arrays()
    flat = {"%s/%d" % (k, i): b for k, v in A.items() for i, b in enumerate(v)}
    np.savez(os.path.join(HERE, "tiny_weights.npz"), **flat)
    write_hdf5(os.path.join(HERE, "tiny_weights.caffemodel.h5"), A)
    write_caffemodel(os.path.join(HERE, "tiny_weights.caffemodel"), A)
    for n in ("tiny_weights.npz", "tiny_weights.caffemodel.h5", "tiny_weights.caffemodel"):
        print(n, os.path.getsize(os.path.join(HERE, n)))
