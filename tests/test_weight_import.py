"""CPU: the weight containers at the input edge of the path: `.caffemodel` (protobuf) and
`.caffemodel.h5` (HDF5, the format the reference snapshots MNC in) are read without protobuf/h5py and checked against
fixtures produced by INDEPENDENT writers -- the real libhdf5 making Caffe's own calls, and google.protobuf
(tests/golden/make_weight_fixtures.py)."""
import os

import numpy as np
import pytest

from mnc_amd import caffemodel, hdf5_min

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SHARED = {"fc6_mask": "fc6", "seg_cls_score_ext": "seg_cls_score"}


@pytest.fixture(scope="module")
def want():
    out = {}
    for k, v in np.load(os.path.join(G, "tiny_weights.npz")).items():
        name, i = k.rsplit("/", 1)
        out.setdefault(name, {})[int(i)] = v
    return {k: [v[i] for i in sorted(v)] for k, v in out.items()}


def test_hdf5_written_by_libhdf5_reads_back_exactly(want):
    got = caffemodel.load_weights(os.path.join(G, "tiny_weights.caffemodel.h5"))
    assert set(got) == set(want) | set(SHARED)                  # incl. the nested-group name 'rpn/cls_score'
    for name, blobs in want.items():
        assert len(got[name]) == len(blobs)
        for a, b in zip(got[name], blobs):
            assert a.dtype == np.float32 and a.shape == b.shape and np.array_equal(a, b), name
    for name, owner in SHARED.items():                          # soft links (shared parameters) resolve to their owners
        for a, b in zip(got[name], want[owner]):
            assert np.array_equal(a, b), name
    tree = hdf5_min.read_tree(os.path.join(G, "tiny_weights.caffemodel.h5"))
    assert "data/conv1_1/0" in tree and not any(k.startswith("diff/") for k in tree)


def test_caffemodel_serialised_by_protobuf_reads_back_exactly(want):
    got = caffemodel.load_weights(os.path.join(G, "tiny_weights.caffemodel"))
    assert set(got) == set(want)                                # 'relu1_1' (no blobs) is not a weight entry
    for name, blobs in want.items():
        for a, b in zip(got[name], blobs):
            assert a.dtype == np.float32 and a.shape == b.shape and np.array_equal(a, b), name
    # seg_cls_score is stored as a legacy V1 layer: [1,1,N,K] weights and a [1,1,1,N] double_data bias
    assert got["seg_cls_score"][0].shape == (21, 80) and got["seg_cls_score"][1].shape == (21,)


def test_npz_roundtrip_and_engine_dict(tmp_path, want):
    path = str(tmp_path / "w.npz")
    caffemodel.save_npz(want, path)
    again = caffemodel.load_weights(path)
    assert set(again) == set(want)
    for name in want:
        for a, b in zip(again[name], want[name]):
            assert np.array_equal(a, b)
    from mnc_amd.engine import Net
    for src in ("tiny_weights.caffemodel.h5", "tiny_weights.caffemodel", "tiny_weights.npz"):
        w = Net._read_weights(os.path.join(G, src))             # what caffe.Net(prototxt, path, TEST) does with a path
        assert np.array_equal(w["fc6"][0], want["fc6"][0]) and np.array_equal(w["conv1_2"][1], want["conv1_2"][1])


def test_malformed_containers_fail_loudly(tmp_path):
    bad = tmp_path / "x.caffemodel"
    bad.write_bytes(b"\x0a\xff\xff\xff\xff\x0f")                 # length-delimited field running past the end
    with pytest.raises(caffemodel.CaffemodelError):
        caffemodel.load_weights(str(bad))
    empty = tmp_path / "e.caffemodel"
    empty.write_bytes(b"\x0a\x04tiny")                           # a NetParameter with a name and no layers
    with pytest.raises(caffemodel.CaffemodelError):
        caffemodel.load_weights(str(empty))
    noth5 = tmp_path / "n.h5"
    noth5.write_bytes(b"not hdf5 at all" * 100)
    with pytest.raises(hdf5_min.Hdf5Error):
        caffemodel.load_weights(str(noth5))
    data = bytearray(open(os.path.join(G, "tiny_weights.caffemodel.h5"), "rb").read())
    data[8] = 2                                                   # superblock version 2 (libver=latest): unsupported, says so
    v2 = tmp_path / "v2.h5"
    v2.write_bytes(bytes(data))
    with pytest.raises(hdf5_min.Hdf5Error, match="superblock version 2"):
        caffemodel.load_weights(str(v2))
    with pytest.raises(ValueError):
        caffemodel.load_weights("weights.bin")


@pytest.mark.skipif(not os.path.exists("/opt/conda/lib/libhdf5.so.103"), reason="needs the image's libhdf5 to write the file")
def test_hdf5_group_with_a_two_level_btree(tmp_path):
    """400 layers under /data: libhdf5 splits the group's B-tree (32 children per node by default) into two levels."""
    import sys
    sys.path.insert(0, G)
    import make_weight_fixtures as mk
    rng = np.random.default_rng(7)
    A = {"layer_%03d" % i: [rng.normal(size=(3, 2)).astype(np.float32)] for i in range(400)}
    path = str(tmp_path / "many.h5")
    saved = dict(mk.SHARED)
    mk.SHARED.clear()
    try:
        mk.write_hdf5(path, A)
    finally:
        mk.SHARED.update(saved)
    f = hdf5_min._File(open(path, "rb").read())
    got = caffemodel.load_weights(path)
    assert len(got) == 400
    for k, v in A.items():
        assert np.array_equal(got[k][0], v[0])
