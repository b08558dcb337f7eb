"""CPU: the graph emitted by mnc_amd.models equals the reference's models/VGG16/mnc_5stage/test.prototxt layer by
layer (only where /root/reference is mounted), and the prototxt parser / shape tracer behave."""
import os

import numpy as np
import pytest

from mnc_amd import models, prototxt, synth

REF = "/root/reference/models/VGG16/mnc_5stage/test.prototxt"


def _norm(msg):
    """Layer -> comparable tuple of everything the inference engine reads."""
    def hp(m):
        if m is None:
            return None
        return tuple(sorted((k, tuple(hp(x) if isinstance(x, dict) else x for x in v)) for k, v in m.items()))
    keys = ("convolution_param", "pooling_param", "inner_product_param", "roi_warping_param", "roi_pooling_param",
            "mask_resize_param", "reshape_param", "python_param", "concat_param")
    out = {}
    for k in keys:
        m = msg.get1(k)
        if m is not None:
            d = dict(m)
            if k == "convolution_param":
                d = {a: d[a] for a in ("num_output", "kernel_size", "pad", "stride") if a in d}
                d.setdefault("stride", [1]); d.setdefault("pad", [0])
            if k == "pooling_param":
                d.setdefault("pad", [0])
            out[k] = hp(prototxt.Message(d))
    return (msg.get1("name"), msg.get1("type"), tuple(msg.all("bottom")), tuple(msg.all("top")),
            tuple(p.get1("name") for p in msg.all("param") if p.get1("name")), tuple(sorted(out.items())))


@pytest.mark.skipif(not os.path.isfile(REF), reason="reference checkout not mounted")
def test_emitted_graph_equals_reference():
    ref = prototxt.parse_file(REF)
    mine = prototxt.parse(models.mnc_5stage_test_prototxt())
    assert ref.all("input") == mine.all("input")
    a, b = [_norm(l) for l in ref.all("layer")], [_norm(l) for l in mine.all("layer")]
    assert len(a) == len(b) == 88
    for x, y in zip(a, b):
        assert x == y


@pytest.mark.skipif(not os.path.isfile(REF), reason="reference checkout not mounted")
def test_emitted_cfm_graph_equals_reference():
    ref = prototxt.parse_file(REF.replace("mnc_5stage", "cfm"))
    mine = prototxt.parse(models.cfm_test_prototxt())
    assert ref.all("input") == mine.all("input") == ["data", "rois", "masks"]
    a, b = [_norm(l) for l in ref.all("layer")], [_norm(l) for l in mine.all("layer")]
    assert len(a) == len(b)
    for x, y in zip(a, b):
        assert x == y


def test_parser_and_shapes():
    path = models.write_mnc_5stage_test_prototxt()
    shapes = {n: (k, w, b) for n, k, w, b in synth.layer_shapes(path)}
    assert shapes["conv1_1"][1] == (64, 3, 3, 3) and shapes["fc6_maskest"][1] == (256, 100352)
    assert shapes["fc6"][1] == (4096, 25088) and shapes["bbox_pred"][1] == (84, 8192)
    assert "fc6_ext" not in shapes                      # shared by param name with fc6
    assert sum(int(np.prod(w)) for _, w, _ in shapes.values()) == 283007936
    small = synth.layer_shapes(models.write_mnc_5stage_test_prototxt(width_div=8))
    assert dict((n, w) for n, _, w, _ in small)["fc6"] == (512, 64 * 49)


def test_parser_errors():
    with pytest.raises(ValueError):
        prototxt.parse("layer { name: 'x' ")
    m = prototxt.parse("a: 1 a: 2 b { c: 'q' d: MAX e: 0.5 f: true }")
    assert m.all("a") == [1, 2] and m.get1("b").get1("d") == "MAX" and m.get1("b").get1("f") is True
