"""CPU: the C restatement (oracle/mnc_oracle.c) against the reference's own nms_kernel.cu / mv_kernel.cu compiled
for the CPU (oracle/_ref/libmnc_ref.so, built by oracle/build_ref.py where /root/reference is mounted; the
prebuilt library travels to the GPU box).  Bit-exact."""
import numpy as np
import pytest

import golden_inputs as GI
from oracle import native

pytestmark = pytest.mark.skipif(not native.ref_available(), reason="oracle/_ref/libmnc_ref.so not built")


@pytest.mark.parametrize("n,thr,seed", [(600, 0.3, 100), (2500, 0.7, 101), (129, 0.5, 102), (63, 0.9, 103)])
def test_nms(n, thr, seed):
    d = GI.nms_case(n, seed)
    assert native.gpu_nms(d, thr) == native.ref_gpu_nms(d, thr)


def test_mv_small_canvas():
    mc = GI.mv_case(55)
    a = native.mv(mc["boxes"], mc["masks"], mc["inds"], mc["start"], mc["weights"], mc["H"], mc["W"])
    b = native.ref_mv(mc["boxes"], mc["masks"], mc["inds"], mc["start"], mc["weights"], mc["H"], mc["W"])
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
