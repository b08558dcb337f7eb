"""GPU: the whole-image C entry points (mnc_net_* / mnc_forward_image, csrc/pipeline.hip) against the Python engine running the
prototxt layer by layer (same kernels, same fused plan -> the same bits), against the oracle voting, and from a plain C program
with no Python in the process."""
import ctypes
import os
import shutil
import subprocess

import numpy as np
import pytest

import mnc_amd
from gpu_util import from_c8
from mnc_amd import caffemodel, models, synth
from mnc_amd.native_net import NativeNet
from oracle import host as ohost

pytestmark = pytest.mark.gpu
mnc_amd.install_paths()
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _engine_reference(net, im):
    """tools/demo.py per-image body on the Python engine: (blobs, voting lists)."""
    import demo
    from transform.mask_transform import gpu_mask_voting
    boxes, masks, scores = demo.im_detect(im, net)
    g = lambda n: net.blobs[n]._host_read().copy()
    blobs = {"rois": g("rois"), "rois_ext": g("rois_ext"), "conv5_3": g("conv5_3"),
             "rpn_cls_prob_reshape": g("rpn_cls_prob_reshape"), "rpn_bbox_pred": g("rpn_bbox_pred"),
             "boxes": np.asarray(boxes).copy(), "mask_proposal": np.asarray(masks).copy(), "seg_cls_prob": np.asarray(scores).copy()}
    lm, lb = gpu_mask_voting(masks, boxes, scores, 21, 100, im.shape[1], im.shape[0])
    return blobs, (lm, lb)


def _check_against_engine(nat, net, im):
    blobs, (lm, lb) = _engine_reference(net, im)
    got_m, got_b = nat.detect(im)
    c5 = nat.blob("conv5_3")
    _, C, h, w = c5.shape
    assert np.array_equal(from_c8(c5.reshape(-1), C, h, w)[None], blobs["conv5_3"])
    for name in ("rpn_cls_prob_reshape", "rpn_bbox_pred", "rois", "rois_ext", "boxes", "mask_proposal", "seg_cls_prob"):
        a = nat.blob(name)
        assert a.shape == blobs[name].shape and np.array_equal(a, blobs[name]), name
    assert [len(b) for b in got_b] == [len(b) for b in lb]
    assert np.array_equal(np.concatenate(got_b, 0), np.concatenate(lb, 0))
    assert np.array_equal(np.concatenate(got_m, 0), np.concatenate(lm, 0), equal_nan=True)
    # and the voting itself against the oracle on the same device outputs
    om, ob = ohost.gpu_mask_voting(blobs["mask_proposal"], blobs["boxes"], blobs["seg_cls_prob"], 21, 100, im.shape[1], im.shape[0])
    assert np.array_equal(np.concatenate(got_b, 0), np.concatenate(ob, 0))
    assert np.array_equal(np.concatenate(got_m, 0), np.concatenate(om, 0), equal_nan=True)
    return got_m, got_b


@pytest.mark.parametrize("math", ["fp32", "bf16x3", "f16", "mixed", "bf16"])
def test_native_pipeline_equals_python_engine(math):
    """Reduced-width graph, three image sizes, several images per size: call 1 of a size launches every kernel, call 2 captures
    the HIP graph, later calls replay it -- every call equals the Python engine bit for bit (intermediate blobs, rois of both
    stages, voted instances)."""
    from mnc_amd.engine import Net
    path = models.write_mnc_5stage_test_prototxt(width_div=8)
    w = synth.synthetic_weights(path, seed=1)
    net = Net(path, w, 1, math=math)
    nat = NativeNet(w, math=math)
    try:
        rng = np.random.default_rng(7)
        for (H, W), reps in (((75, 100), 4), ((120, 90), 3), ((75, 100), 2)):
            for _ in range(reps):
                im = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
                _check_against_engine(nat, net, im)
    finally:
        nat.close()
        net.close()


def test_native_pipeline_without_graph_and_with_few_proposals(monkeypatch):
    """use_graph=0 (direct launches every time) gives the same results; with pre_nms_topn = 40 fewer proposals survive than
    post_nms_topn = 300 -- the speculative heads are re-run on the exact count, as the reference (and the engine) compute it."""
    from mnc_amd.engine import Net
    from mnc_config import cfg
    path = models.write_mnc_5stage_test_prototxt(width_div=8)
    w = synth.synthetic_weights(path, seed=2)
    monkeypatch.setitem(cfg.TEST, "RPN_PRE_NMS_TOP_N", 40)
    net = Net(path, w, 1)
    nat = NativeNet(w, use_graph=False, pre_nms_topn=40)
    nat_g = NativeNet(w, use_graph=True, pre_nms_topn=40)
    try:
        rng = np.random.default_rng(8)
        for _ in range(3):
            im = rng.integers(0, 256, (75, 100, 3), dtype=np.uint8)
            a = _check_against_engine(nat, net, im)
            b = _check_against_engine(nat_g, net, im)
            assert nat.blob("rois").shape[0] <= 40
            assert np.array_equal(np.concatenate(a[1], 0), np.concatenate(b[1], 0))
    finally:
        nat.close()
        nat_g.close()
        net.close()


@pytest.mark.parametrize("math", ["fp32", "bf16x3", "f16"])
def test_native_pipeline_full_size_vgg16(math):
    """BASELINE configs[1] shape through the one-call path: 600x1000, VGG-16 widths, 300 RoIs per stage.  In the bf16x3 and f16
    modes the native trunk keeps 2-byte activation tensors between the convolutions while the Python engine keeps fp32 tensors
    and splits / rounds them in each consumer: the results are the same bits."""
    from mnc_amd.engine import Net
    path = models.write_mnc_5stage_test_prototxt()
    w = synth.synthetic_weights(path, seed=0)
    net = Net(path, w, 1, math=math)
    nat = NativeNet(w, math=math)
    try:
        for seed in ((0, 1, 2) if math == "fp32" else (0, 1)):
            im = np.random.default_rng(seed).integers(0, 256, (600, 1000, 3), dtype=np.uint8)
            _check_against_engine(nat, net, im)
            assert nat.blob("rois").shape == (300, 5)
    finally:
        nat.close()
        net.close()


@pytest.mark.parametrize("math", ["fp32", "mixed", "f16"])
def test_latency_plan_equals_python_engine_and_throughput_plan(math, monkeypatch):
    """PLAN (round 6): the launch plans for latency (MNC_PLAN=1 at context creation: every product cut until it fills the chip --
    rounds 1-5's K ranges and conv_sw plans) against the default plans for CU time, full size.  Both executors of the graph follow the
    context's plan, so native == Python engine bit for bit under either; between the plans only the grouping of the K ranges differs:
    the RPN outputs agree to 1e-5 of their range (fp32) / the reduced-precision modes' own noise."""
    from mnc_amd.engine import Net
    path = models.write_mnc_5stage_test_prototxt()
    w = synth.synthetic_weights(path, seed=0)
    im = np.random.default_rng(3).integers(0, 256, (600, 1000, 3), dtype=np.uint8)
    feats = {}
    for plan in ("0", "1"):
        monkeypatch.setenv("MNC_PLAN", plan)
        net = Net(path, w, 1, math=math)
        nat = NativeNet(w, math=math)
        try:
            _check_against_engine(nat, net, im)
            feats[plan] = {n: nat.blob(n).copy() for n in ("rpn_cls_prob_reshape", "rpn_bbox_pred")}
        finally:
            nat.close()
            net.close()
    monkeypatch.delenv("MNC_PLAN")
    differs = False
    for n, a in feats["0"].items():
        b = feats["1"][n]
        rel = float(np.abs(a - b).max() / max(np.abs(a).max(), 1e-30))
        # (f16 rounds every activation to 11 bits: another summation order flips roundings; the mode itself sits 4e-3 from fp32)
        assert rel < {"fp32": 1e-5, "mixed": 1e-3, "f16": 8e-3}[math], (n, rel)
        differs = differs or not np.array_equal(a, b)
    if math == "fp32":
        assert differs, "the two plans cut the F(4x4) layers' K ranges differently: identical bits mean the switch did nothing"


def test_graph_is_dropped_when_a_context_arena_moves():
    """A, A (graph captured), B (seen once: runs eagerly), A again.  With full VGG widths the Winograd tail plan's split-K scratch
    is not monotonic in the image size (net input 600x898 needs less than the SMALLER 562x1000), so B re-allocates the context's
    scratch arena while every net buffer still fits its head-room: the graph captured for A holds the freed address and must not
    be replayed.  Every call equals a net that never uses a graph."""
    path = models.write_mnc_5stage_test_prototxt()
    w = synth.synthetic_weights(path, seed=0)
    ref = NativeNet(w, use_graph=False)
    nat = NativeNet(w, use_graph=True)
    try:
        rng = np.random.default_rng(21)
        gens = []
        for (H, W) in ((334, 500), (334, 500), (281, 500), (334, 500), (334, 500), (334, 500)):
            im = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
            c0, r0 = ref.forward_image(im)
            c1, r1 = nat.forward_image(im)
            assert np.array_equal(c0, c1) and np.array_equal(r0, r1, equal_nan=True), (H, W, len(gens))
            gens.append(nat.arena_generation())
        assert gens[0] == gens[1], "the second image of a size must not move an arena (it is the captured one)"
        assert gens[2] > gens[1], "this test needs B to grow a context arena after A's graph was captured: %r" % (gens,)
        assert gens[3] == gens[4] == gens[5] == gens[2]
    finally:
        nat.close()
        ref.close()


def test_two_images_in_flight_equal_one_at_a_time():
    """mnc_forward_image_async / mnc_net_fetch: two nets (own context, stream, buffers), image k+1 launched before image k is
    fetched -- overlapping independent images on the GPU changes nothing in any result."""
    path = models.write_mnc_5stage_test_prototxt(width_div=8)
    w = synth.synthetic_weights(path, seed=4)
    ref = NativeNet(w, use_graph=False)
    nets = [NativeNet(w), NativeNet(w)]
    try:
        rng = np.random.default_rng(11)
        images = [rng.integers(0, 256, (75, 100, 3), dtype=np.uint8) for _ in range(7)]
        want = [ref.forward_image(im) for im in images]
        got = []
        nets[0].launch(images[0])
        for k in range(1, len(images)):
            nets[k % 2].launch(images[k])
            got.append(nets[(k - 1) % 2].fetch())
        got.append(nets[(len(images) - 1) % 2].fetch())
        for (c0, r0), (c1, r1) in zip(want, got):
            assert np.array_equal(c0, c1) and np.array_equal(r0, r1, equal_nan=True)
        # launch without fetch: the net waits for its own previous image before reusing the staging buffers
        nets[0].launch(images[1])
        nets[0].launch(images[2])
        c, r = nets[0].fetch()
        assert np.array_equal(c, want[2][0]) and np.array_equal(r, want[2][1], equal_nan=True)
    finally:
        ref.close()
        for n in nets:
            n.close()


def test_image_stream_returns_results_in_order():
    """native_net.ImageStream (1, 2 and 3 images in flight; mixed image sizes): the results of map() are those of one net run
    serially, in submission order."""
    from mnc_amd.native_net import ImageStream
    path = models.write_mnc_5stage_test_prototxt(width_div=8)
    w = synth.synthetic_weights(path, seed=4)
    rng = np.random.default_rng(12)
    images = [rng.integers(0, 256, ((75, 100) if k % 3 else (90, 120)) + (3,), dtype=np.uint8) for k in range(9)]
    ref = NativeNet(w, use_graph=False)
    try:
        want = [ref.forward_image(im) for im in images]
    finally:
        ref.close()
    for n in (1, 2, 3):
        st = ImageStream(w, in_flight=n)
        try:
            got = list(st.map(images))
        finally:
            st.close()
        assert len(got) == len(want)
        for (c0, r0), (c1, r1) in zip(want, got):
            assert np.array_equal(c0, c1) and np.array_equal(r0, r1, equal_nan=True), n


def test_nets_sharing_one_weight_set():
    """mnc_net_create_shared (NativeNet(<another NativeNet>)): a second net on the first one's device weights -- the same results
    as a net with its own copy, on a graph replay too; no weights of its own (device memory grows by its buffers only); the owner
    cannot be destroyed under it, and it takes no parameters."""
    from mnc_amd import _lib
    path = models.write_mnc_5stage_test_prototxt(width_div=8)
    w = synth.synthetic_weights(path, seed=6)
    wbytes = sum(a.nbytes for v in w.values() for a in v)
    rng = np.random.default_rng(3)
    images = [rng.integers(0, 256, (90, 120, 3), dtype=np.uint8) for _ in range(4)]
    free_a, _ = _lib.device_mem_info(0)
    own = NativeNet(w)
    try:
        want = [own.forward_image(im) for im in images]
        free0, _ = _lib.device_mem_info(0)
        sh = NativeNet(own)
        try:
            got = [sh.forward_image(im) for im in images]          # eager, capture, replay, replay
            free1, _ = _lib.device_mem_info(0)
            for (c0, r0), (c1, r1) in zip(want, got):
                assert np.array_equal(c0, c1) and np.array_equal(r0, r1, equal_nan=True)
            # activations + arenas only, no second weight set: the sharer costs at least half a weight set less than the owner did
            assert (free_a - free0) - (free0 - free1) > wbytes // 2, (free_a - free0, free0 - free1, wbytes)
            with pytest.raises(RuntimeError, match="sharing"):
                own.close()
            with pytest.raises(_lib.MncError, match="still use"):
                _lib.call("mnc_net_destroy", own.h)
            a = np.zeros(4, np.float32)
            with pytest.raises(_lib.MncError, match="shares"):
                _lib.call("mnc_net_set_param", sh.h, ctypes.cast(ctypes.c_char_p(b"fc7"), ctypes.c_void_p), 1, _lib.ptr(a), 4)
            c, r = own.forward_image(images[0])                      # the owner still works beside its sharer
            assert np.array_equal(c, want[0][0]) and np.array_equal(r, want[0][1], equal_nan=True)
        finally:
            sh.close()
    finally:
        own.close()


@pytest.mark.skipif(shutil.which("gcc") is None, reason="needs gcc to build the C host program")
def test_c_program_drives_one_image_without_python(tmp_path):
    """tests/c/forward_image_main.c: weights from the flat container, one image, three calls (eager, graph capture, graph replay),
    records to a file -- equal to what the Python wrapper of the same entry points returns."""
    path = models.write_mnc_5stage_test_prototxt(width_div=8)
    w = synth.synthetic_weights(path, seed=3)
    exe = str(tmp_path / "forward_image_main")
    subprocess.check_call(["gcc", "-O2", "-I", os.path.join(REPO, "include"), os.path.join(REPO, "tests", "c", "forward_image_main.c"),
                           "-o", exe, "-L", os.path.join(REPO, "mnc_amd"), "-lmnc_hip", "-Wl,-rpath," + os.path.join(REPO, "mnc_amd"),
                           "-Wl,-rpath-link,/opt/rocm/lib"])
    caffemodel.save_flat({k: v for k, v in w.items()}, str(tmp_path / "w.mncw"))
    im = np.random.default_rng(5).integers(0, 256, (90, 120, 3), dtype=np.uint8)
    im.tofile(str(tmp_path / "im.raw"))
    nat = NativeNet(w)
    try:
        cfg = nat.cfg
        with open(str(tmp_path / "cfg.txt"), "w") as f:
            for i in range(5):
                f.write("trunk%d %d\n" % (i, cfg.trunk_channels[i]))
            f.write("rpn_channels %d\nmask_fc %d\nfc_dim %d\nmath 0\nuse_graph 1\nwinograd %d\n"
                    % (cfg.rpn_channels, cfg.mask_fc, cfg.fc_dim, cfg.winograd))
        env = dict(os.environ)
        env.pop("PYTHONPATH", None)
        r = subprocess.run([exe, str(tmp_path / "w.mncw"), str(tmp_path / "im.raw"), "90", "120", str(tmp_path / "cfg.txt"),
                            str(tmp_path / "out.bin")], capture_output=True, text=True, timeout=300, env=env)
        assert r.returncode == 0, r.stdout + r.stderr
        raw = np.fromfile(str(tmp_path / "out.bin"), dtype=np.uint8)
        counts = raw[:84].view(np.int32)
        rec = raw[84:].view(np.float32).reshape(100, 447)
        want_counts, want_rec = nat.forward_image(im, record_cap=100)
        assert np.array_equal(counts, want_counts)
        assert np.array_equal(rec[:len(want_rec)], want_rec, equal_nan=True) and not rec[len(want_rec):].any()
        assert "instances %d" % counts[0] in r.stdout
    finally:
        nat.close()


def test_bench_launcher_two_ranks_native_engine_on_one_gpu():
    """De-risking the first 8-rank run on hardware this box cannot provide: `bench.py --gpus 2 --dist-backend gloo` starts two
    ranks of the NATIVE engine under torch.distributed.run on the one GPU here -- the launcher path (self-launch, torch-first import
    order, rank-0-only weight synthesis + the shared flat container mapped by rank 1, image sharding, per-step gather of the
    instance blocks, barrier + max-over-ranks timing, per-rank figures in the JSON line) end to end; only the transport of the
    gather differs from the RCCL run (host tensors through gloo: two RCCL ranks need two devices)."""
    import json
    import sys
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="4")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--dist-backend", "gloo", "--steps", "6",
                        "--warmup", "2", "--no-cpu-baseline", "--no-alt-math", "--no-resnet", "--no-resident"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    c = json.loads(line)
    # the LAST stdout line is the compact one (< 4 KB, ranks summarised); the full result -- every rank's own figures -- is the detail file
    assert len(line) < 4096 and c["n_gpus"] == 2 and c["scaling"] == "weak" and c["config"]["images_per_step"] == 2
    assert c["ranks_ms_per_step"]["n"] == 2 and c["ranks_ms_per_step"]["min"] > 0 and "ranks" not in c
    assert c["gather_transport"] == "gloo"
    with open(os.path.join(REPO, c["detail"])) as f:
        d = json.load(f)
    assert d["n_gpus"] == 2 and d["dist_backend"] == "gloo" and d["value"] == pytest.approx(c["value"], rel=1e-4)
    assert [x["rank"] for x in d["ranks"]] == [0, 1] and all(x["ms_per_step"] > 0 for x in d["ranks"])
    assert all(x.get("device_mem_gb") is None or x["device_mem_gb"] > 0 for x in d["ranks"])
    assert abs(d["value"] - 2 * 1e3 / d["ms_per_step"]) < 1e-6 * d["value"]
    assert d["ms_per_step"] >= max(x["ms_per_step"] for x in d["ranks"]) * 0.999       # the max over ranks is what is reported
    assert "native" in d["config"]["engine"]
