#!/usr/bin/env python3
"""bench.py -- images/sec of the MNC 5-stage inference hot path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config vgg16|resnet50] [--math fp32|bf16x3|f16|mixed|bf16]

`--gpus N` with N > 1 and no launcher in the environment re-executes itself under `python -m torch.distributed.run
--nnodes=1 --nproc-per-node N --master-addr 127.0.0.1`, so the plain command really starts N ranks, one per GPU (it fails
loudly when the node has fewer devices than ranks).  Under a launcher (RANK / LOCAL_RANK / WORLD_SIZE set) it is one rank.

A "step" is one pass of the whole hot path over one synthetic image per GPU, measured the way BASELINE.md section 3 asks
(image H2D and result D2H inside the timed region):
    a DIFFERENT uint8 600x1000 image every step (seeds 0..7 rotating, host memory)
    -> upload + mean subtraction / layout on the GPU (tools/demo.py:prepare_mnc_args -> mnc_prep_image)
    -> net.forward: VGG-16 trunk, RPN + proposal NMS, 300 RoIs through both head stages
    -> im_detect's tail (un-scale / clip / concat) -> gpu_mask_voting of the 600 instances on the 600x1000 canvas
    -> the voted masks / boxes / scores copied to host numpy arrays (N = 1), or
       the RCCL all-gather of every rank's [100, 447] instance block over xGMI, issued on the engine's stream from the
       device-resident block (mnc_gather_instances), and the gathered blocks copied to the host (N > 1)
Images are sharded one per rank (weak scaling, no data-path collective).  `value` = images of all ranks / max-over-ranks time.
Every GPU keeps TWELVE images in flight by default (--in-flight: one mnc_net + context + stream per image in flight, ONE shared set
of device weights, 16 GB resident).  Streams map one to one onto hardware queues while there are enough of them; the runtime's default
is four (four images in flight then: fp32 2 / 3 / 4 in flight = 265.9 / 267.0 / 268.1 images/s, a fifth stream shares a queue and costs
3-4 %: profiles/r06_streams.txt), the library asks for 16 (GPU_MAX_HW_QUEUES, set when libmnc_hip.so is loaded unless the host
exported its own), and with round 6's launch plans -- made for CU time, not for the duration of a launch -- 4 / 8 / 12 / 16 images in
flight measure 270.0 / 278.2 / 280.6 / 280.3 images/s.  Round 5's sweep (three the best, four a dip) is profiles/r05_in_flight_sweep.txt;
mnc_forward_image_async of image k+1 is issued before mnc_net_fetch of image k - 11): a sixth of an image's GPU time is spent in
kernels of one or a few workgroups (proposal top-k, NMS scan, voting) that leave the chip idle -- other images' convolutions run
there.  Every image still goes through the whole path, upload to results; K steps = K images.  `one_image_at_a_time` is the
rounds 1-2 protocol (--in-flight 1 makes it the headline).

OUTPUT (round 4).  The LAST stdout line is ONE compact JSON object, < 4 KB (compact_line): metric .. config, `roofline`,
`cpu_baseline`, `kernel_ms_per_image`, `conv_roofline`, and one scalar per other protocol / math mode / configuration.  The full
result -- roofline_by_kernel, alt_math*, config_resnet50*, python_engine, per-rank figures -- is written to bench_detail.json
(repo root and gpurun_out/).  Round 3's line had grown to 24.5 KB and the driver, which keeps an 8 KB tail, could parse none of it.

The TIMED REGION holds exactly K steps of the protocol above and nothing else.  `roofline` is computed from HIP events the engine
records on ITS stream around every launch of the MFMA kernels (mnc_prof_*) during an EVENT PASS that follows the timed region:
clamp(K / 8, 8, 40) further images (--event-steps), one at a time, as direct launches (an event pair costs the stream ~3 us per
bracketed launch and cannot be captured into the HIP graph the library replays by default).

The headline `value` is measured with fp32 MFMA arithmetic (--math fp32, BASELINE configs[1]); at N = 1 the same run also measures
bf16x3, f16, mixed and plain bf16 (BASELINE configs[2] as written) and reports one scalar each (`alt_math*` in the detail file).
`--config resnet50` measures BASELINE configs[4] (ResNet-50 C4 trunk, 800x1333, 1000 proposals, fp16 math) with the same step and
schema; the default N = 1 run appends that measurement (child processes of this file, 40 steps: f16 and mixed; --no-resnet skips).
`cpu_baseline` times the CPU oracle (torch-CPU restatement of the graph + the reference's nms/mv code compiled for the CPU when
oracle/_ref is present) on the same workload, thread count stated.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

PEAK_FP32_MATRIX_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_BF16_MATRIX_TFLOPS = 2500.0     # same guide: v_mfma_f32_32x32x16_bf16 / _f16, dense
PEAK_HBM_GBS = 8000.0
DTYPE = {"fp32": "f32", "bf16x3": "bf16x3 (fp32 operands split into hi+lo bf16, 3 bf16 MFMAs per product, f32 accumulate)",
         "f16": "f16 (3x3 convolutions and InnerProducts: operands rounded to fp16, f32 accumulate)",
         "mixed": "mixed (3x3 convolutions bf16x3 = fp32-class, large InnerProducts fp16, f32 accumulate)",
         "bf16": "bf16 (3x3 convolutions and large InnerProducts: operands rounded to bf16, ONE product per term, f32 accumulate)"}
MATH_NOTE = {"fp32": "fp32 MFMA", "bf16x3": "3x3 convs and large InnerProducts on the bf16 matrix pipe with split operands "
                                            "(fp32-class accuracy), everything else fp32",
             "f16": "convolutions and large InnerProducts in fp16 with fp32 accumulation, everything else fp32",
             "mixed": "3x3 convolutions bf16x3 (split operands, fp32-class), large InnerProducts fp16, everything else fp32: the "
                      "reduced-precision mode that keeps the 1e-3 bar (tests/test_gpu_parity8.py)",
             "bf16": "3x3 convolutions and large InnerProducts in plain bf16 (one product per term, BASELINE configs[2] as written), fp32 "
                     "tensors between the layers, everything else fp32; measured, outside the 1e-3 bar"}
CONFIGS = {
    "vgg16": {"metric": "images/sec (600x1000, 300 RoIs) VGG16 MNC-5stage", "hw": (600, 1000), "rois": 300, "math": "fp32",
              "what": "VGG16 MNC 5-stage inference + gpu_mask_voting"},
    "resnet50": {"metric": "images/sec (800x1333, 1000 RoIs) ResNet50-C4 MNC-5stage", "hw": (800, 1333), "rois": 1000,
                 "math": "f16", "what": "ResNet-50 C4 trunk + MNC 5-stage heads + gpu_mask_voting (BASELINE configs[4]; no such "
                                        "model in the reference)"},
}
N_IMAGES = 8                         # seeds 0..7


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=300)
    p.add_argument("--warmup", type=int, default=10)
    p.add_argument("--config", default="vgg16", choices=sorted(CONFIGS))
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--cpu-images", type=int, default=3, help="images timed on the CPU oracle (after 1 warm-up)")
    p.add_argument("--ramp-ms", type=float, default=-1.0,
                   help="with more than four images in flight the pipeline is filled four images at once, then one more per this "
                        "many ms (-1, default: two thirds of one image's own time; 0: all at once)")
    p.add_argument("--no-latency-plan", action="store_true",
                   help="skip the measurements on the launch plans for latency (MNC_PLAN=1): `one_image_at_a_time` is then the figure on "
                        "the headline's plans and the per-mode `roofline_latency_plan` passes are not run (tools/prof_round.sh: the "
                        "counter passes need one fixed kernel sequence per image)")
    p.add_argument("--no-events", action="store_true", help="do not record per-kernel HIP events in the timed region")
    p.add_argument("--all-events", action="store_true", help="time every launch (default: only the MFMA kernels)")
    p.add_argument("--no-graph", action="store_true", help="native engine: direct launches on every step (counter-profiling runs)")
    p.add_argument("--event-steps", type=int, default=0,
                   help="images of the event pass that follows the timed region (one at a time, direct launches, HIP events around "
                        "the MFMA launches on the engine's stream: what `roofline` is computed from); 0 = clamp(steps / 8, 8, 40)")
    p.add_argument("--math", default=os.environ.get("MNC_MATH"), choices=["fp32", "bf16x3", "f16", "mixed", "bf16"],
                   help="arithmetic of the dense contractions for the headline number (default: fp32; resnet50: f16)")
    p.add_argument("--no-alt-math", action="store_true", help="skip the bf16x3 / f16 measurements (N = 1, vgg16)")
    p.add_argument("--no-resident", action="store_true", help="skip the secondary resident-input measurement")
    p.add_argument("--no-share-weights", action="store_true",
                   help="every image in flight gets its own copy of the device weights (rounds 2-4) instead of sharing one set")
    p.add_argument("--no-repeats", action="store_true", help="steps < 100: do not run the timed loop three more times for value_min/max")
    p.add_argument("--no-resnet", action="store_true",
                   help="skip the BASELINE configs[4] measurement (ResNet-50 C4, 800x1333, 1000 RoIs, f16) that the default N = 1 "
                        "run appends as `config_resnet50` (a child process: python bench.py --config resnet50)")
    p.add_argument("--engine", default="native", choices=["native", "graph", "python"],
                   help="native: one C call per image (mnc_forward_image, csrc/pipeline.hip; vgg16 only); python: the caffe-shaped "
                        "Net executing the prototxt layer by layer + demo.im_detect + gpu_mask_voting (tools/demo.py's own body); "
                        "graph: the same Net's launch sequence for an image, captured into a HIP graph per image size and replayed "
                        "(Net.detect_image: one graph launch + one synchronisation per image, any prototxt)")
    p.add_argument("--in-flight", type=int, default=0, choices=[0, 1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 16],
                   help="native engine: images in flight per GPU (own mnc_net + context + stream each; image k+1 is launched before "
                        "image k is fetched, so the latency-bound stretches of one image -- proposal top-k, NMS scan, voting: one or a "
                        "few workgroups -- run beside the other's convolutions).  1 = one image at a time (rounds 1-2 headline); "
                        "0 (default) = 12: one hardware queue per stream, 16 queues requested by the library "
                        "(profiles/r06_streams.txt)")
    p.add_argument("--dist-backend", default="nccl",
                   help="transport of the instance blocks: nccl = RCCL all-gather issued by libmnc_hip.so on a device stream (one rank "
                        "per GPU) | gloo = host tensors (functional test on fewer GPUs than ranks).  torch.distributed itself -- the "
                        "control plane: rendezvous, barrier, the ncclUniqueId -- always runs on gloo")
    return p.parse_args()


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: start N ranks of this file under torch.distributed.run."""
    from mnc_amd import _lib
    have = _lib.device_count()
    if args.dist_backend == "nccl" and have < args.gpus:
        raise SystemExit("bench.py --gpus %d: this node has %d GPU(s); one rank per GPU is required" % (args.gpus, have))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % args.gpus,
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    env.setdefault("OMP_NUM_THREADS", "4")
    raise SystemExit(subprocess.call(cmd, env=env))


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("WORLD_SIZE %d != --gpus %d" % (world, args.gpus))
    conf = CONFIGS[args.config]
    math = args.math or conf["math"]
    if args.config != "vgg16" and args.engine == "native":
        # the hand-written native pipeline is the VGG-16 5-stage graph; any other prototxt runs on the engine's plan, captured
        # into a HIP graph per image size and replayed (Net.detect_image)
        args.engine = "graph"
    launched = "WORLD_SIZE" in os.environ
    if args.in_flight == 0:
        # one hardware queue per stream (profiles/r06_streams.txt).  The runtime's default is four queues: four images in flight then
        # (fp32 2 / 3 / 4 in flight = 265.9 / 267.0 / 268.1 images/s; a fifth stream shares a queue and costs 3-4 %).  The library asks
        # for 16 queues when it is loaded (GPU_MAX_HW_QUEUES, csrc/ctx.hip), and with the launch plans made for CU time there is room
        # beside every launch: 4 / 8 / 12 / 16 images in flight = 270.0 / 278.2 / 280.6 / 280.3 (fp32, one box; 272.5 -> 284.6 on
        # another; f16 1033 -> 1047, mixed 613 -> 624 with 12; 1 rank under the launcher, the RCCL gather on a stream of its own: 284.3).
        # (the caffe-shaped Net on HIP graphs -- `--engine graph`, the ResNet-50 configuration -- keeps four: every Net in flight holds
        # its own weight copy, and in the driver's 20-step runs a dozen 1000-RoI images draining cost more than they gain: 303 vs 286)
        args.in_flight = 12 if args.engine == "native" else 4
    dist = torch = None
    on_gpu = args.dist_backend == "nccl"
    if launched:
        # libmnc_hip.so BEFORE torch, and torch.distributed on gloo: torch is only the control plane here (rendezvous, barrier,
        # max-over-ranks, carrying the 128-byte ncclUniqueId) and never touches a GPU; the data path -- kernels, streams, the RCCL
        # all-gather of the instance blocks -- is libmnc_hip.so's, bound to /opt/rocm's runtime exactly as in the N = 1 run without a
        # launcher.  (The torch wheel bundles an older ROCm runtime under the same sonames: loaded first it becomes the process's
        # runtime -- measured in round 3: HIP graphs on two streams do not overlap there and the 1-rank launcher run lost the whole
        # gain of the second image in flight, 191 vs 200 images/s.)
        from mnc_amd import _lib as _early
        _early.load()
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from mnc_amd import _lib
    ndev = _lib.device_count()
    if on_gpu and ndev < (local + 1):
        raise SystemExit("rank %d (local %d): only %d GPU(s) visible" % (rank, local, ndev))
    dev_id = local if on_gpu else local % max(ndev, 1)

    import _init_paths  # noqa: F401
    import caffe
    import demo
    from mnc_amd import dist as mdist
    from mnc_amd import models, synth
    from mnc_amd.engine import Net
    from mnc_config import cfg
    from transform.mask_transform import gpu_mask_voting
    from mnc_amd.instances import split_records

    caffe.set_mode_gpu()
    caffe.set_device(dev_id)
    cfg.GPU_ID = dev_id
    H, W = conf["hw"]
    if args.config == "resnet50":
        cfg.TEST.SCALES, cfg.TRAIN.MAX_SIZE, cfg.TEST.RPN_POST_NMS_TOP_N = (H,), W, conf["rois"]
        proto = models.write_mnc_resnet50_test_prototxt()
    else:
        proto = models.write_mnc_5stage_test_prototxt()
    weights = shared_weights(proto, synth, rank, world, dist if launched else None)
    # BASELINE.md section 3 inputs: uint8 images ~ U{0..255}, seeds 0..7; rank r starts the rotation at image r
    images = [np.random.default_rng(s).integers(0, 256, (H, W, 3), dtype=np.uint8) for s in range(N_IMAGES)]
    nms_t, iou_t = float(cfg.TEST.MASK_MERGE_NMS_THRESH), float(cfg.TEST.MASK_MERGE_IOU_THRESH)

    def measure(math, steps, warmup, resident_steps=0, engine=None, pipelined_steps=0, in_flight=None, plan=None):
        """Build the net in `math` mode; warmup + `steps` timed steps of the full protocol (upload .. results on the host), then
        optionally `resident_steps` steps of the old resident-input protocol.  -> dict.
        plan="1": the contexts are created under MNC_PLAN=1 (launch plans for latency); only the timed
        steps and the event pass run."""
        engine = engine or args.engine
        native = engine == "native"
        had_plan = os.environ.get("MNC_PLAN")
        if plan is not None and had_plan is None:
            os.environ["MNC_PLAN"] = plan
        try:
            return _measure(math, steps, warmup, resident_steps, engine, pipelined_steps, in_flight, plan)
        finally:
            if plan is not None and had_plan is None:
                del os.environ["MNC_PLAN"]

    def _measure(math, steps, warmup, resident_steps, engine, pipelined_steps, in_flight, plan):
        native = engine == "native"
        if native:
            from mnc_amd.native_net import NativeNet
            # the library's default path: the image size's HIP graph is captured on its second image and replayed from then on;
            # a step with per-kernel HIP events runs as direct launches (events are not captured), see --event-every
            net = NativeNet(weights, device_id=dev_id, math=math, use_graph=not args.no_graph)
        else:
            net = Net(proto, weights, caffe.TEST, device_id=dev_id, math=math)
        inflight = args.in_flight if ((native or engine == "graph") and in_flight is None) else (in_flight or 1)
        if native:
            # one set of device weights for all images in flight (mnc_net_create_shared): the other nets own a context, a stream
            # and their activation buffers only
            nets = [net] + [NativeNet(weights, device_id=dev_id, math=math, use_graph=not args.no_graph) if args.no_share_weights
                            else NativeNet(net) for _ in range(inflight - 1)]
        else:
            nets = [net] + [Net(proto, weights, caffe.TEST, device_id=dev_id, math=math) for _ in range(inflight - 1)]

        def launch_on(which, im):
            if native:
                nets[which].launch(im)
            else:
                nets[which].launch_image(im, 21, 100, nms_t, iou_t, use_graph=not args.no_graph)

        def fetch_from(which):
            return nets[which].fetch(record_cap=100) if native else nets[which].fetch_image()
        holder = net
        if launched and on_gpu and inflight > 1:
            # the communicator lives on its own context / stream: a gather enqueued on one image's stream would wait behind the
            # other image that is running there
            import types
            from mnc_amd.engine import _Ctx
            holder = types.SimpleNamespace(_ctx=_Ctx(dev_id))
        gatherer, transport = None, None
        if launched and on_gpu:
            # RCCL communicator (ncclCommInitRank through libmnc_hip.so).  If it cannot be brought up on ANY rank (librccl not
            # loadable, no peer access ...) every rank falls back to gathering the blocks as host tensors over the gloo control
            # plane, and the JSON line says so -- a scaling run then still measures the sharded compute instead of dying.
            err = None
            try:
                gatherer = mdist.InstanceGatherer(net=holder, rank=rank, world=world)
            except Exception as e:  # noqa: BLE001
                err = "%s: %s" % (type(e).__name__, e)
            flag = torch.tensor([0 if err is None else 1], dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            if int(flag.item()):
                if gatherer is not None:
                    gatherer.close()
                gatherer = mdist.InstanceGatherer(device=None)
                transport = "gloo fallback (RCCL communicator failed%s)" % ((": " + err) if err else " on another rank")
            else:
                transport = "rccl"
        elif launched:
            gatherer = mdist.InstanceGatherer(device=None)
            transport = "gloo"
        phase = {"prep+forward+tail": 0.0, "voting": 0.0, "gather": 0.0, "results_to_host": 0.0}
        last = {}

        def finish(counts, rec, blk, which, t_c):
            """what happens to one image's voted instances: the RCCL / gloo gather of the [100,447] block (N > 1) or numpy lists"""
            if gatherer is not None and gatherer.net is not None:   # RCCL, device block -> device blocks
                gatherer.gather_block(nets[which].block() if native else (nets[which]._inst.view() if blk is None else blk))
                t_d = time.perf_counter()
                last["gathered"] = gatherer.fetch(rows=int(counts[0]) if blk is None else None)   # [world, 100, 447] on the host
            elif gatherer is not None:                              # gloo functional path (host tensors)
                lists = split_records(rec, counts[1:], 21) if blk is None else blk.lists()
                packed, total = mdist.pack_instances(*lists, lossless=True)
                t_d = time.perf_counter()
                last["gathered"] = np.stack([t.numpy() for t in gatherer.gather(packed, total)])
            else:
                t_d = t_c
                # numpy lists per class, exactly what gpu_mask_voting returns
                last["masks"], last["boxes"] = split_records(rec, counts[1:], 21) if blk is None else blk.lists()
            t_e = time.perf_counter()
            phase["gather"] += t_d - t_c; phase["results_to_host"] += t_e - t_d

        def step(k):
            im = images[(rank + k) % N_IMAGES]
            t_a = time.perf_counter()
            if native:
                # ONE call: pinned staging + H2D + prep + forward + tail + voting + D2H of the records + stream sync
                counts, rec = net.forward_image(im, record_cap=100)
                t_b = t_c = time.perf_counter()
                blk = None
            elif engine == "graph":
                counts, rec = net.detect_image(im, 21, 100, nms_t, iou_t, use_graph=not args.no_graph)
                t_b = t_c = time.perf_counter()
                blk = None
            else:
                boxes, masks, scores = demo.im_detect(im, net)      # H2D + device prep + forward + device tail (DeviceArrays)
                t_b = time.perf_counter()
                blk = net.vote_instances(boxes, masks, scores, 21, 100, im.shape[1], im.shape[0], nms_t, iou_t)
                t_c = time.perf_counter()
                counts = rec = None
            phase["prep+forward+tail"] += t_b - t_a; phase["voting"] += t_c - t_b
            finish(counts, rec, blk, 0, t_c)

        pending = []                                 # (which net) of the images launched and not yet fetched, oldest first
        ramp = {"pace": 0.0, "next": 0.0, "left": 0}  # see step_pipelined

        def drain():
            while pending:
                which = pending.pop(0)
                t_a = time.perf_counter()
                counts, rec = fetch_from(which)
                t_c = time.perf_counter()
                phase["prep+forward+tail"] += t_c - t_a
                finish(counts, rec, None, which, t_c)

        def step_pipelined(k, with_events):
            """image k on net k % inflight: launched BEFORE the previous image is fetched (mnc_forward_image_async /
            mnc_net_fetch); an event step first drains the pipeline and runs alone, so that its per-kernel durations are not
            inflated by the other image's kernels"""
            if with_events:
                drain()
                step(k)
                return
            which = k % inflight
            if ramp["pace"] and ramp["left"] > 0 and 4 <= len(pending) < inflight:
                # filling the pipeline behind a fence: the first four images start together (they saturate the chip), the next
                # inflight - 4 follow one per `pace` instead of all at once -- a dozen images launched in the same instant run in
                # lock step (their latency-bound stretches coincide) for tens of images; staggered they are spread as in the steady
                # state (profiles/r06_streams.txt: 20 steps 271 -> 277 images/s, 40 steps 277 -> 281; a pace above the steady
                # interval would throttle, hence only these launches are paced)
                ramp["left"] -= 1
                ramp["next"] = max(ramp["next"] + ramp["pace"], time.perf_counter())
                while time.perf_counter() < ramp["next"]:
                    pass
            elif len(pending) < 4:
                ramp["next"] = time.perf_counter()
            t_a = time.perf_counter()
            launch_on(which, images[(rank + k) % N_IMAGES])
            phase["prep+forward+tail"] += time.perf_counter() - t_a
            pending.append(which)
            while len(pending) >= inflight:
                which0 = pending.pop(0)
                t_a = time.perf_counter()
                counts, rec = fetch_from(which0)
                t_c = time.perf_counter()
                phase["prep+forward+tail"] += t_c - t_a
                finish(counts, rec, None, which0, t_c)

        def fence():
            drain()
            for nn in nets:
                nn.sync()
            if launched:
                dist.barrier()
            ramp["left"] = inflight - 4              # the pipeline is empty: its next fill is paced

        if native and not args.no_graph:
            # part of building the nets, not of the W warm-up steps: every net of the pipeline sees the image size three times (eager,
            # the capture of its HIP graph, the first replay), so that neither the warm-up nor a short timed region (the driver's --steps 20
            # --warmup 5 with three nets) contains a graph capture
            for which, nn in enumerate(nets):
                for _ in range(3):                   # eager, the capture, the graph's first replay (it uploads the executable graph)
                    counts, rec = nn.forward_image(images[rank % N_IMAGES], record_cap=100)
                if gatherer is not None:
                    # ... and its instance block goes through the gather once (N > 1: the first collective on a buffer carries one-time
                    # set-up; with twelve nets the W warm-up steps do not reach every net's block -- the first timed loop of a 20-step
                    # run under the launcher measured 269 images/s, its repeats 276-280)
                    finish(counts, rec, None, which, time.perf_counter())
        elif engine == "graph" and not args.no_graph:
            # the same for the caffe-shaped Net's captured launch sequence (Net.detect_image): eager, then the capture (round 6: with
            # a dozen nets in flight a 20-step run otherwise times little else than captures)
            for which in range(len(nets)):
                for _ in range(3):
                    launch_on(which, images[rank % N_IMAGES])
                    fetch_from(which)
        if launched:
            # the control plane's first barrier sets its connections up (37 ms on one rank): here, not between the warm-up and the timed
            # region, where the GPU would sit idle and start the timed steps from its idle clocks (the first timed loop of a 20-step
            # run under the launcher measured 269 images/s, the three repeats behind it 277-279)
            dist.barrier()
        if native and inflight > 4 and args.ramp_ms != 0:
            if args.ramp_ms > 0:
                ramp["pace"] = args.ramp_ms * 1e-3
            else:                                    # two thirds of one image's own time (one image alone on the chip: fp32 4.2 ms -> 2.8)
                t_i = time.perf_counter()
                nets[0].forward_image(images[rank % N_IMAGES], record_cap=100)
                ramp["pace"] = (time.perf_counter() - t_i) / 1.5
        for k in range(warmup):
            if inflight > 1:
                step_pipelined(k, False)
            else:
                step(k)
        events = not args.no_events
        fence()
        level = 1 if args.all_events else 2
        for k in phase:
            phase[k] = 0.0
        # the timed region: `steps` steps of the named protocol and nothing else (no event steps inside it since round 4)
        t0 = time.perf_counter()
        for k in range(steps):
            if inflight > 1:
                step_pipelined(warmup + k, False)
            else:
                step(warmup + k)
        fence()
        elapsed = time.perf_counter() - t0
        phase_ms = {k: 1e3 * v / steps for k, v in phase.items()}
        # A short timed region (the driver's --steps 20 is 0.08 s) says nothing about its own spread: the same K steps are run
        # three more times, untimed for `value`, and reported as value_min / value_max next to it.
        repeats = [elapsed]
        if steps < 100 and not args.no_repeats:
            for rep in range(3):
                t1 = time.perf_counter()
                for k in range(steps):
                    if inflight > 1:
                        step_pipelined(warmup + (rep + 1) * steps + k, False)
                    else:
                        step(warmup + (rep + 1) * steps + k)
                fence()
                repeats.append(time.perf_counter() - t1)
        # the event pass, AFTER the timed region: `event_steps` more images, one at a time (pipeline drained), as direct
        # launches with a HIP event pair on the engine's stream around every MFMA launch -- what `roofline` is computed from
        records, event_steps = [], 0
        if events:
            event_steps = args.event_steps if args.event_steps > 0 else max(8, min(40, steps // 8))
            net.profile(level)                       # (resets the record list)
            for k in range(event_steps):
                step(warmup + steps + k)
            fence()
            records = net.profile_records()
            net.profile(False)
        try:                                             # device memory in use on this rank's GPU with every net of the run alive
            fr, tot = _lib.device_mem_info(dev_id)
            mem_gb = (tot - fr) / 1e9
        except Exception:  # noqa: BLE001
            mem_gb = None
        out = {"elapsed": elapsed, "repeats": repeats, "math": math, "phase_ms": phase_ms, "records": records, "device_mem_gb": mem_gb,
               "event_steps": event_steps, "rccl_version": getattr(gatherer, "rccl_version", None), "in_flight": inflight,
               "gather_transport": transport}
        for nn in nets[1:]:
            nn.close()
        if holder is not net and gatherer is not None:
            gatherer.close()
            holder._ctx.close()
            gatherer = None
        out["feats"] = {n: (net.blob(n) if native else net.blobs[n]._host_read().copy())
                        for n in ("rpn_bbox_pred", "rpn_cls_prob_reshape")}
        if native and not launched and plan is None:
            # the same step with every image on the captured HIP graph (no event steps): one hipGraphLaunch + one
            # synchronisation per image
            # Round 6: measured twice -- on the launch plans made for this protocol (MNC_PLAN=1: every launch cut until it fills the
            # chip, the shortest launch; rounds 1-5) and on the headline's plans (least CU time per launch: DESIGN.md section 9 item 10)
            from mnc_amd.native_net import NativeNet
            net.close()
            gsteps = min(steps, 100)
            for plan_key, plan in ((("graph_tp_s", None),) if args.no_latency_plan else (("graph_s", "1"), ("graph_tp_s", None))):
                had = os.environ.get("MNC_PLAN")
                if plan is not None and had is None:
                    os.environ["MNC_PLAN"] = plan
                try:
                    net = NativeNet(weights, device_id=dev_id, math=math, use_graph=True)
                finally:
                    if plan is not None and had is None:
                        del os.environ["MNC_PLAN"]
                for k in range(5):
                    net.forward_image(images[k % N_IMAGES], record_cap=100)
                t0 = time.perf_counter()
                for k in range(gsteps):
                    net.forward_image(images[(5 + k) % N_IMAGES], record_cap=100)
                out[plan_key] = (time.perf_counter() - t0) / gsteps
                if plan_key == "graph_s":
                    net.close()
            if args.no_latency_plan:
                out["graph_s"] = out["graph_tp_s"]
        if native and not launched and pipelined_steps and plan is None:
            # two images in flight: one mnc_net (own context / stream / buffers) per image in flight, launch image k+1 before
            # fetching image k -- independent images overlap on the GPU, the latency-bound stretches of one (proposal top-k, NMS
            # scan, voting) run beside the other's convolutions.  Direct launches, no events; drained inside the timed region.
            from mnc_amd.native_net import NativeNet
            nets = [net, NativeNet(net)]
            for k in range(6):
                nets[k % 2].forward_image(images[k % N_IMAGES], record_cap=100)
            for nn in nets:
                nn.sync()
            t0 = time.perf_counter()
            nets[0].launch(images[0])
            for k in range(1, pipelined_steps):
                nets[k % 2].launch(images[k % N_IMAGES])
                counts, rec = nets[(k - 1) % 2].fetch(record_cap=100)
                split_records(rec, counts[1:], 21)
            counts, rec = nets[(pipelined_steps - 1) % 2].fetch(record_cap=100)
            split_records(rec, counts[1:], 21)
            out["pipelined_s"] = (time.perf_counter() - t0) / pipelined_steps
            nets[1].close()
        if resident_steps and not native:
            # the round-1 protocol, for comparison: ONE image, blob already in HBM, no upload; results still come down
            im = images[rank % N_IMAGES]
            kwargs, im_scales = demo.prepare_mnc_args(im, net)
            net.forward(**kwargs)
            scale = np.float32(im_scales[0])

            def rstep():
                net.forward()
                b, m, s = net.detect_tail(scale, im.shape)
                gpu_mask_voting(m, b, s, 21, 100, im.shape[1], im.shape[0])
            for _ in range(3):
                rstep()
            net.sync()
            t0 = time.perf_counter()
            for _ in range(resident_steps):
                rstep()
            net.sync()
            out["resident_s"] = (time.perf_counter() - t0) / resident_steps
        if gatherer is not None:
            gatherer.close()
        net.close()
        return out

    def latency_plan_roofline(math_):
        """The event pass once more on the launch plans for latency (MNC_PLAN=1: every launch cut until it fills the chip).  The headline's
        plans minimise CU time, so their launches deliberately do NOT fill the chip when measured one at a time; this is what the same
        kernels reach when they do.  -> {"roofline": .., "roofline_by_kernel": ..} of 8 images."""
        ml = measure(math_, 8, 3, in_flight=1, plan="1")
        sm = summarise(8, ml)
        return {"plan": "MNC_PLAN=1 (latency): 8 images, one at a time, direct launches with events",
                "roofline": sm.get("roofline"), "kernel_ms_per_image": sm.get("kernel_ms_per_image"),
                "roofline_by_kernel": sm.get("roofline_by_kernel"), "conv3_x": sm.get("conv3_x")}

    def summarise(steps, m):
        out = {"host_phase_ms_per_image": {k: round(v, 3) for k, v in m["phase_ms"].items()}}
        records = m["records"]
        steps = m.get("event_steps") or steps        # the images of the event pass (their launches carry events)
        out["event_steps"] = steps if records else 0
        if records:
            agg = {}
            for name, kms, fl, by in records:
                a = agg.setdefault(name, [0, 0.0, 0.0, 0.0])
                a[0] += 1; a[1] += kms; a[2] += fl; a[3] += by
            out["kernel_ms_per_image"] = {k: round(v[1] / steps, 4) for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])}
            name, (cnt, tot_ms, tot_fl, tot_by) = max(agg.items(), key=lambda kv: kv[1][1])
            if tot_fl > 0:
                ach = tot_fl / (tot_ms * 1e-3) / 1e12
                peak, basis = mfma_peak(name)
                out["roofline"] = {"kernel": name, "bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s",
                                   "frac": ach / peak, "traffic": None, "launches_per_image": cnt / steps,
                                   "avg_launch_ms": tot_ms / cnt, "algorithmic_gflop_per_launch": tot_fl / cnt / 1e9,
                                   "peak_basis": basis}
            else:
                ach = tot_by / (tot_ms * 1e-3) / 1e9
                out["roofline"] = {"kernel": name, "bound": "hbm", "achieved": ach, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                   "frac": ach / PEAK_HBM_GBS, "traffic": None, "launches_per_image": cnt / steps,
                                   "avg_launch_ms": tot_ms / cnt}
            out["roofline"]["traffic"], out["roofline"]["traffic_source"] = pmc_traffic(out["roofline"]["kernel"])
            out["roofline_by_kernel"] = roofline_by_kernel(records, steps)
            c3 = conv3x_summary(records, steps, m.get("math", "fp32"))
            if c3:
                out["conv3_x"] = c3
        return out

    want_resident = world == 1 and not args.no_resident and args.engine == "python"
    headline_pipelined = args.engine in ("native", "graph") and args.in_flight > 1
    m = measure(math, args.steps, args.warmup, resident_steps=min(args.steps, 50) if want_resident else 0,
                pipelined_steps=0 if (args.no_resident or headline_pipelined) else min(args.steps, 100))
    elapsed = m["elapsed"]
    # every rank's own clock next to the max-over-ranks one: a straggler shows up as one large ms_per_step
    ranks = [{"rank": rank, "device": dev_id, "host": socket.gethostname(), "ms_per_step": 1e3 * elapsed / args.steps,
              "cpu_affinity": len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else None,
              "device_mem_gb": m.get("device_mem_gb")}]
    repeats = list(m.get("repeats") or [elapsed])
    if launched:
        t = torch.tensor([elapsed] + repeats, dtype=torch.float64)       # (every rank ran the same number of repeats)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t[0].item())
        repeats = [float(x) for x in t[1:]]
        box = [None] * world
        dist.all_gather_object(box, ranks[0])
        ranks = box

    if rank == 0:
        ms = 1e3 * elapsed / args.steps
        out = {
            "metric": conf["metric"], "value": world * args.steps / elapsed, "unit": "images/s",
            "n_gpus": dist.get_world_size() if launched else 1,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": DTYPE[math], "data": "synthetic",
            "config": {"workload": "%s, one %dx%d uint8 image per GPU per step (8 seeded images rotating), %d RoIs per stage "
                                   "(%d instances voted on a %dx%d canvas); H2D + device prep + D2H of the voted instances "
                                   "included in the timed region; %s; seeded synthetic weights (no trained model here: "
                                   "mAP unverifiable)" % (conf["what"], H, W, conf["rois"], 2 * conf["rois"], H, W,
                                                          MATH_NOTE[math]),
                       "what": conf["what"], "image": "%dx%d" % (H, W), "config_name": args.config,
                       "images_per_step": world, "rois_per_stage": conf["rois"], "math": math, "images_in_flight_per_gpu": m["in_flight"],
                       "parallelism": (("images sharded 1/GPU, %d ranks; ncclAllGather of [100,447] instance blocks on the "
                                        "engine stream" % world) if on_gpu else
                                       ("images sharded over %d ranks on %d GPU(s); functional run: [100,447] instance blocks "
                                        "gathered through torch.distributed/%s on the host" % (world, ndev, args.dist_backend)))
                                      if world > 1 else
                                      ("single GPU (1 rank under the launcher, RCCL gather of the block included)" if launched
                                       else "single GPU") + ("; %d images in flight per GPU (own stream each: image k+1 is launched "
                                                             "before image k is fetched)" % m["in_flight"] if m["in_flight"] > 1 else "")},
            "ranks": ranks, "rccl_version": m["rccl_version"], "dist_backend": args.dist_backend if launched else None,
            "control_plane": "torch.distributed/gloo" if launched else None, "gather_transport": m.get("gather_transport"),
        }
        if len(repeats) > 1:
            vals = [world * args.steps / e for e in repeats]
            out["value_repeats"] = {"values": [round(v, 3) for v in vals], "min": min(vals), "max": max(vals),
                                    "note": "the timed loop of `steps` steps, then three more runs of it; `value` is the first"}

        out.update(summarise(args.steps, m))
        conv = [r for r in m["records"] if r[0].startswith("conv3x3")]
        if conv and out.get("event_steps"):
            # north_star: images/s "as fraction of the conv roofline".  conv roofline = the trunk + RPN convolutions' algorithmic
            # (direct-form) flop at the dense MFMA peak of the math mode; Winograd executes 2.25x fewer (roofline_by_kernel).
            ev = out["event_steps"]
            fl, ms_c = sum(r[2] for r in conv) / ev, sum(r[1] for r in conv) / ev
            peak, basis = mfma_peak(max(conv, key=lambda r: r[2])[0])
            at_peak = peak * 1e12 / fl
            out["conv_roofline"] = {
                "conv_gflop_per_image": fl / 1e9, "conv_ms_per_image": ms_c, "conv_achieved_tflops": fl / ms_c / 1e9,
                "peak_tflops": peak, "peak_basis": basis, "conv_kernels_frac_of_peak": fl / ms_c / 1e9 / peak,
                "images_per_s_at_conv_roofline": at_peak, "value_as_frac_of_conv_roofline": out["value"] / world / at_peak,
                "definition": "conv roofline = images/s one GPU would reach if the 3x3 convolutions (trunk + rpn_conv, algorithmic "
                              "direct-form flop) ran at the dense MFMA peak and nothing else took time; per-GPU value / that"}
        out["config"]["engine"] = (("native: one mnc_forward_image call per image (csrc/pipeline.hip), the image size's captured HIP graph "
                                    "replayed on every timed step; the roofline events come from %d extra images run after the timed "
                                    "region as direct launches, one at a time" % m["event_steps"])
                                   if args.engine == "native" else
                                   ("graph: mnc_amd.engine.Net's own plan for the prototxt (Net.detect_image), captured into a HIP graph "
                                    "per image size and replayed on every timed step; roofline events from %d extra images after the "
                                    "timed region (direct launches)" % m["event_steps"])
                                   if args.engine == "graph" else "python: mnc_amd.engine.Net layer by layer (tools/demo.py body)")
        if "graph_s" in m:
            out["graph_replay"] = {"value": 1.0 / m["graph_s"], "unit": "images/s", "ms_per_step": 1e3 * m["graph_s"],
                                   "protocol": "ONE image at a time (the rounds 1-2 headline protocol; latency of an image): same step, "
                                               "every image on the captured HIP graph (one hipGraphLaunch + one synchronisation per "
                                               "image; no event steps); launch plans for latency (MNC_PLAN=1: every launch cut until it "
                                               "fills the chip -- rounds 1-5's plans)",
                                   "value_on_the_headline_plans": 1.0 / m["graph_tp_s"] if "graph_tp_s" in m else None}
            out["one_image_at_a_time"] = out["graph_replay"]
        if args.engine == "native" and world == 1 and not launched and not args.no_resident:
            mp = measure(math, min(args.steps, 100), args.warmup, resident_steps=50, engine="python")
            out["python_engine"] = {"value": min(args.steps, 100) / mp["elapsed"], "unit": "images/s",
                                    "ms_per_step": 1e3 * mp["elapsed"] / min(args.steps, 100),
                                    "host_phase_ms_per_image": {k: round(v, 3) for k, v in mp["phase_ms"].items()},
                                    "protocol": "same timed region through mnc_amd.engine.Net + demo.im_detect + gpu_mask_voting "
                                                "(the caffe-shaped drop-in, ~100 C-ABI calls per image)"}
            m["resident_s"] = mp["resident_s"]
            # the same caffe-shaped Net with every image on its captured HIP graph (Net.detect_image) and the headline's images in
            # flight: what the drop-in API reaches when the host keeps several images going
            mg = measure(math, min(args.steps, 100), args.warmup, engine="graph", in_flight=4)
            out["python_engine_graph"] = {"value": min(args.steps, 100) / mg["elapsed"], "unit": "images/s",
                                          "ms_per_step": 1e3 * mg["elapsed"] / min(args.steps, 100),
                                          "images_in_flight_per_gpu": mg["in_flight"],
                                          "protocol": "same timed region through mnc_amd.engine.Net.detect_image: the prototxt's launch "
                                                      "sequence captured into a HIP graph per image size, one Net per image in flight"}
        if "pipelined_s" in m:
            out["two_images_in_flight"] = {
                "value": 1.0 / m["pipelined_s"], "unit": "images/s", "ms_per_step": 1e3 * m["pipelined_s"],
                "protocol": "same step with two images in flight on two streams (mnc_forward_image_async of image k+1 before "
                            "mnc_net_fetch of image k, HIP-graph replay): throughput of the batched-image path on ONE GPU; the "
                            "headline stays one image at a time so that per-kernel durations are not inflated by overlap"}
        if "resident_s" in m:
            out["resident_input"] = {"value": 1.0 / m["resident_s"], "unit": "images/s", "ms_per_step": 1e3 * m["resident_s"],
                                     "protocol": "round-1 protocol: one image, input blob resident in HBM, no upload / device "
                                                 "prep; voted results copied to the host"}
        if world == 1 and not launched and args.config == "vgg16" and not args.no_alt_math and not args.no_events and args.engine == "native" \
                and not args.no_latency_plan:
            out["roofline_latency_plan"] = latency_plan_roofline(math)
        if world == 1 and not launched and args.config == "vgg16" and math == "fp32" and not args.no_alt_math:
            # BASELINE configs[2] ("bf16 convs via MFMA") and the fp16 mode measured in the same run, same protocol; their RPN
            # outputs on the LAST image are compared with the fp32 run's (blobs that do not depend on which RoIs survived)
            for key, alt in (("alt_math", "bf16x3"), ("alt_math_f16", "f16"), ("alt_math_mixed", "mixed"), ("alt_math_bf16", "bf16")):
                # (an image takes 1-2 ms in these modes: in a run of fewer than 100 steps the fill and the drain of twelve images
                # outweigh what they gain -- K = 20: f16 1021 / 994, mixed 619 / 611 with 4 / 12 in flight -- so short runs keep four)
                alt_in_flight = None if (args.steps >= 100 or args.in_flight <= 4 or args.engine != "native") else 4
                m2 = measure(alt, args.steps, args.warmup, in_flight=alt_in_flight,
                             pipelined_steps=0 if (args.no_resident or headline_pipelined) else min(args.steps, 100))
                a = {"math": alt, "dtype": DTYPE[alt], "value": args.steps / m2["elapsed"], "unit": "images/s",
                     "ms_per_step": 1e3 * m2["elapsed"] / args.steps, "images_in_flight_per_gpu": m2["in_flight"]}
                a.update(summarise(args.steps, m2))
                if "graph_s" in m2:
                    a["graph_replay"] = {"value": 1.0 / m2["graph_s"], "unit": "images/s", "ms_per_step": 1e3 * m2["graph_s"]}
                if "pipelined_s" in m2:
                    a["two_images_in_flight"] = {"value": 1.0 / m2["pipelined_s"], "unit": "images/s",
                                                 "ms_per_step": 1e3 * m2["pipelined_s"]}
                a["max_rel_diff_vs_fp32"] = {n: float(np.abs(m2["feats"][n] - m["feats"][n]).max() /
                                                      max(np.abs(m["feats"][n]).max(), 1e-30)) for n in m["feats"]}
                if not args.no_events and not args.no_latency_plan:
                    a["roofline_latency_plan"] = latency_plan_roofline(alt)
                out[key] = a
        if world == 1 and not launched and not args.no_cpu_baseline and args.config == "vgg16":
            out["cpu_baseline"] = cpu_baseline(weights, images[0], args.cpu_images)
        if (world == 1 and not launched and args.config == "vgg16" and math == "fp32" and not args.no_resnet
                and not args.no_alt_math):
            out["config_resnet50"] = resnet50_line()
            # the same configuration in the reduced-precision mode that keeps the 1e-3 bar (3x3 convolutions bf16x3, 1x1 / stem fp32,
            # InnerProducts fp16): what configs[4] costs when the parity claim is kept
            out["config_resnet50_mixed"] = resnet50_line("mixed")
        emit(out)
    if launched:
        dist.barrier()
        dist.destroy_process_group()


def _r(x, nd=4):
    """round floats for the compact line (significant figures, not decimals)"""
    if isinstance(x, float):
        return float("%.*g" % (nd + 2, x))
    return x


def compact_line(out):
    """The LAST stdout line of a run: one JSON object well under 4 KB that carries what the driver parses (metric .. config,
    `roofline`, `cpu_baseline`) and one scalar per alternative measurement.  Everything else stays in the detail file
    (bench_detail.json).  Pure function of the full result dict, so a recorded result can be re-formatted (tests/test_bench_line.py).
    VERDICT r3: the round-3 line was 24.5 KB, the driver keeps an 8 KB tail, and nothing could be parsed."""
    c = out.get("config", {})
    world = out.get("n_gpus", 1)
    conf = {"workload": "%s; one %s uint8 image per GPU per step (8 seeded images rotating), %s RoIs per stage, H2D + device prep + "
                        "forward + tail + gpu_mask_voting + D2H of the voted instances all inside the timed region; synthetic weights"
                        % (c.get("what", out.get("metric", "")), c.get("image", "600x1000"), c.get("rois_per_stage")),
            "images_per_step": c.get("images_per_step", world), "rois_per_stage": c.get("rois_per_stage"),
            "math": c.get("math"), "images_in_flight_per_gpu": c.get("images_in_flight_per_gpu"),
            "engine": (c.get("engine") or "").split(":")[0].split(" ")[0],
            "parallelism": ("single GPU" if world == 1 else
                            "images sharded 1 per rank, %d ranks, %s gather of [100,447] instance blocks" %
                            (world, out.get("gather_transport") or out.get("dist_backend")))}
    line = {k: _r(out.get(k)) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                        "scaling", "vs_baseline")}
    if out.get("value_repeats"):
        line["value_min"], line["value_max"] = _r(out["value_repeats"]["min"]), _r(out["value_repeats"]["max"])
    line["dtype"] = (out.get("dtype") or "").split(" ")[0]
    line["data"] = out.get("data")
    line["config"] = conf
    rf = out.get("roofline")
    if rf:
        line["roofline"] = {k: _r(rf.get(k)) for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic",
                                                       "launches_per_image", "avg_launch_ms", "algorithmic_gflop_per_launch")
                            if k in rf}
        line["roofline"]["event_images"] = out.get("event_steps")
    cb = out.get("cpu_baseline")
    if cb:
        line["cpu_baseline"] = {"value": _r(cb.get("value")), "unit": cb.get("unit"), "cores": cb.get("cores"),
                                "kind": cb.get("kind"), "sample": (cb.get("sample") or "")[:160]}
    if out.get("kernel_ms_per_image"):
        line["kernel_ms_per_image"] = {k: _r(v, 3) for k, v in list(out["kernel_ms_per_image"].items())[:4]}
    cr = out.get("conv_roofline")
    if cr:
        # algorithmic_over_peak: direct-form flop / time / peak -- above 1 under Winograd, which executes 1/4 of them (conv3_x below)
        line["conv_roofline"] = {"conv_tflops": _r(cr.get("conv_achieved_tflops")), "peak_tflops": cr.get("peak_tflops"),
                                 "algorithmic_over_peak": _r(cr.get("conv_kernels_frac_of_peak")),
                                 "value_frac_of_conv_roofline": _r(cr.get("value_as_frac_of_conv_roofline"))}
    if out.get("conv3_x"):
        line["conv3_x"] = {k: _r(v, 3) for k, v in out["conv3_x"].items() if not isinstance(v, str)}
        for key in sorted(k for k in out if k.startswith("alt_math")):
            c3 = out[key].get("conv3_x") or {}
            if "mfma_util_pct" in c3:
                line["conv3_x"][out[key].get("math", key) + "_mfma_util_pct"] = _r(c3["mfma_util_pct"], 3)
    tr = [r for r in out.get("roofline_by_kernel", []) if r.get("what", "").endswith("(all launches of an image)")
          and "traffic_over_algorithmic" in r]
    if tr:
        line["trunk_traffic_over_algorithmic"] = {r["scope"]: _r(r["traffic_over_algorithmic"], 3) for r in tr}
    alt = {}
    if out.get("one_image_at_a_time"):
        alt["one_image_at_a_time"] = _r(out["one_image_at_a_time"].get("value"))
        if out["one_image_at_a_time"].get("value_on_the_headline_plans"):
            alt["one_image_at_a_time_headline_plans"] = _r(out["one_image_at_a_time"]["value_on_the_headline_plans"])
    if out.get("python_engine"):
        alt["python_engine"] = _r(out["python_engine"].get("value"))
    if out.get("python_engine_graph"):
        alt["python_engine_graph_in_flight"] = _r(out["python_engine_graph"].get("value"))
    for key in sorted(k for k in out if k.startswith("alt_math")):
        a = out[key]
        alt[a.get("math", key)] = _r(a.get("value"))
        if a.get("roofline"):
            alt[a.get("math", key) + "_roofline_frac"] = _r(a["roofline"].get("frac"), 3)
            alt[a.get("math", key) + "_roofline_kernel"] = a["roofline"].get("kernel")
        lp = (a.get("roofline_latency_plan") or {}).get("roofline")
        if lp:
            alt[a.get("math", key) + "_roofline_frac_latency_plan"] = _r(lp.get("frac"), 3)
            if lp.get("kernel") != (a.get("roofline") or {}).get("kernel"):
                alt[a.get("math", key) + "_roofline_kernel_latency_plan"] = lp.get("kernel")
        if a.get("max_rel_diff_vs_fp32"):
            alt[a.get("math", key) + "_max_rel_diff_vs_fp32"] = _r(max(a["max_rel_diff_vs_fp32"].values()), 2)
    if isinstance(out.get("config_resnet50"), dict):
        alt["resnet50_800x1333_1000rois"] = _r(out["config_resnet50"].get("value")) if "value" in out["config_resnet50"] \
            else "error"
        if "math" in out["config_resnet50"].get("config", {}):
            alt["resnet50_math"] = out["config_resnet50"]["config"]["math"]
    if isinstance(out.get("config_resnet50_mixed"), dict) and "value" in out["config_resnet50_mixed"]:
        alt["resnet50_mixed"] = _r(out["config_resnet50_mixed"]["value"])
    if alt:
        line["images_per_s_other_protocols"] = alt
    if out.get("gather_transport"):
        line["gather_transport"] = out.get("gather_transport")
    dm = [r.get("device_mem_gb") for r in (out.get("ranks") or []) if r.get("device_mem_gb") is not None]
    if dm:
        line["device_mem_gb"] = _r(max(dm), 3)
    ranks = out.get("ranks") or []
    if len(ranks) > 1:
        ms = [r["ms_per_step"] for r in ranks]
        line["ranks_ms_per_step"] = {"min": _r(min(ms)), "max": _r(max(ms)), "mean": _r(sum(ms) / len(ms)), "n": len(ms)}
        mem = [r["device_mem_gb"] for r in ranks if r.get("device_mem_gb") is not None]
        if mem:
            line["ranks_device_mem_gb"] = {"max": _r(max(mem), 3)}
        line["rccl_version"] = out.get("rccl_version")
    line["detail"] = out.get("_detail_path", "bench_detail.json")
    text = json.dumps(line, separators=(",", ":"))
    if len(text) >= 4000:                               # never let an optional block push the line past the driver's window
        for k in ("kernel_ms_per_image", "images_per_s_other_protocols", "conv_roofline", "trunk_traffic_over_algorithmic", "conv3_x"):
            line.pop(k, None)
            text = json.dumps(line, separators=(",", ":"))
            if len(text) < 4000:
                break
    return text


def emit(out):
    """Full result -> bench_detail.json (repo root, and gpurun_out/ when it exists so that it comes back from the GPU box); the
    compact line is the LAST thing on stdout."""
    name = "bench_detail.json" if out.get("n_gpus", 1) == 1 and out["config"].get("config_name", "vgg16") == "vgg16" else \
        "bench_detail_%s_n%d.json" % (out["config"].get("config_name", "vgg16"), out.get("n_gpus", 1))
    out["_detail_path"] = name
    for d in (ROOT, os.path.join(ROOT, "gpurun_out")):
        if os.path.isdir(d):
            try:
                with open(os.path.join(d, name), "w") as f:
                    json.dump(out, f, indent=1)
            except OSError as e:
                print("bench.py: could not write %s: %s" % (os.path.join(d, name), e), file=sys.stderr)
    sys.stdout.flush()
    print(compact_line(out), flush=True)


def shared_weights(proto, synth, rank, world, dist):
    """Seeded synthetic weights, generated ONCE per node: under a launcher with several ranks, rank 0 synthesises them (1.1 GB of
    normal deviates on up to 16 threads -- eight ranks doing that at once would oversubscribe the host at start-up) and writes the
    flat MNCW0001 container (mnc_amd.caffemodel.save_flat, what mnc_net_load_file reads) to shared memory; the other ranks map
    it.  Every rank then uploads its own copy to its GPU (the replicated-weights layout of SURVEY 8e)."""
    if dist is None or world == 1:
        return synth.synthetic_weights(proto, seed=0)
    import tempfile
    from mnc_amd import caffemodel
    # One writer per NODE (LOCAL_RANK 0), a name nobody can predict or pre-create (mkstemp: O_EXCL, mode 0600), carried to the
    # node's other ranks through the process group (a fixed name written by global rank 0 only failed on other nodes
    # and could collide with a stale file).
    local = int(os.environ.get("LOCAL_RANK", rank))
    base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else tempfile.gettempdir()
    host = socket.gethostname()
    # Every rank makes the SAME three collective calls on every path (a barrier in a `finally` met the other ranks'
    # all_gather_object when one rank failed early, and the job hung until the process-group timeout): a failure travels in the
    # gathered payload and is raised on all ranks; the file is removed without a collective.  The last barrier stands BEHIND the
    # unlink (a rank used to return while LOCAL_RANK 0 had not removed the file yet -- anything that looked at the
    # directory right after the call, e.g. tests/dist_worker8.py, raced with the removal).
    mine, w, err = None, None, None
    try:
        if local == 0:
            fd, mine = tempfile.mkstemp(prefix="mnc_bench_weights_", suffix=".mncw", dir=base)
            os.close(fd)
            w = synth.synthetic_weights(proto, seed=0)
            caffemodel.save_flat(w, mine)
    except Exception as e:  # noqa: BLE001 -- reported to every rank below
        err = "rank %d: writing the weight container: %s: %s" % (rank, type(e).__name__, e)
    names = [None] * world
    dist.all_gather_object(names, (host, mine, err))                # doubles as the barrier: the files are complete
    if not any(e for _, _, e in names) and local != 0:
        try:
            path = next(p for h, p, _ in names if h == host and p is not None)
            w = caffemodel.load_flat(path)
        except Exception as e:  # noqa: BLE001
            err = "rank %d: reading the weight container: %s: %s" % (rank, type(e).__name__, e)
    errs = [None] * world
    dist.all_gather_object(errs, err)                               # everybody has the file mapped (or has failed): the name can go
    if mine is not None:
        try:
            os.remove(mine)                                         # (the pages stay while mapped)
        except OSError:
            pass
    dist.barrier()                                                  # the name is gone on every node before any rank returns
    failed = [e for e in errs if e] or [e for _, _, e in names if e]
    if failed:
        raise RuntimeError("shared_weights: " + "; ".join(sorted(set(failed))))
    return w


def resnet50_line(math=None):
    """BASELINE configs[4] (ResNet-50 C4 trunk, 800x1333, 1000 proposals; fp16 math as the configuration names it, or `math`)
    measured with the same step and schema by a child process of this file, so that the driver's plain `python bench.py` line
    carries it too.  -> the child's JSON line, reduced to the fields that identify and size the measurement (or {"error": ...})."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--config", "resnet50", "--steps", "40", "--warmup", "5",
           "--no-cpu-baseline", "--no-resident"] + (["--math", math] if math else [])
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
        d = json.loads(line)
    except Exception as e:  # noqa: BLE001 -- the headline must not die with the secondary measurement
        return {"error": "%s: %s" % (type(e).__name__, e)}
    keep = ("metric", "value", "unit", "steps", "warmup", "ms_per_step", "dtype", "data", "config", "host_phase_ms_per_image",
            "kernel_ms_per_image", "roofline")
    return {k: d[k] for k in keep if k in d}


def roofline_by_kernel(records, steps):
    """One entry per (profiling scope, shape): the dominant InnerProduct split into its three shapes, the Winograd trunk with
    algorithmic AND executed flop, conv1_1 against HBM.  Same events as `roofline` (HIP events on the engine's stream)."""
    # one group per (scope, algorithmic flop, algorithmic bytes): layers of equal flop but different bytes (conv1_2 / conv2_2 /
    # conv3_2 / conv4_2 are all 44.24 GFLOP and move 307 / 154 / 79 / 48 MB) are different rows, and every row carries ITS OWN
    # bytes (the first member's bytes used to stand for the whole flop class)
    groups = {}
    for name, kms, fl, by in records:
        g = groups.setdefault((name, round(fl / 1e7), round(by / 1e5)), [name, 0, 0.0, fl, by])
        g[1] += 1; g[2] += kms
    rows = []
    for (name, _, _), (_, cnt, tot_ms, fl, by) in sorted(groups.items(), key=lambda kv: -kv[1][2]):
        avg_s = tot_ms / cnt * 1e-3
        label = FC_SHAPES.get((name, round(fl / 1e9, 2)), name)
        e = {"scope": name, "what": label, "launches_per_image": round(cnt / steps, 3), "avg_launch_ms": tot_ms / cnt,
             "ms_per_image": tot_ms / steps, "algorithmic_gflop_per_launch": fl / 1e9, "algorithmic_mb_per_launch": by / 1e6}
        hbm_frac = by / avg_s / 1e9 / PEAK_HBM_GBS if by else None
        if fl >= 1e9 and name not in HBM_BOUND_SCOPES:
            peak, basis = mfma_peak(name)
            e.update(bound="mfma", achieved=fl / avg_s / 1e12, peak=peak, unit="TFLOP/s", frac=fl / avg_s / 1e12 / peak,
                     peak_basis=basis)
            if "wino" in name:          # F(2x2,3x3): 16 multiplies per 2x2 outputs instead of 36; F(4x4,3x3): 36 per 4x4 instead of 144
                ex = fl / wino_factor(name)
                e.update(executed_gflop_per_launch=ex / 1e9, executed_tflops=ex / avg_s / 1e12,
                         executed_frac_of_peak=ex / avg_s / 1e12 / peak,
                         note="frac = algorithmic (direct-form) flop / time / peak, can exceed what the pipe executes; "
                              "executed_* = the MFMA work Winograd actually issues (algorithmic / %.4g)" % wino_factor(name))
            e["hbm_frac_algorithmic"] = hbm_frac
        else:
            e.update(bound="hbm", achieved=by / avg_s / 1e9, peak=PEAK_HBM_GBS, unit="GB/s", frac=hbm_frac)
        pos = fc_positions(label)
        if name.startswith("conv3x3") and name not in HBM_BOUND_SCOPES:
            # the trunk's layer classes share two template instantiations: counter traffic cannot be told apart per class, it is
            # reported for the whole scope in the aggregate row below
            e["traffic"], e["traffic_source"] = None, "per-layer-class traffic is not separable in the PMC profile; see the '(all launches)' row"
        else:
            e["traffic"], e["traffic_source"] = pmc_traffic(name, positions=pos)
        if e["traffic"] and by:
            e["traffic_over_algorithmic"] = e["traffic"] / by
        rows.append(e)
    # one aggregate row per multi-shape convolution scope: all its launches of an image together
    for scope in sorted({r["scope"] for r in rows if r["scope"].startswith("conv3x3") and r["scope"] not in HBM_BOUND_SCOPES}):
        part = [r for r in rows if r["scope"] == scope]
        n = sum(r["launches_per_image"] for r in part)
        ms = sum(r["ms_per_image"] for r in part)
        fl = sum(r["algorithmic_gflop_per_launch"] * r["launches_per_image"] for r in part)
        mb = sum(r["algorithmic_mb_per_launch"] * r["launches_per_image"] for r in part)
        peak, basis = mfma_peak(scope)
        agg = {"scope": scope, "what": "%s (all launches of an image)" % scope, "launches_per_image": n, "ms_per_image": ms,
               "algorithmic_gflop_per_image": fl, "algorithmic_mb_per_image": mb, "bound": "mfma", "achieved": fl / ms, "peak": peak,
               "unit": "TFLOP/s", "frac": fl / ms / peak, "peak_basis": basis}
        if "wino" in scope:
            agg.update(executed_tflops=fl / wino_factor(scope) / ms, executed_frac_of_peak=fl / wino_factor(scope) / ms / peak)
        t, src = pmc_traffic(scope)
        agg["traffic_source"] = src
        if t is not None:
            agg["traffic_mb_per_image"] = t * n / 1e6
            agg["traffic_over_algorithmic"] = t * n / 1e6 / mb
        rows.append(agg)
    return rows


# profiling scope of the engine -> the kernels (rocprofv3 names, regular expressions) launched inside it
PMC_KERNEL = {"conv3x3_c8_mfma": r"conv3x3_c8_kernel", "conv3x3_wino_mfma": r"conv3x3_wino2?_kernel", "conv3x3_wino4_mfma": r"conv3x3_wino4_kernel",
              "fc_mfma": r"fc_mfma_(dma(16)?_)?kernel<(10|5)[,>]", "fc_mfma_small": r"fc_mfma_kernel<2,", "conv3x3_c3": r"conv3x3_c3_kernel",
              "conv3x3_bf16x3": r"conv3x3_sw_kernel<0,", "conv3x3_f16": r"conv3x3_sw_kernel<1,",
              "fc_bf16x3": r"(fc_x3_kernel<\d+, \d+, \d+, 0>|fc_lowp_dma_kernel<\d+, 0>)", "fc_f16": r"(fc_x3_kernel<\d+, \d+, \d+, 1>|fc_lowp_dma_kernel<\d+, 1>)",
              "conv3x3_bf16": r"conv3x3_sw_kernel<2,", "fc_bf16": r"(fc_x3_kernel<\d+, \d+, \d+, 2>|fc_lowp_dma_kernel<\d+, 2>)"}
HBM_BOUND_SCOPES = {"conv3x3_c3"}          # conv1_1: 2 GFLOP over 161 MB -- bound by writing its output
# the big InnerProducts of one 300-RoI head stage by algorithmic GFLOP, and their positions in the per-image
# launch cycle of the InnerProduct kernel (tools/pmc_report.py --cycle).  Round 5: the box and the mask branch are launched in
# pairs (mnc_fc_pair) -- six launches per image: fc6_maskest, fc6 + fc6_mask, fc7 + fc7_mask, twice; unpaired (MNC_FUSE_SMALL=0,
# rounds 1-4): ten -- fc6_maskest, fc6, fc7, fc6_mask, fc7_mask, twice.
FC_SHAPES = {("fc_mfma", 15.41): "fc6_maskest (300 x 256 x 100352)", ("fc_mfma", 61.66): "fc6 / fc6_mask (300 x 4096 x 25088)",
             ("fc_mfma", 10.07): "fc7 / fc7_mask (300 x 4096 x 4096)",
             ("fc_mfma", 123.31): "fc6 + fc6_mask, one launch (2 x 300 x 4096 x 25088)",
             ("fc_mfma", 20.13): "fc7 + fc7_mask, one launch (2 x 300 x 4096 x 4096)"}
FC_POSITIONS_BY_PERIOD = {
    10: {"fc6_maskest (300 x 256 x 100352)": (0, 5), "fc6 / fc6_mask (300 x 4096 x 25088)": (1, 3, 6, 8),
         "fc7 / fc7_mask (300 x 4096 x 4096)": (2, 4, 7, 9)},
    6: {"fc6_maskest (300 x 256 x 100352)": (0, 3), "fc6 + fc6_mask, one launch (2 x 300 x 4096 x 25088)": (1, 4),
        "fc7 + fc7_mask, one launch (2 x 300 x 4096 x 4096)": (2, 5)}}


def fc_positions(label):
    """positions of an InnerProduct shape in the kernel's per-image cycle, for the period the committed PMC profile was cut with"""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_latest.json")) as f:
            data = json.load(f)
    except (OSError, ValueError):
        return None
    for k, v in data.items():
        if k.startswith("fc_mfma_dma") and isinstance(v, dict) and v.get("by_position"):
            return FC_POSITIONS_BY_PERIOD.get(len(v["by_position"]), {}).get(label)
    return None


def wino_factor(scope_name):
    """direct-form multiplies / Winograd multiplies: F(4x4,3x3) 144 / 36, F(2x2,3x3) 36 / 16"""
    return 4.0 if "wino4" in scope_name else 2.25


def mfma_peak(scope_name):
    x3, f16 = "bf16x3" in scope_name, ("f16" in scope_name or scope_name.endswith("_bf16") or "_bf16_" in scope_name)
    if x3:
        return PEAK_BF16_MATRIX_TFLOPS / 3.0, "bf16 dense MFMA peak %.0f TFLOP/s / 3 bf16 products per fp32-class product" % PEAK_BF16_MATRIX_TFLOPS
    if f16:
        return PEAK_BF16_MATRIX_TFLOPS, "fp16 / bf16 dense MFMA peak (v_mfma_f32_32x32x16_f16 / _bf16)"
    return PEAK_FP32_MATRIX_TFLOPS, "fp32 dense MFMA peak (v_mfma_f32_32x32x2_f32)"


# conv3_x (north_star: ">= 50 % MFMA util on conv3_x") = conv3_1, conv3_2, conv3_3 at 600x1000: told apart from the other
# layers of their flop class by their algorithmic bytes; in every trunk kernel's per-image launch cycle of the LARGE-map template
# instantiation (conv1_2, conv2_1, conv2_2, conv3_1, conv3_2, conv3_3: six launches of the fp32 F(4x4) kernel; the reduced-precision
# kernel runs all 13 layers on one instantiation since the CU-time plans of round 6, tools/prof_round.sh cuts its cycle at 13) they are
# positions 3, 4, 5.
def _conv3x_bytes():
    hw, out = 150 * 250, set()
    for cin, cout in ((128, 256), (256, 256)):
        for ib in (2, 4):
            for ob in (2, 4):
                out.add(hw * (cin * ib + cout * ob) + 36.0 * cin * cout)      # the kernels' LaunchScope formula (4-byte weights)
        out.add(4.0 * (hw * cin + (hw // 4) * cout + 9 * cin * cout))         # fp32 with the Pooling fused (conv3_3)
        for ib in (2, 4):                                                      # reduced precision, Pooling fused (round 6: conv_sw.hip)
            out.add(hw * cin * ib + 75 * 125 * cout * ib + 36.0 * cin * cout)
    return tuple(sorted(out))


CONV3X_BYTES = _conv3x_bytes()
CONV3X_PMC = {"fp32": "conv3x3_wino4_kernel<1, 0>", "bf16": "conv3x3_sw_kernel<2, 5, 2, 2, 1>", "f16": "conv3x3_sw_kernel<1, 5, 2, 2, 1>",
              "bf16x3": "conv3x3_sw_kernel<0, 5, 2, 2, 1>", "mixed": "conv3x3_sw_kernel<0, 5, 2, 2, 1>"}
CONV_EXECUTED_DIVISOR = {"conv3x3_wino4_mfma": 4.0, "conv3x3_wino_mfma": 2.25}


def conv3x_summary(records, steps, math):
    """-> {"ms", "algorithmic_tflops", "executed_frac", "mfma_util_pct"} of the three conv3_x launches of an image (HIP events of
    the event pass; MfmaUtil from the PMC profile of this build, positions 3..5 of the large-map kernel's cycle), or None."""
    rows = [r for r in records if r[0].startswith("conv3x3") and any(abs(r[3] - b) < 1.0 for b in CONV3X_BYTES)
            and r[2] in (2.0 * 150 * 250 * 9 * 128 * 256, 2.0 * 150 * 250 * 9 * 256 * 256)]
    if not rows or not steps:
        return None
    ms = sum(r[1] for r in rows) / steps
    fl = sum(r[2] for r in rows) / steps
    peak, _ = mfma_peak(rows[0][0])
    div = CONV_EXECUTED_DIVISOR.get(rows[0][0], 1.0)
    out = {"ms": ms, "algorithmic_tflops": fl / ms / 1e9, "executed_frac": fl / div / ms / 1e9 / peak, "scope": rows[0][0]}
    util, _src = pmc_value(CONV3X_PMC.get(math, ""), "mfma_util_pct", positions=(3, 4, 5))
    if util is not None:
        out["mfma_util_pct"] = util
    return out


def pmc_value(kernel_name, field, positions=None):
    """-> (average `field` of one kernel of profiles/pmc_latest.json, source) when that profile is of this build, else (None, why)."""
    from mnc_amd import _build
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_latest.json")) as f:
            data = json.load(f)
    except (OSError, ValueError):
        return None, "no profiles/pmc_latest.json"
    if data.get("_build") != _build.source_hash():
        return None, "profiles/pmc_latest.json is of another build"
    v = data.get(kernel_name)
    if not v:
        return None, "kernel not in profiles/pmc_latest.json"
    if positions is not None and v.get("by_position"):
        return sum(v["by_position"][i][field] for i in positions) / len(positions), "pmc_latest.json by_position"
    return v.get(field), "pmc_latest.json (all launches of the kernel)"


def pmc_traffic(scope_name, positions=None):
    """-> (HBM bytes per launch, source) of a scope's kernel from the rocprofv3 PMC passes committed in profiles/pmc_latest.json
    (tools/prof_round.sh + tools/pmc_report.py: FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for wide coalesced reads on
    gfx950, + WRITE_SIZE) -- ONLY when that profile was taken on the build this process runs (the JSON's "_build" equals
    mnc_amd._build.source_hash()); otherwise (None, why).  bench.py cannot run rocprofv3 on itself.  positions: average only these
    positions of the kernel's per-image launch cycle (shapes sharing one template instantiation)."""
    import re
    from mnc_amd import _build
    path = os.path.join(ROOT, "profiles", "pmc_latest.json")
    try:
        with open(path) as f:
            data = json.load(f)
    except (OSError, ValueError):
        return None, "no profiles/pmc_latest.json"
    have, want_build = data.get("_build"), _build.source_hash()
    if have != want_build:
        return None, "profiles/pmc_latest.json is of build %s, this is build %s" % (have, want_build)
    want = re.compile(PMC_KERNEL.get(scope_name, re.escape(scope_name)))
    calls = tot = 0.0
    for k, v in data.items():
        if k.startswith("_") or not want.match(k):
            continue
        if positions is not None and v.get("by_position"):
            for pidx in positions:
                calls += 1
                tot += v["by_position"][pidx]["hbm_bytes_corrected"]
        else:
            calls += v["calls"]
            tot += v["calls"] * v["hbm_bytes_corrected"]
    if not calls:
        return None, "kernel not in profiles/pmc_latest.json"
    return tot / calls, "rocprofv3 --pmc FETCH_SIZE (x2) + WRITE_SIZE, profiles/pmc_latest.json, build %s" % have


def cpu_baseline(weights, im, n_images):
    """The oracle (CPU restatement of the same graph on the same weights/image) timed on this host's cores."""
    import torch
    from oracle import host as ohost
    from oracle import native
    from oracle import net as onet
    # 32 threads is where torch-CPU convolutions peak on the 256-core bench host (tools/cpu_probe.py: 16 -> 0.94 s,
    # 32 -> 0.81 s, 64 -> 1.30 s, 128 -> 2.43 s per image); `cores` reports the threads actually used
    cores = min(32, os.cpu_count() or 1)
    torch.set_num_threads(cores)
    use_ref = native.ref_available()
    nms_fn = native.ref_gpu_nms if use_ref else native.gpu_nms
    mv_fn = native.ref_mv if use_ref else native.mv

    def one():
        b, m, s = onet.im_detect(weights, im, nms_fn=nms_fn)
        ohost.gpu_mask_voting(m, b, s, 21, 100, im.shape[1], im.shape[0], nms_fn=nms_fn, mv_fn=mv_fn)

    one()
    t0 = time.perf_counter()
    for _ in range(n_images):
        one()
    dt = (time.perf_counter() - t0) / n_images
    return {"value": 1.0 / dt, "unit": "images/s", "cores": cores, "kind": "port",
            "sample": "%d x one 600x1000 image after 1 warm-up: torch-CPU (%d threads) restatement of the Caffe graph "
                      "(Caffe itself is not buildable here) + %s for nms/mask voting"
                      % (n_images, cores, "the reference's nms_kernel.cu/mv_kernel.cu compiled for the CPU (oracle/_ref)"
                         if use_ref else "oracle/mnc_oracle.c"),
            "s_per_image": dt}


if __name__ == "__main__":
    main()
