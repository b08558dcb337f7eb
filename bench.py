#!/usr/bin/env python3
"""bench.py -- images/sec of the MNC 5-stage inference hot path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

A "step" is one pass of the whole hot path over one synthetic 600x1000 image per GPU (BASELINE configs[2] shape: VGG-16
trunk, RPN + proposal NMS, 300 RoIs per stage through both head stages, un-scale/clip/concat, gpu_mask_voting of the
600 instances on the 600x1000 canvas).  The image blob is resident in HBM before the timed region.  With N > 1 the
images are sharded one per rank (weak scaling, no data-path collective) and every step ends with the RCCL gather of
the padded [100, 447] instance block (box 4 + score + class + 21x21 mask) over xGMI, as north_star describes.

The headline `value` is measured with fp32 MFMA arithmetic (--math fp32, BASELINE configs[1]); at N = 1 the same run
also measures BASELINE configs[2] ("bf16 convs via MFMA": --math bf16x3, split-precision bf16 MFMA for the 3x3 convs and
the large InnerProducts) and reports it under `alt_math`, with its feature-level difference from the fp32 run.

One JSON line is printed by rank 0.  `roofline` is computed from HIP events recorded by the engine on ITS stream around
every launch of the dominant kernel inside the timed region; `cpu_baseline` times the CPU oracle (torch-CPU restatement
of the graph + the reference's nms/mv code compiled for the CPU when oracle/_ref is present) on the same workload.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

METRIC = "images/sec (600x1000, 300 RoIs) VGG16 MNC-5stage"
PEAK_FP32_MATRIX_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_BF16_MATRIX_TFLOPS = 2500.0     # same guide: v_mfma_f32_32x32x16_bf16, dense
PEAK_HBM_GBS = 8000.0
DTYPE = {"fp32": "f32", "bf16x3": "bf16x3 (fp32 operands split into hi+lo bf16, 3 bf16 MFMAs per product, f32 accumulate)",
         "f16": "f16 (3x3 convolutions and InnerProducts: operands rounded to fp16, f32 accumulate)"}
MATH_NOTE = {"fp32": "fp32 MFMA", "bf16x3": "3x3 convs and large InnerProducts on the bf16 matrix pipe with split operands "
                                            "(fp32-class accuracy), everything else fp32",
             "f16": "3x3 convs and large InnerProducts in fp16 with fp32 accumulation, everything else fp32"}


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=20)
    p.add_argument("--warmup", type=int, default=3)
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--cpu-images", type=int, default=3, help="images timed on the CPU oracle (after 1 warm-up)")
    p.add_argument("--no-events", action="store_true", help="do not record per-kernel HIP events in the timed region")
    p.add_argument("--all-events", action="store_true", help="time every launch (default: only the MFMA kernels)")
    p.add_argument("--math", default=os.environ.get("MNC_MATH", "fp32"), choices=["fp32", "bf16x3", "f16"],
                   help="arithmetic of the dense contractions for the headline number (default fp32)")
    p.add_argument("--no-alt-math", action="store_true", help="skip the bf16x3 (BASELINE configs[2]) measurement")
    p.add_argument("--host-results", action="store_true",
                   help="round-trip boxes/masks/scores through numpy between forward and voting (cfg.TEST.DEVICE_RESULTS=False)")
    p.add_argument("--dist-backend", default="nccl", help="nccl (RCCL) | gloo (functional test on fewer GPUs than ranks)")
    return p.parse_args()


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("WORLD_SIZE %d != --gpus %d" % (world, args.gpus))
    dist = None
    torch = None
    if world > 1:
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(args.dist_backend, rank=rank, world_size=world)

    import _init_paths  # noqa: F401
    import caffe
    import demo
    from mnc_amd import models, synth
    from mnc_config import cfg
    from transform.mask_transform import gpu_mask_voting

    from mnc_amd import _lib
    dev_id = local % max(_lib.device_count(), 1) if args.dist_backend != "nccl" else local
    caffe.set_mode_gpu()
    caffe.set_device(dev_id)
    cfg.GPU_ID = dev_id
    proto = models.write_mnc_5stage_test_prototxt()
    weights = synth.synthetic_weights(proto, seed=0)
    im = np.random.default_rng(rank).integers(0, 256, (600, 1000, 3), dtype=np.uint8)   # BASELINE.md section 3 inputs
    from transform.bbox_transform import clip_boxes
    from mnc_amd import dist as mdist
    from mnc_amd.engine import Net
    on_gpu = args.dist_backend == "nccl"
    gatherer = mdist.InstanceGatherer(device="cuda" if on_gpu else None) if world > 1 else None

    def measure(math, steps, warmup):
        """Build the net in `math` mode, run warmup + steps timed steps; returns (elapsed_s, phase_ms, records, last)."""
        net = Net(proto, weights, caffe.TEST, device_id=dev_id, math=math)
        # input resident in HBM before the timed region: prepare once, upload once, forward() re-uses the device blob
        kwargs, im_scales = demo.prepare_mnc_args(im, net)
        net.blobs["data"].set_host(kwargs["data"])
        net.blobs["im_info"].set_host(kwargs["im_info"])
        net.blobs["data"].dev_in("plain")
        scale = np.float32(im_scales[0])
        phase_ms = {"forward": 0.0, "tail": 0.0, "voting": 0.0, "gather": 0.0}
        device_results = bool(cfg.TEST.get("DEVICE_RESULTS", True)) and not args.host_results

        def step():
            t_a = time.perf_counter()
            net.forward()
            t_b = time.perf_counter()
            # demo.im_detect's tail (un-scale, clip, stack both stages): on the device by default, as tools/demo.py runs it
            if device_results:
                all_boxes, masks, scores = net.detect_tail(scale, im.shape)
            else:
                boxes = []
                for name in ("rois", "rois_ext"):
                    r = net.blobs[name]._host_read()
                    boxes.append(clip_boxes(r[:, 1:5] / scale, im.shape)[0])
                masks = np.concatenate((net.blobs["mask_proposal"]._host_read(), net.blobs["mask_proposal_ext"]._host_read()), 0)
                scores = np.concatenate((net.blobs["seg_cls_prob"]._host_read(), net.blobs["seg_cls_prob_ext"]._host_read()), 0)
                all_boxes = np.concatenate(boxes, 0)
            t_c = time.perf_counter()
            rm, rb = gpu_mask_voting(masks, all_boxes, scores, 21, 100, im.shape[1], im.shape[0])
            t_d = time.perf_counter()
            if world > 1:
                rec, _ = mdist.pack_instances(rm, rb)
                gatherer.gather(rec)
            t_e = time.perf_counter()
            phase_ms["forward"] += 1e3 * (t_b - t_a); phase_ms["tail"] += 1e3 * (t_c - t_b)
            phase_ms["voting"] += 1e3 * (t_d - t_c); phase_ms["gather"] += 1e3 * (t_e - t_d)
            return masks, all_boxes, scores

        def fence():
            net.sync()
            if world > 1:
                if on_gpu:
                    torch.cuda.synchronize()
                dist.barrier()
                if on_gpu:
                    torch.cuda.synchronize()

        for _ in range(warmup):
            step()
        events = not args.no_events
        fence()
        if events:
            net.profile(1 if args.all_events else 2)
        for k in phase_ms:
            phase_ms[k] = 0.0
        t0 = time.perf_counter()
        for _ in range(steps):
            last = step()
        fence()
        elapsed = time.perf_counter() - t0
        records = net.profile_records() if events else []
        if events:
            net.profile(False)
        feats = {n: net.blobs[n]._host_read().copy() for n in ("conv5_3", "rpn_bbox_pred", "rpn_cls_prob_reshape")}
        net.close()
        return elapsed, phase_ms, records, last, feats

    def summarise(math, steps, elapsed, phase_ms, records):
        out = {"host_phase_ms_per_image": {k: round(v / steps, 3) for k, v in phase_ms.items()}}
        if records:
            agg = {}
            for name, kms, fl, by in records:
                a = agg.setdefault(name, [0, 0.0, 0.0, 0.0])
                a[0] += 1; a[1] += kms; a[2] += fl; a[3] += by
            out["kernel_ms_per_image"] = {k: round(v[1] / steps, 4) for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])}
            dom = max(agg.items(), key=lambda kv: kv[1][1])
            name, (cnt, tot_ms, tot_fl, tot_by) = dom
            if tot_fl > 0:
                ach = tot_fl / (tot_ms * 1e-3) / 1e12
                x3 = "bf16x3" in name
                peak = PEAK_BF16_MATRIX_TFLOPS / 3.0 if x3 else PEAK_FP32_MATRIX_TFLOPS
                out["roofline"] = {"kernel": name, "bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s",
                                   "frac": ach / peak, "traffic": None, "launches_per_image": cnt / steps,
                                   "avg_launch_ms": tot_ms / cnt, "algorithmic_gflop_per_launch": tot_fl / cnt / 1e9,
                                   "peak_basis": ("bf16 dense MFMA peak %.0f TFLOP/s / 3 bf16 products per fp32-class product"
                                                  % PEAK_BF16_MATRIX_TFLOPS) if x3 else
                                                 "fp32 dense MFMA peak (v_mfma_f32_32x32x2_f32)"}
            else:
                ach = tot_by / (tot_ms * 1e-3) / 1e9
                out["roofline"] = {"kernel": name, "bound": "hbm", "achieved": ach, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                   "frac": ach / PEAK_HBM_GBS, "traffic": None, "launches_per_image": cnt / steps,
                                   "avg_launch_ms": tot_ms / cnt}
            out["roofline"]["traffic"] = pmc_traffic(out["roofline"]["kernel"])
        return out

    math = args.math
    elapsed, phase_ms, records, last, feats = measure(math, args.steps, args.warmup)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if on_gpu else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        ms = 1e3 * elapsed / args.steps
        out = {
            "metric": METRIC, "value": world * args.steps / elapsed, "unit": "images/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": DTYPE[math], "data": "synthetic",
            "config": {"workload": "VGG16 MNC 5-stage inference + gpu_mask_voting, one 600x1000 image per GPU per step, "
                                   "300 RoIs per stage (600 instances voted on a 600x1000 canvas), %s, "
                                   "seeded synthetic weights" % MATH_NOTE[math], "images_per_step": world,
                       "rois_per_stage": 300, "math": math,
                       "parallelism": "images sharded 1/GPU; RCCL all_gather of [100,447] instance blocks"
                       if world > 1 else "single GPU"},
        }
        out.update(summarise(math, args.steps, elapsed, phase_ms, records))
        if world == 1 and math == "fp32" and not args.no_alt_math:
            # BASELINE configs[2] ("bf16 convs via MFMA") measured in the same run, next to the fp32 headline: same image,
            # same weights, same step; its outputs are compared with the fp32 run's (trunk features, and the head outputs
            # that do not depend on which boxes survived NMS)
            e2, p2, r2, last2, feats2 = measure("bf16x3", args.steps, args.warmup)
            alt = {"math": "bf16x3", "dtype": DTYPE["bf16x3"], "value": args.steps / e2, "unit": "images/s",
                   "ms_per_step": 1e3 * e2 / args.steps}
            alt.update(summarise("bf16x3", args.steps, e2, p2, r2))
            alt["max_rel_diff_vs_fp32"] = {
                n: float(np.abs(feats2[n] - feats[n]).max() / max(np.abs(feats[n]).max(), 1e-30))
                for n in ("conv5_3", "rpn_bbox_pred", "rpn_cls_prob_reshape")}       # blobs that do not depend on which RoIs survived
            out["alt_math"] = alt
            # the fp16 mode (BASELINE configs[4] names fp16): 3x3 convolutions and large InnerProducts with one fp16 product
            # per term; compared with the fp32 run on the blobs that do not depend on which RoIs survived
            e3, p3, r3, last3, feats3 = measure("f16", args.steps, args.warmup)
            alt16 = {"math": "f16", "dtype": DTYPE["f16"], "value": args.steps / e3, "unit": "images/s",
                     "ms_per_step": 1e3 * e3 / args.steps}
            alt16.update({k: v for k, v in summarise("f16", args.steps, e3, p3, r3).items() if k != "roofline"})
            alt16["max_rel_diff_vs_fp32"] = {
                n: float(np.abs(feats3[n] - feats[n]).max() / max(np.abs(feats[n]).max(), 1e-30))
                for n in ("conv5_3", "rpn_bbox_pred", "rpn_cls_prob_reshape")}
            out["alt_math_f16"] = alt16
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(weights, im, args.cpu_images)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


PMC_KERNEL = {"conv3x3_c8_mfma": "conv3x3_c8_kernel", "fc_mfma": "fc_mfma_kernel<10>", "conv3x3_bf16x3": "conv3x3_x3_kernel",
              "fc_bf16x3": "fc_x3_kernel"}


def pmc_traffic(scope_name):
    """HBM bytes per launch of the dominant kernel from the rocprofv3 PMC passes of this build, as committed in
    profiles/pmc_latest.json by tools/prof_round.sh + tools/pmc_report.py (FETCH_SIZE doubled as
    MI355X_MICROARCH.md prescribes for wide coalesced reads on gfx950, + WRITE_SIZE).  bench.py cannot run rocprofv3 on
    itself, so this is null when no profile has been committed."""
    path = os.path.join(ROOT, "profiles", "pmc_latest.json")
    try:
        with open(path) as f:
            data = json.load(f)
    except (OSError, ValueError):
        return None
    want = PMC_KERNEL.get(scope_name, scope_name)
    calls = tot = 0.0
    for k, v in data.items():
        if k.startswith(want):
            calls += v["calls"]
            tot += v["calls"] * v["hbm_bytes_corrected"]
    return tot / calls if calls else None


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
            "sample": "%d x the same 600x1000 image after 1 warm-up: torch-CPU (%d threads) restatement of the Caffe graph "
                      "(Caffe itself is not buildable here) + %s for nms/mask voting"
                      % (n_images, cores, "the reference's nms_kernel.cu/mv_kernel.cu compiled for the CPU (oracle/_ref)"
                         if use_ref else "oracle/mnc_oracle.c"),
            "s_per_image": dt}


if __name__ == "__main__":
    main()
