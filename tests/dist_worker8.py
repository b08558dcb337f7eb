"""Worker for tests/test_dist_gloo.py::test_world_size_8_gloo: what an 8-rank node does outside the GPU, on the CPU over gloo --
image sharding, the per-step all-gather of [100, 447] instance blocks with a ragged tail, and the life cycle of the weight
container bench.py shares between the ranks of a node (one writer per node, an unpredictable name, mapped by all, unlinked)."""
import glob
import hashlib
import importlib.util
import os
import sys
import tempfile

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mnc_amd import dist as mdist  # noqa: E402
from mnc_amd import models, synth  # noqa: E402

from dist_worker import fake_results  # noqa: E402


def digest(w):
    h = hashlib.sha256()
    for k in sorted(w):
        for a in w[k]:
            h.update(k.encode())
            h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    assert world == 8
    # ---- weight container: written once per node by LOCAL_RANK 0, read by the other seven, gone afterwards
    spec = importlib.util.spec_from_file_location("bench_for_dist_test", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else tempfile.gettempdir()
    before = set(glob.glob(os.path.join(base, "mnc_bench_weights_*")))
    dist.barrier()
    proto = models.write_mnc_5stage_test_prototxt(width_div=16)
    w = bench.shared_weights(proto, synth, rank, world, dist)
    after = set(glob.glob(os.path.join(base, "mnc_bench_weights_*")))
    assert after <= before, "rank %d: the weight container was left behind: %r" % (rank, sorted(after - before))
    mine = digest(w)
    box = [None] * world
    dist.all_gather_object(box, mine)
    assert len(set(box)) == 1, "ranks hold different weights"
    if rank == 0:
        assert mine == digest(synth.synthetic_weights(proto, seed=0))          # and they are the seeded ones
    # ---- 19 images over 8 ranks: 3 + 3 + 3 + 2 + 2 + 2 + 2 + 2, one fixed-shape block per rank per step
    images = list(range(19))
    idx = mdist.shard_indices(len(images), rank, world)
    assert idx == [i for i in images if i % world == rank]
    assert sorted(i for r in range(world) for i in mdist.shard_indices(len(images), r, world)) == images
    g = mdist.InstanceGatherer(device=None)
    steps = max(len(mdist.shard_indices(len(images), r, world)) for r in range(world))
    got = {}
    for s in range(steps):
        if s < len(idx):
            rec, _ = mdist.pack_instances(*fake_results(100 + idx[s]))
        else:
            rec = np.zeros((mdist.REC_CAP, mdist.REC_DIM), np.float32)             # ragged tail: empty block
        blocks = g.gather(rec)
        assert len(blocks) == world and all(tuple(b.shape) == (mdist.REC_CAP, mdist.REC_DIM) for b in blocks)
        for r, blk in enumerate(blocks):
            ridx = mdist.shard_indices(len(images), r, world)
            if s < len(ridx):
                got[ridx[s]] = blk.numpy().copy()
            else:
                assert not blk.numpy().any()
    assert sorted(got) == images
    for i in images:
        lm, lb = fake_results(100 + i)
        boxes, classes, masks = mdist.unpack_instances(got[i])
        assert np.array_equal(boxes, np.concatenate(lb, 0).astype(np.float32))
        assert np.array_equal(masks, np.concatenate(lm, 0))
    # ---- an image whose scores tie at the voting threshold has MORE than 100 instances (gpu_mask_voting keeps
    # every box with cls_score >= thresh: lib/transform/mask_transform.py:242-258).  Rank 5's image has 131; the lossless gather
    # (counts first, then blocks of max(100, largest count) rows) returns every one of them on every rank, the others' unchanged.
    def tied_results(seed, per_class):
        rng = np.random.default_rng(seed)
        lb = [np.hstack([rng.integers(0, 500, (n, 4)).astype(np.float64), np.full((n, 1), 0.5)]) for n in per_class]
        lm = [rng.uniform(0, 1, (n, 1, 21, 21)).astype(np.float32) for n in per_class]
        return lm, lb
    per_class = [7] * 18 + [5, 0] if rank == 5 else None
    lm, lb = tied_results(900 + rank, per_class) if rank == 5 else fake_results(900 + rank)
    rec, total = mdist.pack_instances(lm, lb, lossless=True)
    assert total == (131 if rank == 5 else sum(len(b) for b in lb)) and rec.shape[0] == max(mdist.REC_CAP, total)
    blocks = g.gather(rec, total)
    assert list(g.last_counts)[5] == 131 and all(tuple(b.shape) == (131, mdist.REC_DIM) for b in blocks)
    for r, blk in enumerate(blocks):
        wm, wb = tied_results(900 + r, [7] * 18 + [5, 0]) if r == 5 else fake_results(900 + r)
        boxes, classes, masks = mdist.unpack_instances(blk.numpy())
        assert len(boxes) == int(g.last_counts[r]) == sum(len(b) for b in wb), (r, len(boxes))
        assert np.array_equal(boxes, np.concatenate(wb, 0).astype(np.float32)) and np.array_equal(masks, np.concatenate(wm, 0))
    # the next step (nobody above 100) is back to [100, 447] blocks
    rec, total = mdist.pack_instances(*fake_results(950 + rank), lossless=True)
    blocks = g.gather(rec, total)
    assert all(tuple(b.shape) == (mdist.REC_CAP, mdist.REC_DIM) for b in blocks)
    # ---- max-over-ranks timing as bench.py reports it
    t = torch.tensor([1.0 + 0.1 * rank], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    assert abs(float(t.item()) - 1.7) < 1e-12
    dist.barrier()
    if rank == 0:
        print("DIST8_OK")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
