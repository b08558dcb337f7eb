"""Worker for tests/test_dist_gloo.py: world_size-2 gloo run of the image sharding + instance gather."""
import os
import sys

import numpy as np
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mnc_amd import dist as mdist  # noqa: E402


def fake_results(seed):
    rng = np.random.default_rng(seed)
    lm, lb = [], []
    for c in range(20):
        n = int(rng.integers(0, 4))
        lb.append(np.hstack([rng.integers(0, 500, (n, 4)).astype(np.float64), rng.uniform(0, 1, (n, 1))]))
        lm.append(rng.uniform(0, 1, (n, 1, 21, 21)).astype(np.float32))
    return lm, lb


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    images = list(range(5))                          # 5 images over 2 ranks: 3 + 2
    mine = mdist.shard_indices(len(images), rank, world)
    assert mine == [i for i in images if i % world == rank]
    g = mdist.InstanceGatherer(device=None)
    steps = max(len(mdist.shard_indices(len(images), r, world)) for r in range(world))
    got = {}
    for s in range(steps):
        if s < len(mine):
            rec, n = mdist.pack_instances(*fake_results(100 + mine[s]))
        else:
            rec, n = np.zeros((mdist.REC_CAP, mdist.REC_DIM), np.float32), 0     # ragged tail: empty block
        blocks = g.gather(rec)
        for r, blk in enumerate(blocks):
            idx = mdist.shard_indices(len(images), r, world)
            if s < len(idx):
                got[idx[s]] = blk.numpy().copy()
    assert sorted(got) == images
    for i in images:                                 # every rank sees every image's instances, bit-exact
        lm, lb = fake_results(100 + i)
        boxes, classes, masks = mdist.unpack_instances(got[i])
        want_b = np.concatenate(lb, 0).astype(np.float32)
        assert np.array_equal(boxes, want_b)
        assert np.array_equal(masks, np.concatenate(lm, 0))
        assert list(classes) == [c + 1 for c, b in enumerate(lb) for _ in range(len(b))]
    dist.barrier()
    if rank == 0:
        print("DIST_OK")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
