"""Multi-GPU plumbing for the batched-image path: independent images are sharded round-robin over one
process per GPU (no data-path collective); the only exchange is an all-gather of each image's final instances as a
fixed-shape block over RCCL/xGMI.  torch.distributed is the transport (backend "nccl" == RCCL on ROCm; "gloo" in the CPU
tests) -- plumbing, not compute.

Record layout per instance (447 float32): x1, y1, x2, y2, score, class id (1..20), 21x21 mask row-major."""
import numpy as np

REC_CAP = 100            # gpu_mask_voting returns at most max_per_image = 100 instances (mask_transform.py:242-244)
REC_DIM = 4 + 1 + 1 + 21 * 21


def shard_indices(n_items, rank, world):
    """Image i -> rank i mod world."""
    return list(range(rank, n_items, world))


def pack_instances(result_mask, result_box, cap=REC_CAP):
    """(list_result_mask[20], list_result_box[20]) of gpu_mask_voting -> ([cap, 447] float32 block, count)."""
    rec = np.zeros((cap, REC_DIM), np.float32)
    n = 0
    for c, (m, b) in enumerate(zip(result_mask, result_box)):
        k = min(len(b), cap - n)
        if k <= 0:
            continue
        rec[n:n + k, :5] = b[:k]
        rec[n:n + k, 5] = c + 1
        rec[n:n + k, 6:] = np.asarray(m[:k], np.float32).reshape(k, -1)
        n += k
    return rec, n


def unpack_instances(rec):
    """[cap, 447] block -> (boxes [n,5], classes [n] int, masks [n,1,21,21]); rows with class 0 are padding."""
    keep = rec[:, 5] > 0
    r = rec[keep]
    return r[:, :5].copy(), r[:, 5].astype(np.int64), r[:, 6:].reshape(-1, 1, 21, 21).copy()


class InstanceGatherer(object):
    """Re-usable all-gather of one [cap, 447] block per rank (device tensors for nccl, host tensors for gloo)."""

    def __init__(self, device=None, cap=REC_CAP):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.world = dist.get_world_size()
        self.device = device
        self.send = torch.empty((cap, REC_DIM), dtype=torch.float32, device=device)
        self.recv = [torch.empty((cap, REC_DIM), dtype=torch.float32, device=device) for _ in range(self.world)]
        # first collective = communicator setup (seconds with RCCL): pay it here, never inside a timed or latency-critical step
        self.send.zero_()
        dist.all_gather(self.recv, self.send)
        if device is not None and str(device).startswith("cuda"):
            torch.cuda.synchronize()

    def gather(self, rec):
        """rec: numpy [cap, 447].  Returns the list of per-rank blocks as tensors (valid on every rank)."""
        self.send.copy_(self.torch.from_numpy(rec))
        self.dist.all_gather(self.recv, self.send)
        return self.recv
