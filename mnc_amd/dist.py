"""Multi-GPU plumbing for the batched-image path: independent images are sharded round-robin over one
process per GPU (no data-path collective); the only exchange is an all-gather of each image's final instances as a
fixed-shape block over RCCL/xGMI.

Two transports behind one class:
  * device (the product path): the block is the InstanceBlock mnc_vote_instances wrote on the GPU; the all-gather is
    ncclAllGather issued by libmnc_hip.so on the engine's own stream (mnc_gather_instances, csrc/comm.hip) -- device pointer to
    device pointer, no numpy hop, no torch tensor.  torch.distributed only carries the 128-byte ncclUniqueId at start-up.
  * host ("gloo", the CPU tests; no GPU here): the same records as numpy arrays through torch.distributed.all_gather.

Record layout per instance (447 float32): x1, y1, x2, y2, score, class id (1..20), 21x21 mask row-major."""
import ctypes

import numpy as np

from .instances import records_from_lists

REC_CAP = 100            # gpu_mask_voting returns max_per_image = 100 instances (mask_transform.py:242-244) unless scores tie
REC_DIM = 4 + 1 + 1 + 21 * 21


def shard_indices(n_items, rank, world):
    """Image i -> rank i mod world."""
    return list(range(rank, n_items, world))


def pack_instances(result_mask, result_box, cap=REC_CAP, lossless=False):
    """(list_result_mask[20], list_result_box[20]) of gpu_mask_voting -> ([cap, 447] float32 block, rows packed).  More than
    `cap` result rows (scores tied at the global threshold: gpu_mask_voting keeps every box with cls_score >= thresh,
    lib/transform/mask_transform.py:242-258) are truncated class-major, and the truncation is reported -- or, lossless=True
    (round 6), ALL rows are packed into a [max(cap, rows), 447] block for InstanceGatherer.gather(rec, total)."""
    if lossless:
        total = int(sum(len(b) for b in result_box))
        rec, total = records_from_lists(result_mask, result_box, max(cap, total))
        return rec, total
    rec, total = records_from_lists(result_mask, result_box, cap)
    if total > cap:
        import warnings
        warnings.warn("pack_instances: %d instances (scores tied at the voting threshold), block holds %d -- %d dropped"
                      % (total, cap, total - cap))
    return rec, min(total, cap)


def unpack_instances(rec):
    """[cap, 447] block -> (boxes [n,5], classes [n] int, masks [n,1,21,21]); rows with class 0 are padding."""
    keep = rec[:, 5] > 0
    r = rec[keep]
    return r[:, :5].copy(), r[:, 5].astype(np.int64), r[:, 6:].reshape(-1, 1, 21, 21).copy()


def _exchange_unique_id(make_id, rank):
    """rank 0's ncclUniqueId to every rank through the torch.distributed process group the launcher set up."""
    import torch.distributed as dist
    box = [make_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    return box[0]


class InstanceGatherer(object):
    """All-gather of one [cap, 447] block per rank.

    InstanceGatherer(net=net)        device transport: RCCL communicator on the net's context (created here -- communicator
                                     setup takes seconds and must never land in a timed step), ncclAllGather on its stream
    InstanceGatherer(device=None)    host transport over the initialised torch.distributed group (gloo)"""

    def __init__(self, net=None, device=None, cap=REC_CAP, rank=None, world=None, unique_id=None):
        self.cap = cap
        self.net = net
        if net is not None:
            from . import _lib
            from .engine import _DevBuf
            self._lib = _lib
            if world is None:
                import torch.distributed as dist
                rank, world = dist.get_rank(), dist.get_world_size()
            self.rank, self.world = rank, world

            def make_id():
                buf = ctypes.create_string_buffer(128)
                _lib.call("mnc_comm_unique_id", ctypes.addressof(buf), 128)
                return buf.raw
            if unique_id is None:
                unique_id = make_id() if world == 1 else _exchange_unique_id(make_id, rank)
            idbuf = ctypes.create_string_buffer(unique_id, 128)
            _lib.call("mnc_comm_init", net._ctx.h, ctypes.addressof(idbuf), world, rank)
            v = ctypes.c_int(0)
            _lib.call("mnc_comm_info", net._ctx.h, None, None, ctypes.addressof(v))
            self.rccl_version = v.value
            self._recv = _DevBuf(net._ctx)
            self._recv.ensure(world * cap * REC_DIM * 4)
            # the gathered blocks come down into PINNED memory: a copy into pageable memory goes through the runtime's staging
            # path, which waits for the whole device -- i.e. for the other image in flight on another stream
            pin = ctypes.c_void_p()
            _lib.call("mnc_host_alloc", net._ctx.h, world * cap * REC_DIM * 4 + world * 4, ctypes.addressof(pin))
            self._pin = pin.value
            # every rank's instance COUNT travels with its block (one 4-byte element per rank, a second all-gather on the same
            # stream): a rank whose image has more than `cap` rows (scores tied at the voting threshold) is then known to everybody,
            # and fetch() gathers the rows past `cap` in a second round -- the N > 1 result is what N = 1 returns (round 6)
            self._recv_cnt = _DevBuf(net._ctx)
            self._recv_cnt.ensure(world * 4)
            self._recv2 = _DevBuf(net._ctx)
            self._send0 = _DevBuf(net._ctx)                      # warm-up block: the first collective builds the rings
            p = self._send0.ensure(cap * REC_DIM * 4)
            _lib.call("mnc_dev_zero", net._ctx.h, p, cap * REC_DIM * 4)
            _lib.call("mnc_gather_instances", net._ctx.h, p, self._recv.ptr, cap * REC_DIM)
            _lib.call("mnc_gather_instances", net._ctx.h, p, self._recv_cnt.ptr, 1)
            _lib.call("mnc_ctx_sync", net._ctx.h)
            self.last_counts = None
            return
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.device = device
        self.rccl_version = None
        self.send = torch.empty((cap, REC_DIM), dtype=torch.float32, device=device)
        self.recv = [torch.empty((cap, REC_DIM), dtype=torch.float32, device=device) for _ in range(self.world)]
        self.send.zero_()
        dist.all_gather(self.recv, self.send)

    def gather_block(self, block):
        """Device transport: enqueue the all-gather of an InstanceBlock's first `cap` records and of its instance count on the
        net's stream (asynchronous; fetch() synchronises and, when some rank's count exceeds `cap`, gathers the remaining rows)."""
        if block.gather_rows != self.cap or block.rec_dim != REC_DIM:
            raise ValueError("block shape [%d,%d] does not match the gatherer's [%d,%d]"
                             % (block.gather_rows, block.rec_dim, self.cap, REC_DIM))
        self._lib.call("mnc_gather_instances", self.net._ctx.h, block.records_ptr, self._recv.ptr, self.cap * REC_DIM)
        cp = getattr(block, "counts_ptr", None)
        self._has_counts = cp is not None
        if cp is not None:
            self._lib.call("mnc_gather_instances", self.net._ctx.h, cp, self._recv_cnt.ptr, 1)
        self._sent = block

    def fetch(self, rows=None):
        """Device transport: the gathered blocks as one numpy array [world, rows, 447] (one copy, one synchronisation): rows =
        `cap` unless some rank's image has more instances than that (scores tied at the voting threshold) -- then every rank
        takes part in a second all-gather of the rows past `cap` and the array holds max-over-ranks rows, padding rows zero.
        `last_counts` = every rank's instance count of this step."""
        nblk = self.world * self.cap * REC_DIM * 4
        nbytes = nblk + self.world * 4
        self._lib.call("mnc_d2h_async", self.net._ctx.h, self._pin, self._recv.ptr, nblk)
        if getattr(self, "_has_counts", False):
            self._lib.call("mnc_d2h_async", self.net._ctx.h, self._pin + nblk, self._recv_cnt.ptr, self.world * 4)
        self._lib.call("mnc_ctx_sync", self.net._ctx.h)
        raw = np.frombuffer((ctypes.c_char * nbytes).from_address(self._pin), dtype=np.uint8)
        out = raw[:nblk].view(np.float32).reshape(self.world, self.cap, REC_DIM).copy()
        blk, self._sent = getattr(self, "_sent", None), None
        if not getattr(self, "_has_counts", False):               # a block without a device-side count: rounds 2-5 behaviour
            self.last_counts = None
            if rows is None and blk is not None and hasattr(blk, "head"):
                rows = int(blk.head()[0])
            if rows is not None and rows > self.cap:
                import warnings
                warnings.warn("gather_block: rank %d has %d instances (scores tied at the voting threshold), the gathered block "
                              "holds %d -- %d dropped" % (self.rank, rows, self.cap, rows - self.cap))
            return out
        counts = raw[nblk:].view(np.int32).copy()
        self.last_counts = counts
        extra = int(counts.max()) - self.cap
        if extra <= 0:
            return out
        # second round (rare): rows cap .. max count of every rank's block -- all ranks see the same counts and call it together
        if blk is None or getattr(blk, "rows_cap", self.cap) < self.cap + extra:
            raise RuntimeError("a rank reports %d instances but the sent block holds %s rows" % (int(counts.max()),
                                                                                              getattr(blk, "rows_cap", None)))
        self._recv2.ensure(self.world * extra * REC_DIM * 4)
        self._lib.call("mnc_gather_instances", self.net._ctx.h, blk.records_ptr + self.cap * REC_DIM * 4, self._recv2.ptr,
                       extra * REC_DIM)
        more = np.zeros((self.world, extra, REC_DIM), np.float32)
        self._lib.call("mnc_d2h", self.net._ctx.h, self._lib.ptr(more), self._recv2.ptr, more.nbytes)     # (synchronises)
        out = np.concatenate((out, more), 1)
        for r in range(self.world):
            out[r, max(int(counts[r]), 0):] = 0.0               # rows past a rank's count: whatever an earlier image left there
        return out

    def gather(self, rec, total=None):
        """Host transport: rec numpy [cap, 447] -> list of per-rank blocks as tensors (valid on every rank).  With `total` (this
        rank's instance count, pack_instances(..., lossless=True)): the counts are gathered first and the blocks travel with
        max(cap, largest count) rows, so no instance is dropped when scores tie at the voting threshold."""
        if total is None:
            self.send.copy_(self.torch.from_numpy(rec))
            self.dist.all_gather(self.recv, self.send)
            return self.recv
        torch, dist = self.torch, self.dist
        cnt = [torch.zeros(1, dtype=torch.int32) for _ in range(self.world)]
        dist.all_gather(cnt, torch.tensor([int(total)], dtype=torch.int32))
        self.last_counts = np.array([int(c.item()) for c in cnt], np.int32)
        rows = max(self.cap, int(self.last_counts.max()))
        if rows == self.cap and rec.shape[0] == self.cap:
            self.send.copy_(torch.from_numpy(rec))
            dist.all_gather(self.recv, self.send)
            return self.recv
        send = torch.zeros((rows, REC_DIM), dtype=torch.float32)
        k = min(rows, rec.shape[0])
        send[:k] = torch.from_numpy(np.ascontiguousarray(rec[:k]))
        recv = [torch.empty((rows, REC_DIM), dtype=torch.float32) for _ in range(self.world)]
        dist.all_gather(recv, send)
        return recv

    def close(self):
        if self.net is not None and getattr(self, "_recv", None) is not None:
            self._lib.call("mnc_comm_destroy", self.net._ctx.h)
            if getattr(self, "_pin", None):
                self._lib.call("mnc_host_free", self.net._ctx.h, self._pin)
                self._pin = None
            self._recv.release()
            self._recv_cnt.release()
            self._recv2.release()
            self._send0.release()
            self._recv = None
