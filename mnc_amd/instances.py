"""Final instances of one image as a fixed-shape device block: what gpu_mask_voting returns
(lib/transform/mask_transform.py:213-286), produced and kept on the GPU by mnc_vote_instances.

Device layout of one InstanceBlock:  [HEAD_BYTES: int32 counts[num_classes]: counts[0] = R, counts[c] = rows of class c]
                                     [rows_cap records of REC_DIM float32: x1, y1, x2, y2, score, class id (1..20), 21x21 mask]
Rows past R are zero (class id 0 == padding).  The first `gather_rows` (= max_per_image = 100) records are the block every rank
contributes to the RCCL all-gather; R exceeds that only when scores tie at the global threshold (the reference's
`cls_score >= thresh`, :258, admits every tied row), which a gather truncates -- and says so."""
import numpy as np

from . import _lib

HEAD_BYTES = 256


class InstanceBlock(object):
    def __init__(self, net, num_classes, mask_size, max_per_image, n):
        from .engine import _DevBuf
        self._net = net
        self.num_classes, self.S = int(num_classes), int(mask_size)
        self.rec_dim = 6 + self.S * self.S
        self.gather_rows = int(max_per_image)
        self.rows_cap = max((self.num_classes - 1) * min(int(max_per_image), int(n)), self.gather_rows, 1)
        if self.num_classes * 4 > HEAD_BYTES:
            raise ValueError("at most %d classes" % (HEAD_BYTES // 4))
        self._buf = _DevBuf(net._ctx)
        self.ptr = self._buf.ensure(HEAD_BYTES + self.rows_cap * self.rec_dim * 4)
        self._host = None
        self.generation = 0            # bumped by invalidate(): every image voted into this buffer is one generation

    counts_ptr = property(lambda self: self.ptr)
    records_ptr = property(lambda self: self.ptr + HEAD_BYTES)

    def fits(self, num_classes, mask_size, max_per_image, n):
        return (self.num_classes == num_classes and self.S == mask_size and self.gather_rows == max_per_image
                and self.rows_cap >= (num_classes - 1) * min(max_per_image, n))

    def invalidate(self):
        """The buffer is about to receive another image's instances."""
        self._host = None
        self.generation += 1

    def view(self):
        """What Net.vote_instances returns: this image's handle on the (reused) buffer.  It reads the device records lazily like
        the block itself, keeps what it has copied, and refuses to hand out a LATER image's data (as DeviceArray.is_current)."""
        return InstanceView(self)

    def head(self):
        """counts int32[num_classes] alone (a 256-byte copy; synchronises the net's stream)."""
        if self._host is not None:
            return self._host[0]
        raw = np.zeros(HEAD_BYTES, np.uint8)
        _lib.call("mnc_d2h", self._net._ctx.h, _lib.ptr(raw), self.ptr, HEAD_BYTES)
        return raw[:self.num_classes * 4].view(np.int32).copy()

    def fetch(self):
        """-> (counts int32[num_classes], records float32[R, rec_dim]): ONE device-to-host copy of the head and the first
        gather_rows records (a second one only when more rows tied at the threshold); synchronises the net's stream."""
        if self._host is None:
            import ctypes
            first = min(self.gather_rows, self.rows_cap)
            nbytes = HEAD_BYTES + first * self.rec_dim * 4
            # through pinned memory: a copy into pageable memory takes the runtime's staging path (0.45 ms for these 179 KB at
            # 1000 RoIs, and it waits for every stream of the device)
            if getattr(self, "_pin", None) is None or self._pin_cap < nbytes:
                if getattr(self, "_pin", None):
                    _lib.call("mnc_host_free", self._net._ctx.h, self._pin)
                p = ctypes.c_void_p()
                _lib.call("mnc_host_alloc", self._net._ctx.h, nbytes, ctypes.addressof(p))
                self._pin, self._pin_cap = p.value, nbytes
            _lib.call("mnc_d2h_async", self._net._ctx.h, self._pin, self.ptr, nbytes)
            _lib.call("mnc_ctx_sync", self._net._ctx.h)
            raw = np.frombuffer((ctypes.c_char * nbytes).from_address(self._pin), dtype=np.uint8).copy()
            counts = raw[:self.num_classes * 4].view(np.int32).copy()
            rec = raw[HEAD_BYTES:].view(np.float32).reshape(first, self.rec_dim)
            R = int(counts[0])
            if R > first:
                more = np.zeros((R - first, self.rec_dim), np.float32)
                _lib.call("mnc_d2h", self._net._ctx.h, _lib.ptr(more), self.records_ptr + first * self.rec_dim * 4, more.nbytes)
                rec = np.concatenate((rec, more), 0)
            self._host = (counts, rec[:R])
        return self._host

    def lists(self):
        """-> (list_result_mask, list_result_box) exactly as the reference's gpu_mask_voting returns them: one entry per
        foreground class, masks [k,1,S,S] float32, boxes [k,5] = (x1, y1, x2, y2, score) (int32 | float32 hstack -> float64)."""
        counts, rec = self.fetch()
        return split_records(rec, counts[1:self.num_classes], self.S)

    def release(self):
        if getattr(self, "_pin", None):
            _lib.call("mnc_host_free", self._net._ctx.h, self._pin)
            self._pin = None
        self._buf.release()


class InstanceView(object):
    """One image's instances in an InstanceBlock whose buffer later images reuse."""

    def __init__(self, block):
        self._blk = block
        self._gen = block.generation
        self._host = None
        self.num_classes, self.S, self.rec_dim = block.num_classes, block.S, block.rec_dim
        self.gather_rows, self.rows_cap = block.gather_rows, block.rows_cap

    def is_current(self):
        return self._blk.generation == self._gen

    def _check(self):
        if not self.is_current():
            raise RuntimeError("this image's instance block has been reused by a later vote_instances (call .fetch() / .lists() "
                               "before the next image if the results must outlive it)")

    counts_ptr = property(lambda self: (self._check(), self._blk.counts_ptr)[1])
    records_ptr = property(lambda self: (self._check(), self._blk.records_ptr)[1])

    def fetch(self):
        if self._host is None:
            self._check()
            self._host = self._blk.fetch()
        return self._host

    def head(self):
        if self._host is not None:
            return self._host[0]
        self._check()
        return self._blk.head()

    def lists(self):
        counts, rec = self.fetch()
        return split_records(rec, counts[1:self.num_classes], self.S)


def split_records(rec, class_counts, S):
    boxes = np.hstack((rec[:, :4].astype(np.int32), rec[:, 4:5]))          # int32 | float32 -> float64, as the reference
    masks = np.ascontiguousarray(rec[:, 6:]).reshape(-1, 1, S, S)
    list_mask, list_box, lo = [], [], 0
    for k in class_counts:
        hi = lo + int(k)
        list_box.append(boxes[lo:hi, :])
        list_mask.append(masks[lo:hi])
        lo = hi
    return list_mask, list_box


def records_from_lists(list_mask, list_box, cap, S=21):
    """Host-side twin of the device block (the gloo tests and the numpy result path): ([cap, 6+S*S] float32, total rows)."""
    rec = np.zeros((cap, 6 + S * S), np.float32)
    n = total = 0
    for c, (m, b) in enumerate(zip(list_mask, list_box)):
        total += len(b)
        k = min(len(b), cap - n)
        if k <= 0:
            continue
        rec[n:n + k, :5] = b[:k]
        rec[n:n + k, 5] = c + 1
        rec[n:n + k, 6:] = np.asarray(m[:k], np.float32).reshape(k, -1)
        n += k
    return rec, total
