"""Input upload overlapped with compute (the device half of the reference's queue-runner input pipeline,
`trainer.py:537-564`: `tf.train.batch` keeps batches ready while the session runs the optimizer ops).

`DevicePrefetcher(source, device)` iterates over `source` (dicts of host tensors, ideally pinned) and yields dicts of
device tensors.  Uploads run on their own HIP stream into a small ring of device slots, `depth` batches ahead of the
consumer, so the PCIe copy of batch n+1 overlaps the kernels of batch n; the consumer's stream only waits on the
slot's "ready" event.  A slot is recycled once the consumer has asked for the batch after it: by then everything
that reads the slot (the trainers copy it into their graphs' static inputs first thing) has been enqueued on the
consumer's stream, and the uploader waits on an event recorded there before overwriting."""
import collections

import torch


class _Slot(object):
    __slots__ = ("buf", "ready", "released")

    def __init__(self):
        self.buf, self.ready, self.released = None, torch.cuda.Event(), None


class DevicePrefetcher(object):
    def __init__(self, source, device, depth=2):
        if depth < 1:
            raise ValueError("depth must be >= 1")
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("DevicePrefetcher uploads to a HIP device; got %s" % self.device)
        self.source = iter(source)
        self.stream = torch.cuda.Stream(self.device)
        self.free = collections.deque(_Slot() for _ in range(depth + 1))
        self.inflight = collections.deque()
        self.held = None
        for _ in range(depth):
            self._issue()

    def _issue(self):
        if not self.free:
            return
        try:
            host = next(self.source)
        except StopIteration:
            return
        slot = self.free.popleft()
        with torch.cuda.stream(self.stream):
            if slot.released is not None:
                self.stream.wait_event(slot.released)
            if slot.buf is None or any(k not in slot.buf or slot.buf[k].shape != v.shape or slot.buf[k].dtype != v.dtype
                                       for k, v in host.items()) or len(slot.buf) != len(host):
                slot.buf = {k: torch.empty(v.shape, dtype=v.dtype, device=self.device) for k, v in host.items()}
            for k, v in host.items():
                slot.buf[k].copy_(v, non_blocking=True)
            slot.ready.record(self.stream)
        self.inflight.append(slot)

    def __iter__(self):
        return self

    def __next__(self):
        cur = torch.cuda.current_stream(self.device)
        if self.held is not None:                      # the previous batch's readers are all enqueued by now
            self.held.released = torch.cuda.Event()
            self.held.released.record(cur)
            self.free.append(self.held)
            self.held = None
        if not self.inflight:
            self._issue()
        if not self.inflight:
            raise StopIteration
        slot = self.inflight.popleft()
        cur.wait_event(slot.ready)
        self.held = slot
        self._issue()
        return slot.buf

    next = __next__
