"""Input upload overlapped with compute (the device half of the reference's queue-runner input pipeline,
`trainer.py:537-564`: `tf.train.batch` keeps batches ready while the session runs the optimizer ops).

`DevicePrefetcher(source, device)` iterates over `source` (dicts of host tensors, ideally pinned) and yields dicts of
device tensors.  Uploads run on their own HIP stream into a small ring of device slots, `depth` batches ahead of the
consumer, so the PCIe copy of batch n+1 overlaps the kernels of batch n; the consumer's stream only waits on the
slot's "ready" event.  A slot is recycled once the consumer has asked for the batch after it: by then everything
that reads the slot (the trainers copy it into their graphs' static inputs first thing) has been enqueued on the
consumer's stream, and the uploader waits on an event recorded there before overwriting.

`packed=True` moves a batch as ONE copy: the host tensors are packed into a pinned staging buffer (`PackedLayout`,
256-byte aligned fields) and the device tensors handed out are views of one device buffer: one transfer and one
event pair per batch instead of five (measured on the Market step: no different from five copies)."""
import collections

import torch


class PackedLayout(object):
    """Byte layout of a dict of tensors inside one flat uint8 buffer."""
    ALIGN = 256

    def __init__(self, sample):
        self.fields, off = [], 0
        for k, v in sample.items():
            nbytes = v.numel() * v.element_size()
            self.fields.append((k, v.dtype, tuple(v.shape), off, nbytes))
            off += -(-nbytes // self.ALIGN) * self.ALIGN
        self.nbytes = max(off, self.ALIGN)

    def matches(self, batch):
        return len(batch) == len(self.fields) and all(
            k in batch and batch[k].dtype == dt and tuple(batch[k].shape) == sh for k, dt, sh, _, _ in self.fields)

    def views(self, flat):
        """{name: view of `flat` (uint8, >= nbytes) with the field's dtype and shape}."""
        return {k: flat[off:off + n].view(dt).reshape(sh) for k, dt, sh, off, n in self.fields}

    def pack(self, batch, flat):
        for k, view in self.views(flat).items():
            view.copy_(batch[k])
        return flat


class _Slot(object):
    __slots__ = ("buf", "ready", "released", "stage", "flat", "layout", "used")

    def __init__(self):
        self.buf, self.ready, self.released = None, torch.cuda.Event(), None
        self.stage = self.flat = self.layout = None
        self.used = False


class DevicePrefetcher(object):
    def __init__(self, source, device, depth=2, packed=False):
        self.packed = bool(packed)
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
        if self.packed:
            return self._issue_packed(slot, host)
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

    def _issue_packed(self, slot, host):
        if slot.layout is None or not slot.layout.matches(host):
            slot.layout = PackedLayout(host)
            slot.stage = torch.empty(slot.layout.nbytes, dtype=torch.uint8).pin_memory()
            slot.used = False
        elif slot.used:
            slot.ready.synchronize()               # the previous upload out of this staging buffer (long finished)
        slot.layout.pack(host, slot.stage)
        with torch.cuda.stream(self.stream):
            if slot.released is not None:
                self.stream.wait_event(slot.released)
            if slot.flat is None or slot.flat.numel() != slot.layout.nbytes:
                slot.flat = torch.empty(slot.layout.nbytes, dtype=torch.uint8, device=self.device)
                slot.buf = None
            if slot.buf is None or not slot.used:
                slot.buf = slot.layout.views(slot.flat)
            slot.flat.copy_(slot.stage, non_blocking=True)
            slot.ready.record(self.stream)
        slot.used = True
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
