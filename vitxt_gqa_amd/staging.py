"""Host -> device input staging for the T2S boundary.

The reference collates a batch by stacking per-sample tensors into pageable host tensors
(``pythia/common/batch_collator.py:5-15`` -> ``SampleList.__init__`` ``pythia/common/sample.py:78-118``) and moves it
with one ``.to(device)`` per field (``sample.py:296-326``, called from ``base_trainer.py:251-258``): at the BASELINE shape
(B=64, 100 frames x 100 OCR) that is 2.4 GB in ~20 synchronous pageable copies per step.

Here a batch lives in ONE arena with a fixed layout (every field 256-byte aligned):

* the host side of the arena is pinned memory; ``collate`` writes each sample's fields straight into its row of the
  batch views (no intermediate stacked tensor),
* ``upload`` is a single asynchronous H2D copy of the used prefix of the arena on a side stream, followed by an event,
* the device side is a same-layout arena in HBM; the ``SampleList`` handed to the model is made of views into it, and
  the consumer stream waits on the event, so the copy of batch i+1 overlaps the train step of batch i,
* ``depth`` arenas form a ring; an arena is reused only after the consumer stream has passed the point where its batch
  was last used (``release`` records that point).

* fields can be produced ON the device instead of being transferred: ``device_only`` adds fields that exist only in the
  HBM arena and ``post_upload`` hooks fill them on the copy stream right after the H2D copy - e.g. the 604-d PHOC rows
  of the OCR tokens from their bytes (``phoc_expander``): 64 bytes instead of 2416 per token cross PCIe.

Non-tensor fields (question ids, encoded strings ...) travel in the ``SampleList`` untouched.  On a CPU-only host
(``device="cpu"``) the same layout is used without pinning and without streams, which is what the CPU tests exercise.
"""
import threading
from collections import OrderedDict

import numpy as np
import torch

from .sample import SampleList

ALIGN = 256


def _nbytes(shape, dtype):
    n = 1
    for s in shape:
        n *= int(s)
    return n * torch.empty(0, dtype=dtype).element_size()


class ArenaLayout:
    """Byte layout of one batch: ``spec`` maps field name -> (per-batch shape, torch dtype), in order."""

    def __init__(self, spec):
        self.fields = OrderedDict()
        off = 0
        for name, (shape, dtype) in spec.items():
            nb = _nbytes(shape, dtype)
            self.fields[name] = (off, nb, tuple(int(s) for s in shape), dtype)
            off += (nb + ALIGN - 1) // ALIGN * ALIGN
        self.nbytes = off

    @staticmethod
    def from_batch(batch):
        """Layout of an example batch (dict of tensors / numpy arrays; other values are skipped)."""
        spec = OrderedDict()
        for k, v in batch.items():
            if isinstance(v, np.ndarray):
                v = torch.from_numpy(v)
            if torch.is_tensor(v):
                spec[k] = (tuple(v.shape), v.dtype)
        return ArenaLayout(spec)

    def views(self, arena):
        """Typed views of a flat uint8 arena tensor."""
        out = OrderedDict()
        for name, (off, nb, shape, dtype) in self.fields.items():
            out[name] = arena[off:off + nb].view(dtype).view(shape)
        return out


class _Slot:
    def __init__(self, layout, device, pin, dev_layout=None):
        self.layout = layout
        self.host_arena = torch.empty(max(layout.nbytes, ALIGN), dtype=torch.uint8, pin_memory=pin)
        self.host = layout.views(self.host_arena)
        extra = dev_layout.nbytes if dev_layout is not None else 0
        if device.type == "cpu" and not extra:
            self.dev_arena, self.dev = self.host_arena, self.host
        else:
            self.dev_arena = torch.empty(max(layout.nbytes, ALIGN) + extra, dtype=torch.uint8, device=device)
            self.dev = layout.views(self.dev_arena)
            if extra:       # device-only fields live behind the transferred prefix
                self.dev.update(dev_layout.views(self.dev_arena[max(layout.nbytes, ALIGN):]))
        self.extras = {}
        self.ready = None        # event: H2D copy complete (recorded on the copy stream)
        self.released = None     # event: consumer done with the device views (recorded on the consumer stream)


class StagedBatch(SampleList):
    """Device ``SampleList`` whose tensors are views into an arena slot.  ``wait()`` makes the current stream wait for
    the upload; ``release()`` tells the stager the views are no longer needed (call after the step's last use)."""

    def wait(self):
        slot = self.__dict__.get("_slot")
        if slot is not None and slot.ready is not None:
            torch.cuda.current_stream().wait_event(slot.ready)
        return self

    def release(self):
        slot = self.__dict__.get("_slot")
        if slot is not None and slot.dev_arena.is_cuda:
            slot.released = torch.cuda.Event()
            slot.released.record(torch.cuda.current_stream())


class BatchStager:
    def __init__(self, layout, device="cuda:0", depth=2, device_only=None, post_upload=()):
        """``device_only``: spec {name: (shape, dtype)} of fields that exist only in the device arena; ``post_upload``:
        callables ``f(dev_views)`` run on the copy stream after the H2D copy (they fill the device-only fields)."""
        if depth < 1:
            raise ValueError("BatchStager needs at least one arena slot")
        self.layout = layout
        self.device = torch.device(device)
        self.cuda = self.device.type == "cuda"
        self.dev_layout = ArenaLayout(device_only) if device_only else None
        self.post_upload = tuple(post_upload)
        self.slots = [_Slot(layout, self.device, pin=self.cuda, dev_layout=self.dev_layout) for _ in range(depth)]
        self.copy_stream = torch.cuda.Stream(device=self.device) if self.cuda else None
        self._next = 0
        self._lock = threading.Lock()

    # -- producer side -------------------------------------------------------------------------------------------
    def acquire(self):
        """Next arena slot of the ring, safe to overwrite: its previous upload has completed and the consumer has
        released it (both waited for on the host: this runs in the loader thread, not in the training loop)."""
        with self._lock:
            slot = self.slots[self._next]
            self._next = (self._next + 1) % len(self.slots)
        if self.cuda:
            if slot.ready is not None:
                slot.ready.synchronize()
            if slot.released is not None:
                slot.released.synchronize()
                slot.released = None
        slot.extras = {}
        return slot

    def collate(self, samples, slot=None):
        """Write a list of per-sample dicts straight into the pinned batch views (the reference stacks them into fresh
        pageable tensors, ``sample.py:78-118``).  Tensor fields missing from the layout raise; non-tensor fields are
        gathered into lists."""
        slot = slot or self.acquire()
        B = len(samples)
        for name, (_, _, shape, dtype) in self.layout.fields.items():
            if shape[0] != B:
                raise ValueError("layout batch size %d != %d samples (field %s)" % (shape[0], B, name))
            dst = slot.host[name]
            for i, smp in enumerate(samples):
                v = smp[name]
                if isinstance(v, np.ndarray):
                    v = torch.from_numpy(v)
                dst[i].copy_(v if torch.is_tensor(v) else torch.as_tensor(v, dtype=dtype))
        for k in samples[0]:
            if k not in self.layout.fields:
                slot.extras[k] = [smp[k] for smp in samples]
        return slot

    def fill(self, batch, slot=None):
        """Copy an already stacked host batch (dict of tensors) into the pinned views."""
        slot = slot or self.acquire()
        for k, v in batch.items():
            if k in self.layout.fields:
                if isinstance(v, np.ndarray):
                    v = torch.from_numpy(v)
                slot.host[k].copy_(v)
            else:
                slot.extras[k] = v
        return slot

    def upload(self, slot):
        """One async H2D copy of the whole arena on the copy stream; returns the device ``StagedBatch``."""
        if self.cuda:
            with torch.cuda.stream(self.copy_stream):
                slot.dev_arena[:slot.host_arena.numel()].copy_(slot.host_arena, non_blocking=True)
                for hook in self.post_upload:
                    hook(slot.dev)
                slot.ready = torch.cuda.Event()
                slot.ready.record(self.copy_stream)
        else:
            if slot.dev_arena is not slot.host_arena:
                slot.dev_arena[:slot.host_arena.numel()].copy_(slot.host_arena)
            for hook in self.post_upload:
                hook(slot.dev)
        out = StagedBatch(slot.dev)
        for k, v in slot.extras.items():
            out[k] = v
        out.__dict__["_slot"] = slot
        return out

    # -- consumer side -------------------------------------------------------------------------------------------
    def prefetch(self, batches, stacked=True):
        """Generator over device batches with one upload in flight: batch i+1 is filled and uploaded (by a loader thread)
        while the caller works on batch i.  ``batches`` yields stacked host batches (``stacked=True``) or lists of
        per-sample dicts.  Each yielded batch has been ``wait()``-ed on the current stream; it is released when the next
        one is requested."""
        if len(self.slots) < 2:
            raise ValueError("prefetch() needs a ring of depth >= 2: the next batch is uploaded while the current one is in use")
        it = iter(batches)
        box = {}

        def produce():
            box.clear()                               # never leave the previous (already released) batch behind
            try:
                try:
                    b = next(it)
                except StopIteration:
                    box["next"] = None
                    return
                slot = self.fill(b) if stacked else self.collate(b)
                box["next"] = self.upload(slot)
            except BaseException as e:                # fill / collate / upload failed: hand the error to the consumer
                box["error"] = e

        th = threading.Thread(target=produce)
        th.start()
        while True:
            th.join()
            if "error" in box:
                raise RuntimeError("BatchStager.prefetch: the loader thread failed") from box["error"]
            cur = box.get("next")
            if cur is None:
                return
            th = threading.Thread(target=produce)
            th.start()
            if self.cuda:
                cur.wait()
            try:
                yield cur
            finally:
                cur.release()


def phoc_expander(slots_field="ocr_token_slots", out_field="context_feature_1"):
    """``post_upload`` hook: context_feature_1 [B, N, 604] (device-only field) from the uploaded OCR token bytes
    [B, N, width] with the ``t2s_phoc`` kernel (the reference computes these rows on the host, one C call per token:
    ``pythia/datasets/processors.py:904-928``)."""
    from . import ops

    def hook(dev):
        ops.phoc(dev[slots_field], out=dev[out_field])

    return hook
