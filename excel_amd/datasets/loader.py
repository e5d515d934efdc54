"""Ragged batches of decoded samples, prefetched by background worker processes.

The reference feeds its loop with `DataLoader(dataset, batch_size=1, num_workers=2, pin_memory=False)` (tools/infer_lam.py:167):
decode (JPEG / PNG) runs in worker processes, one image per step.  Here the workers decode WHOLE batches of images of different
sizes and hand them over packed - the uint8 HWC images back to back, the uint8 label maps back to back, the sizes - in pinned
host memory, so the training-free pipeline (excel_amd/pipeline.TrainingFreePipeline.run_batch_ragged) gets one H2D copy per
tensor per batch and never waits for a decoder.

Any dataset with `__getitem__(i) -> (name, image uint8 [h,w,3], label uint8 [h,w], cls_label f32 [F])` works
(datasets/voc.VOC12SegDataset, datasets/coco.CocoSegDataset, tools/synthetic.SyntheticSegDataset(u8_images=True)).
torch.utils.data is used as process / shared-memory plumbing only.
"""
import numpy as np
import torch
from threading import current_thread as threading_current
from torch.utils.data import DataLoader, Dataset


class RaggedBatch:
    """names, hw int32 [B,2], images uint8 [sum 3 H W], labels uint8 [sum H W], cls f32 [B,F] (host tensors, pinned when requested)."""
    __slots__ = ("names", "hw", "images", "labels", "cls")

    def __init__(self, names, hw, images, labels, cls):
        self.names, self.hw, self.images, self.labels, self.cls = names, hw, images, labels, cls

    def pin_memory(self):                       # DataLoader(pin_memory=True) calls this in its pinning thread
        self.images, self.labels, self.cls = self.images.pin_memory(), self.labels.pin_memory(), self.cls.pin_memory()
        return self

    def __len__(self):
        return len(self.names)


def pack_samples(items):
    """[(name, image [h,w,3] u8, label [h,w] u8, cls [F])] -> RaggedBatch.  Label maps must have their image's size."""
    names, hw, ims, labs, cls = [], [], [], [], []
    for name, image, label, c in items:
        image, label = np.ascontiguousarray(image, np.uint8), np.ascontiguousarray(label, np.uint8)
        if image.ndim != 3 or image.shape[2] != 3 or image.shape[:2] != label.shape:
            raise ValueError(f"{name}: image {image.shape} / label {label.shape}: expected [h,w,3] uint8 and [h,w] uint8 of one size")
        names.append(str(name))
        hw.append(image.shape[:2])
        ims.append(image.reshape(-1))
        labs.append(label.reshape(-1))
        cls.append(np.asarray(c, np.float32))
    return RaggedBatch(names, np.asarray(hw, np.int32), torch.from_numpy(np.concatenate(ims)), torch.from_numpy(np.concatenate(labs)),
                       torch.from_numpy(np.stack(cls)))


class _Batches(Dataset):
    def __init__(self, dataset, chunks):
        self.dataset, self.chunks = dataset, chunks

    def __len__(self):
        return len(self.chunks)

    def __getitem__(self, i):
        return pack_samples([self.dataset[int(j)] for j in self.chunks[i]])


def ragged_batches(dataset, indices, batch_size, num_workers=2, pin_memory=True, prefetch_factor=2, mp_context=None):
    """Iterator of RaggedBatch over `indices` in order, `batch_size` samples each (the last one may be smaller), decoded by
    `num_workers` background processes (0: in the calling process), `prefetch_factor` batches ahead per worker."""
    indices = list(indices)
    chunks = [indices[s:s + batch_size] for s in range(0, len(indices), batch_size)]
    kw = dict(batch_size=None, shuffle=False, num_workers=num_workers, pin_memory=pin_memory and torch.cuda.is_available(),
              collate_fn=None)
    if num_workers > 0:
        kw.update(prefetch_factor=prefetch_factor, persistent_workers=False)
        if mp_context is not None:
            kw.update(multiprocessing_context=mp_context)
    return DataLoader(_Batches(dataset, chunks), **kw)


def threaded_batches(dataset, indices, batch_size, num_threads=16, ahead=3):
    """Same contract as ragged_batches, decoded by a pool of THREADS of this process (`ahead` batches in flight).  The JPEG / PNG
    decoders of Pillow release the GIL, so a thread pool scales for this workload, hands its arrays over without pickling or shared
    memory, and - unlike forked worker processes - leaves the GPU runtime of the process alone: after the forked workers of a
    DataLoader had exited, host-side event waits of the parent were measured at ~350 ms each on this ROCm stack."""
    from collections import deque
    from concurrent.futures import ThreadPoolExecutor
    indices = list(indices)
    chunks = [indices[s:s + batch_size] for s in range(0, len(indices), batch_size)]
    with ThreadPoolExecutor(max_workers=max(1, num_threads), thread_name_prefix="excel_decode") as pool:
        pending = deque()
        nxt = 0
        while nxt < len(chunks) or pending:
            while nxt < len(chunks) and len(pending) < ahead:
                pending.append([pool.submit(dataset.__getitem__, int(j)) for j in chunks[nxt]])
                nxt += 1
            yield pack_samples([f.result() for f in pending.popleft()])


class DeviceFeeder:
    """Host -> device staging of ragged batches on a COPY stream, a few batches ahead of the compute stream.

    A background thread takes RaggedBatches from `batches` (ragged_batches(..., pin_memory=False): worker processes decode into
    shared memory), copies them into a small ring of pinned staging buffers that are allocated once (pinning fresh memory for
    every batch - what DataLoader(pin_memory=True) does - page-locks 24 MB per batch and was measured to hold the whole loop at
    ~700 img/s), builds the batch's tile map (ops.RaggedPlan, a host function) and issues the H2D copies on its own stream.  The
    consumer gets device tensors that its stream already waits for; copies of batch i+1 overlap the kernels of batch i.

        for names, plan, images, cls, labels in DeviceFeeder(ragged_batches(...), device):
            pipe.run_batch_ragged(images, plan, cls, labels)

    A slot (pinned + device buffers) is reused only after the kernels that read it have finished: the CONSUMER's thread records an
    event on its stream when it asks for the next batch and waits for the oldest such event before it lets more than `slots` - 2
    batches be in flight (every host-side event wait stays in the thread that recorded the event: waiting for it from the staging
    thread was measured at ~350 ms per call on this stack)."""

    def __init__(self, batches, device, slots=4):
        import queue
        import threading
        from .. import ops
        self._ops = ops
        self.device = torch.device(device)
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self._batches = batches
        self._nslots = slots
        self._free = queue.Queue()
        self._ready = queue.Queue(maxsize=slots)
        self._slots = [dict(pin={}, dev={}) for _ in range(slots)]
        for i in range(slots):
            self._free.put(i)
        self._copy_stream = torch.cuda.Stream(device=self.device)
        self._error = None
        self._stop = threading.Event()
        self._last_done = None          # event after the last kernels that read a ring buffer (see close)
        self._thread = threading.Thread(target=self._run, daemon=True)
        self._thread.start()

    def close(self):
        """Stop the staging thread and make the ring safe to free: waits (on the host) for the kernels of the batch handed out last,
        so the buffers - allocated on the copy stream, read on the consumer's - do not return to the caching allocator while a kernel
        still reads them; drains the queues so a staging thread blocked on a full / empty queue exits.  Called when iteration ends
        (normally, by `break`, or by an exception in the consumer) and from __del__."""
        import queue
        self._stop.set()
        try:
            while True:                  # unblock a producer waiting on `_ready.put`
                self._ready.get_nowait()
        except queue.Empty:
            pass
        self._free.put(-1)               # unblock a producer waiting on `_free.get`
        if self._last_done is not None:
            self._last_done.synchronize()
            self._last_done = None
        if self._thread.is_alive() and self._thread is not threading_current():
            self._thread.join(timeout=5.0)
        try:
            while True:
                self._ready.get_nowait()
        except queue.Empty:
            pass

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @staticmethod
    def _grow(store, name, numel, dtype, make):
        t = store.get(name)
        if t is None or t.numel() < numel:
            t = store[name] = make(int(numel * 1.25) + 16, dtype)
        return t[:numel]

    def _stage(self, slot, name, src):
        """host tensor -> pinned ring buffer -> device ring buffer (async on the copy stream) -> device view"""
        n = src.numel()
        pin = self._grow(slot["pin"], name, n, src.dtype, lambda m, dt: torch.empty(m, dtype=dt).pin_memory())
        dev = self._grow(slot["dev"], name, n, src.dtype, lambda m, dt: torch.empty(m, dtype=dt, device=self.device))
        np.copyto(pin.numpy(), src.reshape(-1).numpy())       # plain memcpy (a torch copy_ here spins up an OpenMP team per thread)
        dev.copy_(pin, non_blocking=True)
        return dev.view(src.shape)

    def _run(self):
        try:
            torch.cuda.set_device(self.device)
            for rb in self._batches:
                i = self._free.get()
                if self._stop.is_set() or i < 0:
                    break
                slot = self._slots[i]                          # (the consumer frees a slot only after its kernels finished)
                plan = self._ops.RaggedPlan(rb.hw, None)
                with torch.cuda.stream(self._copy_stream):
                    images = self._stage(slot, "images", rb.images)
                    labels = self._stage(slot, "labels", rb.labels)
                    cls = self._stage(slot, "cls", rb.cls)
                    plan.table = self._stage(slot, "table", torch.from_numpy(plan.table_host))
                    ev = torch.cuda.Event()
                    ev.record(self._copy_stream)
                self._ready.put((i, ev, rb.names, plan, images, cls, labels))
        except BaseException as e:      # surfaced in the consumer
            self._error = e
        finally:
            if not self._stop.is_set():
                self._ready.put(None)

    def __iter__(self):
        from collections import deque
        inflight = deque()              # (slot, event after the last kernel that reads it), oldest first
        cur_slot = None
        cur = torch.cuda.current_stream(self.device)
        try:
            while True:
                cur = torch.cuda.current_stream(self.device)
                if cur_slot is not None:    # everything enqueued for the batch handed out last is in `cur`
                    done = torch.cuda.Event()
                    done.record(cur)
                    inflight.append((cur_slot, done))
                    self._last_done = done
                    cur_slot = None
                while inflight and (len(inflight) > self._nslots - 2 or inflight[0][1].query()):
                    i, done = inflight.popleft()
                    done.synchronize()
                    self._free.put(i)
                item = self._ready.get()
                if item is None:
                    if self._error is not None:
                        raise self._error
                    return
                i, ev, names, plan, images, cls, labels = item
                cur.wait_event(ev)
                cur_slot = i
                yield names, plan, images, cls, labels
        finally:
            # the consumer is done with the iterator (exhausted, `break`, exception): whatever it enqueued for the last batch is in
            # `cur` - record it, then close() waits for it before the ring can be freed
            if cur_slot is not None:
                done = torch.cuda.Event()
                done.record(cur)
                self._last_done = done
            self.close()
