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


def ragged_batches(dataset, indices, batch_size, num_workers=2, pin_memory=True, prefetch_factor=2):
    """Iterator of RaggedBatch over `indices` in order, `batch_size` samples each (the last one may be smaller), decoded by
    `num_workers` background processes (0: in the calling process), `prefetch_factor` batches ahead per worker."""
    indices = list(indices)
    chunks = [indices[s:s + batch_size] for s in range(0, len(indices), batch_size)]
    kw = dict(batch_size=None, shuffle=False, num_workers=num_workers, pin_memory=pin_memory and torch.cuda.is_available(),
              collate_fn=None)
    if num_workers > 0:
        kw.update(prefetch_factor=prefetch_factor, persistent_workers=False)
    return DataLoader(_Batches(dataset, chunks), **kw)
