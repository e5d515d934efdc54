"""On-disk MS COCO input format (inference stages): mirror of datasets/coco.py:22-80, :190-230.

  JPEGImages/{train,val}/<COCO_train2014_000000xxxxxx>.jpg, SegmentationClass/{train,val}/<000000xxxxxx>.png (the label file name
  drops the "COCO_train2014_" / "COCO_val2014_" prefix, :62,:68), grey-scale JPEGs are replicated to 3 channels (:22-26).
Samples carry the decoded uint8 image (normalised on the device, ops.normalize_img_u8) like excel_amd/datasets/voc.py.
"""
import os

import numpy as np

from .voc import _imread, load_cls_label_list, load_img_name_list

# the 80 COCO category names in the reference's order (data)
class_list = ['_background_', 'person', 'bicycle', 'car', 'motorcycle', 'airplane', 'bus', 'train', 'truck', 'boat', 'traffic light', 'fire hydrant', 'stop sign', 'parking meter', 'bench', 'bird', 'cat', 'dog', 'horse', 'sheep', 'cow', 'elephant', 'bear', 'zebra', 'giraffe', 'backpack', 'umbrella', 'handbag', 'tie', 'suitcase', 'frisbee', 'skis', 'snowboard', 'sports ball', 'kite', 'baseball bat', 'baseball glove', 'skateboard', 'surfboard', 'tennis racket', 'bottle', 'wine glass', 'cup', 'fork', 'knife', 'spoon', 'bowl', 'banana', 'apple', 'sandwich', 'orange', 'broccoli', 'carrot', 'hot dog', 'pizza', 'donut', 'cake', 'chair', 'couch', 'potted plant', 'bed', 'dining table', 'toilet', 'tv', 'laptop', 'mouse', 'remote', 'keyboard', 'cell phone', 'microwave', 'oven', 'toaster', 'sink', 'refrigerator', 'book', 'clock', 'vase', 'scissors', 'teddy bear', 'hair drier', 'toothbrush']


def robust_read_image(path):
    image = _imread(path)
    return np.stack((image, image, image), axis=-1) if image.ndim < 3 else image


class CocoDataset:
    def __init__(self, root_dir=None, name_list_dir=None, split="train", stage="train"):
        self.root_dir, self.stage = root_dir, stage
        sub = "train" if "train" in split else ("val" if "val" in split else "")
        self.img_dir = os.path.join(root_dir, "JPEGImages", sub)
        self.label_dir = os.path.join(root_dir, "SegmentationClass", sub)
        self.name_list = load_img_name_list(os.path.join(name_list_dir, split + ".txt"))

    def __len__(self):
        return len(self.name_list)

    def __getitem__(self, idx):
        full = str(self.name_list[idx])
        image = robust_read_image(os.path.join(self.img_dir, full + ".jpg"))
        if self.stage == "test":
            return full, full, image, image[:, :, 0]
        short = full[15:] if self.stage == "train" else full[13:]          # strip "COCO_train2014_" / "COCO_val2014_"
        return full, short, image, _imread(os.path.join(self.label_dir, short + ".png"))


class CocoSegDataset(CocoDataset):
    """(name, image uint8 [h,w,3], label uint8 [h,w], cls_label f32 [80]); `batch` as in datasets/voc.VOC12SegDataset."""

    def __init__(self, root_dir=None, name_list_dir=None, split="val", stage="val", ignore_index=255, **kwargs):
        super().__init__(root_dir, name_list_dir, split, stage)
        self.ignore_index = ignore_index
        self.label_list = load_cls_label_list(name_list_dir) if stage != "test" else None

    def __getitem__(self, idx):
        full, short, image, label = super().__getitem__(idx)
        cls = np.zeros(len(class_list) - 1, np.float32) if self.stage == "test" else np.asarray(self.label_list[full], np.float32)
        return full, np.ascontiguousarray(image[..., :3], np.uint8), np.ascontiguousarray(label, np.uint8), cls

    def max_k(self):
        return 18           # the largest number of present classes of a COCO 2014 image

    def batch(self, indices):
        items = [self[i] for i in indices]
        if len({it[1].shape for it in items}) != 1:
            raise ValueError("COCO images have different sizes: use batch_size 1 or resize first")
        return ([it[0] for it in items], np.stack([it[1] for it in items]), np.stack([it[2] for it in items]), np.stack([it[3] for it in items]))
