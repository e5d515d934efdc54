"""On-disk PASCAL VOC input format (the data side before the path): mirror of the inference-time parts of datasets/voc.py.

  load_img_name_list / load_cls_label_list   :21-27   (split .txt of image ids; cls_labels_onehot.npy = pickled {id: one-hot[20]})
  VOC12Dataset.__getitem__                   :49-70   (JPEGImages/<id>.jpg, SegmentationClassAug/<id>.png; test: no label)
  VOC12SegDataset (stage val/test, aug off)  :133-230 (normalize_img + HWC->CHW happen on the DEVICE here: samples carry the
                                                        decoded uint8 image, 3 B/pixel over PCIe; ops.normalize_img_u8)

Decoding uses PIL (the reference's imageio.v2.imread delegates to the same Pillow decoder for .jpg/.png).  Training-time
augmentation (random crop / flip / colour jitter, :176-200) belongs to the training loop and is not mirrored.
"""
import os

import numpy as np

class_list = ["_background_", "aeroplane", "bicycle", "bird", "boat", "bottle", "bus", "car", "cat", "chair", "cow", "diningtable",
              "dog", "horse", "motorbike", "person", "pottedplant", "sheep", "sofa", "train", "tvmonitor"]


def load_img_name_list(img_name_list_path):
    return np.loadtxt(img_name_list_path, dtype=str, ndmin=1)


def load_cls_label_list(name_list_dir):
    return np.load(os.path.join(name_list_dir, "cls_labels_onehot.npy"), allow_pickle=True).item()


def _imread(path):
    from PIL import Image
    with Image.open(path) as im:
        return np.asarray(im)          # palette PNGs give the index map, JPEGs the RGB array (like imageio.v2.imread)


class VOC12Dataset:
    def __init__(self, root_dir=None, name_list_dir=None, split="train", stage="train"):
        self.root_dir, self.stage = root_dir, stage
        self.img_dir = os.path.join(root_dir, "JPEGImages")
        self.label_dir = os.path.join(root_dir, "SegmentationClassAug")
        self.name_list = load_img_name_list(os.path.join(name_list_dir, split + ".txt"))

    def __len__(self):
        return len(self.name_list)

    def __getitem__(self, idx):
        name = str(self.name_list[idx])
        image = _imread(os.path.join(self.img_dir, name + ".jpg"))
        if image.ndim == 2:
            image = np.stack([image] * 3, -1)
        label = image[:, :, 0] if self.stage == "test" else _imread(os.path.join(self.label_dir, name + ".png"))
        return name, image, label


class VOC12SegDataset(VOC12Dataset):
    """Inference-time samples: (name, image uint8 [h,w,3], label uint8 [h,w], cls_label f32 [20]).  `batch` keeps the calling
    convention of tools/synthetic.SyntheticSegDataset so tools/infer_lam.build_validation can consume either."""

    def __init__(self, root_dir=None, name_list_dir=None, split="val", stage="val", ignore_index=255, **kwargs):
        super().__init__(root_dir, name_list_dir, split, stage)
        self.ignore_index = ignore_index
        self.label_list = load_cls_label_list(name_list_dir) if stage != "test" else None

    def __getitem__(self, idx):
        name, image, label = super().__getitem__(idx)
        cls = np.zeros(len(class_list) - 1, np.float32) if self.stage == "test" else np.asarray(self.label_list[name], np.float32)
        return name, np.ascontiguousarray(image[..., :3], np.uint8), np.ascontiguousarray(label, np.uint8), cls

    def max_k(self):
        return 6            # the largest number of present classes of a VOC train_aug image

    def batch(self, indices):
        items = [self[i] for i in indices]
        if len({it[1].shape for it in items}) != 1:
            raise ValueError("VOC images have different sizes: use batch_size 1 (tools/infer_lam.py:167 does) or resize first")
        return ([it[0] for it in items], np.stack([it[1] for it in items]), np.stack([it[2] for it in items]), np.stack([it[3] for it in items]))
