"""Seeded synthetic stand-ins for the data the path consumes (no VOC images / CLIP checkpoint exist offline).

Shapes and statistics follow BASELINE.md section 2 / SURVEY 8(d): images N(0,1) already "normalised", gt uniform in
{0..nc-1} with 2 % ignore (255), present-class count k drawn from the VOC train_aug histogram, ViT weights with
the CLIP ViT-B/16 shapes, text bank = the shipped attribute bank through attr_aggregate with random unit-norm
class/background text features.  numpy RandomState only, so every box regenerates identical data.
"""
import numpy as np

VOC_K_HIST = {1: 6249, 2: 3104, 3: 969, 4: 210, 5: 46, 6: 4}     # datasets/voc/cls_labels_onehot.npy, train_aug


def vit_b16_config():
    return dict(width=768, layers=12, heads=12, patch=16, output_dim=512, input_resolution=224)


def make_vit_state_dict(cfg=None, seed=0, attn_gain=2.0):
    """Random weights keyed like the reference VisionTransformer.state_dict() (clip/clip_surgery_model.py:374-394)."""
    c = dict(vit_b16_config())
    c.update(cfg or {})
    rs = np.random.RandomState(seed)
    D, L, ps, out = c["width"], c["layers"], c["patch"], c["output_dim"]
    g = c["input_resolution"] // ps
    f32 = np.float32

    def rn(*shape, std=1.0):
        return (rs.standard_normal(shape) * std).astype(f32)

    w = {"conv1.weight": rn(D, 3, ps, ps, std=(3 * ps * ps) ** -0.5), "class_embedding": rn(D, std=D ** -0.5),
         "positional_embedding": rn(g * g + 1, D, std=D ** -0.5 * 4)}
    for nm in ("ln_pre", "ln_post"):
        w[nm + ".weight"] = (1.0 + 0.1 * rs.standard_normal(D)).astype(f32)
        w[nm + ".bias"] = rn(D, std=0.1)
    for i in range(L):
        p = f"transformer.resblocks.{i}."
        for nm in ("ln_1", "ln_2"):
            w[p + nm + ".weight"] = (1.0 + 0.1 * rs.standard_normal(D)).astype(f32)
            w[p + nm + ".bias"] = rn(D, std=0.1)
        w[p + "attn.in_proj_weight"] = rn(3 * D, D, std=attn_gain ** 0.5 * D ** -0.5)
        w[p + "attn.in_proj_bias"] = rn(3 * D, std=0.1)
        w[p + "attn.out_proj.weight"] = rn(D, D, std=0.5 * D ** -0.5)
        w[p + "attn.out_proj.bias"] = rn(D, std=0.02)
        w[p + "mlp.c_fc.weight"] = rn(4 * D, D, std=D ** -0.5)
        w[p + "mlp.c_fc.bias"] = rn(4 * D, std=0.1)
        w[p + "mlp.c_proj.weight"] = rn(D, 4 * D, std=0.5 * (4 * D) ** -0.5)
        w[p + "mlp.c_proj.bias"] = rn(D, std=0.02)
    w["proj"] = rn(D, out, std=D ** -0.5)
    return w


def make_text_features(T=45, C=512, seed=7):
    rs = np.random.RandomState(seed)
    t = rs.standard_normal((T, C)).astype(np.float32)
    return t / np.linalg.norm(t, axis=1, keepdims=True)


def draw_k(rs, hist=VOC_K_HIST):
    ks = np.array(sorted(hist))
    p = np.array([hist[k] for k in ks], np.float64)
    return int(rs.choice(ks, p=p / p.sum()))


# Image sizes of a VOC-like data set: PASCAL VOC images have their longer side at 500; most are 500 x 375 (landscape) or 375 x 500
# (portrait), the rest varies.  (H, W, weight) - the last entry draws a uniform random size with the longer side at 500.
VOC_LIKE_SIZES = [(375, 500, 0.55), (500, 375, 0.17), (333, 500, 0.10), (500, 333, 0.04), (334, 500, 0.03), (374, 500, 0.03), (None, None, 0.08)]


def draw_voc_like_size(rs):
    r, acc = rs.rand(), 0.0
    for h, w, p in VOC_LIKE_SIZES:
        acc += p
        if r < acc and h is not None:
            return h, w
    short = int(rs.randint(180, 501))
    return (short, 500) if rs.rand() < 0.7 else (500, short)


class SyntheticSegDataset:
    """Index -> (name, image [3,h,w] f32, label [H,W] u8, cls_label [F] f32), like VOC12SegDataset.__getitem__
    (datasets/voc.py:212-230).  Every sample is generated from its own seed, so shards are order independent."""

    def __init__(self, n, image_hw=(448, 448), label_hw=None, num_classes=21, seed=1234, fixed_k=None, u8_images=False, ragged=False):
        """u8_images: samples carry the DECODED image (uint8 [h,w,3], what imageio.imread returns, datasets/voc.py:52) instead of
        the normalised float tensor; the harness then normalises on the device (ops.normalize_img_u8).
        ragged: every sample has its own VOC-like size (draw_voc_like_size; image and label of one size, uint8 images) - what real
        VOC data looks like to the harness (image_hw / label_hw are ignored)."""
        self.u8_images = u8_images or ragged
        self.ragged = ragged
        self.n = n
        self.image_hw = tuple(image_hw)
        self.label_hw = tuple(label_hw or image_hw)
        self.num_classes = num_classes
        self.seed = seed
        self.fixed_k = fixed_k

    def __len__(self):
        return self.n

    def max_k(self):
        return self.fixed_k or max(VOC_K_HIST)

    def size_of(self, i):
        """(H, W) of sample i without generating it."""
        if not self.ragged:
            return self.label_hw
        return draw_voc_like_size(np.random.RandomState((self.seed * 7919 + 31 * i + 5) % (2 ** 31 - 1)))

    def __getitem__(self, i):
        rs = np.random.RandomState((self.seed * 1000003 + i) % (2 ** 31 - 1))
        F = self.num_classes - 1
        image_hw = label_hw = self.size_of(i) if self.ragged else None
        if not self.ragged:
            image_hw, label_hw = self.image_hw, self.label_hw
        if self.ragged:
            # smooth-ish content (a coarse random field, up-sampled) + noise: JPEG-like statistics at a fraction of the cost of
            # drawing h*w*3 normals per sample
            h, w = image_hw
            coarse = rs.randint(0, 256, (h // 8 + 2, w // 8 + 2, 3)).astype(np.int16)
            img = np.repeat(np.repeat(coarse, 8, 0), 8, 1)[:h, :w]
            img = np.clip(img + rs.randint(-24, 25, (h, w, 1)), 0, 255).astype(np.uint8)
        else:
            img = rs.standard_normal((3,) + image_hw).astype(np.float32)
            if self.u8_images:
                img = np.clip(img.transpose(1, 2, 0) * 58.0 + 116.0, 0, 255).astype(np.uint8)      # same stream of random numbers
        gt = rs.randint(0, self.num_classes, label_hw).astype(np.uint8)
        gt[rs.rand(*label_hw) < 0.02] = 255
        k = self.fixed_k or draw_k(rs)
        cls = np.zeros(F, np.float32)
        cls[rs.choice(F, size=k, replace=False)] = 1
        return f"syn_{i:06d}", img, gt, cls

    def batch(self, indices):
        items = [self[i] for i in indices]
        return ([it[0] for it in items], np.stack([it[1] for it in items]), np.stack([it[2] for it in items]),
                np.stack([it[3] for it in items]))


def write_voc_tree(root, lists, n, seed=1234, split="train", num_classes=21, quality=90):
    """Write `n` ragged synthetic samples as an on-disk PASCAL-VOC-format tree (what datasets/voc.VOC12SegDataset reads):
    <root>/JPEGImages/<id>.jpg, <root>/SegmentationClassAug/<id>.png (palette PNG, 255 = ignore), <lists>/<split>.txt and
    <lists>/cls_labels_onehot.npy (pickled {id: one-hot[20]}).  -> (ids, number of scored pixels)."""
    import os
    from PIL import Image
    from ..utils import imutils
    os.makedirs(os.path.join(root, "JPEGImages"), exist_ok=True)
    os.makedirs(os.path.join(root, "SegmentationClassAug"), exist_ok=True)
    os.makedirs(lists, exist_ok=True)
    ds = SyntheticSegDataset(n, num_classes=num_classes, seed=seed, ragged=True)
    palette = imutils.colormap().flatten().tolist()
    ids, onehot, npix = [], {}, 0
    for i in range(n):
        name, img, gt, cls = ds[i]
        name = "2008_%06d" % i
        ids.append(name)
        Image.fromarray(img).save(os.path.join(root, "JPEGImages", name + ".jpg"), quality=quality)
        im = Image.fromarray(gt, mode="P")
        im.putpalette(palette)
        im.save(os.path.join(root, "SegmentationClassAug", name + ".png"))
        onehot[name] = cls
        npix += int((gt < num_classes).sum())
    with open(os.path.join(lists, split + ".txt"), "w") as f:
        f.write("\n".join(ids) + "\n")
    np.save(os.path.join(lists, "cls_labels_onehot.npy"), onehot)
    return ids, npix
