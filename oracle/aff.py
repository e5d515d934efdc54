"""Affinity random walk + background/PAR label step restated in numpy fp32
(oracle; test infrastructure only).

Follows utils/affutils.py of the reference:
  compute_trans_mat            :8-24
  scoremap2bbox                :26-53
  generate_cam_label           :55-67
  scale_cam_image              :69-78
  _refine_cams                 :80-89
  refine_cams_with_bkg_weclip  :161-174
  refine_cams_with_aff         :177-223 (seg_attn is None branch :196-198)

OpenCV (cv2.threshold / findContours / boundingRect / resize) is a third-party
dependency that is absent from /root/reference and from this image, version
unpinned upstream -> PARITY UNPINNED for those calls.  Their published
semantics are restated below (`_contour_boxes`, interp.cv2_resize_linear).
"""
import numpy as np

from .interp import cv2_resize_linear


def compute_trans_mat(aff_mat):
    t = np.asarray(aff_mat, np.float32)
    t = t / t.sum(0, keepdims=True, dtype=np.float32)                  # :11
    t = t / t.sum(1, keepdims=True, dtype=np.float32)                  # :12
    for _ in range(2):                                                 # :14-16
        t = t / t.sum(0, keepdims=True, dtype=np.float32)
        t = t / t.sum(1, keepdims=True, dtype=np.float32)
    t = (t + t.T) / np.float32(2)                                      # :17
    t = t @ t                                                          # :19-20
    return t.astype(np.float32)


def _contour_boxes(binary):
    """Bounding rects (x, y, w, h) of cv2.findContours(RETR_TREE, CHAIN_APPROX_SIMPLE)
    on a 0/255 image, restated: every 8-connected foreground component yields one outer
    contour whose boundingRect is the component's tight box.  Hole contours (RETR_TREE
    also returns them) are traced on foreground pixels of the enclosing component, so
    their rects lie inside that component's rect and never change the UNION of boxes,
    which is all refine_cams_with_aff consumes (:210-212).  Only outer boxes are returned.
    """
    H, W = binary.shape
    seen = np.zeros((H, W), bool)
    boxes = []
    for y in range(H):
        for x in range(W):
            if binary[y, x] and not seen[y, x]:
                stack = [(y, x)]
                seen[y, x] = True
                x0 = x1 = x
                y0 = y1 = y
                while stack:
                    cy, cx = stack.pop()
                    x0, x1 = min(x0, cx), max(x1, cx)
                    y0, y1 = min(y0, cy), max(y1, cy)
                    for dy in (-1, 0, 1):
                        for dx in (-1, 0, 1):
                            ny, nx = cy + dy, cx + dx
                            if 0 <= ny < H and 0 <= nx < W and binary[ny, nx] and not seen[ny, nx]:
                                seen[ny, nx] = True
                                stack.append((ny, nx))
                boxes.append((x0, y0, x1 - x0 + 1, y1 - y0 + 1))
    return boxes


def scoremap2bbox(scoremap, threshold):
    """:26-53 with multi_contour_eval=True -> (boxes [cnt,4] as x0,y0,x1,y1; cnt)."""
    height, width = scoremap.shape
    # (scoremap * 255).astype(np.uint8): C truncation toward zero (values are in [0,1])
    img = (np.asarray(scoremap, np.float32) * np.float32(255)).astype(np.uint8)       # :28
    thr = int(threshold * np.max(img))                                                 # :31
    binary = img > thr                                                                 # THRESH_BINARY: src > thresh
    rects = _contour_boxes(binary)                                                     # :34-37
    if len(rects) == 0:
        return np.asarray([[0, 0, 0, 0]]), 1                                           # :39-40
    boxes = []
    for (x, y, w, h) in rects:                                                         # :46-51
        x0, y0, x1, y1 = x, y, x + w, y + h
        x1 = min(x1, width - 1)
        y1 = min(y1, height - 1)
        boxes.append([x0, y0, x1, y1])
    return np.asarray(boxes), len(rects)


def box_mask(scoremap, threshold):
    """The aff_mask built at :209-212 (end-exclusive fill of every box)."""
    boxes, cnt = scoremap2bbox(scoremap, threshold)
    m = np.zeros(scoremap.shape, np.float32)
    for i in range(cnt):
        x0, y0, x1, y1 = boxes[i]
        m[y0:y1, x0:x1] = 1
    return m


def select_attn_layers(aw, seg_attn):
    """:182-195: keep the layers whose attention is closest (in total mass) to the decoder's affinity prediction,
    average them, gate by the prediction.  aw [L,P,P], seg_attn [1,P,P] or [P,P] -> [P,P]."""
    sa = np.asarray(seg_attn, np.float32).reshape(1, *aw.shape[1:])
    diff = (sa - aw).reshape(aw.shape[0], -1).sum(1, dtype=np.float32)                 # :183-184
    th = diff.mean(dtype=np.float32)                                                   # :185
    mask = (diff <= th).astype(np.float32).reshape(-1, 1, 1)                           # :187-190
    sel = (mask * aw).sum(0, dtype=np.float32) / (mask.sum(0, dtype=np.float32) + np.float32(1e-5))   # :193
    return (sel * sa[0]).astype(np.float32)                                            # :195


def refine_cams_with_aff(attr_map, attn_weights, cls_label, size, caa_thre=0.79, attn_layers=6, seg_attn=None):
    """attr_map [P,F], attn_weights [L,N,N], cls_label [F] -> (list of [g,g], cls_lst)."""
    h, w = size
    aw = np.asarray(attn_weights, np.float32)[:, 1:, 1:][-attn_layers:]               # :180
    if seg_attn is not None:
        aw = select_attn_layers(aw, seg_attn)                                          # :182-195
    else:
        aw = aw.mean(0, dtype=np.float32)                                              # :197
    trans = compute_trans_mat(aw)                                                      # :200
    cls_lst = np.where(np.asarray(cls_label) != 0)[0]                                  # :203
    out = []
    for cls in cls_lst:                                                                # :206
        g = np.asarray(attr_map[:, cls], np.float32).reshape(h // 16, w // 16)         # :207
        mask = box_mask(g, caa_thre).reshape(1, -1)                                    # :208-214
        tm = trans * mask                                                              # :215
        out.append((tm @ g.reshape(-1, 1)).reshape(h // 16, w // 16).astype(np.float32))  # :217-220
    return out, cls_lst


def scale_cam_image(img, target_wh):
    img = img - np.min(img)                                                            # :72
    img = img / (np.float32(1e-7) + np.max(img))                                       # :73
    return cv2_resize_linear(img.astype(np.float32), target_wh[0], target_wh[1])       # :75


def refine_cams_with_bkg_weclip(cam_refined_list, img, cls_lst, par, size):
    """-> (cam_labels [1,H,W] int64, cams [k+1,H,W] f32).  `size` = labels.shape[-2:] = (H,W);
    the reference's w/h names are swapped twice and end up right (:163-164,:60)."""
    H, W = size
    cams = np.stack([scale_cam_image(np.asarray(c, np.float32), (W, H)) for c in cam_refined_list], 0)
    bg = np.float32(1) - cams.max(0, keepdims=True)                                    # :165
    cams = np.concatenate([bg, cams], 0).astype(np.float32)                            # :166
    valid_key = np.pad(np.asarray(cls_lst) + 1, (1, 0), mode="constant")               # :168
    refined = par(np.asarray(img, np.float32)[None], cams[None])                       # :84
    label = valid_key[refined.argmax(1)]                                               # :86-87
    return label.astype(np.int64), cams
