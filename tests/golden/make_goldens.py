#!/usr/bin/env python3
"""Mint golden vectors by running the REFERENCE's own Python in the build container.

Runs only where /root/reference exists (never on the GPU box, never from tests).
It imports the reference modules by file path with the shims SURVEY.md 8(c)
lists, feeds them seeded synthetic inputs, and writes small .npz fixtures next
to this file.  Only arrays are written - no reference source text.

Shims (harness-side only):
  * clip/clip_surgery_model.py is loaded by path (imports only torch/numpy).
  * clip/clip.py cannot be imported (torchvision missing): the two pure-torch
    functions on the hot path are executed from the reference file's own text
    via ast (function bodies are taken from /root/reference at run time).
  * utils/affutils.py needs `cv2`: a stub module implements threshold /
    findContours / boundingRect / contourArea / resize with scipy.ndimage and
    torch (so everything AROUND the cv2 calls is the reference's code; the cv2
    calls themselves stay "parity unpinned").
  * Tensor.cuda / Module.cuda are patched to identity (no GPU here).
  * model/load_attr.py: stub `clip` module in sys.modules, cwd=/root/reference.

Usage:  python tests/golden/make_goldens.py
"""
import ast
import importlib.util
import os
import sys
import types

import numpy as np
import torch
import torch.nn.functional as F

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle.vit import VitConfig, make_vit_weights  # noqa: E402  (seeded weight generator only)

torch.manual_seed(0)
torch.set_num_threads(8)

# ---------------------------------------------------------------- shims
torch.Tensor.cuda = lambda self, *a, **k: self
torch.nn.Module.cuda = lambda self, *a, **k: self


def _load_by_path(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _functions_from_source(path, names, glb):
    tree = ast.parse(open(path).read())
    body = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in names]
    code = compile(ast.Module(body=body, type_ignores=[]), path, "exec")
    exec(code, glb)
    return [glb[n] for n in names]


def _make_cv2_stub():
    from scipy import ndimage
    cv2 = types.ModuleType("cv2")
    cv2.__version__ = "4.0.0-stub"
    cv2.THRESH_BINARY = 0
    cv2.RETR_TREE = 3
    cv2.CHAIN_APPROX_SIMPLE = 2

    def threshold(src, thresh, maxval, type):
        return thresh, np.where(src > thresh, maxval, 0).astype(np.uint8)

    def findContours(image, mode, method):
        img = np.asarray(image)
        if img.ndim == 3:
            img = img[..., 0]
        lab, n = ndimage.label(img > 0, structure=np.ones((3, 3), int))
        contours = []
        for sl in ndimage.find_objects(lab):
            ys, xs = sl
            contours.append(("box", xs.start, ys.start, xs.stop - xs.start, ys.stop - ys.start))
        return contours, None

    def boundingRect(c):
        return c[1], c[2], c[3], c[4]

    def contourArea(c):
        return c[3] * c[4]

    def resize(img, dsize):
        w, h = dsize
        t = torch.from_numpy(np.asarray(img, np.float32))[None, None]
        return F.interpolate(t, size=(h, w), mode="bilinear", align_corners=False)[0, 0].numpy()

    cv2.threshold, cv2.findContours, cv2.boundingRect = threshold, findContours, boundingRect
    cv2.contourArea, cv2.resize = contourArea, resize
    return cv2


sys.modules["cv2"] = _make_cv2_stub()
sys.path.insert(0, REF)
csm = _load_by_path("ref_clip_surgery_model", os.path.join(REF, "clip/clip_surgery_model.py"))
ref_clip_fs, = _functions_from_source(os.path.join(REF, "clip/clip.py"), ["clip_feature_surgery"],
                                      {"torch": torch, "np": np})
from utils import PAR as ref_PAR_mod          # noqa: E402
from utils import evaluate as ref_eval        # noqa: E402
from utils import affutils as ref_aff         # noqa: E402

sys.modules.setdefault("clip", types.ModuleType("clip"))
os.chdir(REF)
from model.load_attr import attr_aggregate as ref_attr_aggregate  # noqa: E402


def build_ref_vit(cfg: VitConfig, weights, feat_size, mode):
    m = csm.VisionTransformer(cfg.input_resolution, cfg.patch, cfg.width, cfg.layers, cfg.heads, cfg.out_dim)
    sd = {k: torch.from_numpy(v.copy()) for k, v in weights.items()}
    missing = m.load_state_dict(sd, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    m.eval()
    m.reload_self_attn(layers=6, feat_size=feat_size, mode=mode)
    return m


def ref_generate_clip_fts(vit, inputs):
    # clip/clip.py:348-358 with model.encode_image == visual(...) (clip_surgery_model.py:548-549)
    image_features, attn_list, feat_list = vit(inputs, True, None)
    image_features_n = image_features / image_features.norm(dim=1, keepdim=True)
    return image_features, image_features_n, torch.stack(attn_list, 0), [f.clone() for f in feat_list]


def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **out)
    print(f"wrote {name}: {os.path.getsize(path)/1024:.1f} KB")


# ---------------------------------------------------------------- 1. tiny ViT + CAM
TINY = VitConfig(width=128, layers=8, heads=2, patch=16, out_dim=64, input_resolution=64, n_surgery=5)


def gold_vit_cam():
    w = make_vit_weights(TINY, seed=11)
    rs = np.random.RandomState(21)
    imgs = rs.standard_normal((2, 3, 96, 96)).astype(np.float32)
    out = {}
    for mode in ("train", "val"):
        vit = build_ref_vit(TINY, w, feat_size=6, mode=mode)
        x, f, attn, feats = ref_generate_clip_fts(vit, torch.from_numpy(imgs))
        text = rs.standard_normal((9, 64)).astype(np.float32)
        text /= np.linalg.norm(text, axis=1, keepdims=True)
        cam = ref_clip_fs(f, torch.from_numpy(text))
        out.update({f"{mode}_x": x, f"{mode}_image_features": f, f"{mode}_attn": attn,
                    f"{mode}_feat0": feats[0], f"{mode}_feat5": feats[2], f"{mode}_feat_last": feats[-1],
                    f"{mode}_text": text, f"{mode}_cam": cam})
    # a second resolution through the 'train' model (pos-emb re-interpolated from the 6x6 grid, quirk Q5)
    vit = build_ref_vit(TINY, w, feat_size=6, mode="train")
    imgs2 = rs.standard_normal((1, 3, 64, 64)).astype(np.float32)
    x2, f2, attn2, _ = ref_generate_clip_fts(vit, torch.from_numpy(imgs2))
    out.update(res2_imgs=imgs2, res2_x=x2, res2_attn_last=attn2[-1])
    save("vit_cam_tiny.npz", seed_w=11, imgs=imgs, **out)


# ---------------------------------------------------------------- 2. op-level goldens
def gold_ops():
    rs = np.random.RandomState(5)
    # clip_feature_surgery on random normalised features
    f = rs.standard_normal((2, 37, 64)).astype(np.float32)
    f = f / np.linalg.norm(f, axis=1, keepdims=True)
    t = rs.standard_normal((9, 64)).astype(np.float32)
    t /= np.linalg.norm(t, axis=1, keepdims=True)
    cam = ref_clip_fs(torch.from_numpy(f), torch.from_numpy(t))
    # compute_trans_mat
    a = rs.rand(36, 36).astype(np.float32) + 0.01
    tm = ref_aff.compute_trans_mat(torch.from_numpy(a))
    # PAR, non-square, odd sizes, dilation 24 > size/2
    par = ref_PAR_mod.PAR(num_iter=20, dilations=[1, 2, 4, 8, 12, 24])
    pimg = rs.standard_normal((1, 3, 24, 24)).astype(np.float32)
    pmask = rs.rand(1, 3, 31, 45).astype(np.float32)
    pout = par(torch.from_numpy(pimg), torch.from_numpy(pmask))
    par3 = ref_PAR_mod.PAR(num_iter=3, dilations=[1, 2, 4, 8, 12, 24])
    pimg2 = rs.standard_normal((2, 3, 40, 56)).astype(np.float32)
    pmask2 = rs.rand(2, 4, 40, 56).astype(np.float32)
    pout2 = par3(torch.from_numpy(pimg2), torch.from_numpy(pmask2))
    # scores with 255s
    gts = [rs.randint(0, 21, (17, 23)).astype(np.int16) for _ in range(3)]
    for g in gts:
        g[rs.rand(*g.shape) < 0.05] = 255
    preds = [rs.randint(0, 21, (17, 23)).astype(np.int16) for _ in range(3)]
    with np.errstate(all="ignore"):
        sc = ref_eval.scores(gts, preds, num_classes=21)
    hist = sum(ref_eval._fast_hist(g.flatten(), p.flatten(), 21) for g, p in zip(gts, preds))
    save("ops.npz", cfs_f=f, cfs_t=t, cfs_out=cam, tm_in=a, tm_out=tm,
         par_img=pimg, par_mask=pmask, par_out=pout, par2_img=pimg2, par2_mask=pmask2, par2_out=pout2,
         sc_gts=np.stack(gts), sc_preds=np.stack(preds), sc_hist=hist,
         sc_miou=sc["miou"], sc_pacc=sc["pAcc"], sc_macc=sc["mAcc"],
         sc_iou=np.array(list(sc["iou"].values())), sc_prec=np.array(list(sc["precision"].values())),
         sc_rec=np.array(list(sc["recall"].values())), sc_conf=np.array(list(sc["confusion"].values())))


# ---------------------------------------------------------------- 3. attr_aggregate with the shipped banks
def gold_attr():
    rs = np.random.RandomState(9)
    out = {}
    for ds, F_, T, K in (("pascal_voc", 20, 45, 112), ("ms_coco", 80, 103, 224)):
        bank, flag = torch.load(os.path.join(REF, f"attributes_text/{ds}_desc_clip_ViT-B-16_gpt4.0_cluster_{K}_embedding_bank.pth"))
        tf = rs.standard_normal((T, 512)).astype(np.float32)
        tf /= np.linalg.norm(tf, axis=1, keepdims=True)
        agg, _ = ref_attr_aggregate(torch.from_numpy(tf), ds, F_, K, "unused.json")
        out[f"{ds}_text"] = tf
        out[f"{ds}_agg"] = agg
        # the shipped bank itself is reference DATA (not source): the product keeps a copy inside the package
        # (excel_amd/attributes_text/), so the GPU box, which has no /root/reference, builds the same text bank.
        np.savez_compressed(os.path.join(HERE, "..", "..", "excel_amd", "attributes_text", f"attr_bank_{ds}.npz"), bank=bank.numpy(), flag=flag.numpy())
    save("attr_aggregate.npz", **out)


# ---------------------------------------------------------------- 4. refine + bkg + PAR + label, and e2e harness trace
def gold_pipeline():
    w = make_vit_weights(TINY, seed=11)
    vit = build_ref_vit(TINY, w, feat_size=6, mode="train")
    rs = np.random.RandomState(33)
    text = rs.standard_normal((9, 64)).astype(np.float32)
    text /= np.linalg.norm(text, axis=1, keepdims=True)
    par = ref_PAR_mod.PAR(num_iter=20, dilations=[1, 2, 4, 8, 12, 24])
    F_ = 4
    out = dict(text=text)
    gts, preds = [], []
    sizes = [(40, 56), (96, 96), (50, 33), (64, 80)]
    cls_sets = [[0, 2], [1], [0, 1, 3], [3, 2]]
    for i, ((H, W), cls) in enumerate(zip(sizes, cls_sets)):
        img = rs.standard_normal((1, 3, H, W)).astype(np.float32)
        gt = rs.randint(0, F_ + 1, (H, W)).astype(np.uint8)
        gt[:2, :] = 255
        cls_label = np.zeros(F_, np.float32)
        cls_label[cls] = 1
        inputs = F.interpolate(torch.from_numpy(img), size=[96, 96], mode="bilinear", align_corners=False)
        _, f, attn, _ = ref_generate_clip_fts(vit, inputs)
        maps = ref_clip_fs(f, torch.from_numpy(text))[:, 1:, :F_]
        refined, cls_lst = ref_aff.refine_cams_with_aff(maps[0], attn[:, 0], torch.from_numpy(cls_label),
                                                        size=inputs.shape[2:], caa_thre=0.79)
        labels, cams = ref_aff.refine_cams_with_bkg_weclip(refined, inputs[0], cls_lst, par, (H, W))
        out.update({f"s{i}_img": img[0], f"s{i}_gt": gt, f"s{i}_cls": cls_label, f"s{i}_inputs": inputs,
                    f"s{i}_maps": maps, f"s{i}_refined": torch.stack(refined, 0), f"s{i}_cls_lst": cls_lst,
                    f"s{i}_cams": cams, f"s{i}_label": labels[0]})
        preds.append(labels[0].numpy().astype(np.int16))
        gts.append(gt.astype(np.int16))
    hist = sum(ref_eval._fast_hist(g.flatten(), p.flatten(), F_ + 1) for g, p in zip(gts, preds))
    with np.errstate(all="ignore"):
        sc = ref_eval.scores(gts, preds, num_classes=F_ + 1)
    save("pipeline_tiny.npz", seed_w=11, hist=hist, miou=sc["miou"], **out)


# ---------------------------------------------------------------- 5. LVC branch: ex_feats in the surgery attention, seg_attn layer selection
def gold_lvc():
    w = make_vit_weights(TINY, seed=11)
    vit = build_ref_vit(TINY, w, feat_size=6, mode="train")
    rs = np.random.RandomState(41)
    imgs = rs.standard_normal((2, 3, 96, 96)).astype(np.float32)
    ex_feats = rs.standard_normal((2, 32, 6, 6)).astype(np.float32)
    text = rs.standard_normal((9, 64)).astype(np.float32)
    text /= np.linalg.norm(text, axis=1, keepdims=True)
    x, attn_list, _ = vit(torch.from_numpy(imgs), True, torch.from_numpy(ex_feats))       # clip_surgery_model.py:419, :127-141
    f = x / x.norm(dim=1, keepdim=True)                                                  # clip.py:353
    cam = ref_clip_fs(f, torch.from_numpy(text))
    attn = torch.stack(attn_list, 0)
    # attn_pred: model/model_excel.py:70-76 cannot be imported (mmcv); the five tensor lines are re-typed HERE (harness
    # only) -- the same similarity is pinned through the reference's own Attention.forward above.
    fl = F.normalize(torch.from_numpy(ex_feats).reshape(2, 32, 36), dim=1)
    ap = fl.transpose(2, 1).bmm(fl)
    ap = torch.sigmoid((ap - torch.mean(ap) * 1.) * 3.0)
    # refine_cams_with_aff with seg_attn (utils/affutils.py:182-195), image 0
    cls_label = np.array([1, 0, 1, 0], np.float32)
    maps = cam[:, 1:, :4]
    refined, cls_lst = ref_aff.refine_cams_with_aff(maps[0], attn[:, 0], torch.from_numpy(cls_label), size=(96, 96),
                                                    seg_attn=ap[0].unsqueeze(0), caa_thre=0.79)
    save("lvc_tiny.npz", seed_w=11, imgs=imgs, ex_feats=ex_feats, text=text, x=x, cam=cam, attn=attn, attn_pred=ap,
         cls=cls_label, refined=torch.stack(refined, 0), cls_lst=cls_lst)


# ---------------------------------------------------------------- 6. decoder head: SegFormerHead fuse + DecoderTransformer + attn_pred
def gold_decoder():
    mm, mc = types.ModuleType("mmcv"), types.ModuleType("mmcv.cnn")       # segformer_head.py:10 imports ConvModule (unused by the head)
    mc.ConvModule = object
    mm.cnn = mc
    sys.modules.setdefault("mmcv", mm)
    sys.modules.setdefault("mmcv.cnn", mc)
    from model.segformer_head import SegFormerHead
    from model.decoder.TransDecoder import DecoderTransformer
    w = make_vit_weights(TINY, seed=11)
    vit = build_ref_vit(TINY, w, feat_size=6, mode="train")
    rs = np.random.RandomState(51)
    imgs = rs.standard_normal((2, 3, 96, 96)).astype(np.float32)
    # generate_clip_fts (clip/clip.py:348-358): the list is stacked AFTER the forward, i.e. with the in-place aliasing of
    # clip_surgery_model.py:317,329,442 baked in -- this is what the decoder really sees
    _, _, _, feats = ref_generate_clip_fts(vit, torch.from_numpy(imgs))
    all_feats = torch.stack(feats, 0)                                      # [8,2,37,128]
    torch.manual_seed(7)
    fuse = SegFormerHead(in_channels=128, embedding_dim=32, num_classes=5, index=8).eval()        # model_excel.py:28-29
    dec = DecoderTransformer(width=32, layers=3, heads=8, output_dim=5).eval()                   # :30
    b, h, wd = 2, 96, 96
    tok = all_feats[:, :, 1:, ...].permute(0, 1, 3, 2).reshape(8, b, 128, h // 16, wd // 16)     # :60-62
    fts = fuse(tok)                                                        # :64
    seg, attn_list = dec(fts)                                              # :68
    fl = F.normalize(fts.reshape(b, 32, -1), dim=1)                        # :70-73 (re-typed: model_excel.py imports mmcv/clip)
    ap = torch.sigmoid((fl.transpose(2, 1).bmm(fl) - torch.mean(fl.transpose(2, 1).bmm(fl)) * 1.) * 3.0)   # :74-76
    sd = {"fuse." + k: v for k, v in fuse.state_dict().items()}
    sd.update({"dec." + k: v for k, v in dec.state_dict().items()})
    save("decoder_tiny.npz", seed_w=11, imgs=imgs, all_feats=all_feats, fts=fts, seg=seg, attn_pred=ap,
         dec_attn_last=attn_list[-1], **sd)


# ---------------------------------------------------------------- 7. CLIP text tower + tokenizer (one-time text bank)
def gold_text():
    torch.manual_seed(13)
    m = csm.ExCEL_CLIP(embed_dim=32, image_resolution=64, vision_layers=1, vision_width=64, vision_patch_size=16, context_length=77,
                       vocab_size=120, transformer_width=64, transformer_heads=2, transformer_layers=3).eval()
    rs = np.random.RandomState(5)
    tok = np.zeros((6, 77), np.int64)
    for b in range(6):
        n = rs.randint(3, 20)
        tok[b, 0] = 118
        tok[b, 1:n] = rs.randint(1, 117, n - 1)
        tok[b, n] = 119                                   # eot = the largest id (clip_surgery_model.py:561)
    out = m.encode_text(torch.from_numpy(tok))
    emb = out / out.norm(dim=-1, keepdim=True)            # clip.py:263-265 on the six rows as one prompt ensemble
    ens = emb.mean(dim=0)
    ens = ens / ens.norm()
    sd = {"w." + k: v for k, v in m.state_dict().items() if not k.startswith("visual.") and k != "logit_scale"}
    # tokenizer: the reference's SimpleTokenizer on its own vocabulary file (ftfy is absent: identity stub, ASCII inputs)
    ft = types.ModuleType("ftfy")
    ft.fix_text = lambda t: t
    sys.modules.setdefault("ftfy", ft)
    st = _load_by_path("ref_simple_tokenizer", os.path.join(REF, "clip/simple_tokenizer.py"))
    tk = st.SimpleTokenizer(os.path.join(REF, "clip/bpe_simple_vocab_16e6.txt.gz"))
    texts = ["a clean origami aeroplane.", "a clean origami potted plant.", "a photo of a tv/monitor, it's nice!", "Ground  KEYBOARD  floor",
             "a clean origami motorbike.", "there is the diningtable in the scene."]
    ids = np.zeros((len(texts), 77), np.int32)
    for i, t in enumerate(texts):
        e = [tk.encoder["<|startoftext|>"]] + tk.encode(t) + [tk.encoder["<|endoftext|>"]]       # clip.py:233-235
        ids[i, :len(e)] = e
    save("text_tiny.npz", tokens=tok, out=out, ensemble=ens, texts=np.array(texts), token_ids=ids, **sd)


# ---------------------------------------------------------------- 8. one training iteration of the decoder
def gold_train():
    """scripts/train_voc.py:186-220 with the reference's own modules, losses, affinity labels and optimizer: losses, the
    gradients autograd gives for every decoder parameter (and w.r.t. seg / attn_pred), and the parameters after two
    PolyWarmupAdamW steps."""
    mm, mc = types.ModuleType("mmcv"), types.ModuleType("mmcv.cnn")
    mc.ConvModule = object
    mm.cnn = mc
    sys.modules.setdefault("mmcv", mm)
    sys.modules.setdefault("mmcv.cnn", mc)
    from model.segformer_head import SegFormerHead
    from model.decoder.TransDecoder import DecoderTransformer
    losses = _load_by_path("ref_losses", os.path.join(REF, "model/losses.py"))
    glb = {"torch": torch, "F": F, "np": np}
    cams_to_affinity_label, get_mask_by_radius = None, None
    # camutils.py imports torchvision via imutils: take the two pure functions from the reference text (the LAST definition of
    # cams_to_affinity_label, :438, is the one Python binds)
    tree = ast.parse(open(os.path.join(REF, "utils/camutils.py")).read())
    body = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in ("cams_to_affinity_label", "get_mask_by_radius")]
    exec(compile(ast.Module(body=body, type_ignores=[]), "camutils", "exec"), glb)
    cams_to_affinity_label, get_mask_by_radius = glb["cams_to_affinity_label"], glb["get_mask_by_radius"]
    if not hasattr(np, "float"):
        np.float = float                                       # utils/optimizer.py:12 predates numpy 1.24
    ropt = _load_by_path("ref_optimizer", os.path.join(REF, "utils/optimizer.py"))

    g = np.load(os.path.join(HERE, "decoder_tiny.npz"))
    all_feats = torch.from_numpy(g["all_feats"])
    torch.manual_seed(7)
    fuse = SegFormerHead(in_channels=128, embedding_dim=32, num_classes=5, index=8)
    dec = DecoderTransformer(width=32, layers=3, heads=8, output_dim=5)
    fuse.eval(); dec.eval()                                    # dropout off: deterministic (the head has Dropout2d(0.1) in train mode)
    rs = np.random.RandomState(77)
    b, h, wd = 2, 96, 96
    pseudo = rs.randint(0, 5, (b, h, wd)).astype(np.int64)
    pseudo[rs.rand(b, h, wd) < 0.1] = 255
    pseudo_t = torch.from_numpy(pseudo)
    params = list(dec.parameters()) + list(fuse.parameters())  # model_excel.py:40-45 (group 3)
    optim = ropt.PolyWarmupAdamW(params=[{"params": [], "lr": 1e-4, "weight_decay": 1e-2}, {"params": [], "lr": 1e-4, "weight_decay": 1e-2},
                                         {"params": [], "lr": 1e-3, "weight_decay": 1e-2}, {"params": params, "lr": 1e-3, "weight_decay": 1e-2}],
                                 lr=1e-4, weight_decay=1e-2, betas=(0.9, 0.999), warmup_iter=50, max_iter=1000, warmup_ratio=1e-6, power=1)
    out = {}
    torch.set_grad_enabled(True)                               # __main__ runs the generators under no_grad
    for it in range(2):
        tok = all_feats[:, :, 1:, ...].permute(0, 1, 3, 2).reshape(8, b, 128, h // 16, wd // 16)
        fts = fuse(tok)
        seg, _ = dec(fts)
        fl = F.normalize(fts.reshape(b, 32, -1), dim=1)
        ap = fl.transpose(2, 1).bmm(fl)
        ap = torch.sigmoid((ap - torch.mean(ap) * 1.) * 3.0)
        seg.retain_grad(); ap.retain_grad(); fts.retain_grad()
        segs = F.interpolate(seg, size=(h, wd), mode="bilinear", align_corners=False)              # train_voc.py:202
        seg_loss = losses.get_seg_loss(segs, pseudo_t, ignore_index=255)                           # :203
        attn_mask = get_mask_by_radius(h=h // 16, w=wd // 16, radius=2)                             # :207-208
        aff_mask = cams_to_affinity_label(pseudo_t, mask=attn_mask)                                 # :210
        diver_loss, pos_c, neg_c = losses.get_aff_loss(ap, aff_mask)                                # :212
        loss = 1.0 * seg_loss + 0.1 * diver_loss                                                   # :215
        optim.zero_grad()
        loss.backward()
        if it == 0:
            out.update(seg=seg.detach().clone(), attn_pred=ap.detach().clone(), seg_loss=seg_loss.detach(), diver_loss=diver_loss.detach(),
                       aff_mask=aff_mask.numpy().astype(np.int16), attn_mask=attn_mask.astype(np.uint8), pos_count=int(pos_c), neg_count=int(neg_c),
                       d_seg=seg.grad.clone(), d_attn_pred=ap.grad.clone(), d_fts=fts.grad.clone())
            out.update({"w0.fuse." + k: v.detach().clone() for k, v in fuse.state_dict().items()})
            out.update({"w0.dec." + k: v.detach().clone() for k, v in dec.state_dict().items()})
            out.update({"g.fuse." + k: v.grad.clone() for k, v in fuse.named_parameters()})
            out.update({"g.dec." + k: v.grad.clone() for k, v in dec.named_parameters()})
        else:
            out.update(seg_loss_it1=seg_loss.detach(), diver_loss_it1=diver_loss.detach())
        optim.step()
        out.update({f"w{it + 1}.fuse." + k: v.detach().clone() for k, v in fuse.state_dict().items()})
        out.update({f"w{it + 1}.dec." + k: v.detach().clone() for k, v in dec.state_dict().items()})
        out[f"lr_it{it}"] = optim.param_groups[3]["lr"]
    torch.set_grad_enabled(False)
    save("train_tiny.npz", pseudo=pseudo.astype(np.uint8), **out)


if __name__ == "__main__":
    which = sys.argv[1:] or ["vit_cam", "ops", "attr", "pipeline", "lvc", "decoder", "text", "train"]       # e.g. `make_goldens.py lvc` mints one file
    with torch.no_grad():
        for name in which:
            globals()["gold_" + name]()
