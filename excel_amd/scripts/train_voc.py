"""One training iteration of the decoder: mirror of the loop body of scripts/train_voc.py:172-220.

  frozen CLIP surgery forward + CAMs (the hot path)      -> attr_maps_raw, attention            :186
  decoder head in training mode                          -> segs, fts_diver, attn_pred          :186 (model/model_excel.py:60-76)
  pseudo labels: refine_cams_with_aff (+ seg_attn after `lvc_iter`) + refine_cams_with_bkg_weclip on the DE-normalised image  :188-199
  seg loss on the up-sampled logits + affinity ("diver") loss on attn_pred, w_diver = 0.1                                       :202-215
  backward of the head, gradient all-reduce over RCCL (DistributedDataParallel's mean), PolyWarmupAdamW step                    :217-219

All tensor work runs in libexcel_hip.so; torch carries device memory and the one collective.  Data loading, augmentation,
TensorBoard and checkpoint I/O of the reference script are outside this function.
"""
import torch
import torch.distributed as dist

from .. import clip as xclip
from .. import ops
from ..utils import imutils
from ..utils.affutils import refine_cams_with_aff, refine_cams_with_bkg_weclip
from ..utils.camutils import cure_attr_map


def poly_warmup_lr(base_lr, step, warmup_iter, max_iter, warmup_ratio, power):
    """PolyWarmupAdamW's schedule (utils/optimizer.py:52-64); `step` = optimizer.global_step before the update."""
    if step < warmup_iter:
        return base_lr * (1 - (1 - step / warmup_iter) * (1 - warmup_ratio))
    if step < max_iter:
        return base_lr * (1 - step / max_iter) ** power
    return base_lr * (1 - (max_iter - 1) / max_iter) ** power if max_iter > 0 else base_lr     # the reference stops updating lr past max_iter


def allreduce_mean_(flat, group=None):
    """DistributedDataParallel's gradient averaging as one collective on the flat gradient buffer."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        flat.div_(dist.get_world_size(group))
    return flat


class DecoderTrainer:
    def __init__(self, model, par, lr=1e-4, wt_decay=1e-2, betas=(0.9, 0.999), warmup_iters=50, max_iters=30000, warmup_lr=1e-6, power=1, caa_thre=0.79,
                 w_diver=0.1, radius=8, ignore_index=255, lvc_iter=14000, seg_aff_iter=24000, dropout_p=0.1, seed=0):
        """Defaults = scripts/train_voc.py:36-80.  The head's parameters sit in param group 3 (model_excel.py:40-45): lr x 10."""
        if model._dec is None:
            raise RuntimeError("DecoderTrainer needs a model built with decoder_state_dict= (initial head weights)")
        self.model, self.par = model, par
        self.base_lr = lr * 10                                         # engine/optimizer_engine.py: groups 2, 3 use args.lr * 10
        self.wt_decay, self.betas = wt_decay, betas
        self.warmup_iters, self.max_iters, self.warmup_lr, self.power = warmup_iters, max_iters, warmup_lr, power
        self.w_diver, self.radius, self.ignore_index, self.lvc_iter = w_diver, radius, ignore_index, lvc_iter
        self.caa_thre = caa_thre                                       # 0.79 VOC (train_voc.py:195), 0.88 COCO (train_coco.py:193)
        self.seg_aff_iter, self.dropout_p, self.seed = seg_aff_iter, dropout_p, seed
        self.global_step = 0

    @torch.no_grad()
    def pseudo_labels(self, inputs, cls_labels, attr_maps_raw, attn_weights, attn_pred, n_iter):
        """:188-199 -> aff_pseudos [B,H,W] uint8 (PAR guide = denormalize_img2(inputs), :181)."""
        guide = imutils.denormalize_img2(inputs)
        out = []
        for i, attr_map in enumerate(attr_maps_raw):
            seg_attn = attn_pred[i][None] if n_iter >= self.lvc_iter else None                                        # :194
            refined, cls_lst = refine_cams_with_aff(attr_map, attn_weights[:, i, ...], cls_labels[i], size=inputs.shape[2:],
                                                    seg_attn=seg_attn, caa_thre=self.caa_thre)                     # :195
            lab, _ = refine_cams_with_bkg_weclip(refined, guide[i], cls_lst, self.par, size=inputs.shape[2:])     # :196
            out.append(lab)
        return torch.cat(out, dim=0).to(torch.uint8)                                                               # :198

    def train_step(self, inputs, cls_labels, n_iter=None):
        """inputs [B,3,S,S] normalised, cls_labels [B,F] -> dict(seg_loss, diver_loss, lr)."""
        n_iter = self.global_step if n_iter is None else n_iter
        model, dec = self.model, self.model._dec
        with torch.no_grad():
            image_features, attn_weights, all_feats = xclip.generate_clip_fts(
                inputs, model.encoder, return_weights=True, n_attn_out=6 if n_iter >= self.lvc_iter else 0, want_feats=True,
                feats_as_reference=True)                                                                             # model_excel.py:56
            attr_maps_raw = ops.clip_feature_surgery(image_features, model._text_rows, num_fg=model.num_classes - 1, want_full=False)[1]
            segs, attn_pred, ctx = dec.forward_train(all_feats, dropout_p=self.dropout_p,
                                                     dropout_seed=self.seed * 1000003 + self.global_step)            # :60-76 (Dropout2d active: model.train())
            if n_iter >= self.lvc_iter:
                # fts_diver = attn_fts.clone().detach() of THIS train-mode forward (:186-189): the post-Dropout2d features
                attr_maps_raw = cure_attr_map(model, inputs, ex_feats=dec.train_attn_fts(ctx))
            aff_pseudos = self.pseudo_labels(inputs, cls_labels, attr_maps_raw, attn_weights, attn_pred, n_iter)
            aff_src = None
            if n_iter >= self.seg_aff_iter:                                                                          # :204, :210
                aff_src = ops.argmax_label(ops.bilinear_resize(segs, inputs.shape[2], inputs.shape[3], align_corners=False))
            losses, d_seg, d_ap = ops.train_losses(segs, attn_pred, aff_pseudos, radius=self.radius, ignore_index=self.ignore_index,
                                                   w_seg=1.0, w_diver=self.w_diver, aff_labels_u8=aff_src)           # :202-215
            dec.backward(ctx, d_seg, d_ap)                                                                           # :218
            allreduce_mean_(dec.grad_flat)
            lr = poly_warmup_lr(self.base_lr, self.global_step, self.warmup_iters, self.max_iters, self.warmup_lr, self.power)
            dec.adamw_step(lr, self.global_step + 1, betas=self.betas, eps=1e-8, weight_decay=self.wt_decay)         # :219
            self.global_step += 1
        l = losses.tolist()
        return dict(seg_loss=l[0], diver_loss=l[1], lr=lr, aff_pseudos=aff_pseudos)
