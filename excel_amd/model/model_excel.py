"""ExCEL_model: the drop-in boundary of the hot path (mirror of model/model_excel.py:16-78).

Constructor signature and forward contract follow the reference.  The decoder head (SegFormerHead + DecoderTransformer,
:28-30, :60-76) is inference-only here: give its trained weights (`decoder_state_dict`, keys "decoder_fts_fuse.*" /
"decoder.*" as in the reference model's state_dict) and forward returns `seg`, `attn_fts` and `attn_pred`; without
them (training-free mode ignores all three, tools/infer_lam.py:79,92) they are None.
"""
import torch

from .. import clip, ops
from .load_attr import attr_aggregate


class ExCEL_model:
    def __init__(self, clip_model=None, embedding_dim=256, in_channels=512, dataset_name="pascal_voc",
                 num_classes=21, num_atrr_clusters=112, json_file=None, img_size=320, mode="train", device="cuda",
                 state_dict=None, text_features=None, attr_bank=None, vit_cfg=None, text_attr=None, gemm_mode=None,
                 decoder_state_dict=None, decoder_heads=8, class_names=None, tokenizer=None):
        """Extra keyword arguments (no network here): `state_dict` = CLIP visual weights, `text_features` [T,512] =
        output of encode_text_with_prompt_ensemble (clip/clip.py:252-269, one-time, out of scope),
        `attr_bank` [512,K] overrides the bank file, `vit_cfg` overrides the ViT-B/16 shape, `gemm_mode` = "bf16x3"
        (default) | "f32" selects the matrix-core numerics."""
        self.num_classes = num_classes
        self.embedding_dim = embedding_dim
        self.in_channels = in_channels
        self.device = device
        # the tower's shape is read off the checkpoint (clip/build_model.py:33-38); `vit_cfg` overrides individual fields
        self.encoder, _ = clip.load(clip_model, device=device, state_dict=state_dict, gemm_mode=gemm_mode, **(vit_cfg or {}))   # :25
        self.encoder.visual.reload_self_attn(layers=6, feat_size=img_size // self.encoder.visual.patch_size, mode=mode)   # :26
        if text_attr is not None:          # a pre-aggregated [C,T] bank (tests / cached banks)
            self.integral_text_features, self.attr_flag = None, None
            self.text_attr = torch.as_tensor(text_attr).float().to(device)
        else:
            if text_features is None and class_names is not None:
                # :31-33: text_prompts = class names + background categories (clip/clip_text.py lists, passed by the caller),
                # prompt template 'a clean origami {}.' -- runs on the library's text tower (needs the full CLIP state_dict
                # and CLIP's BPE merges file, see excel_amd/clip/bpe.py)
                text_features = clip.encode_text_with_prompt_ensemble(self.encoder, list(class_names), device,
                                                                      prompt_templates=["a clean origami {}."], tokenizer=tokenizer)
            if text_features is None:
                # the reference's own default (:31-33): the dataset's class + background prompt lists
                from ..datasets.clip_text import text_prompts
                text_features = clip.encode_text_with_prompt_ensemble(self.encoder, text_prompts(num_classes), device,
                                                                      prompt_templates=["a clean origami {}."], tokenizer=tokenizer)
            self.integral_text_features = torch.as_tensor(text_features).float().to(device)
            self.text_attr, self.attr_flag = attr_aggregate(self.integral_text_features, dataset_name, num_classes - 1,
                                                            num_atrr_clusters, json_file, bank=attr_bank, device=device)   # :34
        self._text_rows = self.text_attr.permute(1, 0).contiguous()     # [T,C] view the CAM kernel consumes (:58)
        self.decoder_fts_fuse = self.decoder = self._dec = None
        if decoder_state_dict is not None:
            from .decoder.TransDecoder import DecoderTransformer
            from .segformer_head import SegFormerHead
            fuse_sd = {k[len("decoder_fts_fuse."):]: v for k, v in decoder_state_dict.items() if k.startswith("decoder_fts_fuse.")}
            dec_sd = {k[len("decoder."):]: v for k, v in decoder_state_dict.items() if k.startswith("decoder.")}
            n_fuse = sum(1 for k in fuse_sd if k.endswith(".proj.weight"))
            n_dec = sum(1 for k in dec_sd if k.endswith(".ln_1.weight"))
            self.decoder_fts_fuse = SegFormerHead(in_channels=in_channels, embedding_dim=embedding_dim, num_classes=num_classes,
                                                  index=n_fuse).load_state_dict(fuse_sd)                               # :28-29
            self.decoder = DecoderTransformer(width=embedding_dim, layers=n_dec, heads=decoder_heads,
                                              output_dim=num_classes).load_state_dict(dec_sd)                      # :30
            self._dec = ops.DecoderHandle(fuse_sd, dec_sd, heads=decoder_heads, device=device)

    def eval(self):
        return self

    def train(self):
        return self

    def state_dict(self):
        """The trainable part, keyed like the reference model's state_dict ("decoder_fts_fuse.*", "decoder.*"): what the
        training script checkpoints (scripts/train_voc.py torch.save(model.state_dict(), ...))."""
        if self._dec is None:
            return {}
        fuse, dec = self._dec.state_dicts()
        out = {"decoder_fts_fuse." + k: v for k, v in fuse.items()}
        out.update({"decoder." + k: v for k, v in dec.items()})
        return out

    def to(self, *a, **k):
        return self

    # The learned decoder (decoder_fts_fuse + SegFormerHead, :60-68) runs inside the library when `decoder_state_dict` is given
    # (excel_decoder_forward).  A caller that owns a different head can still plug it in here:
    # feature_head(all_feats [L,B,N,D]) -> attn_fts [B,C,g,g]; forward then also returns attn_fts and attn_pred like the reference.
    feature_head = None

    @staticmethod
    def attn_pred_from(attn_fts):
        """:70-76: sigmoid((f^T f - mean) * 3) of the channel-normalised decoder features -> [B,P,P]."""
        return ops.feature_affinity(attn_fts, "sigmoid", beta=1.0, gamma=3.0)

    def forward(self, img, ex_feats=None, n_attn_out=0, want_feats=False):
        """model(img) -> (seg, attn_fts, attr_maps_raw [B,P,F], attn_weights, attn_pred)      (:48-78)
        model(img, ex_feats=[B,C,g,g]) -> attr_maps_raw of the LVC branch                       (:50-53)"""
        if ex_feats is not None:
            r = self.encoder.encode_image(img, True, ex_feats, want_w_aff=False, want_raw=True, want_features=False)          # :51
            return ops.patch_text_cam(r["x_raw"], self._text_rows, num_fg=self.num_classes - 1,
                                      mode=self.encoder.visual.handle().gemm_mode())[1]                                  # :52
        want_feats = want_feats or self.feature_head is not None or self._dec is not None
        # the decoder consumes the list exactly as the reference stacks it (in-place aliasing quirk, include/excel_hip.h)
        # :57-58 back to back -> the fused kernel: the tower hands over the un-normalised token features, the token-axis norm,
        # the similarity GEMM and the surgery epilogue are one C-ABI call (excel_patch_text_cam)
        r = self.encoder.encode_image(img, True, None, want_w_aff=True, aff_layers=6, n_attn_out=n_attn_out, want_feats=want_feats,
                                      feats_as_reference=self._dec is not None, want_raw=True, want_features=False)
        attn_weights, all_feats = clip.clip.LazyAttnWeights(r["w_aff"], r["attn"], 6), r["feats"]
        _, attr_maps_raw, _ = ops.patch_text_cam(r["x_raw"], self._text_rows, num_fg=self.num_classes - 1,
                                                 mode=self.encoder.visual.handle().gemm_mode())
        self.last_all_feats = all_feats
        seg = attn_fts = attn_pred = None
        if self._dec is not None:
            attn_fts, seg = self._dec.forward(all_feats)                                                                # :60-68
            attn_pred = self.attn_pred_from(attn_fts)                                                                   # :70-76
        elif self.feature_head is not None:
            attn_fts = self.feature_head(all_feats)                                                                     # :60-66
            attn_pred = self.attn_pred_from(attn_fts)                                                                   # :70-76
        return seg, attn_fts, attr_maps_raw, attn_weights, attn_pred

    __call__ = forward

    def check_numerics(self, img, tol=5e-4, fallback=True, reduce=None):
        """Guard of the fast matrix-core modes on the caller's OWN weights and images.  "bf16x3" carries 16 mantissa bits per operand:
        measured against a float64 run of the oracle its CAM error is ~12x that of fp32 arithmetic - 1e-5 on well-conditioned networks
        (gate 1e-3), but on an ill-conditioned one (massive-activation channels + near-one-hot attention rows:
        tests/test_gpu_ops.py::test_vit_b16_448_clip_like_outlier_net) fp32 itself sits at 2e-4 and bf16x3 exceeds the gate; "f16x3"
        (IEEE-half planes, 22 bits) stays within 1.4x of fp32 there at ~2.5 % less throughput.  This runs ViT + CAM of `img`
        [b,3,S,S] in the current mode and in exact fp32 ("f32") and compares the attr maps - the quantity the gate is stated on.
        With `fallback`, a difference above `tol` (half the gate by default) moves this model one step down the ladder
        bf16x3 -> f16x3 -> f32 (re-checked at every step) for everything that follows.  On fp16-VALUED weights (every published CLIP
        archive) the f16 rung is "f16x2": the bits of f16x3 with two MFMAs per product in the nn.Linear GEMMs - it is the mode such a
        model starts in (ops.VitHandle: "auto"), so its ladder is f16x2 -> f32.
        `reduce(diff) -> diff` shares a rung's verdict between the ranks of a job (infer_lam: all-reduce MAX, a NaN counts as +inf);
        it is called exactly once per rung on EVERY rank, also on a rank whose shard is empty (`img=None`: contributes 0), so the
        collective sequence is identical everywhere.  This is the one implementation of the ladder.
        -> {"max_abs_diff" (of the mode it started in), "tol", "mode_before", "mode_after", "ladder": [(mode, diff), ...]}"""
        h = self.encoder.visual.handle()
        before = h.gemm_mode()
        if before == "f32":
            return {"max_abs_diff": 0.0, "tol": tol, "mode_before": before, "mode_after": before, "ladder": []}
        exact = None
        if img is not None:
            h.set_gemm_mode("f32")
            try:
                exact = self.forward(img)[2].clone()
            finally:
                h.set_gemm_mode(before)
        ladder, after = [], before
        f16_rung = "f16x2" if getattr(h, "weights_fp16_exact", lambda: False)() else "f16x3"
        for mode in (["bf16x3", f16_rung] if before == "bf16x3" else [before]):
            diff = 0.0
            if img is not None:
                h.set_gemm_mode(mode)
                diff = float((self.forward(img)[2] - exact).abs().max())
            if reduce is not None:
                diff = float(reduce(diff if diff == diff else float("inf")))
            ladder.append((mode, diff))
            after = mode
            if diff <= tol or not fallback:              # (NaN compares false: moves on)
                break
        else:
            after = "f32"
        h.set_gemm_mode(after if fallback else before)
        return {"max_abs_diff": ladder[0][1], "tol": tol, "mode_before": before, "mode_after": after if fallback else before, "ladder": ladder}
