"""Host-side mirror of the reference's surgery ViT (clip/clip_surgery_model.py:374-448, :479-564).

The module owns nothing but weights and a handle; the forward pass is ONE call into
libexcel_hip (excel_vit_forward).  Names and argument meaning follow the reference so that
model_excel / clip.generate_clip_fts read like the original.
"""
import torch

from .. import ops


class VisionTransformer:
    """VisionTransformer(input_resolution, patch_size, width, layers, heads, output_dim)  (:375)"""

    def __init__(self, input_resolution, patch_size, width, layers, heads, output_dim, state_dict=None, device="cuda",
                 gemm_mode=None):
        self.input_resolution = input_resolution
        self.patch_size = patch_size
        self.embed_dim = width
        self.layers = layers
        self.num_heads = heads
        self.output_dim = output_dim
        self.device = device
        self.n_surgery = 0
        self.gemm_mode = gemm_mode
        self._sd = dict(state_dict) if state_dict is not None else None
        self._handle = None
        self.attn = None

    def load_state_dict(self, sd):
        self._sd = dict(sd)
        self._handle = None

    def state_dict(self):
        return self._sd

    @torch.no_grad()
    def reload_self_attn(self, layers=6, feat_size=20, mode="train"):
        """:396-416.  range(1, layers) -> the last layers-1 blocks use q-q/k-k/v-v attention (weights re-used
        verbatim); a mode containing 'train' permanently resizes the positional grid to feat_size."""
        if self.attn is None:
            self.n_surgery = max(layers - 1, 0)
            self.attn = "surgery"
        if "train" in mode:
            pos = torch.as_tensor(self._sd["positional_embedding"]).float()
            side = int((pos.shape[0] - 1) ** 0.5)
            if side != feat_size:
                self._sd["positional_embedding"] = ops.pos_embed_resize(pos.to(self.device), feat_size)   # :407-414
        self._handle = None

    def handle(self):
        if self._handle is None:
            if self._sd is None:
                raise RuntimeError("VisionTransformer has no weights (load_state_dict first)")
            self._handle = ops.VitHandle(self._sd, self.embed_dim, self.layers, self.num_heads, self.patch_size,
                                         self.output_dim, n_surgery=self.n_surgery, device=self.device, gemm_mode=self.gemm_mode)
        return self._handle

    @torch.no_grad()
    def forward(self, x, return_weights=False, ex_feats=None, **kw):
        """:419-448.  Returns the handle's dict (image_features [B,N,C], w_aff, attn, feats, x_raw - ops.VitHandle.forward);
        clip.generate_clip_fts turns it into the reference's (image_features, attn_weights, all_feats) triple."""
        if ex_feats is not None:
            # Attention.forward :127-137: the cue is the same for every head and every surgery block -> built once
            kw["ex_attn"] = ops.feature_affinity(ex_feats, "mask_softmax", beta=1.0, gamma=3.0)
        return self.handle().forward(x, **kw)

    __call__ = forward

    def cuda(self, *a, **k):
        return self

    def eval(self):
        return self


class ExCEL_CLIP:
    """The visual tower lives on the hot path (encode_image, :548-549); the text tower (encode_text, :551-564) is the one-time
    text-bank step and exists when the checkpoint's text-side weights were given."""

    def __init__(self, visual, text_state_dict=None, text_heads=None):
        self.visual = visual
        self._text = None
        self.context_length = 77
        if text_state_dict is not None:
            width = int(text_state_dict["ln_final.weight"].shape[0])
            self._text = ops.TextHandle(text_state_dict, heads=text_heads or max(width // 64, 1), device=visual.device)   # build_model: heads = width // 64
            self.context_length = self._text.cfg["context_length"]

    @torch.no_grad()
    def encode_text(self, text):
        """text: token ids [B, context_length] -> [B, embed_dim]   (:551-564)"""
        if self._text is None:
            raise RuntimeError("encode_text needs the text tower: clip.load(..., state_dict=<full CLIP state_dict>) keeps it when the "
                               "text-side keys (token_embedding.weight, transformer.*, ln_final.*, text_projection) are present")
        return self._text.encode(text)

    def encode_image(self, image, return_weights=True, ex_feats=None, **kw):
        return self.visual(image, return_weights, ex_feats, **kw)

    def cuda(self, *a, **k):
        return self

    def eval(self):
        return self
