"""Mirror of the hot functions of the reference's clip/clip.py.

  generate_clip_fts      clip/clip.py:348-358
  clip_feature_surgery   clip/clip.py:288-310
  load                   clip/clip.py:106-154 (weights come from a state_dict; no download, no JIT archive)
"""
import torch

from .. import ops
from .clip_surgery_model import ExCEL_CLIP, VisionTransformer


class LazyAttnWeights:
    """What generate_clip_fts returns as `attn_weights` on the fast path.

    The reference stacks 12 x [B,N,N] matrices (29.6 MB/image) although only
    attn_weights[-6:, 1:, 1:].mean(0) is ever consumed (utils/affutils.py:180,197).  The HIP ViT reduces
    that mean inside the attention kernel; this object carries it (`w_aff`) and still supports the
    reference's `attn_weights[:, i]` indexing.  Per-layer matrices are materialised only when asked for
    (generate_clip_fts(..., n_attn_out=k)).
    """

    def __init__(self, w_aff, stacked=None, aff_layers=6):
        self.w_aff = w_aff            # [B,P,P]
        self.stacked = stacked        # [n,B,N,N] or None
        self.aff_layers = aff_layers

    def __getitem__(self, key):
        if isinstance(key, tuple) and len(key) >= 2 and key[0] == slice(None):
            i = key[1]
            if isinstance(i, int):
                return LazyAttnWeights(self.w_aff[i:i + 1], None if self.stacked is None else self.stacked[:, i:i + 1],
                                       self.aff_layers)
        if self.stacked is not None:
            return self.stacked[key]
        raise IndexError("per-layer attention weights were not materialised; call generate_clip_fts(..., n_attn_out=k)")

    @property
    def shape(self):
        B, P, _ = self.w_aff.shape
        return (self.aff_layers if self.stacked is None else self.stacked.shape[0], B, P + 1, P + 1)


def generate_clip_fts(inputs, model, return_weights=True, ex_feats=None, n_attn_out=0, want_feats=False, aff_layers=6,
                      feats_as_reference=False):
    """-> (image_features [B,N,C] L2-normalised over the TOKEN axis (:353), attn_weights, all_feats).

    attn_weights is a LazyAttnWeights (fast path) unless n_attn_out > 0, in which case it also holds the stacked
    [n_attn_out,B,N,N] tensor of the last layers.  all_feats is [L,B,N,D] when want_feats else None."""
    r = model.encode_image(inputs, return_weights, ex_feats, want_w_aff=True, aff_layers=aff_layers,
                           n_attn_out=n_attn_out, want_feats=want_feats, feats_as_reference=feats_as_reference)
    return r["image_features"], LazyAttnWeights(r["w_aff"], r["attn"], aff_layers), r["feats"]


def clip_feature_surgery(image_features, text_features, redundant_feats=None, t=2):
    """[B,N,C] x [T,C] -> attr maps [B,N,T] (:288-310)."""
    if redundant_feats is not None:
        # :289-290 similarity = image_features @ (text_features - redundant_feats).t().  The reference then returns the unassigned
        # name `attr_maps` (:310, UnboundLocalError), so this branch cannot run there; the evident intent (the CLIP-Surgery function
        # it derives from) is to return the similarity itself, which is what this does: one NT GEMM over all B*N token rows.
        B, N, Cc = image_features.shape
        txt = (torch.as_tensor(text_features).float() - torch.as_tensor(redundant_feats).float()).to(image_features.device)
        return ops.gemm(ops.f32c(image_features).reshape(B * N, Cc), ops.f32c(txt.reshape(-1, Cc))).reshape(B, N, -1)
    full, _ = ops.clip_feature_surgery(image_features, text_features, t=float(t))
    return full


def infer_vit_config(sd):
    """Architecture from a visual-tower state_dict (keys without the "visual." prefix), as clip/build_model.py:33-38 reads it."""
    width = int(sd["conv1.weight"].shape[0])
    layers = len([k for k in sd if k.endswith(".attn.in_proj_weight") or k.endswith(".attn.qkv.weight")])
    patch = int(sd["conv1.weight"].shape[-1])
    grid = round((int(sd["positional_embedding"].shape[0]) - 1) ** 0.5)
    return dict(width=width, layers=layers, heads=width // 64, patch=patch, output_dim=int(sd["proj"].shape[1]), input_resolution=patch * grid)


# checkpoint file names of the published CLIP archives (the table of clip/clip.py:32-52 maps a model name to a download URL whose
# last path component is this file; the reference's downloader stores it under ~/.cache/clip, :120)
_MODEL_FILES = {"ViT-B/16": "ViT-B-16.pt", "CS-ViT-B/16": "ViT-B-16.pt", "ExCEL_ViT-B/16": "ViT-B-16.pt",
                "ViT-B/32": "ViT-B-32.pt", "CS-ViT-B/32": "ViT-B-32.pt", "ViT-L/14": "ViT-L-14.pt", "CS-ViT-L/14": "ViT-L-14.pt",
                "ViT-L/14@336px": "ViT-L-14-336px.pt", "CS-ViT-L/14@336px": "ViT-L-14-336px.pt"}


def available_models():
    return list(_MODEL_FILES)


def find_checkpoint(name, download_root=None):
    """Path of the CLIP checkpoint `name` denotes, or None: a file path as is (clip/clip.py:122-123); a model name is looked up where
    the reference's downloader would have put it (`download_root`, $EXCEL_CLIP_ROOT, ~/.cache/clip; :120).  Nothing is downloaded."""
    import os
    if not isinstance(name, str):
        return None
    if os.path.isfile(name):
        return name
    if name in _MODEL_FILES:
        for root in (download_root, os.environ.get("EXCEL_CLIP_ROOT"), os.path.expanduser("~/.cache/clip")):
            if root and os.path.isfile(os.path.join(root, _MODEL_FILES[name])):
                return os.path.join(root, _MODEL_FILES[name])
    return None


def read_checkpoint(path):
    """JIT archive (:138-141, the published files) or plain state_dict (:147) -> state_dict on the CPU."""
    try:
        return torch.jit.load(path, map_location="cpu").state_dict()
    except RuntimeError:
        return torch.load(path, map_location="cpu")


def load(name, device="cuda", state_dict=None, width=None, layers=None, heads=None, patch=None, output_dim=None,
         input_resolution=None, gemm_mode=None, download_root=None):
    """clip.load("ExCEL_ViT-B/16") counterpart (clip/clip.py:104-154).  `state_dict`: the CLIP checkpoint's state_dict
    ("visual.conv1.weight" ... or a bare visual-tower dict); otherwise `name` is a local checkpoint path or a model name resolved
    by find_checkpoint (there is no network: nothing is downloaded).  The text tower is kept when its keys are
    present; the architecture is read off the tensors like clip/build_model.py:30-50 unless given.  Returns (model, None)."""
    if state_dict is None:
        path = find_checkpoint(name, download_root)
        if path is None:
            raise RuntimeError(f"Model {name} not found: excel_amd.clip.load needs state_dict=, a local checkpoint path, or the published "
                               f"archive under download_root / $EXCEL_CLIP_ROOT / ~/.cache/clip (no network: nothing is downloaded); "
                               f"known names = {available_models()}")
        state_dict = read_checkpoint(path)
    sd, text_sd = {}, {}
    has_prefix = any(k.startswith("visual.") for k in state_dict)
    for k, v in state_dict.items():
        if k.startswith("visual."):
            sd[k[len("visual."):]] = v
        elif has_prefix and (k.startswith(("transformer.", "token_embedding.", "ln_final.")) or k in ("positional_embedding", "text_projection")):
            text_sd[k] = v
        elif not has_prefix:
            sd[k] = v
    cfg = infer_vit_config(sd)                                    # build_model.py:33-38; explicit arguments override
    for k, v in dict(width=width, layers=layers, heads=heads, patch=patch, output_dim=output_dim, input_resolution=input_resolution).items():
        if v is not None:
            cfg[k] = v
    vis = VisionTransformer(cfg["input_resolution"], cfg["patch"], cfg["width"], cfg["layers"], cfg["heads"], cfg["output_dim"], state_dict=sd,
                            device=device, gemm_mode=gemm_mode)
    return ExCEL_CLIP(vis, text_sd if "text_projection" in text_sd else None), None


def tokenize(texts, context_length=77, truncate=False, tokenizer=None, vocab_path=None):
    """clip/clip.py:206-248 -> int32 tensor [n, context_length].  Needs CLIP's merges file (excel_amd/clip/bpe.py)."""
    from . import bpe
    global _tokenizer
    tk = tokenizer or _tokenizer
    if tk is None:
        tk = _tokenizer = bpe.BPETokenizer(vocab_path)
    return torch.from_numpy(bpe.tokenize(texts, tk, context_length, truncate))


_tokenizer = None



def default_prompt_templates():
    """The default list of clip/clip.py:256 (85 strings; data file next to this module)."""
    import json
    import os
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "prompt_templates.json")) as f:
        return json.load(f)["templates"]


@torch.no_grad()
def encode_text_with_prompt_ensemble(model, texts, device=None, prompt_templates=None, tokenizer=None):
    """clip/clip.py:252-269: per class: tokenize the prompted strings, encode_text, normalise, mean, normalise -> [n_classes, C]."""
    templates = list(prompt_templates) if prompt_templates is not None else default_prompt_templates()
    feats = []
    for t in texts:
        prompted = tokenize([tpl.format(t) for tpl in templates], context_length=model.context_length, tokenizer=tokenizer)   # :261-262
        feats.append(ops.prompt_ensemble(model.encode_text(prompted)))                                                        # :263-266
    return torch.stack(feats, dim=0)                                                                                         # :268
