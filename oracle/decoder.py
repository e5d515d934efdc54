"""Decoder head of the reference restated in numpy fp32 (oracle; test infrastructure only).

  SegFormerHead.forward        model/segformer_head.py:47-77   (MLP :12-27: Linear, ReLU, Linear per ViT layer; 1x1 fuse)
  DecoderTransformer.forward   model/decoder/TransDecoder.py:105-124 (ResidualAttentionBlock :62-84: pre-LN MHA + QuickGELU MLP)
  ExCEL_model.forward          model/model_excel.py:60-68 (token slicing / reshape feeding the head)

Weights are dicts keyed like the reference modules' state_dict, prefixed "fuse." / "dec.".
"""
import numpy as np

from .vit import layer_norm, quick_gelu, softmax


def segformer_fuse(all_feats, w, prefix="fuse."):
    """all_feats [L,B,N,D] (what generate_clip_fts stacks) -> fts [B,E,g,g]   (model_excel.py:60-64)."""
    L, B, N, D = all_feats.shape
    g = int(round((N - 1) ** 0.5))
    tok = np.asarray(all_feats, np.float32)[:, :, 1:, :]                      # :60  [L,B,P,D]
    outs = []
    for l in range(L):                                                          # segformer_head.py:68-72
        p = f"{prefix}linears_modulelist.{l}."
        h = tok[l] @ w[p + "proj.weight"].T + w[p + "proj.bias"]               # :23
        h = np.maximum(h, 0)                                                    # :24
        h = h @ w[p + "proj_2.weight"].T + w[p + "proj_2.bias"]               # :25
        outs.append(h.astype(np.float32))
    cat = np.concatenate(outs, -1)                                              # :73 (channel concat)
    Wf = w[prefix + "linear_fuse.weight"].reshape(w[prefix + "linear_fuse.weight"].shape[0], -1)
    fts = cat @ Wf.T + w[prefix + "linear_fuse.bias"]                          # :74 1x1 conv; dropout inactive in eval
    return fts.astype(np.float32).transpose(0, 2, 1).reshape(B, -1, g, g)


def decoder_transformer(fts, w, heads, prefix="dec."):
    """fts [B,E,g,g] -> (seg logits [B,nc,g,g], list of head-averaged attention [B,P,P])   (TransDecoder.py:113-124)."""
    B, E, g, _ = fts.shape
    x = np.asarray(fts, np.float32).reshape(B, E, g * g).transpose(0, 2, 1)   # [B,P,E]  (:115-116, batch-major here)
    hd = E // heads
    scale = np.float32(hd ** -0.5)
    attns = []
    layer = 0
    while f"{prefix}transformer.resblocks.{layer}.ln_1.weight" in w:
        p = f"{prefix}transformer.resblocks.{layer}."
        y = layer_norm(x, w[p + "ln_1.weight"], w[p + "ln_1.bias"])            # :79
        qkv = y @ w[p + "attn.in_proj_weight"].T + w[p + "attn.in_proj_bias"]
        q, k, v = (qkv[..., i * E:(i + 1) * E].reshape(B, -1, heads, hd).transpose(0, 2, 1, 3) for i in range(3))
        a = softmax((q * scale) @ k.transpose(0, 1, 3, 2))                      # [B,h,P,P]
        o = (a @ v).transpose(0, 2, 1, 3).reshape(B, -1, E)
        o = o @ w[p + "attn.out_proj.weight"].T + w[p + "attn.out_proj.bias"]
        x = (x + o).astype(np.float32)                                          # :80
        m = quick_gelu(layer_norm(x, w[p + "ln_2.weight"], w[p + "ln_2.bias"]) @ w[p + "mlp.c_fc.weight"].T + w[p + "mlp.c_fc.bias"])
        x = (x + m @ w[p + "mlp.c_proj.weight"].T + w[p + "mlp.c_proj.bias"]).astype(np.float32)   # :81
        attns.append(a.mean(1, dtype=np.float32))
        layer += 1
    Wp = w[prefix + "linear_pred.weight"].reshape(w[prefix + "linear_pred.weight"].shape[0], -1)
    seg = x @ Wp.T + w[prefix + "linear_pred.bias"]                            # :122 1x1 conv
    return seg.astype(np.float32).transpose(0, 2, 1).reshape(B, -1, g, g), attns
