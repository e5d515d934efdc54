"""CLIP text tower restated in numpy fp32 (oracle; test infrastructure only).

  ExCEL_CLIP.encode_text               clip/clip_surgery_model.py:551-564
  build_attention_mask (causal)        :537-543
  ResidualAttentionBlock (MHA branch)  :299-337 (single path; the text blocks are never "surgery" blocks)
  encode_text_with_prompt_ensemble     clip/clip.py:252-269 (the reduction :262-266)
"""
import numpy as np

from .vit import layer_norm, quick_gelu, softmax


def encode_text(tokens, w, heads):
    """tokens [B,ctx] int, w: state_dict of the text side -> [B, embed_dim]."""
    tokens = np.asarray(tokens)
    B, ctx = tokens.shape
    x = w["token_embedding.weight"][tokens] + w["positional_embedding"][None]          # :552-554
    x = x.astype(np.float32)
    E = x.shape[-1]
    hd = E // heads
    scale = np.float32(hd ** -0.5)
    mask = np.triu(np.full((ctx, ctx), -np.inf, np.float32), 1)                         # :540-542
    layer = 0
    while f"transformer.resblocks.{layer}.ln_1.weight" in w:
        p = f"transformer.resblocks.{layer}."
        y = layer_norm(x, w[p + "ln_1.weight"], w[p + "ln_1.bias"])
        qkv = y @ w[p + "attn.in_proj_weight"].T + w[p + "attn.in_proj_bias"]
        q, k, v = (qkv[..., i * E:(i + 1) * E].reshape(B, ctx, heads, hd).transpose(0, 2, 1, 3) for i in range(3))
        a = softmax((q * scale) @ k.transpose(0, 1, 3, 2) + mask)
        o = (a @ v).transpose(0, 2, 1, 3).reshape(B, ctx, E)
        x = (x + o @ w[p + "attn.out_proj.weight"].T + w[p + "attn.out_proj.bias"]).astype(np.float32)
        m = quick_gelu(layer_norm(x, w[p + "ln_2.weight"], w[p + "ln_2.bias"]) @ w[p + "mlp.c_fc.weight"].T + w[p + "mlp.c_fc.bias"])
        x = (x + m @ w[p + "mlp.c_proj.weight"].T + w[p + "mlp.c_proj.bias"]).astype(np.float32)
        layer += 1
    x = layer_norm(x, w["ln_final.weight"], w["ln_final.bias"])                         # :558
    eot = tokens.argmax(-1)                                                              # :562
    return (x[np.arange(B), eot] @ w["text_projection"]).astype(np.float32)


def prompt_ensemble(class_embeddings):
    """clip.py:262-266: [n,E] -> [E]."""
    e = np.asarray(class_embeddings, np.float32)
    e = e / np.linalg.norm(e, axis=-1, keepdims=True)
    m = e.mean(0, dtype=np.float32)
    return (m / np.linalg.norm(m)).astype(np.float32)
