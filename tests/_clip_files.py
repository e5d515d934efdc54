"""Test fixture: a tiny but COMPLETE CLIP checkpoint on disk (visual tower + text tower, keyed like the published archives'
state_dict) and a tiny BPE merges file in CLIP's format, so the harness can be driven exactly as with ViT-B-16.pt."""
import gzip
import os

import numpy as np
import torch

from oracle.vit import VitConfig, make_vit_weights

TINY512 = VitConfig(width=128, layers=8, heads=2, patch=16, out_dim=512, input_resolution=64, n_surgery=5)
MERGES = ["a n", "t h", "i n", "e r", "o n", "r e", "th e</w>", "c l", "o r", "in g</w>", "a t", "e n", "o u", "a r", "e s</w>", "cl e", "an d</w>"]


def make_text_tower(vocab, width=64, layers=2, embed=512, ctx=77, seed=5):
    rs = np.random.RandomState(seed)
    f = np.float32
    rn = lambda *s, std=1.0: (rs.standard_normal(s) * std).astype(f)
    w = {"token_embedding.weight": rn(vocab, width, std=0.5), "positional_embedding": rn(ctx, width, std=0.1),
         "ln_final.weight": (1 + 0.1 * rs.standard_normal(width)).astype(f), "ln_final.bias": rn(width, std=0.1),
         "text_projection": rn(width, embed, std=width ** -0.5), "logit_scale": np.array(4.6, f)}
    for i in range(layers):
        p = f"transformer.resblocks.{i}."
        for nm in ("ln_1", "ln_2"):
            w[p + nm + ".weight"] = (1 + 0.1 * rs.standard_normal(width)).astype(f)
            w[p + nm + ".bias"] = rn(width, std=0.1)
        w[p + "attn.in_proj_weight"] = rn(3 * width, width, std=2 * width ** -0.5)
        w[p + "attn.in_proj_bias"] = rn(3 * width, std=0.1)
        w[p + "attn.out_proj.weight"] = rn(width, width, std=0.5 * width ** -0.5)
        w[p + "attn.out_proj.bias"] = rn(width, std=0.02)
        w[p + "mlp.c_fc.weight"] = rn(4 * width, width, std=width ** -0.5)
        w[p + "mlp.c_fc.bias"] = rn(4 * width, std=0.1)
        w[p + "mlp.c_proj.weight"] = rn(width, 4 * width, std=0.5 * (4 * width) ** -0.5)
        w[p + "mlp.c_proj.bias"] = rn(width, std=0.02)
    return w


def write_tiny_clip(tmp_dir, seed=11, half=False):
    """-> (checkpoint path, merges path, full state_dict of numpy arrays).  half: store the parameters as fp16 tensors, like the published
    archives do (clip/clip.py:138-154 loads them, clip/build_model.py:72 keeps the model fp32: the weights are then fp16-VALUED)."""
    tmp_dir = str(tmp_dir)
    bpe_path = os.path.join(tmp_dir, "bpe_tiny_vocab.txt.gz")
    with gzip.open(bpe_path, "wb") as f:
        f.write(('"bpe_simple_vocab" - version: tiny\n' + "\n".join(MERGES)).encode("utf-8"))
    vocab = 256 + 256 + len(MERGES) + 2
    vis = make_vit_weights(TINY512, seed=seed)
    full = {"visual." + k: v for k, v in vis.items()}
    full.update(make_text_tower(vocab))
    ckpt = os.path.join(tmp_dir, "ViT-B-16.pt")                  # the published archive's file name (state_dict form, clip/clip.py:147)
    if half:
        full = {k: np.asarray(v, np.float32).astype(np.float16).astype(np.float32) if np.asarray(v).dtype == np.float32 else v for k, v in full.items()}
        torch.save({k: (torch.from_numpy(np.asarray(v)).half() if np.asarray(v).dtype == np.float32 else torch.from_numpy(np.asarray(v))) for k, v in full.items()}, ckpt)
    else:
        torch.save({k: torch.from_numpy(np.asarray(v)) for k, v in full.items()}, ckpt)
    return ckpt, bpe_path, full


class _Node(torch.nn.Module):
    """One level of the parameter tree of a TorchScript archive (the published CLIP files are scripted modules, clip/clip.py:138-141)."""

    def forward(self, x):
        return x


def write_tiny_clip_jit(tmp_dir, seed=11):
    """The same tiny checkpoint as a TorchScript ARCHIVE whose state_dict carries the CLIP keys -> (archive path, full state_dict)."""
    _, _, full = write_tiny_clip(tmp_dir, seed)
    root = _Node()
    for k, v in full.items():
        parts, m = k.split("."), root
        for p in parts[:-1]:
            if p not in m._modules:
                m.add_module(p, _Node())
            m = m._modules[p]
        m.register_parameter(parts[-1], torch.nn.Parameter(torch.from_numpy(np.asarray(v)).clone(), requires_grad=False))
    path = os.path.join(str(tmp_dir), "ViT-B-16.jit.pt")
    torch.jit.script(root).save(path)
    return path, full
