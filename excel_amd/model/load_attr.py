"""attr_aggregate (TSE): fuse class text embeddings with the cached k-means attribute bank.
Mirror of model/load_attr.py:86-119; the bank file is the reference's shipped data
(attributes_text/*_embedding_bank.pth, or the .npz copy under tests/golden/)."""
import os

import numpy as np
import torch

from .. import ops


def load_bank(dataset_name="pascal_voc", num_atrr_clusters=112, search_dirs=None):
    """-> (bank [512,K] f32 tensor, flag [F,K]).  Looks for the reference's .pth next to the cwd (as the reference
    does, load_attr.py:88) and for the .npz fixture shipped with this repo."""
    here = os.path.dirname(os.path.abspath(__file__))
    root = os.path.dirname(os.path.dirname(here))
    dirs = list(search_dirs or []) + ["./attributes_text", os.path.join(root, "tests", "golden")]
    for d in dirs:
        p = os.path.join(d, f"{dataset_name}_desc_clip_ViT-B-16_gpt4.0_cluster_{num_atrr_clusters}_embedding_bank.pth")
        if os.path.exists(p):
            bank, flag = torch.load(p, map_location="cpu")
            return bank.float(), flag.float()
        p = os.path.join(d, f"attr_bank_{dataset_name}.npz")
        if os.path.exists(p):
            z = np.load(p)
            if z["bank"].shape[1] == num_atrr_clusters:
                return torch.from_numpy(z["bank"]), torch.from_numpy(z["flag"])
    raise FileNotFoundError(f"attribute bank for {dataset_name}/{num_atrr_clusters} not found in {dirs} "
                            "(attr_clustering, load_attr.py:10-84, is an offline step and out of scope)")


def attr_aggregate(text_features, dataset_name="pascal_voc", num_classes=20, num_atrr_clusters=112,
                   json_file=None, topK=0.9, bank=None, device="cuda"):
    """-> (text_attr [C,T] with unit-norm columns, attr_flag).  Same signature as the reference."""
    flag = None
    if bank is None:
        bank, flag = load_bank(dataset_name, num_atrr_clusters)
    out = ops.attr_aggregate(torch.as_tensor(text_features).float().to(device), torch.as_tensor(bank).float().to(device),
                             num_classes, topK)
    return out, flag
