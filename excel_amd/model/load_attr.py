"""attr_aggregate (TSE): fuse class text embeddings with the cached k-means attribute bank.
Mirror of model/load_attr.py:86-119; the bank file is the reference's shipped data
(attributes_text/*_embedding_bank.pth, or the .npz copy of those two tables shipped inside the package:
excel_amd/attributes_text/attr_bank_<dataset>.npz).  Nothing here reads from tests/."""
import os

import numpy as np
import torch

from .. import ops


BANK_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "attributes_text")


def load_bank(dataset_name="pascal_voc", num_atrr_clusters=112, search_dirs=None):
    """-> (bank [512,K] f32 tensor, flag [F,K]).  Looks for the reference's .pth next to the cwd (as the reference
    does, load_attr.py:88) and for the .npz copy shipped inside the package (excel_amd/attributes_text/)."""
    dirs = list(search_dirs or []) + ["./attributes_text", BANK_DIR]
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


def attr_clustering(descriptions, model, num_atrr_clusters=112, tokenizer=None, save_path=None):
    """Mirror of attr_clustering (model/load_attr.py:10-84), the one-time builder of the attribute bank file:
    descriptions {class name: [sentences]} (the GPT-4 JSON the reference ships) -> lower-cased sentences through the text tower
    (`model.encode_text`, HIP), L2-normalised (:32), k-means over all sentences (scikit-learn KMeans(random_state=0), the same
    third-party call as the reference, :39-40), per-class activated-cluster flags (:45-53)
    -> [bank [C,K] float32, flags [num_classes,K] float32] (the pair torch.save'd at :72; written when save_path is given)."""
    import numpy as np
    from sklearn.cluster import KMeans
    from .. import clip as xclip
    embs = []
    for _, desc in descriptions.items():                                                   # :24
        ids = xclip.tokenize([str(d).lower() for d in desc], context_length=model.context_length, tokenizer=tokenizer)   # :26-28
        e = model.encode_text(ids).cpu().numpy().astype(np.float32)                        # :31
        embs.append(e / np.maximum(np.linalg.norm(e, axis=1, keepdims=True), 1e-12))       # :32 F.normalize(p=2, dim=1)
    allv = np.concatenate(embs, 0)                                                         # :34
    kmeans = KMeans(n_clusters=num_atrr_clusters, random_state=0).fit(allv)                # :39-40
    owner = np.concatenate([np.full(len(e), i) for i, e in enumerate(embs)])               # :41-44
    flags = np.zeros((len(embs), num_atrr_clusters), np.float32)
    for c in range(len(embs)):                                                             # :47-53
        flags[c, np.unique(kmeans.labels_[owner == c])] = 1
    bank = [torch.tensor(kmeans.cluster_centers_.transpose(1, 0)), torch.tensor(flags)]    # :66-70
    if save_path:
        torch.save(bank, save_path)                                                        # :72
    return bank
