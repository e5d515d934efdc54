"""ctypes binding of libexcel_hip.so (the C ABI declared in include/excel_hip.h).

There is deliberately NO fallback: if the shared library is missing or a symbol
cannot be resolved the import of any compute entry point raises.  The product
path never routes through PyTorch eager ops or the CPU oracle.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libexcel_hip.so")

c_f = C.c_void_p      # device pointers travel as void*
c_i = C.c_int
c_ll = C.c_longlong
c_sz = C.c_size_t


class VitConfig(C.Structure):
    _fields_ = [("width", c_i), ("layers", c_i), ("heads", c_i), ("patch", c_i), ("out_dim", c_i),
                ("n_surgery", c_i), ("pos_grid", c_i)]


class VitBlockWeights(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in (
        "ln1_w", "ln1_b", "in_proj_w", "in_proj_b", "out_proj_w", "out_proj_b",
        "ln2_w", "ln2_b", "fc1_w", "fc1_b", "fc2_w", "fc2_b")]


class VitWeights(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in (
        "conv1_w", "class_emb", "pos_emb", "ln_pre_w", "ln_pre_b", "ln_post_w", "ln_post_b", "proj")] + \
        [("blocks", C.POINTER(VitBlockWeights))]


class DecoderConfig(C.Structure):
    _fields_ = [(n, c_i) for n in ("vit_layers", "vit_width", "embed", "dec_layers", "heads", "num_classes")]


class FuseLayerWeights(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("proj_w", "proj_b", "proj2_w", "proj2_b")]


class DecoderBlockWeights(C.Structure):
    _fields_ = VitBlockWeights._fields_


class DecoderWeights(C.Structure):
    _fields_ = [("fuse", C.POINTER(FuseLayerWeights)), ("fuse_w", C.c_void_p), ("fuse_b", C.c_void_p),
                ("blocks", C.POINTER(DecoderBlockWeights)), ("pred_w", C.c_void_p), ("pred_b", C.c_void_p)]


class TextConfig(C.Structure):
    _fields_ = [(n, c_i) for n in ("vocab_size", "context_length", "width", "layers", "heads", "embed_dim")]


class TextWeights(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("token_embedding", "positional_embedding", "ln_final_w", "ln_final_b", "text_projection")] + \
        [("blocks", C.POINTER(DecoderBlockWeights))]


class RaggedInfo(C.Structure):
    _fields_ = [("B", C.c_int32), ("total_tiles", C.c_int32), ("total_pix", C.c_int64), ("total_label_pix", C.c_int64),
                ("max_plane_pix", C.c_int64), ("table_ints", C.c_int64)]


# name -> (restype, argtypes); mirrors include/excel_hip.h one to one
SIGNATURES = {
    "excel_last_error": (C.c_char_p, []),
    "excel_abi_version": (c_i, []),
    "excel_build_id": (C.c_char_p, []),
    "excel_gemm_f32": (c_i, [c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i,
                             c_ll, c_ll, c_ll, c_ll, c_f]),
    "excel_split_bf16": (c_i, [c_f, c_f, c_ll, c_i, c_f]),
    "excel_gemm_bf16x3": (c_i, [c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_f]),
    "excel_split_f16": (c_i, [c_f, c_f, c_ll, c_i, c_f]),
    "excel_gemm_f16x3": (c_i, [c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_f]),
    "excel_pack_f16": (c_i, [c_f, c_f, c_ll, c_i, c_f, c_f]),
    "excel_gemm_f16x2": (c_i, [c_f, c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_f]),
    "excel_layernorm": (c_i, [c_f, c_f, c_f, c_f, c_i, c_i, C.c_float, c_f]),
    "excel_vit_create": (c_i, [C.POINTER(VitConfig), C.POINTER(VitWeights), C.POINTER(C.c_void_p)]),
    "excel_vit_destroy": (None, [C.c_void_p]),
    "excel_vit_workspace_bytes": (c_sz, [C.c_void_p, c_i, c_i]),
    "excel_vit_set_gemm_mode": (c_i, [C.c_void_p, c_i]),
    "excel_vit_get_gemm_mode": (c_i, [C.c_void_p]),
    "excel_vit_weights_fp16_exact": (c_i, [C.c_void_p]),
    "excel_vit_forward": (c_i, [C.c_void_p, c_f, c_i, c_i, c_f, c_sz, c_f, c_f, c_f, c_i, c_f, c_i, c_f, c_f]),
    "excel_vit_forward_ex": (c_i, [C.c_void_p, c_f, c_i, c_i, c_f, c_sz, c_f, c_f, c_f, c_i, c_f, c_i, c_f, c_f, c_i, c_f]),
    "excel_feature_affinity_workspace_bytes": (c_sz, [c_i, c_i, c_i]),
    "excel_feature_affinity": (c_i, [c_f, c_i, c_i, c_i, C.c_float, C.c_float, c_i, c_f, c_f, c_f]),
    "excel_attn_select_workspace_bytes": (c_sz, [c_i, c_i]),
    "excel_attn_select_mean": (c_i, [c_f, c_i, c_i, c_i, c_i, c_i, c_f, c_f, c_f, c_f]),
    "excel_decoder_create": (c_i, [C.POINTER(DecoderConfig), C.POINTER(DecoderWeights), C.POINTER(C.c_void_p)]),
    "excel_decoder_destroy": (None, [C.c_void_p]),
    "excel_decoder_workspace_bytes": (c_sz, [C.c_void_p, c_i, c_i]),
    "excel_decoder_forward": (c_i, [C.c_void_p, c_f, c_i, c_i, c_f, c_sz, c_f, c_f, c_f]),
    "excel_text_create": (c_i, [C.POINTER(TextConfig), C.POINTER(TextWeights), C.POINTER(C.c_void_p)]),
    "excel_text_destroy": (None, [C.c_void_p]),
    "excel_text_workspace_bytes": (c_sz, [C.c_void_p, c_i]),
    "excel_text_encode": (c_i, [C.c_void_p, c_f, c_i, c_f, c_f, c_sz, c_f]),
    "excel_prompt_ensemble": (c_i, [c_f, c_i, c_i, c_f, c_f]),
    "excel_decoder_train_workspace_bytes": (c_sz, [C.c_void_p, c_i, c_i]),
    "excel_decoder_forward_train": (c_i, [C.c_void_p, c_f, c_i, c_i, c_f, c_sz, c_f, c_f, C.c_float, C.c_uint, c_f]),
    "excel_decoder_train_attn_fts": (c_i, [C.c_void_p, c_i, c_i, c_f, c_sz, c_f, c_f]),
    "excel_decoder_backward": (c_i, [C.c_void_p, c_f, c_i, c_i, c_f, c_sz, c_f, c_f, C.POINTER(DecoderWeights), C.c_float, C.c_uint, c_f]),
    "excel_adamw_step": (c_i, [c_f, c_f, c_f, c_f, c_ll, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, c_i, c_f]),
    "excel_train_losses_workspace_bytes": (c_sz, [c_i, c_i, c_i, c_i]),
    "excel_train_losses": (c_i, [c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, C.c_float, C.c_float, c_f, c_f, c_f, c_f, c_f]),
    "excel_lam_to_label": (c_i, [c_f, c_f, c_f, c_i, c_i, c_i, c_i, C.c_float, C.c_float, C.c_float, c_i, c_i, c_f, c_f, c_f]),
    "excel_normalize_img_u8": (c_i, [c_f, c_i, c_i, c_i, C.POINTER(C.c_double), C.POINTER(C.c_double), c_f, c_f]),
    "excel_denormalize_img": (c_i, [c_f, c_i, c_i, c_i, C.POINTER(C.c_float), C.POINTER(C.c_float), c_f, c_f, c_f]),
    "excel_seg_scale_accumulate": (c_i, [c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, C.c_float, c_f]),
    "excel_cam_workspace_bytes": (c_sz, [c_i, c_i, c_i]),
    "excel_patch_text_cam_workspace_bytes": (c_sz, [c_i, c_i, c_i, c_i]),
    "excel_patch_text_cam": (c_i, [c_f, c_f, c_i, c_i, c_i, c_i, c_i, C.c_float, c_i, c_f, c_f, c_f, c_f, c_f]),
    "excel_clip_feature_surgery": (c_i, [c_f, c_f, c_i, c_i, c_i, c_i, c_i, C.c_float, c_f, c_f, c_f, c_f]),
    "excel_dcrf_workspace_bytes": (c_sz, [c_i, c_i, c_i]),
    "excel_dcrf_inference": (c_i, [c_f, c_f, c_i, c_i, c_i, c_i, c_i, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, c_f, c_f, c_f]),
    "excel_attn_layer_mean": (c_i, [c_f, c_i, c_i, c_i, c_i, c_i, c_f, c_f]),
    "excel_trans_mat_workspace_bytes": (c_sz, [c_i, c_i]),
    "excel_compute_trans_mat": (c_i, [c_f, c_i, c_i, c_f, c_f, c_f]),
    "excel_cls_compact": (c_i, [c_f, c_i, c_i, c_i, c_f, c_f, c_f, c_f]),
    "excel_scoremap_box_mask": (c_i, [c_f, c_f, c_f, c_i, c_i, c_i, c_i, C.c_double, c_f, c_f, c_f]),
    "excel_refine_workspace_bytes": (c_sz, [c_i, c_i, c_i]),
    "excel_refine_cams_with_aff": (c_i, [c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, C.c_double, c_f, c_f, c_f]),
    "excel_cam_upsample_bkg": (c_i, [c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_f, c_f, c_i, c_f]),
    "excel_par_workspace_bytes": (c_sz, [c_i, c_i, c_i, c_i, c_i]),
    "excel_par_forward": (c_i, [c_f, c_i, c_i, c_f, c_f, c_i, c_i, c_i, c_i, C.POINTER(C.c_int32), c_i, c_i,
                                C.c_float, C.c_float, c_f, c_f, c_i, c_f]),
    "excel_ragged_plan": (c_i, [C.POINTER(C.c_int32), c_i, C.POINTER(RaggedInfo), C.POINTER(C.c_int32)]),
    "excel_normalize_resize_u8_ragged": (c_i, [c_f, c_f, c_i, c_i, C.POINTER(C.c_double), C.POINTER(C.c_double), c_f, c_f]),
    "excel_cam_upsample_bkg_ragged": (c_i, [c_f, c_f, c_f, C.POINTER(RaggedInfo), c_i, c_i, c_f, c_f, c_i, c_f]),
    "excel_par_ragged_workspace_bytes": (c_sz, [c_ll, c_i]),
    "excel_par_forward_ragged": (c_i, [c_f, c_i, c_i, c_f, c_f, c_f, C.POINTER(RaggedInfo), c_i, C.POINTER(C.c_int32), c_i, c_i,
                                       C.c_float, C.c_float, c_f, c_f, c_f]),
    "excel_argmax_label_ragged": (c_i, [c_f, c_f, c_f, c_f, C.POINTER(RaggedInfo), c_i, c_i, c_f, c_f]),
    "excel_argmax_label": (c_i, [c_f, c_f, c_f, c_i, c_i, c_i, c_ll, c_f, c_f, c_f]),
    "excel_confusion_accumulate": (c_i, [c_f, c_f, c_ll, c_i, c_f, c_f]),
    "excel_attr_aggregate": (c_i, [c_f, c_f, c_i, c_i, c_i, c_i, C.c_double, c_f, c_f]),
    "excel_bilinear_resize": (c_i, [c_f, c_f, c_ll, c_i, c_i, c_i, c_i, c_i, c_f]),
    "excel_pos_embed_resize": (c_i, [c_f, c_i, c_i, c_i, c_f, c_f]),
    "excel_flip_max_normalize": (c_i, [c_f, c_f, c_i, c_i, c_i, c_f]),
    "excel_lam_scale_accumulate": (c_i, [c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_f]),
    "excel_plane_minmax_normalize": (c_i, [c_f, c_ll, c_ll, c_f]),
    "excel_prof_enable": (c_i, [c_i]),
    "excel_prof_set_mask": (c_i, [C.c_ulonglong]),
    "excel_prof_set_sampling": (c_i, [c_i]),
    "excel_prof_num_categories": (c_i, []),
    "excel_prof_category_name": (C.c_char_p, [c_i]),
    "excel_prof_collect": (c_i, [C.POINTER(C.c_double), C.POINTER(C.c_longlong), C.POINTER(C.c_double)]),
}

_lib = None


def lib():
    """Load (once) and return the bound library; raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    # torch first: its wheel bundles its own libamdhip64; if this library is loaded before it, the process ends up with two HIP
    # runtimes and the second one cannot see the device ("no ROCm-capable device is detected" on the first launch)
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"excel_amd: {LIB_PATH} is missing - the HIP extension is the only compute path. "
            "Build it with `python -m excel_amd.build` (or __graft_entry__.build()).")
    handle = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(handle, name)        # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = handle
    return _lib


def build_id():
    """Source id compiled into the loaded library (include/excel_hip.h: excel_build_id)."""
    return lib().excel_build_id().decode()


def check(rc, what):
    if rc != 0:
        msg = lib().excel_last_error()
        raise RuntimeError(f"{what} failed (rc={rc}): {msg.decode() if msg else '?'}")
