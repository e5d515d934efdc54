"""excel_amd - MI355X-native (gfx950) implementation of ExCEL's training-free patch-text CAM +
affinity random-walk + PAR hot path, behind the reference's own Python call surface.

Layout mirrors the reference repo (zwyang6/ExCEL) for the modules on the path:
    excel_amd.clip          generate_clip_fts, clip_feature_surgery, VisionTransformer   (clip/)
    excel_amd.model         ExCEL_model, attr_aggregate                                  (model/)
    excel_amd.utils         affutils, PAR, camutils, evaluate                            (utils/)
    excel_amd.tools         infer_lam (the training-free evaluation harness)             (tools/)
    excel_amd.pipeline      batched, fully device-resident fast path over the same kernels
    excel_amd.csrc          HIP kernels + the C ABI (libexcel_hip.so, include/excel_hip.h)
All compute goes through libexcel_hip.so; there is no eager/CPU fallback.
"""
__version__ = "0.1.0"
