from .clip import generate_clip_fts, clip_feature_surgery, load, LazyAttnWeights  # noqa: F401
from .clip_surgery_model import VisionTransformer, ExCEL_CLIP  # noqa: F401
