from .clip import (generate_clip_fts, clip_feature_surgery, load, LazyAttnWeights, tokenize,  # noqa: F401
                   encode_text_with_prompt_ensemble)
from .clip_surgery_model import VisionTransformer, ExCEL_CLIP  # noqa: F401
