from .model_excel import ExCEL_model  # noqa: F401
from .load_attr import attr_aggregate  # noqa: F401
