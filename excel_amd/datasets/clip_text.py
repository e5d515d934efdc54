"""Class / background prompt vocabularies of the text bank (data; the lists of datasets/clip_text.py:6-62, stored next to this
module as clip_text.json).  ExCEL_model concatenates `new_class_names + BACKGROUND_CATEGORY` (VOC, 20 + 25 = 45 prompts) or
`new_class_names_coco + BACKGROUND_CATEGORY_COCO` (COCO, 80 + 23 = 103) exactly like model/model_excel.py:31."""
import json
import os

with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "clip_text.json")) as _f:
    _d = json.load(_f)

BACKGROUND_CATEGORY = _d["BACKGROUND_CATEGORY"]
class_names = _d["class_names"]
new_class_names = _d["new_class_names"]
class_names_coco = _d["class_names_coco"]
new_class_names_coco = _d["new_class_names_coco"]
BACKGROUND_CATEGORY_COCO = _d["BACKGROUND_CATEGORY_COCO"]


def text_prompts(num_classes):
    """model/model_excel.py:31"""
    return list(new_class_names + BACKGROUND_CATEGORY) if num_classes <= 21 else list(new_class_names_coco + BACKGROUND_CATEGORY_COCO)
