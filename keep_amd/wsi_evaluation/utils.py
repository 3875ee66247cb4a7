"""``WSI_evaluation/utils.py`` of the reference, on the MI355X engine (same names, arguments, return values)."""
from keep_amd.cohort import WSI_Classification_Dataset                                           # utils.py:11-61
from keep_amd.wsi import (accuracy, cood2str, get_zeroshot_classifier, rank_cls_score, str2cood,  # utils.py:64-156
                          zero_shot_classifier, zero_shot_prompt_select)

__all__ = ["WSI_Classification_Dataset", "zero_shot_classifier", "get_zeroshot_classifier", "rank_cls_score",
           "zero_shot_prompt_select", "cood2str", "str2cood", "accuracy"]
