"""``WSI_evaluation/detection_utils.py`` of the reference, on the MI355X engine (same names, arguments, return values).
``calculate_metric`` (sensitivity / specificity from sklearn's confusion matrix, :76-86) is evaluation code outside the hot path."""
from keep_amd.cohort import run_detection as run                                                  # detection_utils.py:12-36
from keep_amd.wsi import refine_seg_detection as refine_seg, zero_shot_detection                  # :39-74, :88-100

__all__ = ["run", "refine_seg", "zero_shot_detection"]
