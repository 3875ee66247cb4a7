"""``WSI_evaluation/segment_utils.py`` of the reference, on the MI355X engine (same names, arguments, return values).
``eval_seg_auc`` / ``eval_seg_coarse`` (:91-152: AUC / Dice against an openslide mask) are outside the hot path."""
from keep_amd.cohort import run_segmentation as run                                               # segment_utils.py:16-42
from keep_amd.wsi import refine_seg_segment as refine_seg, zero_shot_segment                      # :63-89, :44-60

__all__ = ["run", "refine_seg", "zero_shot_segment"]
