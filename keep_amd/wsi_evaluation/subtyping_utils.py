"""``WSI_evaluation/subtyping_utils.py`` of the reference, on the MI355X engine (same names, arguments, return values)."""
from keep_amd.cohort import run_subtyping as run                                                  # subtyping_utils.py:12-35
from keep_amd.wsi import refine_seg_subtyping as refine_seg, zero_shot_subtyping                  # :38-65, :67-83

__all__ = ["run", "refine_seg", "zero_shot_subtyping"]
