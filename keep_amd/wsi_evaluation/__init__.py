"""Drop-in modules for the reference's ``WSI_evaluation/`` helpers, under the reference's module names.

The three scripts import ``from utils import ...``, ``from subtyping_utils import zero_shot_subtyping`` ... with their own
directory on ``sys.path`` (``WSI_evaluation/zeroshot_subtyping_WSI.py:3-4``).  Either put THIS directory first on ``sys.path``
(``sys.path.insert(0, keep_amd.wsi_evaluation.PATH)``: the scripts' import lines then resolve here unchanged) or import
``keep_amd.wsi_evaluation.utils`` etc. explicitly.  Every function has the reference's name, arguments and return value; the
arithmetic runs on the MI355X through ``libkeep_hip.so``.
"""
import os

PATH = os.path.dirname(os.path.abspath(__file__))
