"""keep_amd: MI355X (gfx950) native zero-shot WSI inference engine for the KEEP hot path.

Public surface mirrors the reference model API (quick_start/keep_inference.py:25-73):

    from keep_amd import KEEPModel
    model = KEEPModel.from_pretrained(release_dir).to("cuda").eval()
    img = model.encode_image(tiles)          # [B,768] unit-norm fp32
    txt = model.encode_text(token_inputs)    # [P,768] unit-norm fp32
    sim = img @ txt.T                        # or model.similarity(img, txt, mode=...)
"""
from .config import KEEPShape, TextShape, VisionShape, bert_flops_per_prompt, vit_flops_per_tile
from .model import KEEPModel, PROFILE_TAGS

__all__ = ["KEEPModel", "KEEPShape", "VisionShape", "TextShape", "PROFILE_TAGS",
           "vit_flops_per_tile", "bert_flops_per_prompt"]
__version__ = "0.1.0"
