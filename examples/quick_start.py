#!/usr/bin/env python3
"""The reference's quick start (quick_start/keep_inference.py:79-104) on keep_amd: one tile x three prompts.

    python examples/quick_start.py [--model-path ../KEEP_release/] [--image tests/golden/example.tif]

With --model-path (config.json + pytorch_model.bin / model.safetensors + tokenizer files) this is the reference
script with the model class swapped.  Without it (no weights or vocabulary exist offline) it runs on seeded
synthetic weights and a stand-in tokenizer, which exercises exactly the same code path.
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from keep_amd import KEEPModel                        # noqa: E402
from keep_amd.preprocess import preprocess            # noqa: E402


class StandInTokenizer:
    def __call__(self, texts, max_length=256, padding="max_length", truncation=True, return_tensors="pt"):
        ids = torch.zeros(len(texts), max_length, dtype=torch.int64)
        mask = torch.zeros_like(ids)
        for i, t in enumerate(texts):
            toks = [2] + [5 + (sum(map(ord, w)) * 31 + len(w)) % 30000 for w in t.lower().split()][: max_length - 2] + [3]
            ids[i, :len(toks)] = torch.tensor(toks)
            mask[i, :len(toks)] = 1
        return {"input_ids": ids, "token_type_ids": torch.zeros_like(ids), "attention_mask": mask}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model-path", default=None)
    ap.add_argument("--image", default=os.path.join(ROOT, "tests", "golden", "example.tif"))
    args = ap.parse_args()
    if args.model_path:
        from keep_amd.tokenizer import load_tokenizer
        model = KEEPModel.from_pretrained(args.model_path)
        tokenizer = load_tokenizer(args.model_path)
    else:
        from keep_amd.synth import synth_state_dict
        model = KEEPModel()
        model.load_state_dict(synth_state_dict(seed=0), strict=True)
        tokenizer = StandInTokenizer()
    model.to("cuda").eval()

    example_text = ['an H&E image of breast invasive carcinoma.', 'an H&E image of normal tissue.', 'an H&E image of lung adenocarcinoma.']
    img_input = preprocess(args.image).unsqueeze(0)
    token_input = tokenizer(example_text, max_length=256, padding='max_length', truncation=True, return_tensors='pt')
    img_feature = model.encode_image(img_input)
    text_feature = model.encode_text(token_input)
    print(img_feature @ text_feature.T)


if __name__ == "__main__":
    main()
