"""Deterministic stand-ins shared by tools/make_golden.py (which feeds them to the REFERENCE's own functions) and the GPU
tests (which feed them to keep_amd): the PubMedBERT vocabulary and real feature files do not exist offline."""
import torch


class HashTokenizer:
    """``tokenizer(texts, max_length=256, padding='max_length', truncation=True, return_tensors='pt')`` -> BatchEncoding-like
    object with ``.to(device)`` (WSI_evaluation/utils.py:73 calls exactly that).  Token ids are a hash of the lower-cased words."""
    calls = 0

    class Encoding(dict):
        def to(self, device):
            return HashTokenizer.Encoding({k: v.to(device) for k, v in self.items()})

    def __call__(self, texts, max_length=256, padding="max_length", truncation=True, return_tensors="pt"):
        texts = [texts] if isinstance(texts, str) else list(texts)
        HashTokenizer.calls += len(texts)
        ids = torch.zeros(len(texts), max_length, dtype=torch.int64)
        mask = torch.zeros_like(ids)
        for i, t in enumerate(texts):
            toks = [2] + [5 + (sum(map(ord, w)) * 31 + len(w)) % 30000 for w in t.lower().split()][: max_length - 2] + [3]
            ids[i, :len(toks)] = torch.tensor(toks)
            mask[i, :len(toks)] = 1
        return HashTokenizer.Encoding({"input_ids": ids, "token_type_ids": torch.zeros_like(ids), "attention_mask": mask})


CALLER_LABEL_MAP = {"CHRCC": 0, "CCRCC": 1, "PRCC": 2}
CALLER_PROMPTS = [
    {"classnames": {"CHRCC": "chromophobe renal cell carcinoma", "CCRCC": "clear cell renal cell carcinoma",
                    "PRCC": "papillary renal cell carcinoma", "Normal": "normal kidney tissue"}, "templates": "an H&E image of CLASSNAME."},
    {"classnames": {"CHRCC": "renal chromophobe carcinoma", "CCRCC": "clear cell carcinoma of the kidney",
                    "PRCC": "papillary carcinoma of the kidney", "Normal": "benign renal parenchyma"},
     "templates": "a histopathology slide showing CLASSNAME."},
    # a LIST of templates: the reference keeps only the first one (`encode_text(text_inputs)[0]`, utils.py:74)
    {"classnames": {"CHRCC": "chromophobe renal cell carcinoma", "CCRCC": "clear cell renal cell carcinoma",
                    "PRCC": "papillary renal cell carcinoma", "Normal": "normal kidney tissue"},
     "templates": ["CLASSNAME is present.", "an example of CLASSNAME, H&E stain."]},
]
CALLER_SLIDES = (("slide_0", 21, "A"), ("slide_1", 1, "B"), ("slide_2", 130, "A"))       # (id, tiles, diagnosis)
CALLER_DIAG_MAP = {"A": 0, "B": 1}


def caller_slide_features(seed: int = 77):
    g = torch.Generator().manual_seed(seed)
    return {sid: torch.randn(n, 768, generator=g) * 2.0 for sid, n, _ in CALLER_SLIDES}
