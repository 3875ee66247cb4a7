#!/usr/bin/env python3
"""The reference's WSI_evaluation/zeroshot_subtyping_WSI.py on the MI355X engine.

Everything from `## load model` to the final print is the reference script's own text (zeroshot_subtyping_WSI.py:42-80),
unchanged; what differs is (1) the import block -- `import keep_amd.hf` registers the engine with AutoModel, and
`keep_amd.wsi_evaluation.PATH` goes first on sys.path so that `from utils import ...` / `from subtyping_utils import ...`
resolve to the engine's modules -- and (2) the data: no KEEP release, prompt file or CLAM .h5 exists offline, so a synthetic
release directory (config.json, model.safetensors with seeded weights, a WordPiece vocab.txt), a synthetic prompt file with the
RCC file's structure and a synthetic slide are written first, in the formats the script reads.

    python examples/zeroshot_subtyping_WSI.py [--depth 24] [--prompt-sets 48] [--tiles 2048]
"""
import argparse
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import keep_amd.hf                                                     # noqa: E402,F401  AutoConfig/AutoModel.register (keep_inference.py:75-76)
import keep_amd.wsi_evaluation                                         # noqa: E402
sys.path.insert(0, keep_amd.wsi_evaluation.PATH)                       # `from utils import ...` -> keep_amd/wsi_evaluation/utils.py

# ---- the reference's import block (zeroshot_subtyping_WSI.py:1-10), minus torchvision / h5py (absent in this image, unused below)
from tqdm import tqdm                                                  # noqa: E402
import json                                                            # noqa: E402
from utils import get_zeroshot_classifier, zero_shot_prompt_select     # noqa: E402
from subtyping_utils import zero_shot_subtyping                        # noqa: E402
from transformers import AutoModel, AutoTokenizer                      # noqa: E402
import torch                                                           # noqa: E402
import torch.nn.functional as F                                        # noqa: E402
import random                                                          # noqa: E402


def write_synthetic_inputs(tmp, depth, prompt_sets, tiles):
    """A release directory, a prompt file and a slide in the on-disk formats the reference script opens."""
    from safetensors.torch import save_file
    from keep_amd.config import KEEPShape, small_shape
    from keep_amd.synth import synth_state_dict
    shape = KEEPShape() if depth >= 24 else small_shape(depth, max(1, depth // 2))
    rel = os.path.join(tmp, "KEEP_release")
    os.makedirs(rel)
    sd = {k: v.contiguous() for k, v in synth_state_dict(shape, seed=0).items()}
    save_file(sd, os.path.join(rel, "model.safetensors"))
    json.dump({"model_type": "keep", "projection_dim": 768, "vision_config": None,
               "text_config": {"vocab_size": 30522, "hidden_size": 768, "num_hidden_layers": shape.text.num_hidden_layers,
                               "num_attention_heads": 12, "intermediate_size": 3072, "max_position_embeddings": 512, "type_vocab_size": 2}},
              open(os.path.join(rel, "config.json"), "w"))
    words = ("an h & e image of a histopathology slide showing tissue with features consistent . , chromophobe clear cell papillary renal "
             "carcinoma normal kidney tumor benign parenchyma kind type variant region").split()
    vocab = ["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]"] + sorted(set(words)) + [f"##{c}" for c in "abcdefghijklmnopqrstuvwxyz0123456789"] + \
        list("abcdefghijklmnopqrstuvwxyz0123456789")
    vocab += [f"[unused{i}]" for i in range(30522 - len(vocab))]
    open(os.path.join(rel, "vocab.txt"), "w").write("\n".join(vocab) + "\n")
    json.dump({"tokenizer_class": "BertTokenizer", "do_lower_case": True, "model_max_length": 512}, open(os.path.join(rel, "tokenizer_config.json"), "w"))
    names = {"CHRCC": ["chromophobe renal cell carcinoma", "renal carcinoma chromophobe type"],
             "CCRCC": ["clear cell renal cell carcinoma", "renal carcinoma clear cell type"],
             "PRCC": ["papillary renal cell carcinoma", "renal carcinoma papillary variant"],
             "Normal": ["normal kidney tissue", "benign renal parenchyma"]}
    templates = ["an H&E image of CLASSNAME.", "a histopathology slide showing CLASSNAME.", "tissue with features consistent with CLASSNAME."]
    rnd = random.Random(3)
    prompts = {str(i): {"classnames": {k: rnd.choice(v) for k, v in names.items()}, "templates": rnd.choice(templates)} for i in range(prompt_sets)}
    prompt_file = os.path.join(tmp, "synthetic_rcc_prompts.json")
    json.dump(prompts, open(prompt_file, "w"))
    grid = int(tiles ** 0.5) + 1
    idx = torch.arange(tiles)
    coords = torch.stack([(idx % grid) * 256, (idx // grid) * 256], 1).numpy()
    return rel, prompt_file, coords


ap = argparse.ArgumentParser()
ap.add_argument("--depth", type=int, default=24)
ap.add_argument("--prompt-sets", type=int, default=48)
ap.add_argument("--tiles", type=int, default=2048)
args = ap.parse_args()
_tmp = tempfile.TemporaryDirectory()
model_path, prompt_file, tile_coords = write_synthetic_inputs(_tmp.name, args.depth, args.prompt_sets, args.tiles)

test_data_name = 'RCC'
topn = 50
device = 'cuda:0'
wsi_label = {'CHRCC': 0, 'CCRCC': 1, 'PRCC': 2}
id_label = {0:'CHRCC', 1:'CCRCC', 2:'PRCC'}
prompt_screening = True

with open(prompt_file, 'r') as pf: 
    prompts = json.load(pf)

## load model
KEEP_model = dict()
model = AutoModel.from_pretrained(model_path, trust_remote_code=True).to(device)
model.eval()
tokenizer = AutoTokenizer.from_pretrained(model_path, trust_remote_code=True)
KEEP_model['model'] = model
KEEP_model['tokenizer'] = tokenizer

# the reference reads pre-extracted tile features from a CLAM .h5 (zeroshot_subtyping_WSI.py:38-40); here the extraction is the
# engine's own encode_image over a synthetic slide (h5py is not installed in this image)
from keep_amd.synth import synth_tiles_device                          # noqa: E402
tile_features = torch.cat([model.encode_image(synth_tiles_device(a, min(a + 256, args.tiles), device, torch.bfloat16))
                           for a in range(0, args.tiles, 256)])

## generate prompt classifier
merge_classifier = []
for prompt_idx in (pbar := tqdm(range(len(prompts)))):
    prompt = prompts[str(prompt_idx)]
    classifier = get_zeroshot_classifier(KEEP_model, wsi_label, prompt, device, add_normal=True)
    merge_classifier.append(classifier)

## select prompt classifier
if prompt_screening:
    print('Rank prompts...')
    ensemble_classifier = zero_shot_prompt_select(merge_classifier, tile_features, topn = topn, device = device)
else:
    ensemble_cls = torch.zeros_like(classifier)
    cter = 0
    while cter < topn:
        random.seed(cter)
        rand_id = random.randint(0,len(merge_classifier)-1)
        ensemble_cls += merge_classifier[rand_id]
        cter += 1
    ensemble_classifier = F.normalize(ensemble_cls, p=2, dim=0)

subtyping_preds = zero_shot_subtyping(ensemble_classifier, tile_features, tile_coords, patch_size = 256,  overlap = True)

print('Predicted subtype: ' + id_label[subtyping_preds.item()])
