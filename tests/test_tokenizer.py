"""Row a8: the tokenizer call.  A synthetic vocabulary (the PubMedBERT one is not available offline) through the same
third-party library the reference uses, checked against the oracle's restatement of uncased WordPiece."""
import os

import pytest
import torch

from keep_amd.tokenizer import load_tokenizer, tokenize
from oracle import keep_oracle as O

WORDS = ["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]", "a", "an", "of", "the", "image", "h", "&", "e", ".", ",", "-",
         "histo", "##path", "##ology", "tumor", "tumour", "normal", "tissue", "clear", "cell", "renal", "carcinoma",
         "##s", "papillary", "chromo", "##phobe", "lympho", "##cyte", "##cytes", "slide", "showing", "(", ")", "20", "##x"]


@pytest.fixture()
def vocab_dir(tmp_path):
    with open(tmp_path / "vocab.txt", "w") as f:
        f.write("\n".join(WORDS) + "\n")
    return str(tmp_path)


def test_reference_call_layout_and_wordpiece(vocab_dir):
    tok = load_tokenizer(vocab_dir)
    vocab = {w: i for i, w in enumerate(WORDS)}
    texts = ["An H&E image of Clear Cell Renal Cell Carcinoma.", "a histopathology slide showing lymphocytes (20x)",
             "papillary tumours, chromophobe; unknownword", "", "tumor " * 400, "Tumör-tissue"]
    enc = tokenize(tok, texts)
    assert set(enc.keys()) >= {"input_ids", "token_type_ids", "attention_mask"}
    for k in ("input_ids", "token_type_ids", "attention_mask"):
        assert enc[k].shape == (len(texts), 256) and enc[k].dtype == torch.int64
    assert int(enc["token_type_ids"].abs().sum()) == 0
    for i, t in enumerate(texts):
        ids, mask = O.wordpiece_ids(vocab, t, 256)
        assert enc["input_ids"][i].tolist() == ids, t
        assert enc["attention_mask"][i].tolist() == mask, t
    # [CLS] ... [SEP] [PAD]*; the long text is truncated to 256 with [SEP] last
    assert enc["input_ids"][4, 0] == vocab["[CLS]"] and enc["input_ids"][4, 255] == vocab["[SEP]"] and int(enc["attention_mask"][4].sum()) == 256
    assert enc["input_ids"][3].tolist()[:3] == [vocab["[CLS]"], vocab["[SEP]"], vocab["[PAD]"]]


def test_missing_vocabulary_is_an_error(tmp_path):
    with pytest.raises(FileNotFoundError):
        load_tokenizer(str(tmp_path))
    with pytest.raises(FileNotFoundError):
        load_tokenizer(os.path.join(str(tmp_path), "nope"))
