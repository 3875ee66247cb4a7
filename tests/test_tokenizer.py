"""Row a8: the tokenizer.  keep_amd's native WordPiece implementation against (1) the oracle's restatement of uncased WordPiece
and (2) transformers.BertTokenizerFast -- the class the reference's AutoTokenizer returns -- token for token, on a synthetic
vocabulary (the PubMedBERT one is not available offline)."""
import os

import pytest
import torch

from keep_amd.tokenizer import load_tokenizer, tokenize
from oracle import keep_oracle as O

WORDS = ["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]", "a", "an", "of", "the", "image", "h", "&", "e", ".", ",", "-",
         "histo", "##path", "##ology", "tumor", "tumour", "normal", "tissue", "clear", "cell", "renal", "carcinoma",
         "##s", "papillary", "chromo", "##phobe", "lympho", "##cyte", "##cytes", "slide", "showing", "(", ")", "20", "##x",
         "##-", "癌", "細", "胞", "naive", "cafe", "##e", "µ", "m", "µm", "5", "!", "?", "[", "]", "mask", "x", "##y", "##z", "β", "##β"]


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


TRICKY = ["An H&E image of Clear Cell Renal Cell Carcinoma.", "a histopathology slide showing lymphocytes (20x)", "", "   ",
          "papillary tumours, chromophobe; unknownword", "tumor " * 400, "Tumör-tissue", "naïve café CAFÉ", "癌細胞 tumor細胞", "5µm µ m",
          "a\tb\nc\r\nd\u00a0e\u2003f", "tab\x00null\ufffdrepl\x07bell\u200bzw", "x" * 101 + " " + "x" * 100, "tumor[MASK]cell [SEP] [CLS][PAD]",
          "!?[]", "ǅ İstanbul ß ſ", "xyz xyzz βββ", "e\u0301 a\u0300 normal",
          "a\u2028b a\u2029normal", "tumor\u038bcell a\u0383 \u0378", "x\x0by\x0cz\x85normal", "a\ue000b \u00adnormal\u202e"]


def test_native_tokenizer_matches_bert_tokenizer_fast(vocab_dir):
    """Token for token against the tokenizer class the reference gets from AutoTokenizer (uncased BERT: clean-up, CJK isolation,
    accent stripping, lower-casing, punctuation splitting, greedy WordPiece, 100-character limit, special tokens in the text)."""
    transformers = pytest.importorskip("transformers")
    hf = transformers.BertTokenizerFast.from_pretrained(vocab_dir, local_files_only=True, do_lower_case=True)
    assert hf.vocab_size == len(WORDS)
    nat = load_tokenizer(vocab_dir)
    assert type(nat).__name__ == "WordPieceTokenizer"
    a = hf(TRICKY, max_length=256, padding="max_length", truncation=True, return_tensors="pt")
    b = nat(TRICKY, max_length=256, padding="max_length", truncation=True, return_tensors="pt")
    for i, t in enumerate(TRICKY):
        n = int(a["attention_mask"][i].sum())
        assert b["input_ids"][i].tolist() == a["input_ids"][i].tolist(), (t, hf.convert_ids_to_tokens(a["input_ids"][i][:n].tolist()))
        assert b["attention_mask"][i].tolist() == a["attention_mask"][i].tolist() and int(b["token_type_ids"][i].sum()) == 0
    # the other call shapes of the HF surface that the examples use
    assert nat("a tumor")["input_ids"] == hf("a tumor")["input_ids"] and nat("a tumor")["attention_mask"] == hf("a tumor")["attention_mask"]
    lo = nat(TRICKY[:3], padding=True, return_tensors="pt")
    assert lo["input_ids"].shape == hf(TRICKY[:3], padding=True, return_tensors="pt")["input_ids"].shape
    assert lo.to("cpu").input_ids.dtype == torch.int64
    assert type(load_tokenizer(vocab_dir, backend="hf")).__name__.startswith("Bert")


def test_missing_vocabulary_is_an_error(tmp_path):
    with pytest.raises(FileNotFoundError):
        load_tokenizer(str(tmp_path))
    with pytest.raises(FileNotFoundError):
        load_tokenizer(os.path.join(str(tmp_path), "nope"))


def test_native_tokenizer_fuzz_against_bert_tokenizer_fast(vocab_dir):
    """1 500 random strings over Latin-1 / Latin Extended, every Unicode space and line separator, unassigned and private-use code
    points, soft hyphens, BOMs and non-characters: the native tokenizer gives the ids of BertTokenizerFast on every one."""
    import random
    transformers = pytest.importorskip("transformers")
    hf = transformers.BertTokenizerFast.from_pretrained(vocab_dir, local_files_only=True, do_lower_case=True)
    nat = load_tokenizer(vocab_dir)
    rnd = random.Random(0)
    pool = [chr(c) for c in list(range(0, 0x250)) + [0x2028, 0x2029, 0x85, 0x38B, 0x383, 0x378, 0xE000, 0xAD, 0x200B, 0x3000, 0x1680, 0x205F,
                                                      0x202F, 0xFEFF, 0xFFFD, 0x764C, 0x10FFFF, 0xFDD0]]
    words = [w.replace("##", "") for w in WORDS if not w.startswith("[")]
    for _ in range(1500):
        t = "".join(rnd.choice(pool) if rnd.random() < 0.5 else rnd.choice(words) + (" " if rnd.random() < 0.5 else "")
                    for _ in range(rnd.randint(1, 12)))
        assert nat(t)["input_ids"] == hf(t)["input_ids"], repr(t)
