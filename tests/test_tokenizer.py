"""CLIP byte-level BPE tokenizer (uniir_amd.clip_front, the clip.tokenize call of clip_sf.py:35-41) against G15: ids
produced by transformers' CLIPTokenizer (an independent implementation) on a synthetic merge list, plus the padding /
truncation contract of clip.tokenize.  Integer work: exact."""
import gzip
import json
import os

import pytest
import torch

G = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture()
def bpe_file(tmp_path):
    z = json.load(open(os.path.join(G, "g15_bpe.json"), encoding="utf-8"))
    path = str(tmp_path / "bpe_simple_vocab_16e6.txt.gz")
    with gzip.open(path, "wt", encoding="utf-8") as f:
        f.write(z["merges_file"])
    return path, z


def test_g15_bpe_ids_match_the_independent_implementation(bpe_file):
    from uniir_amd.clip_front import BPETokenizer
    path, z = bpe_file
    tk = BPETokenizer(path)
    assert (tk.sot, tk.eot) == (len(tk.encoder) - 2, len(tk.encoder) - 1) and len(tk.encoder) == 512 + 120 + 2
    for text, want in zip(z["texts"], z["ids"]):
        assert [tk.sot] + tk.encode(text) + [tk.eot] == want, text


def test_tokenize_contract(bpe_file, monkeypatch):
    from uniir_amd import clip_front
    path, z = bpe_file
    monkeypatch.setenv("UNIIR_BPE_PATH", path)
    monkeypatch.setattr(clip_front, "_TOKENIZER", None)
    out = clip_front.tokenize(z["texts"][:3], context_length=77)
    assert out.dtype == torch.int32 and out.shape == (3, 77)
    for row, want in zip(out.tolist(), z["ids"][:3]):
        assert row[:len(want)] == want and not any(row[len(want):])
    single = clip_front.tokenize(z["texts"][2], context_length=8, truncate=True)         # a str, cut, EOT forced last
    assert single.shape == (1, 8) and single[0, :7].tolist() == z["ids"][2][:7] and single[0, 7].item() == z["ids"][2][-1]
    with pytest.raises(RuntimeError):
        clip_front.tokenize(z["texts"][2], context_length=8)
    monkeypatch.setattr(clip_front, "_TOKENIZER", None)
    monkeypatch.delenv("UNIIR_BPE_PATH")
    monkeypatch.setenv("HOME", os.path.dirname(path) + "/nohome")
    with pytest.raises(RuntimeError):                                                     # no vocabulary: loud, not silent
        clip_front.tokenize("a photo")


def test_blip_tokenizer_offline_with_a_vocabulary_dir(tmp_path, monkeypatch):
    """blip.init_tokenizer (backbone/blip.py:221-226): BertTokenizer + "[DEC]" / "[ENC]" from a local vocabulary
    directory (UNIIR_BERT_VOCAB_DIR), and the model-side wrapper's max_length padding gives the prefix masks the BERT
    kernels take as one key length per row; without a vocabulary the call fails loudly"""
    from uniir_amd import blip_front
    vocab = ["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]", "a", "photo", "of", "red", "dog", "##s", "running", "."]
    (tmp_path / "vocab.txt").write_text("\n".join(vocab) + "\n")
    monkeypatch.setenv("UNIIR_BERT_VOCAB_DIR", str(tmp_path))
    tok = blip_front.init_tokenizer()
    assert tok.bos_token == "[DEC]" and tok.enc_token_id == tok.convert_tokens_to_ids("[ENC]") and tok.enc_token_id >= len(vocab)
    out = tok(["a photo of red dogs running.", "dog"], padding="max_length", truncation=True, max_length=12, return_tensors="pt")
    ids, mask = out["input_ids"], out["attention_mask"]
    assert ids.shape == (2, 12) and ids[0, 0].item() == 2          # [CLS] first
    assert ids[0, :10].tolist() == [2, 5, 6, 7, 8, 9, 10, 11, 12, 3]     # wordpiece "dog" + "##s", then [SEP]
    assert bool((mask[:, :-1] >= mask[:, 1:]).all()) and mask.sum(1).tolist() == [10, 3]
    monkeypatch.setenv("UNIIR_BERT_VOCAB_DIR", str(tmp_path / "missing"))
    with pytest.raises(RuntimeError):
        blip_front.init_tokenizer()
