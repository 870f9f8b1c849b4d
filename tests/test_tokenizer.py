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
