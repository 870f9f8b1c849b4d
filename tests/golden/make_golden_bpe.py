"""G15: golden token ids for the CLIP byte-level BPE tokenizer on a synthetic merge list (the real 16e6 vocabulary is not
available offline).  The merge list is trained here on a toy corpus (plain pair-frequency BPE), written in upstream's
file format, and the expected ids come from an independent third-party implementation run in the build container:
transformers' CLIPTokenizer (the Rust `tokenizers` BPE model with CLIP's normaliser / splitter), given the vocabulary in
upstream's order.  tests/test_tokenizer.py holds uniir_amd.clip_front.BPETokenizer / tokenize to these.
    python tests/golden/make_golden_bpe.py"""
import collections
import json
import os

import transformers
from transformers import CLIPTokenizer

HERE = os.path.dirname(os.path.abspath(__file__))
corpus = ("a photo of a red dog running on the beach . the quick brown fox jumps over the lazy dog ! "
          "two cats are sitting on a wooden table , it's raining ; we've seen 3 birds and 42 trees . "
          "fashion dress with floral pattern ? don't stop believing , i'm here , you'll see , they'd go").split()
keep = list(range(33, 127)) + list(range(161, 173)) + list(range(174, 256))
byte_sym, extra = {}, 0
for b in range(256):
    if b in keep:
        byte_sym[b] = chr(b)
    else:
        byte_sym[b] = chr(256 + extra)
        extra += 1
words = collections.Counter()
for w in corpus:
    sym = [byte_sym[b] for b in w.encode()]
    sym[-1] += "</w>"
    words[tuple(sym)] += 1
merges = []
for _ in range(120):
    pairs = collections.Counter()
    for w, c in words.items():
        for a, b in zip(w, w[1:]):
            pairs[(a, b)] += c
    if not pairs:
        break
    best = max(sorted(pairs), key=lambda p: pairs[p])
    merges.append(best)
    new = collections.Counter()
    for w, c in words.items():
        out, i = [], 0
        while i < len(w):
            if i + 1 < len(w) and (w[i], w[i + 1]) == best:
                out.append(w[i] + w[i + 1])
                i += 2
            else:
                out.append(w[i])
                i += 1
        new[tuple(out)] += c
    words = new
base = [byte_sym[b] for b in keep] + [byte_sym[b] for b in range(256) if b not in keep]
vocab = base + [v + "</w>" for v in base] + ["".join(m) for m in merges] + ["<|startoftext|>", "<|endoftext|>"]
ref = CLIPTokenizer(vocab={t: i for i, t in enumerate(vocab)}, merges=merges)
texts = ["a photo of a red dog", "The Quick  brown fox!", "it's raining; we've seen 3 birds and 42 trees.", "don't stop",
         "naïve café — ¿qué?", "  leading and trailing  ", "UPPER lower 007", "<|startoftext|> inline special",
         "emoji 🙂 test", "tabs\tand\nnewlines", "they'd go , you'll see", ""]
out = {"transformers_version": transformers.__version__, "merges_file": "#version: 0.2\n" + "\n".join(" ".join(m) for m in merges) + "\n",
       "texts": texts, "ids": [ref(t)["input_ids"] for t in texts]}
with open(os.path.join(HERE, "g15_bpe.json"), "w", encoding="utf-8") as f:
    json.dump(out, f, ensure_ascii=False, indent=0)
print("wrote g15_bpe.json", len(merges), "merges", len(texts), "texts")
