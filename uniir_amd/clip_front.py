"""The slice of the `clip` package surface that UniIR uses (clip.load, clip.tokenize), backed by uniir_amd.

Reference call sites: src/models/uniir_clip/clip_scorefusion/clip_sf.py:25-26 (clip.load -> (model, preprocess);
clip.tokenize).  Upstream downloads weights / ships a BPE vocabulary; neither exists offline, so:
  * load(): builds the architecture for `name` (random init, upstream std choices) and, when
    <download_root>/<name>.pt (a state dict, or a checkpoint with a "state_dict"/"model" entry) exists, loads it;
  * tokenize(): needs <UNIIR_BPE_PATH or download_root>/bpe_simple_vocab_16e6.txt.gz; without the file it raises
    (tokenizer parity is unpinned offline, SURVEY.md section 7 'No network').
"""
import os

import numpy as np
import torch

from .clip_model import CLIP, CLIP_CONFIGS

_MEAN = (0.48145466, 0.4578275, 0.40821073)
_STD = (0.26862954, 0.26130258, 0.27577711)


def _preprocess(n_px):
    def fn(image):
        """PIL image -> float tensor [3, n_px, n_px]: bicubic resize of the short side, centre crop, RGB,
        CLIP mean/std normalisation (upstream clip._transform)."""
        from PIL import Image
        w, h = image.size
        s = n_px / min(w, h)
        nw, nh = max(n_px, round(w * s)), max(n_px, round(h * s))
        image = image.resize((nw, nh), Image.BICUBIC)
        left, top = (nw - n_px) // 2, (nh - n_px) // 2
        image = image.crop((left, top, left + n_px, top + n_px)).convert("RGB")
        a = torch.from_numpy(np.asarray(image, dtype=np.float32) / 255.0).permute(2, 0, 1)
        return (a - torch.tensor(_MEAN).view(3, 1, 1)) / torch.tensor(_STD).view(3, 1, 1)

    return fn


def load(name="ViT-B/32", device="cuda", jit=False, download_root=None, seed=0):
    if name not in CLIP_CONFIGS:
        raise RuntimeError(f"Model {name} not found; available models = {list(CLIP_CONFIGS)}")
    model = CLIP(CLIP_CONFIGS[name], seed=seed)
    if download_root:
        path = os.path.join(os.path.expanduser(download_root), name.replace("/", "-") + ".pt")
        if os.path.exists(path):
            sd = torch.load(path, map_location="cpu")
            sd = sd.get("state_dict", sd.get("model", sd)) if isinstance(sd, dict) else sd.state_dict()
            sd = {k: v.float() for k, v in sd.items() if k in model.state_dict()}
            model.load_state_dict(sd, strict=True)
    if device is not None and str(device) != "cpu":
        model = model.to(device)
    return model, _preprocess(CLIP_CONFIGS[name]["image_resolution"])


def tokenize(texts, context_length=77, truncate=False):
    raise RuntimeError(
        "clip.tokenize needs OpenAI's bpe_simple_vocab_16e6.txt.gz, which is not available offline; feed token ids "
        "(int32 [n, 77], SOT=49406 ... EOT=49407, zero padded) or install the vocabulary (see INTEGRATION.md)")
