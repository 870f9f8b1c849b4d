"""Load-time / input-side helpers of the BLIP path that are not kernels (drop-in for the pieces of
src/models/uniir_blip/backbone/blip.py:221-226,263-289, backbone/vit.py:359-384 and
backbone/transform/blip_transform.py:9-49 that blip_ff.py imports).

  * init_tokenizer(): the reference builds `BertTokenizer.from_pretrained("bert-base-uncased")` + "[DEC]" / "[ENC]"
    (vocab 30522 + 2 = 30524, med_config.json).  The vocabulary is not available offline: the same call is made with
    local_files_only, pointing at $UNIIR_BERT_VOCAB_DIR when set; without the files it raises (tokenizer parity is
    unpinned offline; feed token ids + attention masks);
  * get_blip_transform(): PIL + torch restatement of the eval transform (bicubic resize to a square, CLIP mean/std) and
    of the train transform's geometric part (RandomResizedCrop scale (min_scale, 1), bicubic; horizontal flip).
    RandomAugment (backbone/transform/randaugment.py) is a CPU data-augmentation policy outside the hot path and is not
    carried; plug the reference's transform into the collator to get it;
  * interpolate_pos_embed() / load_checkpoint(): checkpoint loading with the bicubic position-embedding resize.
"""
import math
import os
import random

import numpy as np
import torch

_MEAN = (0.48145466, 0.4578275, 0.40821073)
_STD = (0.26862954, 0.26130258, 0.27577711)


def init_tokenizer():
    from transformers import BertTokenizer
    src = os.environ.get("UNIIR_BERT_VOCAB_DIR", "bert-base-uncased")
    try:
        tokenizer = BertTokenizer.from_pretrained(src, local_files_only=True)
    except Exception as e:  # noqa: BLE001
        raise RuntimeError("bert-base-uncased vocabulary not found offline; set UNIIR_BERT_VOCAB_DIR to a directory "
                           "holding vocab.txt, or feed input_ids / attention_mask directly") from e
    tokenizer.add_special_tokens({"bos_token": "[DEC]"})
    tokenizer.add_tokens(["[ENC]"], special_tokens=True)      # == add_special_tokens({"additional_special_tokens": [...]}),
    tokenizer.enc_token_id = tokenizer.convert_tokens_to_ids("[ENC]")   # spelled so that transformers 4.x and 5.x both take it
    return tokenizer


def _to_tensor_normalized(image):
    a = torch.from_numpy(np.asarray(image.convert("RGB"), dtype=np.float32) / 255.0).permute(2, 0, 1)
    return (a - torch.tensor(_MEAN).view(3, 1, 1)) / torch.tensor(_STD).view(3, 1, 1)


def get_blip_transform(image_size, min_scale=0.5, is_train=True):
    from PIL import Image

    def eval_fn(image):
        return _to_tensor_normalized(image.resize((image_size, image_size), Image.BICUBIC))

    def train_fn(image):
        w, h = image.size
        area = w * h
        box = None
        for _ in range(10):     # torchvision RandomResizedCrop.get_params: scale (min_scale, 1), ratio (3/4, 4/3)
            target = area * random.uniform(min_scale, 1.0)
            ratio = math.exp(random.uniform(math.log(3 / 4), math.log(4 / 3)))
            cw, ch = int(round(math.sqrt(target * ratio))), int(round(math.sqrt(target / ratio)))
            if 0 < cw <= w and 0 < ch <= h:
                top, left = random.randint(0, h - ch), random.randint(0, w - cw)
                box = (left, top, left + cw, top + ch)
                break
        if box is None:         # fallback: central crop at the closest allowed ratio
            r = w / h
            cw, ch = (w, int(round(w / (3 / 4)))) if r < 3 / 4 else ((int(round(h * (4 / 3))), h) if r > 4 / 3 else (w, h))
            left, top = (w - cw) // 2, (h - ch) // 2
            box = (left, top, left + cw, top + ch)
        image = image.crop(box).resize((image_size, image_size), Image.BICUBIC)
        if random.random() < 0.5:
            image = image.transpose(Image.FLIP_LEFT_RIGHT)
        return _to_tensor_normalized(image)

    return train_fn if is_train else eval_fn


def interpolate_pos_embed(pos_embed_checkpoint, num_patches, num_extra_tokens=1):
    """[1, 1+g0*g0, D] -> [1, 1+g*g, D]: class token kept, grid resized bicubically (align_corners False)"""
    D = pos_embed_checkpoint.shape[-1]
    orig = int((pos_embed_checkpoint.shape[-2] - num_extra_tokens) ** 0.5)
    new = int(num_patches ** 0.5)
    if orig == new:
        return pos_embed_checkpoint
    extra = pos_embed_checkpoint[:, :num_extra_tokens]
    grid = pos_embed_checkpoint[:, num_extra_tokens:].reshape(-1, orig, orig, D).permute(0, 3, 1, 2)
    grid = torch.nn.functional.interpolate(grid.float(), size=(new, new), mode="bicubic", align_corners=False)
    print("reshape position embedding from %d to %d" % (orig ** 2, new ** 2))
    return torch.cat((extra, grid.permute(0, 2, 3, 1).flatten(1, 2).to(extra.dtype)), dim=1)


def load_checkpoint(model, filename):
    """backbone/blip.py:263-289 for local files (there is no network): resize pos_embed, drop shape mismatches,
    load non-strictly, return (model, missing/unexpected report)"""
    if not os.path.isfile(filename):
        raise RuntimeError("checkpoint url or path is invalid")
    from .host_utils import load_checkpoint_file
    state_dict = load_checkpoint_file(filename)["model"]
    own = model.state_dict()
    n_patches = own["visual_encoder.pos_embed"].shape[-2] - 1
    for key in ("visual_encoder.pos_embed", "visual_encoder_m.pos_embed"):
        if key in state_dict and key in own:
            state_dict[key] = interpolate_pos_embed(state_dict[key], n_patches)
    for key in list(state_dict.keys()):
        if key in own and state_dict[key].shape != own[key].shape:
            del state_dict[key]
    msg = model.load_state_dict(state_dict, strict=False)
    print("load checkpoint from %s" % filename)
    return model, msg
