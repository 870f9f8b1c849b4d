"""Drop-in for the names blip_ff.py / train.py import from UniIR src/models/uniir_blip/backbone/blip.py
(init_tokenizer :221-226, create_vit :229-255, load_checkpoint :263-289).  create_vit returns the ViT geometry only:
on the MI355X path the ViT is a launch sequence over a flat parameter store (uniir_amd/blip_model.py), not a module."""
from uniir_amd.blip_front import init_tokenizer, load_checkpoint  # noqa: F401
from uniir_amd.blip_model import VIT_CONFIGS


def create_vit(vit, image_size, use_grad_checkpointing=False, ckpt_layer=0, drop_path_rate=0):
    assert vit in ["base", "large"], "vit parameter must be base or large"
    cfg = dict(VIT_CONFIGS[vit], img_size=image_size)
    return cfg, cfg["embed_dim"]
