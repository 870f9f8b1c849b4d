"""Drop-in for UniIR src/models/uniir_blip/blip_scorefusion/blip_sf.py: `BLIPScoreFusion(med_config, image_size, vit,
vit_grad_ckpt, vit_ckpt_layer, embed_dim, queue_size, momentum, config)` and the factory `blip_sf(pretrained, ...)`.
The class lives in uniir_amd/blip_model.py; `med_config` given as the reference's relative path resolves against this tree."""
import os

from uniir_amd.blip_front import load_checkpoint
from uniir_amd.blip_model import BLIPScoreFusion as _Native

_BLIP_DIR = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class BLIPScoreFusion(_Native):
    def __init__(self, med_config="backbone/configs/med_config.json", **kwargs):
        if isinstance(med_config, str) and not os.path.isfile(med_config):
            med_config = {}      # built-in BERT-base defaults (uniir_amd.blip_model.MED_DEFAULT); an existing file is honoured
        super().__init__(med_config=med_config, **kwargs)


def blip_sf(pretrained="", **kwargs):
    model = BLIPScoreFusion(**kwargs)
    if pretrained:
        model, msg = load_checkpoint(model, pretrained)
        print("missing keys:")
        print(msg.missing_keys)
    return model
