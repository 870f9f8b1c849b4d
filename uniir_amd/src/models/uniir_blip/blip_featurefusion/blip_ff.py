"""Drop-in for UniIR src/models/uniir_blip/blip_featurefusion/blip_ff.py: `BLIPFeatureFusion(med_config, image_size,
vit, vit_grad_ckpt, vit_ckpt_layer, embed_dim, queue_size, momentum, config)` and the factory `blip_ff(pretrained, ...)`.
The class lives in uniir_amd/blip_model.py (launch sequences over libuniir_hip.so); the reference's relative
`med_config` path falls back to the built-in MED defaults."""
import os

from uniir_amd.blip_front import load_checkpoint
from uniir_amd.blip_model import BLIPFeatureFusion as _Native

_BLIP_DIR = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class BLIPFeatureFusion(_Native):
    def __init__(self, med_config="backbone/configs/med_config.json", **kwargs):
        if isinstance(med_config, str) and not os.path.isfile(med_config):
            med_config = {}      # the reference's relative "backbone/configs/med_config.json": BERT-base + 2 BLIP tokens, the
            # values are the built-in defaults (uniir_amd.blip_model.MED_DEFAULT); an existing JSON file is still honoured
        super().__init__(med_config=med_config, **kwargs)


def blip_ff(pretrained="", **kwargs):
    model = BLIPFeatureFusion(**kwargs)
    if pretrained:
        model, msg = load_checkpoint(model, pretrained)
        print("missing keys:")
        print(msg.missing_keys)
    return model
