"""Drop-in for UniIR src/models/uniir_blip/backbone/transform/blip_transform.py (get_blip_transform)."""
from uniir_amd.blip_front import get_blip_transform  # noqa: F401
