"""Drop-in for UniIR src/models/uniir_blip/utils.py: the reference keeps a byte-identical copy of the CLIP utilities
there (SmoothedValue, MetricLogger, distributed helpers); here both module paths share one implementation."""
from models.uniir_clip.utils import *  # noqa: F401,F403
from models.uniir_clip.utils import (MetricLogger, SmoothedValue, get_rank, get_world_size,  # noqa: F401
                                     init_distributed_mode, is_dist_avail_and_initialized, is_main_process,
                                     setup_for_distributed)
