"""Drop-in module path for UniIR src/common/dist_utils.py; the implementation is shared: uniir_amd/host_utils.py."""
from uniir_amd.host_utils import (ContiguousDistributedSampler, get_rank, get_world_size, init_distributed_mode,  # noqa: F401
                                  is_dist_avail_and_initialized, is_main_process, setup_for_distributed)
