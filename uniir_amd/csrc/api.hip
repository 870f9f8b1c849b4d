// Error strings / ABI version of libuniir_hip.so (include/uniir_hip.h).
#include "../../include/uniir_hip.h"

extern "C" const char* uniir_strerror(int code) {
    switch (code) {
        case UNIIR_OK: return "ok";
        case UNIIR_EINVAL: return "invalid argument (null pointer, negative size or bad enum)";
        case UNIIR_ESHAPE: return "shape not supported by the gfx950 kernels";
        case UNIIR_EALIGN: return "pointer or leading dimension is not 16-byte aligned";
        case UNIIR_ELAUNCH: return "HIP kernel launch failed";
        case UNIIR_EUNSUPPORTED: return "unsupported configuration";
        default: return "unknown uniir error code";
    }
}
// 2 (round 6): uniir_clip_tower grew pool_last_block; uniir_gemm_timing_read_ex returns the fallback flag; new entry points
// uniir_reduce_scratch, uniir_attention_{fwd,bwd}_rows, uniir_dropout_{f32,bf16}_rows, uniir_gemm_timing_filter
extern "C" int uniir_abi_version(void) { return 2; }
