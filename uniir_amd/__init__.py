"""uniir_amd -- the MI355X (gfx950) hot path of UniIR: CLIP_SF contrastive train step, embedding extraction and
brute-force top-k retrieval as hand-written HIP kernels behind a C ABI (include/uniir_hip.h)."""
__version__ = "0.1.0"
