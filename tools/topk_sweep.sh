for wm in 1 2; do for bl in 512 1024 2048 4096; do echo "WM=$wm BLOCKS=$bl"; UNIIR_TOPK_WM=$wm UNIIR_TOPK_BLOCKS=$bl python tools/microbench.py 2>&1 | grep "topk nq=64"; done; done
