cd $GRAFT_REPO_ROOT; export PYTHONPATH=$PWD
for F in 1 0 1 0; do python -c "
import torch, bench
from uniir_amd import blip_model
blip_model.FUSE_HIDDEN_DROPOUT = bool($F)
r = bench.bench_blip_ff(torch.device('cuda:0'), steps=6, warmup=2)
print('FUSED=$F', r['value'], r['ms_per_step'], r['mfma_frac'])
" 2>/dev/null | tail -1; done
