#!/bin/bash
# act(f) kept per layer (uniir_clip_tower.stash_act) vs re-materialised in the backward: tests, then the headline A / B / A / B on one box
cd /root/repo
export PYTHONPATH=/root/repo
mkdir -p gpurun_out/r4
timeout 1200 python -m pytest tests/test_kernels_gpu.py tests/test_clip_model_gpu.py tests/test_parity_exact_gpu.py tests/test_fullsize_gpu.py tests/test_dist_device_gpu.py -m gpu -x -q > gpurun_out/r4/stash_pytest.txt 2>&1
tail -3 gpurun_out/r4/stash_pytest.txt
for i in 1 2; do
  for f in "" "--no-stash-act"; do
    timeout 400 python bench.py --no-secondary --no-retrieval --no-cpu-baseline --no-unpacked --steps 8 --warmup 3 $f 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$f' or 'stash', d['value'], d['ms_per_step'], d['roofline']['achieved'], d['config']['mlp_stash'], d['config']['peak_mem_GB'])"
  done
done
