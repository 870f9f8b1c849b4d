#!/bin/bash
# the DACT-only kernel in the models: tower / BLIP / CLIP_FF / parity tests, then the headline
cd /root/repo
export PYTHONPATH=/root/repo
mkdir -p gpurun_out/r4
timeout 1200 python -m pytest tests/test_kernels_gpu.py tests/test_clip_model_gpu.py tests/test_blip_gpu.py tests/test_clipff_gpu.py tests/test_parity_exact_gpu.py tests/test_fullsize_gpu.py tests/test_fp32_parity_gpu.py -m gpu -x -q > gpurun_out/r4/dact8b_pytest.txt 2>&1
tail -3 gpurun_out/r4/dact8b_pytest.txt
timeout 400 python bench.py --no-secondary --no-retrieval --no-cpu-baseline --steps 8 --warmup 3 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('HEADLINE', d['value'], d['ms_per_step'], 'unpacked', d['value_unpacked'], d['ms_per_step_unpacked'], d['roofline']['achieved'], d['roofline']['board']['sclk_mhz_mean'], d['roofline']['board']['power_w_mean'], d['config']['peak_mem_GB'])"
