#!/bin/bash
# weight gradients with a reduction length that is not a multiple of 64 (the packed text tower): tests, the fused-epilogue microbench, headline
cd /root/repo
export PYTHONPATH=/root/repo
mkdir -p gpurun_out/r4
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_clip_model_gpu.py tests/test_parity_exact_gpu.py -m gpu -x -q > gpurun_out/r4/wg_pytest.txt 2>&1
tail -3 gpurun_out/r4/wg_pytest.txt
MB_ITEMS=1024 timeout 300 python tools/microbench.py 2>&1 | grep -E "^gemm" > gpurun_out/r4/wg_microbench.txt; cat gpurun_out/r4/wg_microbench.txt
timeout 600 python bench.py --no-secondary --no-retrieval --no-cpu-baseline > gpurun_out/r4/wg_bench.txt 2>gpurun_out/r4/wg_bench.err
python - <<'PY'
import json
d=json.loads([l for l in open('/root/repo/gpurun_out/r4/wg_bench.txt') if l.startswith('{')][-1])
print('HEADLINE', d['value'], d['ms_per_step'], 'unpacked', d['value_unpacked'], d['ms_per_step_unpacked'], d['roofline']['achieved'], d['roofline']['board']['sclk_mhz_mean'], d['roofline']['board']['power_w_mean'])
PY
