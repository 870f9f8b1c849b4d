#!/bin/bash
# bulk search with one tail per batch of sweeps + the topk.hip / topk_tail.hip split: retrieval tests, then the retrieval block
cd /root/repo
export PYTHONPATH=/root/repo
mkdir -p gpurun_out/r4
timeout 1200 python -m pytest tests/test_topk_gpu.py tests/test_fullsize_gpu.py tests/test_bench_paths_gpu.py tests/test_pipeline_gpu.py -m gpu -x -q > gpurun_out/r4/tail_pytest.txt 2>&1
tail -4 gpurun_out/r4/tail_pytest.txt
timeout 600 python bench.py --steps 2 --warmup 1 --no-secondary --no-cpu-baseline --no-unpacked > gpurun_out/r4/tail_bench.txt 2>gpurun_out/r4/tail_bench.err
python - <<'PY'
import json
d=json.loads([l for l in open('/root/repo/gpurun_out/r4/tail_bench.txt') if l.startswith('{')][-1])
r=d['retrieval']
for k,v in r.items():
    if isinstance(v,dict) and 'ms' in v: print(k, v['ms'], v['hbm']['frac'], v['mfma']['frac'])
for k,v in r.get('full_pool',{}).items():
    if isinstance(v,dict) and 'ms' in v: print('full_pool',k, v['ms'], v['hbm']['frac'], v['mfma']['frac'])
PY
tail -3 gpurun_out/r4/tail_bench.err
