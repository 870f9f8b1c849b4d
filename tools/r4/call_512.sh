#!/bin/bash
# dim-512 pools on the templated streaming scan: oracle tests, the 700 k x 512 property test, the other top-k tests, bench block
cd /root/repo
export PYTHONPATH=/root/repo
mkdir -p gpurun_out/r4
timeout 1200 python -m pytest tests/test_topk_gpu.py tests/test_fullsize_gpu.py tests/test_bench_paths_gpu.py -m gpu -x -q -k "topk or search or retriev or pool or bench" > gpurun_out/r4/d512_pytest.txt 2>&1
tail -4 gpurun_out/r4/d512_pytest.txt
timeout 300 python - <<'PY' 2>&1 | grep -v amdgpu.ids
import json, torch, bench
dev = torch.device("cuda:0")
r = bench.bench_retrieval(dev, 700_000, 512, 10, False)
print({k: (v["ms"], v["hbm"]["frac"], v["mfma"]["frac"]) for k, v in r.items() if isinstance(v, dict)})
r = bench.bench_retrieval(dev, 700_000, 768, 10, False)
print({k: (v["ms"], v["hbm"]["frac"], v["mfma"]["frac"]) for k, v in r.items() if isinstance(v, dict)})
PY
