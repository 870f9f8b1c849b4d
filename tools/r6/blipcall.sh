cd $GRAFT_REPO_ROOT; export PYTHONPATH=$PWD; mkdir -p gpurun_out/r06b
timeout 900 python -m pytest tests/test_blip_gpu.py tests/test_kernels_gpu.py -k "blip or packed or attention or dropout" -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r06b/pytest.log; cat gpurun_out/r06b/pytest.log
timeout 600 python -c "
import json, torch, bench
print(json.dumps(bench.bench_blip_ff(torch.device('cuda:0'), steps=4, warmup=2)))
" > gpurun_out/r06b/blip.json 2> gpurun_out/r06b/blip.err; tail -3 gpurun_out/r06b/blip.err; cat gpurun_out/r06b/blip.json
