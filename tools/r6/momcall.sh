cd $GRAFT_REPO_ROOT; export PYTHONPATH=$PWD; mkdir -p gpurun_out/r06e
timeout 900 python -m pytest tests/test_blip_gpu.py tests/test_bench_paths_gpu.py -m gpu -x -q 2>&1 | tail -12 > gpurun_out/r06e/pytest.log; cat gpurun_out/r06e/pytest.log
for O in 1 0 1 0; do
UNIIR_OVERLAP_MOMENTUM=$O timeout 600 python -c "
import json, torch, bench
r = bench.bench_blip_ff(torch.device('cuda:0'), steps=5, warmup=2)
print('OVERLAP_MOMENTUM=$O', r['value'], r['ms_per_step'], r['mfma_frac'], r['padded_rows'])
" 2>/dev/null | tail -1
done
