cd $GRAFT_REPO_ROOT; export PYTHONPATH=$PWD; mkdir -p gpurun_out/r06c
timeout 1200 python -m pytest tests/test_clip_model_gpu.py tests/test_fullsize_gpu.py tests/test_fp32_parity_gpu.py tests/test_bench_paths_gpu.py -m gpu -x -q 2>&1 | tail -25 > gpurun_out/r06c/pytest.log; cat gpurun_out/r06c/pytest.log
timeout 600 python bench.py --no-secondary --no-retrieval --no-cpu-baseline --steps 8 --warmup 3 2>gpurun_out/r06c/bench.err > gpurun_out/r06c/line.json; python tools/bench_summary.py gpurun_out/r06c/line.json | head -8
timeout 600 python bench.py --no-secondary --no-retrieval --no-cpu-baseline --no-unpacked --no-pool-last --steps 8 --warmup 3 2>>gpurun_out/r06c/bench.err > gpurun_out/r06c/line_nopool.json; python tools/bench_summary.py gpurun_out/r06c/line_nopool.json | head -3
