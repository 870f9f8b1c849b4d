cd $GRAFT_REPO_ROOT; export PYTHONPATH=$PWD; mkdir -p gpurun_out/r06c
timeout 1200 python -m pytest tests/test_clip_model_gpu.py tests/test_bench_paths_gpu.py tests/test_pipeline_gpu.py tests/test_dist_device_gpu.py -m gpu -x -q 2>&1 | tail -25 > gpurun_out/r06c/pytest2.log; cat gpurun_out/r06c/pytest2.log
