cd $GRAFT_REPO_ROOT; export PYTHONPATH=$PWD; mkdir -p gpurun_out/r06d
timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_clip_model_gpu.py tests/test_blip_gpu.py tests/test_clipff_gpu.py tests/test_parity_exact_gpu.py tests/test_fullsize_gpu.py -m gpu -x -q 2>&1 | tail -30 > gpurun_out/r06d/pytest.log; cat gpurun_out/r06d/pytest.log
for D in 1 0 1 0; do
UNIIR_DETERMINISTIC=$D timeout 600 python bench.py --no-secondary --no-retrieval --no-cpu-baseline --no-unpacked --steps 10 --warmup 3 2>>gpurun_out/r06d/bench.err > gpurun_out/r06d/line_det$D.json; echo "DET=$D"; python tools/bench_summary.py gpurun_out/r06d/line_det$D.json | head -1
done
