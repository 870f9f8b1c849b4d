#!/bin/bash
# round-3 GPU call 3: retrieval after the query staging / gather-by-DMA / sort changes: parity, timing, per-kernel table
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3c3
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_topk_gpu.py tests/test_bench_paths_gpu.py::test_search_shard_multi_sweep_loop_equals_the_oracle "tests/test_fullsize_gpu.py::test_topk_full_shard_properties" tests/test_parity_exact_gpu.py tests/test_pipeline_gpu.py -x -q > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log
for cfg in "new::" "nont:UNIIR_TOPK_NT=0:"; do
  name=${cfg%%:*}; rest=${cfg#*:}; e1=${rest%%:*}; e2=${rest#*:}
  env $e1 $e2 NQS=16,64,128,256,1024 timeout 300 python tools/r3/topk_bench.py > $O/tb_$name.txt 2>&1
  echo "== $name"; grep topk $O/tb_$name.txt
done
cd $R
timeout 400 python -m pytest tests/test_kernels_gpu.py -x -q > $O/pytest_kernels.log 2>&1; echo "kernels rc=$?"; tail -2 $O/pytest_kernels.log
ARGS="--steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-retrieval"
for i in a b; do
  UNIIR_GEMM_REMAINDER=0 python bench.py $ARGS > $O/bench_rem0$i.json 2> $O/bench_rem0$i.err
  python bench.py $ARGS > $O/bench_rem1$i.json 2> $O/bench_rem1$i.err
done
for f in $O/bench_rem*.json; do echo $f; python -c "import json,sys; d=json.loads(open('$f').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['end_to_end_frac'])"; done
cd /tmp && export TMPDIR=/tmp
: > $O/topk_table.txt
for NQ in 16 64 128 256 1024; do
  rm -rf /tmp/tkp_$NQ
  NQ=$NQ timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/tkp_$NQ -o t -- python $R/tools/topk_prof.py > /dev/null 2>&1
  DB=$(find /tmp/tkp_$NQ -name "*_results.db" | head -1)
  echo "## nq = $NQ (4 searches; pool 700000 x 768 fp16 = 1.075 GB per sweep)" >> $O/topk_table.txt
  python $R/tools/rocpd_summary.py $DB | grep -v "randn\|distribution\|copyBuffer\|vectorized\|^# rocprofv3\|inv_norm_kernel\|elementwise_kernel" | head -10 >> $O/topk_table.txt
done
cat $O/topk_table.txt
