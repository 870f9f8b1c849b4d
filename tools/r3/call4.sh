#!/bin/bash
# round-3 GPU call 4: retrieval (32-byte group-max stores, tail breakdown), LayerNorm nt loads A/B
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3c4
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_topk_gpu.py "tests/test_fullsize_gpu.py::test_topk_full_shard_properties" -x -q > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
tail -3 $O/pytest.log
for cfg in "new:" "stop1:UNIIR_TOPK_TAIL_STOP=1" "stop2:UNIIR_TOPK_TAIL_STOP=2"; do
  name=${cfg%%:*}; e1=${cfg#*:}
  env $e1 NQS=16,64,128 timeout 300 python tools/r3/topk_bench.py > $O/tb_$name.txt 2>&1
  echo "== $name"; grep topk $O/tb_$name.txt
done
echo "== ln default"; python tools/ln_bench.py > $O/ln_default.txt 2>&1; cat $O/ln_default.txt
echo "== ln nt"; UNIIR_HIP_LIB=$R/experiments/build/libuniir_lnnt.so python tools/ln_bench.py > $O/ln_nt.txt 2>&1; cat $O/ln_nt.txt
ARGS="--steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-retrieval"
python bench.py $ARGS > $O/bench_def_a.json 2> $O/bench_def_a.err
UNIIR_HIP_LIB=$R/experiments/build/libuniir_lnnt.so python bench.py $ARGS > $O/bench_lnnt_a.json 2> $O/bench_lnnt_a.err
python bench.py $ARGS > $O/bench_def_b.json 2> $O/bench_def_b.err
UNIIR_HIP_LIB=$R/experiments/build/libuniir_lnnt.so python bench.py $ARGS > $O/bench_lnnt_b.json 2> $O/bench_lnnt_b.err
for f in $O/bench_*.json; do echo $f; python -c "import json,sys; d=json.loads(open('$f').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['end_to_end_frac'])"; done
