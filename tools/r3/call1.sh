#!/bin/bash
# round-3 GPU call 1: full GPU test suite + two-stream tower A/B on the headline step
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3c1
mkdir -p $O
cd $R
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
ARGS="--steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-retrieval"
UNIIR_TOWER_STREAMS=0 python bench.py $ARGS > $O/bench_streams0.json 2> $O/bench_streams0.err
UNIIR_TOWER_STREAMS=1 python bench.py $ARGS > $O/bench_streams1.json 2> $O/bench_streams1.err
UNIIR_TOWER_STREAMS=0 python bench.py $ARGS > $O/bench_streams0b.json 2> $O/bench_streams0b.err
UNIIR_TOWER_STREAMS=1 python bench.py $ARGS > $O/bench_streams1b.json 2> $O/bench_streams1b.err
for f in $O/bench_streams*.json; do echo $f; python -c "import json,sys; d=json.loads(open('$f').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['end_to_end_frac'])"; done
