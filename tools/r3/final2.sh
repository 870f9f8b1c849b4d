#!/bin/bash
# refresh of the round-3 bench line and the retrieval table after the last kernel changes
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3final2
mkdir -p $O
cd $R
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench.err; echo "bench rc=$?"
true
python - <<'PY'
import json
d=json.loads(open('/root/repo/gpurun_out/r3final2/bench_line.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['end_to_end_frac'], d['roofline']['board'])
for k,v in d['retrieval'].items():
    if isinstance(v,dict) and 'ms' in v: print(k, v['ms'], v['hbm']['frac'], v['mfma']['frac'])
for k in ('embed','blip_ff_large','clip_ff'): print(k, d[k].get('value'), d[k].get('mfma_frac'))
PY
