#!/bin/bash
# round-4 evidence run (GPU box): full GPU test suite, smoke(), the default bench line (timed by the wall clock too), rocprofv3 kernel
# stats + PMC passes of the step, the retrieval per-kernel table, the secondary profiles, the isolated-kernel record -> gpurun_out/r4final/
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4final
mkdir -p $O
cd $R
export PYTHONPATH=$R
if [ "${SKIP_TESTS:-0}" != 1 ]; then
  timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
fi
T0=$(date +%s)
timeout 1200 python bench.py > $O/bench_line.json 2> $O/bench.err; echo "bench rc=$? wall $(( $(date +%s) - T0 )) s"
bash tools/profile_bench.sh > $O/profile_bench.out 2>&1; tail -12 $O/profile_bench.out | cut -c1-200
bash tools/topk_table.sh > /dev/null 2>&1; cp gpurun_out/topk_table.txt $O/
SEC_LIST="embed blip clipff" bash tools/profile_secondary.sh > $O/profile_secondary.out 2>&1; grep -E "^\{" $O/profile_secondary.out | cut -c1-200
MB_ITEMS=1024 timeout 600 python tools/microbench.py > $O/microbench.txt 2>&1; tail -5 $O/microbench.txt
python - <<'PY'
import json
d=json.loads([l for l in open('/root/repo/gpurun_out/r4final/bench_line.json') if l.startswith('{')][-1])
r=d['roofline']
print('HEADLINE', d['value'], d['ms_per_step'], 'unpacked', d.get('value_unpacked'), d.get('ms_per_step_unpacked'), 'gemm', r['achieved'], r['frac'], 'e2e', r['end_to_end_frac'], r.get('end_to_end_frac_unpacked'))
print('board', r.get('board'))
for k,v in d['retrieval'].items():
    if isinstance(v,dict) and 'ms' in v: print(k, v['ms'], v['hbm']['frac'], v['mfma']['frac'])
for kk in ('full_pool','dim512'):
    for k,v in d['retrieval'].get(kk,{}).items():
        if isinstance(v,dict) and 'ms' in v: print(kk,k, v['ms'], v['hbm']['frac'], v['mfma']['frac'])
for k in ('embed','blip_ff_large','clip_ff'): print(k, d[k].get('value'), d[k].get('mfma_frac'))
print('cpu', d['cpu_baseline']['value'], d['retrieval'].get('cpu_baseline',{}).get('value'))
PY
