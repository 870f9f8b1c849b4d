#!/bin/bash
# The evidence run of a round on the GPU box (one gpurun call, or several with STAGES): full GPU test suite, smoke(), the default
# bench line with its wall clock, rocprofv3 kernel stats + PMC passes of the step, the retrieval per-kernel table, the secondary
# profiles, the isolated-kernel record  ->  gpurun_out/<TAG>final/   (copy what is to be judged into profiles/<TAG>_*)
#   tools/evidence.sh [TAG=r05] [STAGES="tests bench profile topk secondary micro"]
# (replaces tools/r3/final*.sh and tools/r4/final.sh)
TAG=${1:-r05}
STAGES=${2:-"tests bench profile topk secondary micro"}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${TAG}final
mkdir -p $O
cd $R
export PYTHONPATH=$R
for S in $STAGES; do
  case $S in
    tests)
      timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log
      timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log ;;
    bench)
      T0=$(date +%s)
      timeout 1500 python bench.py ${BENCH_ARGS:-} > $O/bench_line.json 2> $O/bench.err; echo "bench rc=$? wall $(( $(date +%s) - T0 )) s"
      python tools/bench_summary.py $O/bench_line.json ;;
    profile)
      bash tools/profile_bench.sh $TAG > $O/profile_bench.out 2>&1; tail -14 $O/profile_bench.out | cut -c1-200 ;;
    topk)
      bash tools/topk_table.sh > /dev/null 2>&1; cp gpurun_out/topk_table.txt $O/ ; tail -30 $O/topk_table.txt | cut -c1-160
      rm -f gpurun_out/topk_pmc/topk_pmc.txt; bash tools/topk_pmc.sh > /dev/null 2>&1; cp gpurun_out/topk_pmc/topk_pmc.txt $O/ ; tail -12 $O/topk_pmc.txt | cut -c1-160 ;;
    secondary)
      SEC_LIST="embed blip clipff" bash tools/profile_secondary.sh > $O/profile_secondary.out 2>&1; grep -E "^\{" $O/profile_secondary.out | cut -c1-200 ;;
    micro)
      MB_ITEMS=1024 timeout 600 python tools/microbench.py > $O/microbench.txt 2>&1; tail -5 $O/microbench.txt
      timeout 300 python tools/r5/epi_forms.py > $O/epilogue_forms.txt 2>&1; cat $O/epilogue_forms.txt ;;
  esac
done
