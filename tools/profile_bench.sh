#!/bin/bash
# rocprofv3 evidence for the bench line (GPU box): kernel-trace stats of the train step, then FETCH_SIZE / WRITE_SIZE PMC passes
# (separate passes, --kernel-trace only, as /opt/skills/guides/MI355X_MICROARCH.md prescribes) -> gpurun_out/prof_<TAG>/
#   tools/profile_bench.sh [TAG=r05] [stats|all]      stats = the kernel-trace pass only
TAG=${1:-r05}
WHAT=${2:-all}
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
ARGS="--no-cpu-baseline --no-secondary --no-retrieval --no-unpacked"
# per-kernel accounting: both towers on ONE stream (kernel durations add up to the step only when nothing overlaps; the two-stream
# order runs the same kernel binaries -- its effect on the step is an A/B, tools/r5/overlap_ab.sh).  UNIIR_OVERLAP_TOWERS=1 tools/profile_bench.sh
# profiles the default order instead.
export UNIIR_OVERLAP_TOWERS=${UNIIR_OVERLAP_TOWERS:-0}
rm -rf /tmp/pb_stats /tmp/pb_fetch /tmp/pb_write
rocprofv3 --kernel-trace --stats -d /tmp/pb_stats -o s -- python $R/bench.py --steps 4 --warmup 1 $ARGS > $O/bench_under_rocprof.json 2> $O/stats.err
DB=$(find /tmp/pb_stats -name "*_results.db" | head -1)
python $R/tools/rocpd_summary.py $DB > $O/kernel_stats.txt
python $R/tools/kernel_groups.py $O/kernel_stats.txt 5 > $O/step_breakdown.txt 2>&1 || true
if [ "$WHAT" = "stats" ]; then head -34 $O/kernel_stats.txt; cat $O/step_breakdown.txt; tail -2 $O/bench_under_rocprof.json | cut -c1-400; exit 0; fi
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d /tmp/pb_fetch -o f -- python $R/bench.py --steps 1 --warmup 1 $ARGS > /dev/null 2> $O/fetch.err
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d /tmp/pb_write -o w -- python $R/bench.py --steps 1 --warmup 1 $ARGS > /dev/null 2> $O/write.err
F=$(find /tmp/pb_fetch -name "*counter_collection.csv" | head -1)
W=$(find /tmp/pb_write -name "*counter_collection.csv" | head -1)
python $R/tools/pmc_summary.py $F > $O/pmc_fetch.txt
python $R/tools/pmc_summary.py $W > $O/pmc_write.txt
cd $R && python tools/pmc_gemm_traffic.py $F $W ViT-L/14 512 "profiles/${TAG}_pmc_step.txt" > $O/pmc_gemm_traffic.out
cp $R/profiles/pmc_gemm_traffic.json $O/
head -30 $O/kernel_stats.txt; cat $O/pmc_gemm_traffic.out; tail -2 $O/bench_under_rocprof.json | cut -c1-600
