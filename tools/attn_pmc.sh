#!/bin/bash
# PMC anatomy of the attention kernels (T = 257, 16 heads, MB_ITEMS items): where do the wave cycles go?
# usage (GPU box): bash tools/attn_pmc.sh   -> gpurun_out/attn_pmc.txt
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/attn_pmc${MB_TOKENS:+_$MB_TOKENS}.txt
: > $OUT
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"
P2="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_TRANS_F32"
P3="SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_WAVES GRBM_GUI_ACTIVE SQ_INST_CYCLES_VALU SQ_THREAD_CYCLES_VALU SQ_IFETCH"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1)); rm -rf /tmp/apmc_$i
  MB_ONLY=attn MB_ITEMS=${MB_ITEMS:-256} rocprofv3 --kernel-trace --output-format csv --pmc $P -d /tmp/apmc_$i -o p -- python $R/tools/microbench.py > /tmp/apmc_$i.log 2>&1 || tail -5 /tmp/apmc_$i.log
  CSV=$(find /tmp/apmc_$i -name "*counter_collection.csv" | head -1)
  python - "$CSV" >> $OUT <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    k = r["Kernel_Name"].split("(")[0][:48]
    if "attn" not in k: continue
    a = agg[(k, r["Counter_Name"])]
    a[0] += 1; a[1] += float(r["Counter_Value"])
for (k, c), (n, v) in sorted(agg.items()):
    print(f"{k:48s} {c:28s} launches {n:4d} per-launch {v/n:16.1f}")
PY
done
cat $OUT
