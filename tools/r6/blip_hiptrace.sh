#!/bin/bash
# which HIP runtime calls stand behind BLIP_FF's ~500 __amd_rocclr_copyBuffer kernels per train step?
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
export PYTHONPATH=$R
rm -rf /tmp/bh
rocprofv3 --hip-trace --kernel-trace --stats --output-format csv -d /tmp/bh -o t -- python $R/tools/bench_blip.py --steps 2 --warmup 1 --pairs 64 > /tmp/bh.log 2>&1
ls /tmp/bh/*/ 2>/dev/null | head
F=$(find /tmp/bh -name "*hip_api_stats.csv" | head -1); echo $F; head -25 $F | cut -c1-160
T=$(find /tmp/bh -name "*hip_api_trace.csv" | head -1)
python - "$T" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
c = collections.Counter(r["Function"] for r in rows)
print([(k, v) for k, v in c.most_common(15)])
PY
K=$(find /tmp/bh -name "*kernel_trace.csv" | head -1)
python - "$T" "$K" <<'PY'
import csv, sys, collections
api = list(csv.DictReader(open(sys.argv[1])))
ker = {r["Correlation_Id"]: r["Kernel_Name"].split("(")[0][:50] for r in csv.DictReader(open(sys.argv[2]))}
api.sort(key=lambda r: int(r["Start_Timestamp"]))
tid_main = collections.Counter(r["Thread_Id"] for r in api if r["Function"] == "hipLaunchKernel").most_common(1)[0][0]
print("threads:", collections.Counter((r["Thread_Id"], r["Function"]) for r in api if r["Function"] in ("hipLaunchKernel", "hipMemcpyWithStream", "hipMemcpyAsync")).most_common(8))
seq = [r for r in api if r["Function"] in ("hipLaunchKernel", "hipMemcpyWithStream", "hipMemcpyAsync")]
nxt = collections.Counter()
prev = collections.Counter()
last_k = "?"
pending = []
for r in seq:
    if r["Function"] == "hipLaunchKernel":
        k = ker.get(r["Correlation_Id"], "?")
        for f in pending:
            nxt[(f, k)] += 1
        pending = []
        last_k = k
    else:
        pending.append(r["Function"])
        prev[(r["Function"], last_k)] += 1
print("memcpy -> next kernel:")
for (f, k), n in nxt.most_common(14):
    print(f"{n:5d} {f:22s} {k}")
print("previous kernel -> memcpy:")
for (f, k), n in prev.most_common(10):
    print(f"{n:5d} {f:22s} {k}")
PY
