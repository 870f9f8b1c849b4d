#!/usr/bin/env python
"""After the closing GPU call (tools/gpu_final.sh r06final): copy its records into profiles/ and put the driver-shaped numbers into the
R6_* placeholders of DESIGN.md / README.md.   python tools/r6/fill_docs.py [gpurun_out/r06final]"""
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "r06final")
line = [l for l in open(os.path.join(src, "bench_line.json")).read().splitlines() if l.startswith("{")][-1]
d = json.loads(line)
r = d["roofline"]
log = open(os.path.join(src, "pytest_gpu.log")).read()
m = re.search(r"(\d+) passed", log)
ngpu = m.group(1) if m else "?"
shutil.copy(os.path.join(src, "bench_line.json"), os.path.join(ROOT, "profiles", "r06_bench_line.json"))
open(os.path.join(ROOT, "profiles", "r06_gputest_log.txt"), "w").write(
    "# tools/gpu_final.sh r06final: full `pytest tests -x -q -m gpu`, then smoke(), on the snapshot of the commit named below\n" + log
    + "\n# smoke\n" + open(os.path.join(src, "smoke.log")).read().splitlines()[-1] + "\n")
dense = d["value"] * 1.052e12 / 2.5e15
sub = {
    "R6_MS": f"{d['ms_per_step']:.1f}", "R6_VALUE": f"{d['value']:.1f}", "R6_E2E": f"{r['end_to_end_frac']:.3f}",
    "R6_GEMM": f"{r['achieved']:.0f}", "R6_FRAC": f"{r['frac']:.3f}", "R6_UMS": f"{d['ms_per_step_unpacked']:.1f}",
    "R6_UVALUE": f"{d['value_unpacked']:.1f}", "R6_UE2E": f"{r['end_to_end_frac_unpacked']:.3f}", "R6_DENSE": f"{dense:.3f}",
    "R6_NGPU": ngpu,
}
for name in ("DESIGN.md", "README.md"):
    p = os.path.join(ROOT, name)
    s = open(p).read()
    for k in sorted(sub, key=len, reverse=True):
        s = s.replace(k, sub[k])
    open(p, "w").write(s)
print(sub)
