"""print the numbers of a bench.py JSON line that the round's notes quote:  python tools/bench_summary.py gpurun_out/x/bench_line.json"""
import json
import sys

d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
r = d.get("roofline") or {}
print("HEADLINE", d["value"], "pairs/s", d["ms_per_step"], "ms | unpacked", d.get("value_unpacked"), d.get("ms_per_step_unpacked"),
      "| gemm", r.get("achieved"), r.get("frac"), "| e2e", r.get("end_to_end_frac"), r.get("end_to_end_frac_unpacked"),
      "| peak mem", d["config"].get("peak_mem_GB"))
print("board", r.get("board"))
for line in d["config"].get("mlp_stash_decisions") or []:
    print("stash:", line)
rt = d.get("retrieval")
if isinstance(rt, dict):
    for k, v in rt.items():
        if isinstance(v, dict) and "ms" in v:
            print(k, v["ms"], "ms hbm", v["hbm"]["frac"], "mfma", v["mfma"]["frac"])
    for kk in ("full_pool", "dim512"):
        for k, v in (rt.get(kk) or {}).items():
            if isinstance(v, dict) and "ms" in v:
                print(kk, k, v["ms"], "ms hbm", v["hbm"]["frac"], "mfma", v["mfma"]["frac"])
    print("cpu retrieval", (rt.get("cpu_baseline") or {}).get("value"))
em = d.get("embed")
if isinstance(em, dict) and "modes" in em:
    for k, v in em["modes"].items():
        print("embed", k, v["value"], "items/s", v["mfma_frac"], "(dense count", v.get("mfma_frac_dense_count"), ")")
for k in ("blip_ff_large", "clip_ff"):
    if isinstance(d.get(k), dict):
        print(k, d[k].get("value"), d[k].get("mfma_frac"), d[k].get("ms_per_step"))
if "cpu_baseline" in d:
    print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"].get("config1", {}).get("value"))
