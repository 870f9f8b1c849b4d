"""same-box A/B of the shard search between two builds of the library (UNIIR_HIP_LIB is read at import: one process per build):
    python tools/r5/topk_ab.py   -> runs itself twice per build, interleaved"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "uniir_amd", "src"))
    import torch, bench
    r = bench.bench_retrieval(torch.device("cuda", 0))
    print(json.dumps({k: v["ms"] for k, v in r.items() if k.startswith("q")}))
else:
    libs = {"new": os.path.join(ROOT, "uniir_amd", "libuniir_hip.so"), "old": os.path.join(ROOT, "uniir_amd", "libuniir_var_old.so")}
    for rnd in range(2):
        for name, lib in libs.items():
            env = dict(os.environ, UNIIR_HIP_LIB=lib)
            out = subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True).stdout.strip().splitlines()
            print(name, out[-1] if out else "?", flush=True)
