"""Group a tools/rocpd_summary.py kernel table into GEMM / attention / LayerNorm / rest, per step:
    python tools/kernel_groups.py gpurun_out/prof_r05/kernel_stats.txt 5"""
import re
import sys

path, steps = sys.argv[1], float(sys.argv[2])
groups = {"GEMMs (+ split-K reduce, small GEMMs)": 0.0, "attention": 0.0, "LayerNorm": 0.0, "AdamW + rest": 0.0}
per_kernel = []
for line in open(path):
    m = re.match(r"^(.{100}) +(\d+) +([\d.]+) +([\d.]+) +([\d.]+)\s*$", line.rstrip("\n"))
    if not m:
        continue
    name, calls, total_ms = m.group(1).strip(), int(m.group(2)), float(m.group(3))
    if "gemm" in name or "splitk_reduce" in name:
        g = "GEMMs (+ split-K reduce, small GEMMs)"
    elif "attn" in name:
        g = "attention"
    elif name.startswith("void ln_") or "ln_fwd" in name or "ln_bwd" in name:
        g = "LayerNorm"
    else:
        g = "AdamW + rest"
    groups[g] += total_ms
    per_kernel.append((name, calls, total_ms))
tot = sum(groups.values())
print(f"# per step (total / {steps:g}) from {path}")
for g, v in groups.items():
    print(f"{g:42s} {v / steps:9.2f} ms / step")
print(f"{'total kernel time':42s} {tot / steps:9.2f} ms / step")
