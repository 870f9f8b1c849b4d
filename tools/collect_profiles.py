"""Assemble the judged evidence files profiles/<TAG>_* from what tools/evidence.sh left under gpurun_out/ (run in the build container
after the GPU call):   python tools/collect_profiles.py r05 [bench-line dir, default gpurun_out/r05final]"""
import json
import os
import re
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r05"
fin = os.path.join(ROOT, "gpurun_out", f"{tag}final")
prof = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
sec = os.path.join(ROOT, "gpurun_out", "prof_sec")
P = os.path.join(ROOT, "profiles")


def rd(path):
    return open(path).read() if os.path.exists(path) else ""


def line_of(path):
    ls = [l for l in rd(path).splitlines() if l.startswith("{")]
    return json.loads(ls[-1]) if ls else None


R04 = {"GEMMs (+ split-K reduce, small GEMMs)": 447.32, "attention": 57.25, "LayerNorm": 56.27, "AdamW + rest": 6.44, "total kernel time": 567.28}
# round 5's accounting (profiles/r05_bench_kernel_stats.txt): what round 6 is compared with
R05 = {"GEMMs (+ split-K reduce, small GEMMs)": 425.3, "attention": 57.9, "LayerNorm": 54.0, "AdamW + rest": 6.4, "total kernel time": 543.6}
PREV, PREV_TAG = (R05, "r05") if tag >= "r06" else (R04, "r04")

# ---- kernel stats of the step, with the accounting table
ks = rd(os.path.join(prof, "kernel_stats.txt"))
if ks:
    groups = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "kernel_groups.py"), os.path.join(prof, "kernel_stats.txt"), "5"],
                            capture_output=True, text=True).stdout
    under = line_of(os.path.join(prof, "bench_under_rocprof.json"))
    head = [f"# rocprofv3 --kernel-trace --stats -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-secondary --no-retrieval --no-unpacked",
            f"# (5 train steps of CLIP_SF ViT-L/14, 512 pairs, packed text rows, act(f) kept, rotating caption batches, both towers on ONE stream so that kernel durations add up;",
            f"#  1 x MI355X, build of round {tag[1:]}" + (": the last block of each tower on its pooled rows, reproducible reductions (partial_reduce_kernel, tokd_*)" if tag >= "r06" else "") + f"; tools/profile_bench.sh {tag})."]
    if under:
        b = under["roofline"].get("board") or {}
        head.append(f"# Under the profiler the step ran at {under['ms_per_step']} ms ({under['value']} pairs/s), board {b.get('sclk_mhz_mean')} MHz mean at "
                    f"{b.get('power_w_mean')} W of {b.get('power_cap_w')} W; sampled GEMM rate {under['roofline'].get('achieved')} TFLOP/s.")
    head.append(f"# Accounting, ms of kernel time per step (this file / profiles/{PREV_TAG}_bench_kernel_stats.txt, a box of the same clock class):")
    for ln in groups.splitlines():
        m = re.match(r"^(.{42}) +([\d.]+) ms / step", ln)
        if m:
            k = m.group(1).strip()
            head.append(f"#   {k:42s} {float(m.group(2)):8.2f}   ({PREV_TAG} {PREV.get(k, float('nan')):7.2f})")
    if tag >= "r06":
        head.append("# VERDICT r05 targets: step <= 528 ms (un-profiled, two streams: profiles/r06_bench_line.json), attention <= 50 ms/step.  What moved: the last block of each")
        head.append("#   tower on its pooled rows (-1/24 of the image tower's GEMM / attention / ln_2 time, + small [1024-row] launches), section 2 of DESIGN.md.")
    else:
        head.append("# VERDICT r04 targets: GEMM families <= 415 ms/step, attention <= 48 (experiments/attention_pair/README.md).")
    open(os.path.join(P, f"{tag}_bench_kernel_stats.txt"), "w").write("\n".join(head) + "\n" + ks)
    if under:
        json.dump(under, open(os.path.join(P, f"{tag}_bench_line_profiled_box.json"), "w"))

# ---- PMC traffic of the step
pf, pw = rd(os.path.join(prof, "pmc_fetch.txt")), rd(os.path.join(prof, "pmc_write.txt"))
if pf and pw:
    tr = line_of(os.path.join(prof, "pmc_gemm_traffic.out")) or {}
    head = ["# rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of",
            "#   python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary --no-retrieval --no-unpacked   (2 train steps, CLIP_SF ViT-L/14, 512 pairs)",
            "# Counter unit: KB per dispatch, summed per kernel family (tools/pmc_summary.py).  gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE reports 1/2 of the",
            "# bytes of wide (16 B/lane) reads -> doubled in tools/pmc_gemm_traffic.py; WRITE_SIZE as is.  L2<->fabric requests: MALL hits included (upper bound of HBM bytes).",
            f"# GEMM mean per launch: {json.dumps({k: tr.get(k) for k in ('bytes_per_launch', 'fetch_bytes_per_launch_corrected', 'write_bytes_per_launch', 'launches')})} -> profiles/pmc_gemm_traffic.json (bench.py roofline.traffic)"]
    open(os.path.join(P, f"{tag}_pmc_step.txt"), "w").write("\n".join(head) + "\n## FETCH_SIZE\n" + pf + "## WRITE_SIZE\n" + pw)
    if os.path.exists(os.path.join(prof, "pmc_gemm_traffic.json")):
        shutil.copy(os.path.join(prof, "pmc_gemm_traffic.json"), os.path.join(P, "pmc_gemm_traffic.json"))

# ---- retrieval
tt = rd(os.path.join(fin, "topk_table.txt"))
if tt:
    open(os.path.join(P, f"{tag}_topk_kernel_table.txt"), "w").write(
        "# rocprofv3 --kernel-trace --stats of tools/topk_prof.py (tools/topk_table.sh): one 700 000 x 768 fp16 shard, k = 10, 12 searches per query count;\n"
        "# then the whole 5.6 M x 768 pool as ONE resident shard (uniir_topk_ip_multi: 8 scans + one batched tail + one sort + one merge per sweep), 6 searches.\n" + tt)
tp = rd(os.path.join(fin, "topk_pmc.txt"))
if tp:
    open(os.path.join(P, f"{tag}_topk_pmc.txt"), "w").write(
        "# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, kernel-trace only) of tools/topk_prof.py, 8 searches of a 700 000 x 768 fp16 shard (tools/topk_pmc.sh).\n"
        "# Counter unit KB per dispatch; FETCH_SIZE x 2 = bytes (gfx950 correction for 16-B-per-lane reads).\n" + tp)

    def per_launch(kernel, counter, nq):
        sect = tp.split(f"## nq = {nq}  {counter}")[1].split("## nq")[0]
        for ln in sect.splitlines():
            if kernel in ln:
                return float(ln.split()[-1])
        return None
    f64, w64 = per_launch("topk_stream2_kernel", "FETCH_SIZE", 64), per_launch("topk_stream2_kernel", "WRITE_SIZE", 64)
    f256, w256 = per_launch("topk_stream5_kernel", "FETCH_SIZE", 256), per_launch("topk_stream5_kernel", "WRITE_SIZE", 256)
    if f64 and w64 and f256 and w256:
        note = (f"cited from profiles/{tag}_topk_pmc.txt (rocprofv3 --pmc FETCH_SIZE x2 / WRITE_SIZE of the scan kernels on a 700000 x 768 shard, this round's "
                f"build): {2 * f64 * 1024 / 1e9:.3f} GB fetched per sweep of the 1.075-GB shard at 64 queries (read once), {w64 * 1024 / 1e6:.1f} MB written; "
                f"{2 * f256 * 1024 / 1e9:.3f} GB fetched / {w256 * 1024 / 1e6:.1f} MB written at 256 queries")
        json.dump({"note": note}, open(os.path.join(P, "topk_pmc_note.json"), "w"))
        print(note)

# ---- isolated kernels
ef = rd(os.path.join(fin, "epilogue_forms.txt"))
if ef:
    open(os.path.join(P, f"{tag}_epilogue_forms.txt"), "w").write(
        "# tools/r5/epi_forms.py (MB_ITEMS = 1024: 263 168 rows = ViT-L/14 x 1024 items): the linear-layer shapes with the epilogues the train step runs them with, next to\n"
        "# their plain forms; ms, two timings of 10 launches each.  Round 4 (profiles/r04_microbench.txt): out fwd fused 0.857 / plain 0.519, fc fwd 2.257 / 1.868,\n"
        "# proj fwd 1.969 / 1.642, proj dgrad x act'(f) 2.62-2.81 / 1.897.  VERDICT r04 targets: out <= 0.60, proj fwd <= 1.80, fc fwd <= 2.05, dgrad-act <= 2.20.\n"
        + "\n".join(l for l in ef.splitlines() if "amdgpu.ids" not in l) + "\n")
mb = rd(os.path.join(fin, "microbench.txt"))
if mb:
    open(os.path.join(P, f"{tag}_microbench.txt"), "w").write(f"# MB_ITEMS=1024 python tools/microbench.py (build of round {tag[1:]})\n" + "\n".join(l for l in mb.splitlines() if "amdgpu.ids" not in l) + "\n")
for n in ("blip", "clipff", "embed"):
    k = rd(os.path.join(sec, f"{n}_kernel_stats.txt"))
    j = line_of(os.path.join(sec, f"{n}.json"))
    if k:
        open(os.path.join(P, f"{tag}_{n}_kernel_stats.txt"), "w").write(
            f"# rocprofv3 --kernel-trace --stats -- python tools/bench_{n}.py --steps 3 (tools/profile_secondary.sh); the run's own line: "
            f"{json.dumps({kk: j[kk] for kk in list(j)[:4]}) if j else 'n/a'}\n" + k)
bl = line_of(os.path.join(fin, "bench_line.json"))
if bl:
    json.dump(bl, open(os.path.join(P, f"{tag}_bench_line.json"), "w"))
print("profiles written for", tag)
