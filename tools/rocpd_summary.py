"""Turn a rocprofv3 (ROCm 7.2 rocpd sqlite) kernel trace into the text summary committed under profiles/.
usage: python tools/rocpd_summary.py gpurun_out/prof/x_results.db > profiles/rNN_name.txt"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(top_kernels)")]
rows = [dict(zip(cols, r)) for r in cur.execute("select * from top_kernels")]
tot = sum(r["total_duration"] for r in rows)
print(f"# rocprofv3 --kernel-trace --stats summary ({sys.argv[1]}); top_kernels view stores microseconds")
print(f"# total kernel time {tot/1e6:.4f} s over {sum(r['total_calls'] for r in rows)} dispatches")
print(f"{'kernel':100s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'pct':>6s}")
for r in rows[:40]:
    print(f"{r['name'][:100]:100s} {r['total_calls']:7d} {r['total_duration']/1e3:10.3f} {r['average']:10.2f} {r['percentage']:6.2f}")
