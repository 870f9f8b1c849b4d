"""rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes of `python bench.py ...` -> profiles/pmc_gemm_traffic.json, the record
bench.py cites as roofline.traffic (bytes per GEMM launch).  Corrections per /opt/skills/guides/MI355X_MICROARCH.md (HBM
section): counters in KB; FETCH_SIZE on gfx950 reports 1/2 of wide (16 B/lane) reads -> doubled; WRITE_SIZE as is.
usage: python tools/pmc_gemm_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> MODEL PAIRS [source-note]"""
import csv
import json
import os
import sys


def per_launch(path, counter):
    n, tot = 0, 0.0
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter and r["Kernel_Name"].startswith("void gemm_glds_kernel<ElemBF16"):
            n += 1
            tot += float(r["Counter_Value"])
    return n, tot / max(n, 1) * 1024.0


def main():
    fetch_csv, write_csv, model, pairs = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
    src = sys.argv[5] if len(sys.argv) > 5 else ""
    nf, f = per_launch(fetch_csv, "FETCH_SIZE")
    nw, w = per_launch(write_csv, "WRITE_SIZE")
    rec = {"model": model, "pairs": pairs, "bytes_per_launch": round(2 * f + w), "fetch_bytes_per_launch_corrected": round(2 * f),
           "write_bytes_per_launch": round(w), "launches": [nf, nw],
           "note": f"bytes per GEMM launch (mean over {nf} launches), rocprofv3 --pmc FETCH_SIZE (x2, gfx950 correction) + WRITE_SIZE in "
                   f"separate passes; L2<->fabric requests, MALL hits included (upper bound of HBM bytes); algorithmic mean 2.6e9. {src}"}
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "pmc_gemm_traffic.json")
    json.dump(rec, open(out, "w"), indent=1)
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
