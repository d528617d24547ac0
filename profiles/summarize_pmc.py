#!/usr/bin/env python3
"""Summarise two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE — separate runs, as MI355X_MICROARCH.md prescribes:
they do not fit one pass) into per-kernel HBM bytes per launch:

    hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024

FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 FETCH_SIZE tallies 64 B per 128-B request, hence the factor 2
(MI355X_MICROARCH.md, "HBM"). WRITE_SIZE is uncalibrated there; taken as is.

    python profiles/summarize_pmc.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json>
"""
import collections
import csv
import json
import re
import sys


def short(n):
    n = n.replace("void ", "")
    m = re.search(r"rolo::\(anonymous namespace\)::([A-Za-z_0-9]+)", n)
    if m:
        return m.group(1)
    if "rocprim" in n:
        for tag in ("merge_sort_block_merge", "radix_sort_block_sort", "transform"):
            if tag in n:
                return "rocprim::" + tag
        return "rocprim::other"
    return n.split("(")[0][:40]


def load(path, counter):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            acc[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return acc


def main(fetch_csv, write_csv, out_json):
    f = load(fetch_csv, "FETCH_SIZE")
    w = load(write_csv, "WRITE_SIZE")
    out = {}
    for k in sorted(f, key=lambda k: -sum(f[k])):
        fv, wv = f[k], w.get(k, [0.0])
        fa, wa = sum(fv) / len(fv), sum(wv) / max(len(wv), 1)
        out[k] = {"launches": len(fv), "FETCH_SIZE_KB_avg": round(fa, 1), "WRITE_SIZE_KB_avg": round(wa, 1),
                  "hbm_bytes_per_launch": round((2 * fa + wa) * 1024)}
    import os
    import subprocess
    try:   # the commit the counters were collected at (bench.py cites this file: traffic and avg_launch_ms must describe the same code)
        commit = os.environ.get("ROLO_PROF_COMMIT") or subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or "?"
    except Exception:
        commit = "?"
    out["_collected"] = {"commit": commit, "command": os.environ.get("ROLO_PROF_COMMAND", "")}
    json.dump(out, open(out_json, "w"), indent=1)
    del out["_collected"]
    for k, v in out.items():
        print(f"{k:34s} {v['launches']:5d} {v['FETCH_SIZE_KB_avg']:12.1f} {v['WRITE_SIZE_KB_avg']:12.1f} {v['hbm_bytes_per_launch']:14d}")


if __name__ == "__main__":
    main(*sys.argv[1:4])
