#!/usr/bin/env python3
"""L2 (TCC) counters per kernel from rocprofv3 --pmc passes: hits, misses, hit rate, requests, fabric read / write requests per launch.
    python profiles/tools/tcc_summary.py <counter_collection.csv> [<second pass>] > l2_counters.csv"""
import collections
import csv
import re
import sys


def short(n):
    m = re.search(r"rolo::\(anonymous namespace\)::([A-Za-z_0-9]+)", n)
    return m.group(1) if m else n.split("(")[0][:40]


acc = collections.defaultdict(lambda: collections.defaultdict(list))
for p in sys.argv[1:]:
    for r in csv.DictReader(open(p)):
        acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
w = csv.writer(sys.stdout)
cols = ["TCC_HIT_sum", "TCC_MISS_sum", "TCC_REQ_sum", "TCC_EA0_RDREQ_sum", "TCC_EA0_WRREQ_sum"]
w.writerow(["kernel", "launches"] + [c + "_per_launch" for c in cols] + ["l2_hit_rate"])
mean = lambda v: sum(v) / len(v) if v else float("nan")
for k in sorted(acc, key=lambda k: -mean(acc[k].get("TCC_REQ_sum", [0]))):
    a = acc[k]
    h, m = mean(a["TCC_HIT_sum"]), mean(a["TCC_MISS_sum"])
    w.writerow([k, max(len(v) for v in a.values())] + [round(mean(a[c])) if a[c] else "" for c in cols] + [round(h / (h + m), 4) if a["TCC_HIT_sum"] and (h + m) > 0 else ""])
