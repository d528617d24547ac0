#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc SQ_* passes (counter_collection.csv) into one row per kernel: launches, mean duration, wavefronts, VALU / SALU /
SMEM instructions per wavefront, mean wavefront lifetime as a fraction of the kernel's duration, share of wavefront cycles waiting on
instructions.   python profiles/summarize_sq.py <pass1 counter_collection.csv> [<pass2 ...>] > out.csv"""
import collections
import csv
import re
import sys


def short(n):
    m = re.search(r"rolo::\(anonymous namespace\)::([A-Za-z_0-9]+)", n)
    if m:
        return m.group(1)
    if "rocprim" in n:
        return "rocprim::" + ("merge_sort_block_merge" if "block_merge" in n else "radix_sort_block_sort" if "block_sort" in n else "other")
    return n.split("(")[0][:40]


def main(paths):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    meta = {}
    for p in paths:
        seen = set()
        for r in csv.DictReader(open(p)):
            k = short(r["Kernel_Name"])
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            key = (p, r["Dispatch_Id"])
            if key not in seen:
                seen.add(key)
                acc[k]["_dur_ns"].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
            meta[k] = (r["VGPR_Count"], r["SGPR_Count"], r["LDS_Block_Size"], r["Workgroup_Size"])
    w = csv.writer(sys.stdout)
    w.writerow(["kernel", "launches", "mean_us", "VGPRs", "SGPRs", "LDS_B", "workgroup", "waves", "VALU_per_wave", "SALU_per_wave", "SMEM_per_wave", "mean_wave_cycles",
                "busy_cycles_per_SE_sum", "wait_inst_share_of_wave_cycles", "wait_any_share_of_wave_cycles", "valu_thread_utilisation"])
    mean = lambda v: sum(v) / len(v) if v else float("nan")
    for k in sorted(acc, key=lambda k: -sum(acc[k]["_dur_ns"])):
        a = acc[k]
        waves = mean(a["SQ_WAVES"]); wc = mean(a["SQ_WAVE_CYCLES"])
        row = [k, len(a["SQ_WAVES"]) or len(a["_dur_ns"]), round(mean(a["_dur_ns"]) / 1e3, 2), *meta[k], round(waves), round(mean(a["SQ_INSTS_VALU"]) / waves, 1) if waves else "",
               round(mean(a["SQ_INSTS_SALU"]) / waves, 1) if waves else "", round(mean(a["SQ_INSTS_SMEM"]) / waves, 1) if waves else "", round(wc / waves) if waves else "",
               round(mean(a["SQ_BUSY_CYCLES"])), round(mean(a["SQ_WAIT_INST_ANY"]) / wc, 3) if wc else "",
               round(mean(a["SQ_WAIT_ANY"]) / wc, 3) if wc and a["SQ_WAIT_ANY"] else "",
               round(mean(a["SQ_THREAD_CYCLES_VALU"]) / (64 * mean(a["SQ_ACTIVE_INST_VALU"])), 3) if a["SQ_THREAD_CYCLES_VALU"] and a["SQ_ACTIVE_INST_VALU"] else ""]
        # (valu_thread_utilisation = live lanes per issued VALU instruction / 64; round 2 divided by 64 * 4 and the column saturated at 0.25)
        w.writerow(row)


if __name__ == "__main__":
    main(sys.argv[1:])
