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


def full(n):
    """name with its template arguments: `knn_walk_kernel<20, false, false, true>` (the key of profiles/tools/codeobj_notes.py's table)"""
    m = re.search(r"rolo::\(anonymous namespace\)::([A-Za-z_0-9]+)(<[^()]*>)?\(", n)
    return (m.group(1) + (m.group(2) or "")) if m else None


def load_notes(path):
    """profiles/rNN/codeobj_notes.csv -> {kernel<args>: row}: what the CODE OBJECT says about registers, spills, LDS and scratch. rocprofv3's VGPR_Count column is the
    allocation granule count of the arch VGPR file (58 / 202 / 192 / 110 / 146 registers read 32 / 104 / 96 / 56 / 76 there: round 5's verdict, item 6)."""
    if not path:
        return {}
    return {r["kernel"]: r for r in csv.DictReader(open(path))}


def main(paths):
    notes_path = None
    if "--notes" in paths:
        i = paths.index("--notes"); notes_path = paths[i + 1]; paths = paths[:i] + paths[i + 2:]
    notes = load_notes(notes_path)
    inst = collections.defaultdict(collections.Counter)   # short name -> the instantiations that ran (by dispatches)
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
            f = full(r["Kernel_Name"])
            if f:
                inst[k][f] += 1
    w = csv.writer(sys.stdout)
    # SQ_WAVE_CYCLES (and SQ_WAIT_*, SQ_ACTIVE_INST_*, SQ_BUSY_CYCLES) count QUAD-cycles on gfx950 (MI355X_MICROARCH.md, "s_memtime tick vs SQ PMC units"): one count = 4
    # shader clocks. Rounds 2-4 printed the raw quotient as "mean_wave_cycles", four times too short (30 514 for a walk whose wavefronts live 50-60 us); the shares, being
    # ratios of two quad-cycle counters, were right. mean_wave_us uses the shader clock GRBM_GUI_ACTIVE / kernel duration of the same dispatches when that counter was
    # collected, 2.4 GHz otherwise, and can be held against the s_memrealtime records of profiles/tools/wavestats.py.
    # "rocprof_VGPR_granules" (round 5: "VGPRs") is rocprofv3's VGPR_Count — NOT the registers a wavefront holds; code_object_* come from the library's own metadata (--notes)
    w.writerow(["kernel", "launches", "mean_us", "rocprof_VGPR_granules", "SGPRs", "LDS_B", "workgroup", "waves", "VALU_per_wave", "SALU_per_wave", "SMEM_per_wave", "mean_wave_cycles",
                "mean_wave_us", "clock_GHz", "busy_quad_cycles_per_SE_sum", "wait_inst_share_of_wave_cycles", "wait_any_share_of_wave_cycles", "valu_thread_utilisation",
                "valu_insts_per_launch_M", "salu_insts_per_launch_M", "instantiation", "code_object_vgprs", "code_object_agprs", "code_object_sgprs", "code_object_vgpr_spills",
                "code_object_scratch_bytes_per_lane", "code_object_lds_bytes", "waves_per_simd_by_registers"])
    mean = lambda v: sum(v) / len(v) if v else float("nan")
    for k in sorted(acc, key=lambda k: -sum(acc[k]["_dur_ns"])):
        a = acc[k]
        waves = mean(a["SQ_WAVES"]); wc = mean(a["SQ_WAVE_CYCLES"])
        # shader clock of the run = GRBM_GUI_ACTIVE / kernel duration; rocprofv3 reports the counter summed over the chip's 8 XCDs on this stack (ratio ~17-19 per ns):
        # take the divisor that lands in a plausible clock range, else the nominal 2.4 GHz
        ghz = 2.4
        if a["GRBM_GUI_ACTIVE"] and mean(a["_dur_ns"]) > 0:
            r = mean(a["GRBM_GUI_ACTIVE"]) / mean(a["_dur_ns"])
            for div in (1.0, 8.0, 32.0):
                if 0.8 < r / div < 2.6:
                    ghz = r / div
                    break
        cyc = 4.0 * wc / waves if waves else float("nan")
        row = [k, len(a["SQ_WAVES"]) or len(a["_dur_ns"]), round(mean(a["_dur_ns"]) / 1e3, 2), *meta[k], round(waves), round(mean(a["SQ_INSTS_VALU"]) / waves, 1) if waves else "",
               round(mean(a["SQ_INSTS_SALU"]) / waves, 1) if waves else "", round(mean(a["SQ_INSTS_SMEM"]) / waves, 1) if waves else "", round(cyc) if waves else "",
               round(cyc / ghz / 1e3, 2) if waves else "", round(ghz, 2),
               round(mean(a["SQ_BUSY_CYCLES"])), round(mean(a["SQ_WAIT_INST_ANY"]) / wc, 3) if wc else "",
               round(mean(a["SQ_WAIT_ANY"]) / wc, 3) if wc and a["SQ_WAIT_ANY"] else "",
               round(mean(a["SQ_THREAD_CYCLES_VALU"]) / (64 * mean(a["SQ_ACTIVE_INST_VALU"])), 3) if a["SQ_THREAD_CYCLES_VALU"] and a["SQ_ACTIVE_INST_VALU"] else "",
               round(mean(a["SQ_INSTS_VALU"]) / 1e6, 3) if a["SQ_INSTS_VALU"] else "", round(mean(a["SQ_INSTS_SALU"]) / 1e6, 3) if a["SQ_INSTS_SALU"] else ""]
        # (valu_thread_utilisation = live lanes per issued VALU instruction / 64; round 2 divided by 64 * 4 and the column saturated at 0.25)
        top = inst[k].most_common(1)[0][0] if inst[k] else ""
        nt = notes.get(top, {})
        row += [top, nt.get("vgpr_count", ""), nt.get("agpr_count", ""), nt.get("sgpr_count", ""), nt.get("vgpr_spill", ""), nt.get("scratch_bytes_per_lane", ""), nt.get("lds_bytes", ""),
                nt.get("waves_per_simd_by_registers", "")]
        w.writerow(row)


if __name__ == "__main__":
    main(sys.argv[1:])
