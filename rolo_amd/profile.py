"""bench.py helper: per-kernel durations from HIP events on the context's stream, and the `roofline` object.

ALGORITHMIC bytes (fixed definitions; compact records P = 16 B point, C = 24 B covariance,
R = 48 B voxel record, S = 16 B hash slot):
  knn_walk_kernel  (K5)      : K5 is (P + K*P + C) = 360 B per point; the walk reads the query and its K neighbours
                               (P + K*P = 336 B per point of the cloud(s) it is launched on: one cloud, or the source/target
                               pair of a frame when overlap_knn is set); the C = 24 B are written by knn_tail_kernel
  rot_pass_kernel  (K7+K8+K10): (P + C + S + R) = 104 B per source point per pass
  trans_pass_kernel (K11)     : 104 B per source point per pass
"""
from __future__ import annotations

import ctypes as C
import json
import os

import numpy as np

from ._lib import lib, check

SLOTS = {"knn_build": 0, "knn_walk": 1, "voxel_build": 2, "rot_pass": 3, "trans_pass": 4, "ctrl": 5, "knn_tail": 6, "lm_pass": 7}
# per point of the cloud(s) a launch works on; knn_build (Morton sort + BVH, not in SURVEY 8d) is priced as one read and one write of
# the points; the controller moves no algorithmic bytes at all (pure overhead)
BYTES_PER_POINT = {"knn_walk": 336.0, "knn_tail": 24.0, "knn_build": 32.0, "voxel_build": 136.0, "rot_pass": 104.0, "trans_pass": 104.0,
                   "lm_pass": 104.0, "ctrl": 0.0}
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def pmc_traffic_file():
    """(path, description) of the newest committed PMC summary, profiles/rNN/pmc_traffic.json — the file the `traffic` figures are cited from; it carries the
    commit it was collected at. (Until round 3 this was a root profiles/pmc_traffic.json that went stale.)"""
    import glob
    c = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]", "pmc_traffic.json")))
    if not c:
        return None, None
    path = c[-1]
    try:
        meta = json.load(open(path)).get("_collected", {})
    except Exception:
        meta = {}
    rel = os.path.relpath(path, ROOT)
    return path, f"{rel} (collected at commit {meta.get('commit', '?')}: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of a single-context eager run, (2 FETCH + WRITE) KB; not measured in this run)"


def read_slot(g, name, cap=4096):
    buf = (C.c_float * cap)()
    n = check(lib().rolo_prof_read(g._h, SLOTS[name], buf, cap), "rolo_prof_read")
    return np.array(buf[:min(n, cap)], np.float64)


def kernel_times(g, run_one_step, reps=3):
    """Returns {slot: list of per-launch ms} accumulated over `reps` steps with event bracketing enabled."""
    check(lib().rolo_prof_enable(g._h, 1), "rolo_prof_enable")
    acc = {k: [] for k in SLOTS}
    try:
        for _ in range(reps):
            run_one_step()
            for k in SLOTS:
                acc[k].append(read_slot(g, k))
    finally:
        lib().rolo_prof_enable(g._h, 0)
    return acc


def roofline(g, run_one_step, n_src, n_tgt, passes_per_frame, hbm_peak_gbs, reps=3):
    acc = kernel_times(g, run_one_step, reps)
    rot_real = g.last_stats.n_passes
    trans_real = g.last_translation_stats.n_passes
    per_frame_ms = {}
    avg_ms = {}
    real_runs = {}
    for k, runs in acc.items():
        real = []
        for r in runs:
            if k == "rot_pass":
                r = r[:rot_real]      # the tail of the fixed schedule are predicated no-op launches
            elif k == "trans_pass":
                r = r[:trans_real]
            elif k == "lm_pass":
                r = r[:rot_real + trans_real + 1]   # + the launch that finishes the last trial
            real.append(r)
        real_runs[k] = real
        per_frame_ms[k] = float(np.mean([r.sum() for r in real])) if real else 0.0
        allv = np.concatenate(real) if real else np.zeros(0)
        avg_ms[k] = float(allv.mean()) if allv.size else 0.0
    # top-3 = largest shares of the frame's GPU time among ALL timed slots (controller and tree build included). The roofline object is
    # for the top kernel that MOVES algorithmic bytes (the controller moves none: it is pure overhead and is listed, not put on a roofline).
    # An event pair around a sub-10 us launch also times the launch gap (rot_pass: 6.9 us in rocprofv3, 9.7 us between events), so the
    # ranking of the byte-moving kernels charges every launch 2.7 us less; the reported durations stay the raw event times.
    total = sum(per_frame_ms.values()) or 1.0
    rank = sorted(per_frame_ms, key=per_frame_ms.get, reverse=True)
    top3 = [{"kernel": k + "_kernel", "per_frame_ms": per_frame_ms[k], "share_of_timed_slots": per_frame_ms[k] / total,
             "algorithmic_bytes_per_point": BYTES_PER_POINT.get(k, 0.0)} for k in rank[:3]]
    launches_per_frame = {k: float(np.mean([len(r) for r in real_runs[k]])) if real_runs.get(k) else 0.0 for k in per_frame_ms}
    movers = [k for k in per_frame_ms if BYTES_PER_POINT.get(k, 0.0) > 0.0]
    dom = max(movers, key=lambda k: per_frame_ms[k] - 0.0027 * launches_per_frame[k])
    if dom in ("knn_walk", "knn_tail", "knn_build"):
        # one launch works on one cloud, or on the source/target pair of a frame (overlap_knn)
        launches = np.mean([len(r) for r in acc[dom]]) if acc[dom] else 2.0
        if dom == "knn_build":
            launches = 1.0 if launches else 1.0   # one event pair brackets the whole chain of build launches
        npts = (n_src + n_tgt) / max(launches, 1.0)
    elif dom == "voxel_build":
        npts = n_tgt
    else:
        npts = n_src
    algo_bytes = BYTES_PER_POINT[dom] * npts
    resident = dom == "lm_pass" and launches_per_frame.get("lm_pass", 0.0) <= 1.5 and passes_per_frame > 1
    if resident:   # the resident LM kernel (rolo_params.fused_lm = 2): ONE launch evaluates every pass of the frame — SURVEY 8d's 104 B per point and pass, cost-only passes priced the same
        algo_bytes *= passes_per_frame
    achieved = algo_bytes / (avg_ms[dom] * 1e-3) / 1e9 if avg_ms[dom] > 0 else 0.0
    traffic = None
    pmc, pmc_desc = pmc_traffic_file()
    if pmc and os.path.exists(pmc):
        try:
            tr = json.load(open(pmc))
            names = [dom + "_kernel"] + (["knn_walk_sub_kernel"] if dom == "knn_walk" else [])   # the walk has two kernels: the profile holds the one the launch size picks
            traffic = next((tr[k]["hbm_bytes_per_launch"] for k in names if k in tr), None)
        except Exception:
            traffic = None
    # the same kernel per profiled step (bench.py profiles one step per pair of its input pool, in pool order: step 0 = the nominal pair of SURVEY 8d)
    per_step_ms = [float(r.mean()) for r in real_runs.get(dom, []) if len(r)]
    kname = dom + "_kernel"
    lanes = g.counters().get("walk_lanes", 1)   # which of the walk's kernels the launch size picked (rolo_ctx_counters [8])
    walk_name = "knn_walk_kernel" if lanes <= 1 else f"knn_walk_sub_kernel<{lanes}>"
    if dom == "knn_walk":
        kname = walk_name
    if resident:
        kname = "lm_persist_kernel"
        if traffic is None and pmc and os.path.exists(pmc):
            try:
                traffic = json.load(open(pmc)).get("lm_persist_kernel", {}).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
    out = {"bound": "hbm", "kernel": kname, "achieved": achieved, "peak": hbm_peak_gbs, "unit": "GB/s",
           "frac": achieved / hbm_peak_gbs, "traffic": traffic, "avg_launch_ms_per_profiled_step": per_step_ms,
           "traffic_source": pmc_desc,
           "algorithmic_bytes_per_launch": algo_bytes, "avg_launch_ms": avg_ms[dom], "top3": top3, "per_frame_ms": per_frame_ms, "avg_ms": avg_ms}
    if resident:
        out["passes_per_launch"] = passes_per_frame
    if dom != "knn_walk" and avg_ms.get("knn_walk", 0.0) > 0:   # the neighbour search next to it: rounds 1-5's dominant kernel, kept for the comparison across rounds
        wl = np.mean([len(r) for r in acc["knn_walk"]]) if acc["knn_walk"] else 1.0
        wb = BYTES_PER_POINT["knn_walk"] * (n_src + n_tgt) / max(wl, 1.0)
        wt = None
        if pmc and os.path.exists(pmc):
            try:
                tr = json.load(open(pmc))
                wt = next((tr[k]["hbm_bytes_per_launch"] for k in ("knn_walk_kernel", "knn_walk_sub_kernel") if k in tr), None)
            except Exception:
                wt = None
        out["search"] = {"kernel": walk_name, "algorithmic_bytes_per_launch": wb, "avg_launch_ms": avg_ms["knn_walk"], "achieved": wb / (avg_ms["knn_walk"] * 1e-3) / 1e9,
                         "frac": wb / (avg_ms["knn_walk"] * 1e-3) / 1e9 / hbm_peak_gbs, "traffic": wt,
                         "avg_launch_ms_per_profiled_step": [float(r.mean()) for r in real_runs.get("knn_walk", []) if len(r)]}
    return out


def sq_counters_file():
    """(path, commit) of the newest committed profiles/rNN/sq_counters.csv"""
    import glob
    c = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]", "sq_counters.csv")))
    if not c:
        return None, None
    commit = "?"
    try:
        commit = json.load(open(os.path.join(os.path.dirname(c[-1]), "pmc_traffic.json"))).get("_collected", {}).get("commit", "?")
    except Exception:
        pass
    return c[-1], commit


N_SIMD = 256 * 4          # MI355X: 256 CUs x 4 SIMDs


def valu_ns_per_inst():
    """(ns per wavefront VALU instruction per SIMD, source): the newest committed profiles/rNN/valu_rate.txt (profiles/tools/valu_rate.hip on the box: v_fma_f64, v_min_f64,
    v_max_f32, v_pk_add_f32, v_cmp at four wavefronts per SIMD — the instruction mix of the hot kernels; all within 4 % of each other), or 2.0 ns as measured in round 5"""
    import glob
    import re
    c = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]", "valu_rate.txt")))
    if c:
        v = [float(m.group(1)) for l in open(c[-1]) if "waves/SIMD 4" in l and not l.startswith("v_cndmask") for m in [re.search(r"-> ([0-9.]+) ns per instruction", l)] if m]
        if v:
            return float(np.mean(v)), os.path.relpath(c[-1], ROOT)
    return 2.0, "round-5 measurement (profiles/tools/valu_rate.hip)"


def valu_issue(scans_per_s_per_gpu: float):
    """The instruction-issue roofline of a frame: wave-level VALU (and SALU) instructions per frame, kernel by kernel, from the newest committed rocprofv3 SQ pass
    (profiles/rNN/sq_counters.csv: SQ_INSTS_VALU / SQ_INSTS_SALU per launch x launches per frame of that profile run), against what the chip's 1024 SIMDs can issue:
        issue_ms_per_frame = VALU instructions x (measured ns per instruction per SIMD) / 1024 SIMDs;   frac = issue_ms_per_frame x scans/s.
    Cited like roofline.traffic (collected in its own PMC passes, not in this run); `frac` uses THIS run's scans/s."""
    import csv
    path, commit = sq_counters_file()
    if not path:
        return {"error": "no profiles/rNN/sq_counters.csv"}
    rows = [r for r in csv.DictReader(l for l in open(path) if not l.startswith("#")) if r.get("valu_insts_per_launch_M")]
    if not rows:
        return {"error": f"{os.path.relpath(path, ROOT)} predates the per-launch instruction columns"}
    by = {r["kernel"]: r for r in rows}
    walk = next((by[k] for k in ("knn_walk_kernel", "knn_walk_sub_kernel") if k in by), None)
    frames = float(walk["launches"]) if walk else 1.0
    per, salu = {}, {}
    for k, r in by.items():
        try:
            lf = float(r["launches"]) / frames
            per[k] = round(float(r["valu_insts_per_launch_M"]) * lf, 3)
            salu[k] = round(float(r["salu_insts_per_launch_M"] or 0) * lf, 3)
        except Exception:
            continue
    tot_v, tot_s = sum(per.values()), sum(salu.values())
    ns, ns_src = valu_ns_per_inst()
    issue_ms = tot_v * 1e6 * ns * 1e-9 / N_SIMD * 1e3
    frame_ms = 1e3 / scans_per_s_per_gpu if scans_per_s_per_gpu > 0 else float("nan")
    top = dict(sorted(per.items(), key=lambda kv: -kv[1])[:8])
    return {"bound": "valu-issue", "valu_insts_per_frame_M": round(tot_v, 2), "salu_insts_per_frame_M": round(tot_s, 2), "valu_insts_per_frame_M_by_kernel": top,
            "simds": N_SIMD, "ns_per_valu_inst_per_simd": round(ns, 3), "issue_ms_per_frame": round(issue_ms, 4), "frame_ms_this_run": round(frame_ms, 4),
            "frac": round(issue_ms / frame_ms, 4),
            "what": "share of the chip's VALU issue capacity (1024 SIMDs x one wavefront instruction per measured ns_per_valu_inst_per_simd) that the frames of this run's "
                    "timed region occupy; the rest is latency the contexts in flight did not cover",
            "source": f"{os.path.relpath(path, ROOT)} (collected at commit {commit}: SQ_INSTS_VALU / SQ_INSTS_SALU of a single-context eager run with the busy-device kernels, "
                      f"per launch x launches per frame); issue rate: {ns_src}"}
