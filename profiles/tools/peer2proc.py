#!/usr/bin/env python3
"""W ranks in W PROCESSES on ONE device: the point-sharded frame (BASELINE configs[3]) through the peer exchange (rolo_peer_*), timed.

    python profiles/tools/peer2proc.py [--world 2] [--sensor os1-128x2048] [--frames 20] [--out gpurun_out/peer2proc.json]

The pool's boxes have one GPU, so this is not a scaling measurement (the ranks share the chip); it measures what the exchange COSTS:
  * whole frames per second of the W-rank job against the unsharded frame on the same device, graph replay on;
  * per-launch HIP-event times of the pass and of the controller-with-exchange (eager launches, rolo_prof_*), against the unsharded
    controller — the per-trial latency budget of DESIGN.md section 6.
Rank processes meet through files in a temporary directory (handles, barriers)."""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def file_barrier(d, name, rank, world, timeout=300):
    open(os.path.join(d, f"{name}.{rank}"), "w").close()
    t0 = time.time()
    while not all(os.path.exists(os.path.join(d, f"{name}.{q}")) for q in range(world)):
        if time.time() - t0 > timeout:
            raise SystemExit(f"barrier {name} timed out")
        time.sleep(0.002)


def rank_main(rank, world, d, sensor, frames, leaf):
    from rolo_amd import synth, profile
    from rolo_amd.rotvgicp import RotVGICP
    src, tgt, _ = synth.dense_pair(sensor, seed=synth.SEED)
    G = -np.asarray(synth.PREV_STEP_T); L0 = G * 0.97
    g = RotVGICP(0); g.setResolution(leaf); g.setFixedIterations(20)
    if world > 1:
        h = g.peer_export(world, 2 * src.shape[0])
        with open(os.path.join(d, f"h{rank}.tmp"), "wb") as f:
            f.write(h)
        os.replace(os.path.join(d, f"h{rank}.tmp"), os.path.join(d, f"h{rank}.bin"))
        file_barrier(d, "exported", rank, world)
        g.peer_connect([open(os.path.join(d, f"h{q}.bin"), "rb").read() for q in range(world)], rank, world)

    # inputs resident in HBM, through the HIP runtime directly (no torch in the rank processes: torch's own streams / queues on top of W
    # processes oversubscribe the one device's hardware queues and every kernel then pays a queue switch — measured: 61 us per 12 us pass)
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")

    def to_dev(a):
        p = C.c_void_p(); assert hip.hipMalloc(C.byref(p), C.c_size_t(a.nbytes)) == 0
        assert hip.hipMemcpy(p, a.ctypes.data_as(C.c_void_p), C.c_size_t(a.nbytes), 1) == 0
        return p.value
    src = np.ascontiguousarray(src, np.float32); tgt = np.ascontiguousarray(tgt, np.float32)
    p_src, p_tgt = to_dev(src), to_dev(tgt)

    def frame():   # the whole frame every time (K5 + exchange, map, both LM stages)
        g.setInputTargetDevice(p_tgt, tgt.shape[0], 4); g.setInputSourceDevice(p_src, src.shape[0], 4)
        g.register_async(None, np.zeros(3), G, L0)
        return g.register_wait()

    for _ in range(6):
        frame()
    file_barrier(d, "warm", rank, world)
    t0 = time.perf_counter()
    for _ in range(frames):
        Tf, Td, t = frame()
    dt = time.perf_counter() - t0
    file_barrier(d, "timed", rank, world)
    res = {"rank": rank, "ms_per_frame": 1e3 * dt / frames, "passes": g.last_stats.n_passes + g.last_translation_stats.n_passes, "counters": g.counters(),
           "pose_head": Td.reshape(-1)[:4].tolist(), "mailbox": g.peer_info()[2] if world > 1 else ""}
    # per-launch event times (eager launches while profiling)
    acc = profile.kernel_times(g, frame, reps=3)
    rot, tr = g.last_stats.n_passes, g.last_translation_stats.n_passes
    for k in ("rot_pass", "trans_pass", "ctrl", "knn_walk", "knn_tail", "knn_build", "voxel_build"):
        v = np.concatenate([r[:rot + tr] if k == "ctrl" else (r[:rot] if k == "rot_pass" else (r[:tr] if k == "trans_pass" else r)) for r in acc[k]]) if acc[k] else np.zeros(0)
        res[k + "_us"] = {"mean": float(1e3 * v.mean()) if v.size else None, "median": float(1e3 * np.median(v)) if v.size else None, "n": int(v.size)}
    file_barrier(d, "profiled", rank, world)
    json.dump(res, open(os.path.join(d, f"res{rank}.json"), "w"))
    file_barrier(d, "done", rank, world)
    g.close()


def run_world_threads(world, sensor, frames, leaf):
    """the same ranks as THREADS of this process (handles resolve through the library's process-local registry): no second process competes
    for the device's queues, so what is left over the unsharded frame is the exchange itself + the replicated sort / tree / map"""
    import threading
    d = tempfile.mkdtemp(prefix="peer2thr_")
    errs = []

    def body(r):
        try:
            rank_main(r, world, d, sensor, frames, leaf)
        except BaseException as e:  # noqa: BLE001
            errs.append((r, repr(e)))
    th = [threading.Thread(target=body, args=(r,)) for r in range(world)]
    [t.start() for t in th]; [t.join(timeout=600) for t in th]
    if errs:
        return {"error": repr(errs)}
    return [json.load(open(os.path.join(d, f"res{r}.json"))) for r in range(world)]


def run_world(world, sensor, frames, leaf):
    d = tempfile.mkdtemp(prefix="peer2proc_")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    code = "import sys; sys.path.insert(0, %r); sys.path.insert(0, %r); from peer2proc import rank_main; rank_main(int(sys.argv[1]), %d, %r, %r, %d, %r)" % (
        ROOT, os.path.dirname(os.path.abspath(__file__)), world, d, sensor, frames, leaf)
    procs = [subprocess.Popen([sys.executable, "-c", code, str(r)], env=env) for r in range(world)]
    rc = [p.wait(timeout=600) for p in procs]
    if any(rc):
        return {"error": f"rank exit codes {rc}"}
    return [json.load(open(os.path.join(d, f"res{r}.json"))) for r in range(world)]


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=2)
    ap.add_argument("--sensor", default="os1-128x2048")
    ap.add_argument("--frames", type=int, default=20)
    ap.add_argument("--leaf", type=float, default=0.5)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "peer2proc.json"))
    a = ap.parse_args()
    out = {"workload": f"{a.sensor} dense pair, leaf {a.leaf} m, 20 SO(3) LM iterations + CT translation; all ranks share ONE device",
           "unsharded_1proc": run_world(1, a.sensor, a.frames, a.leaf), f"peer_{a.world}proc": run_world(a.world, a.sensor, a.frames, a.leaf)}
    if a.world == 2:
        out["peer_2threads_one_process"] = run_world_threads(2, a.sensor, a.frames, a.leaf)
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump(out, open(a.out, "w"), indent=1)
    print(json.dumps(out))
