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


from rolo_amd.peerbench import file_barrier, rank_main  # noqa: E402,F401


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
    procs = [subprocess.Popen([sys.executable, "-m", "rolo_amd.peerbench", str(r), str(world), d, sensor, str(frames), str(leaf), "0"], env=env, cwd=ROOT) for r in range(world)]
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
