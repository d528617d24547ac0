"""Round 6 (round 5's verdict, item 7): is the library's load heuristic right OUTSIDE the bench's regime?

`rolo_register_async` counts the frames other contexts OF THIS PROCESS have in flight (api.hip g_frames_in_flight) and picks the kernels that share the chip
(64-query packet walk, 64 resident LM workgroups) or the ones that finish soonest alone (two lanes per query, 256 workgroups). Two regimes it cannot see:
  (1) two PROCESSES, one context each, on the same GPU;
  (2) one context beside a FOREIGN compute stream of the same process (here: a loop of 1 GiB device-to-device copies on a torch stream).
For both: scans/s with the hint pinned to idle (0), pinned to busy (1) and left to the library (-1: per frame, plus what the resident kernel's admission learns — a launch that
could not become resident makes the context take the busy sizing for the next 64 frames). Usage: python profiles/tools/load_regimes.py > gpurun_out/load_regimes.json"""
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

WORKER = r"""
import sys, time, json, numpy as np
sys.path.insert(0, sys.argv[1])
import torch
from rolo_amd import synth
from rolo_amd.rotvgicp import RotVGICP
hint, seconds, foreign = int(sys.argv[2]), float(sys.argv[3]), int(sys.argv[4])
src, tgt, _ = synth.dense_pair("os1-128", seed=synth.SEED)
d = (torch.from_numpy(src).cuda(), torch.from_numpy(tgt).cuda(), src.shape[0])
G = -np.asarray(synth.PREV_STEP_T, np.float64)
g = RotVGICP(0); g.setResolution(0.5); g.setFixedIterations(20); g.setLoadHint(hint)
def frame():
    g.setInputTargetDevice(d[1].data_ptr(), d[2], 4); g.setInputSourceDevice(d[0].data_ptr(), d[2], 4)
    g.register_async(None, np.zeros(3), G, G * 0.97); g.register_wait()
for _ in range(10): frame()
fs = None
if foreign:
    a = torch.empty(1 << 30, dtype=torch.uint8, device="cuda"); b = torch.empty_like(a)
    fs = torch.cuda.Stream()
print("READY", flush=True)
sys.stdin.readline()
n = 0; copies = 0; t0 = time.perf_counter()
while time.perf_counter() - t0 < seconds:
    if fs is not None:
        with torch.cuda.stream(fs):
            for _ in range(2): b.copy_(a); copies += 1
    for _ in range(20): frame(); n += 1
if fs is not None: fs.synchronize()
dt = time.perf_counter() - t0
print(json.dumps({"scans_per_s": n / dt, "frames": n, "copy_GBps": copies * 2 * (1 << 30) / dt / 1e9, "counters": g.counters()}), flush=True)
"""


def run(n_proc, hint, foreign, seconds=4.0):
    ps = [subprocess.Popen([sys.executable, "-c", WORKER, ROOT, str(hint), str(seconds), str(foreign)], stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True, cwd=ROOT) for _ in range(n_proc)]
    for p in ps:
        assert p.stdout.readline().strip() == "READY"
    for p in ps:
        p.stdin.write("go\n"); p.stdin.flush()
    out = []
    for p in ps:
        line = p.stdout.readline()
        out.append(json.loads(line))
        p.wait(timeout=60)
    return out


def main():
    res = {}
    if len(sys.argv) > 1 and sys.argv[1] == "--two-auto":
        r2 = run(2, -1, 0, seconds=float(sys.argv[2]) if len(sys.argv) > 2 else 8.0)
        print(json.dumps({"scans_per_s_each": [r["scans_per_s"] for r in r2], "counters": [r["counters"] for r in r2]}))
        return
    for hint in (0, 1, -1):
        r1 = run(1, hint, 0)
        res[f"one_process_alone_hint{hint}"] = {"scans_per_s": r1[0]["scans_per_s"], "persist_bails": r1[0]["counters"]["persist_bails"], "load_mode_at_end": r1[0]["counters"].get("load_mode")}
        r2 = run(2, hint, 0)
        res[f"two_processes_hint{hint}"] = {"scans_per_s_each": [r["scans_per_s"] for r in r2], "sum": sum(r["scans_per_s"] for r in r2), "persist_bails": [r["counters"]["persist_bails"] for r in r2],
                                            "load_mode_at_end": [r["counters"].get("load_mode") for r in r2], "graph_captures": [r["counters"]["graph_captures"] for r in r2]}
        rf = run(1, hint, 1)
        res[f"beside_foreign_copies_hint{hint}"] = {"scans_per_s": rf[0]["scans_per_s"], "copy_GBps": rf[0]["copy_GBps"], "persist_bails": rf[0]["counters"]["persist_bails"]}
        print(hint, json.dumps({k: v for k, v in res.items() if k.endswith(str(hint))}), file=sys.stderr, flush=True)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
