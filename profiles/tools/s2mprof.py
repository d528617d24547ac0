"""Kernel-level view of the back end's scan-to-submap call: bench.py's backend workload, the resident-sub-map call repeated; run under
rocprofv3 --kernel-trace --stats (profiles/tools/README.md). Prints the wall time per call and the per-iteration host time between launches."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench

ap = argparse.ArgumentParser(); ap.add_argument("--sensor", default="os1-128"); ap.add_argument("--reps", type=int, default=20)
a = ap.parse_args()
t0 = time.perf_counter()
for _ in range(max(1, a.reps // 5)):
    r = bench.backend_leg(a, 0)
print(r, "wall", time.perf_counter() - t0)
