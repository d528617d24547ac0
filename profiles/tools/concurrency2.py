"""Round 6, second concurrency experiment: how many chains of short launches does the device really run at once?

N contexts (the library's stream bank deals their main streams to the hardware queues) each replay the SAME captured chain back to back; the aggregate rate of launch
pairs over N says how the chains share the device:
    emptyLM  empty 512-workgroup + empty 1-workgroup launches   (dispatch + boundary only: the command processor's ceiling)
    lm       the real LM chain (pass + controller pairs)         (latency-bound alone: 2 wavefronts per SIMD for ~5 us, then one workgroup for ~5 us)
    passes   the real pass kernel alone
    k5       the K5 chain (sort + tree + packet walk + tail), eager launches
A chain that is latency-bound alone should scale ~N until the issue slots or the queues run out. Usage: python profiles/tools/concurrency2.py > gpurun_out/conc2.json"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from rolo_amd import synth  # noqa: E402
from rolo_amd.rotvgicp import RotVGICP  # noqa: E402

PAIRS = 31
GUESS = -np.asarray(synth.PREV_STEP_T, np.float64)


def make_ctx(d):
    g = RotVGICP(0)
    g.setResolution(0.5); g.setFixedIterations(20); g.setLoadHint(1)
    g.setInputTargetDevice(d[1].data_ptr(), d[2], 4); g.setInputSourceDevice(d[0].data_ptr(), d[2], 4)
    g.register_async(None, np.zeros(3), GUESS, GUESS * 0.97); g.register_wait()
    return g


def main():
    src, tgt, _ = synth.dense_pair("os1-128", seed=synth.SEED)
    d = (torch.from_numpy(src).cuda(), torch.from_numpy(tgt).cuda(), src.shape[0])
    ctxs = [make_ctx(d) for _ in range(8)]
    out = {}
    for name, kind, reps in (("emptyLM", 2, 200), ("lm", 3, 80), ("passes", 4, 80)):
        for c in ctxs:
            c.debug_chain(kind, PAIRS, 512, 2); c.synchronize()
        res = {}
        for n in (1, 2, 3, 4, 6, 8):
            t0 = time.perf_counter()
            for r in range(0, reps, 10):          # interleave the contexts' enqueues so that all queues fill together
                for c in ctxs[:n]:
                    c.debug_chain(kind, PAIRS, 512, 10)
            for c in ctxs[:n]:
                c.synchronize()
            dt = time.perf_counter() - t0
            res[str(n)] = {"us_per_pair_per_context": 1e6 * dt / (reps * PAIRS), "pairs_per_us_aggregate": n * reps * PAIRS / (1e6 * dt)}
        out[name] = res
        print(name, json.dumps(res), file=sys.stderr, flush=True)
    res = {}
    for n in (1, 2, 3, 4, 6, 8):
        reps = 40
        for c in ctxs[:n]:
            c.synchronize()
        t0 = time.perf_counter()
        for r in range(reps):
            for c in ctxs[:n]:
                c.setInputTargetDevice(d[1].data_ptr(), d[2], 4); c.setInputSourceDevice(d[0].data_ptr(), d[2], 4)
                c.computeCovariances()
        for c in ctxs[:n]:
            c.synchronize()
        dt = time.perf_counter() - t0
        res[str(n)] = {"ms_per_chain_per_context": 1e3 * dt / reps, "chains_per_ms_aggregate": n * reps / (1e3 * dt)}
    out["k5"] = res
    print("k5", json.dumps(res), file=sys.stderr, flush=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
