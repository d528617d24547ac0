"""Round 6 (round 5's verdict, item 2b): what does a second context's chain of short launches do to a neighbour search running beside it?

Context A loops the K5 chain of the bench's dense pair (sort + tree + 64-query packet walk + covariance tail, eager launches, every walk timed with HIP events on
A's stream) while context B replays, from a captured graph on ANOTHER hardware queue,
    empty1     launch pairs of an empty kernel, 1 workgroup each           (dispatch + kernel boundary only, the controller's shape)
    empty512   launch pairs of an empty kernel, 512 workgroups each        (the pass's shape)
    emptyLM    empty 512-workgroup + empty 1-workgroup launches alternating (the LM chain's boundaries without its instructions or traffic)
    passes     the real pass kernel alone (every launch linearises; no controller)
    lm         the real LM chain (frame begin + predicated pass / controller pairs)
    walk       B loops the K5 chain too (two searches side by side)
and, for the layout question, the same with B on A's OWN hardware queue (the bank's third context). Reported per case: A's mean / median walk duration, B's time per
launch pair while A runs and alone. Usage (GPU box):  python profiles/tools/concurrency.py > gpurun_out/concurrency.json"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402  (device buffers)
from rolo_amd import synth  # noqa: E402
from rolo_amd._lib import lib  # noqa: E402
from rolo_amd.rotvgicp import RotVGICP  # noqa: E402
import ctypes as C  # noqa: E402

N_WALKS = int(os.environ.get("CONC_WALKS", "40"))
PAIRS = 31      # a frame's 21 rotation + 10 translation trials
GUESS = -np.asarray(synth.PREV_STEP_T, np.float64)


def make_ctx(d):
    g = RotVGICP(0)
    g.setResolution(0.5); g.setFixedIterations(20); g.setLoadHint(1)
    g.setInputTargetDevice(d[1].data_ptr(), d[2], 4); g.setInputSourceDevice(d[0].data_ptr(), d[2], 4)
    g.register_async(None, np.zeros(3), GUESS, GUESS * 0.97); g.register_wait()
    return g


def walk_loop(g, d, n):
    """n K5 chains on g's stream (asynchronous): the clouds handed over again (a pack launch each), then rolo_compute_covariances"""
    for _ in range(n):
        g.setInputTargetDevice(d[1].data_ptr(), d[2], 4); g.setInputSourceDevice(d[0].data_ptr(), d[2], 4)
        g.computeCovariances()


def read_prof(g, slot):
    buf = (C.c_float * 4096)()
    k = lib().rolo_prof_read(g._h, slot, buf, 4096)
    return np.array(buf[:max(k, 0)], np.float64)


def main():
    src, tgt, _ = synth.dense_pair("os1-128", seed=synth.SEED)
    d = (torch.from_numpy(src).cuda(), torch.from_numpy(tgt).cuda(), src.shape[0])
    # the bank deals main streams to hardware queues in creation order: contexts 0 and 2 share one queue, 1 and 3 the other (api.hip stream bank)
    A = make_ctx(d); B = make_ctx(d); A2 = make_ctx(d)     # A2: on A's queue
    out = {"n_walks": N_WALKS, "pairs_per_chain": PAIRS, "points_per_cloud": int(d[2])}
    KINDS = {"empty1": (0, 1), "empty512": (1, 512), "emptyLM": (2, 512), "passes": (4, 512), "lm": (3, 512)}

    def chain_alone(ctx, kind, grid, reps=60):
        ctx.debug_chain(kind, PAIRS, grid, 2); ctx.synchronize()
        t0 = time.perf_counter(); ctx.debug_chain(kind, PAIRS, grid, reps); ctx.synchronize()
        return 1e6 * (time.perf_counter() - t0) / (reps * PAIRS)

    def a_walks(load=None):
        """A's walks timed by events; load = a callable that enqueues the other context's work first and returns a finisher"""
        lib().rolo_prof_enable(A._h, 1)
        fin = load() if load else None
        t0 = time.perf_counter()
        walk_loop(A, d, N_WALKS)
        A.synchronize()
        wall = time.perf_counter() - t0
        w = read_prof(A, 1); b_ = read_prof(A, 0); t_ = read_prof(A, 6)
        lib().rolo_prof_enable(A._h, 0)
        r = {"walk_ms_mean": float(w.mean()), "walk_ms_median": float(np.median(w)), "walk_ms_p90": float(np.percentile(w, 90)), "build_ms_mean": float(b_.mean()),
             "tail_ms_mean": float(t_.mean()), "a_wall_ms_per_chain": 1e3 * wall / N_WALKS}
        if fin:
            r.update(fin())
        return r

    walk_loop(A, d, 5); A.synchronize()
    out["A_alone"] = a_walks()
    for other, tag in ((B, "other_queue"), (A2, "same_queue")):
        for name, (kind, grid) in KINDS.items():
            alone_us = chain_alone(other, kind, grid)
            reps = max(40, int(1.6 * N_WALKS * 0.45e3 / max(alone_us * PAIRS, 1.0)))   # outlast A's loop

            def load(other=other, kind=kind, grid=grid, reps=reps):
                t0 = time.perf_counter()
                other.debug_chain(kind, PAIRS, grid, reps)

                def fin():
                    other.synchronize()
                    return {"b_us_per_pair_beside_A": 1e6 * (time.perf_counter() - t0) / (reps * PAIRS), "b_reps": reps}
                return fin
            r = a_walks(load)
            r["b_us_per_pair_alone"] = alone_us
            out[f"{tag}:{name}"] = r
            print(f"{tag}:{name}", json.dumps(r), file=sys.stderr, flush=True)

        def load_walk(other=other):
            t0 = time.perf_counter()
            walk_loop(other, d, N_WALKS)

            def fin():
                other.synchronize()
                return {"b_wall_ms_per_chain": 1e3 * (time.perf_counter() - t0) / N_WALKS}
            return fin
        out[f"{tag}:walk"] = a_walks(load_walk)
    # two LM chains beside A (B on the other queue, A2 on A's own); the walk cases above took B's and A2's maps away: register them again
    for c_ in (B, A2):
        c_.setInputTargetDevice(d[1].data_ptr(), d[2], 4); c_.setInputSourceDevice(d[0].data_ptr(), d[2], 4)
        c_.register_async(None, np.zeros(3), GUESS, GUESS * 0.97); c_.register_wait()

    def load2():
        t0 = time.perf_counter()
        B.debug_chain(3, PAIRS, 512, 60); A2.debug_chain(3, PAIRS, 512, 60)

        def fin():
            B.synchronize(); A2.synchronize()
            return {"both_done_ms": 1e3 * (time.perf_counter() - t0)}
        return fin
    out["two_lm_chains"] = a_walks(load2)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
