"""One point-sharded frame (BASELINE configs[3]) through the peer exchange (rolo_peer_*) or through RCCL (rolo_comm_*), one process (or thread) per
rank, timed — the body shared by bench.py's `sharded` leg and profiles/tools/peer2proc.py. The ranks meet through files in a directory (the 64-byte
mailbox handles / the 128-byte RCCL unique id, barriers): no torch, nothing but the C ABI. Measurement harness, not product code.

    python -m rolo_amd.peerbench <rank> <world> <dir> <sensor> <frames> <leaf> <device> [peer|rccl]
"""
import json
import os
import sys
import time

import numpy as np


def file_barrier(d, name, rank, world, timeout=300):
    open(os.path.join(d, f"{name}.{rank}"), "w").close()
    t0 = time.time()
    while not all(os.path.exists(os.path.join(d, f"{name}.{q}")) for q in range(world)):
        if time.time() - t0 > timeout:
            raise SystemExit(f"barrier {name} timed out")
        time.sleep(0.002)


def rank_main(rank, world, d, sensor, frames, leaf, device=0, exchange="peer"):
    from . import synth, profile
    from .rotvgicp import RotVGICP
    src, tgt, _ = synth.dense_pair(sensor, seed=synth.SEED)
    G = -np.asarray(synth.PREV_STEP_T); L0 = G * 0.97
    g = RotVGICP(device); g.setResolution(leaf); g.setFixedIterations(20)
    selftest = None
    if world > 1 and exchange == "rccl":
        # north_star's form: ncclAllGather of the covariances + one ncclAllReduce of the sums per LM pass (rolo_comm_init dlopens librccl); the
        # unique id travels through a file like the mailbox handles do
        if rank == 0:
            uid = RotVGICP.comm_unique_id()
            with open(os.path.join(d, "uid.tmp"), "wb") as f:
                f.write(uid)
            os.replace(os.path.join(d, "uid.tmp"), os.path.join(d, "uid.bin"))
        file_barrier(d, "uid", rank, world)
        g.comm_init(open(os.path.join(d, "uid.bin"), "rb").read(), rank, world)
    elif world > 1:
        h = g.peer_export(world, 2 * src.shape[0])
        with open(os.path.join(d, f"h{rank}.tmp"), "wb") as f:
            f.write(h)
        os.replace(os.path.join(d, f"h{rank}.tmp"), os.path.join(d, f"h{rank}.bin"))
        file_barrier(d, "exported", rank, world)
        g.peer_connect([open(os.path.join(d, f"h{q}.bin"), "rb").read() for q in range(world)], rank, world)
        # both exchanges with known words BEFORE any frame: the first crossing of the link (hipIpc mapping, peer access, fine-grained memory over xGMI)
        # fails here with a named error (rolo_peer_selftest) instead of as a wrong pose or a time-out inside a frame
        selftest = g.peer_selftest(16)

    # inputs resident in HBM, through the HIP runtime directly (no torch in the rank processes: torch's own streams / queues on top of W
    # processes oversubscribe the one device's hardware queues and every kernel then pays a queue switch — measured: 61 us per 12 us pass)
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    assert hip.hipSetDevice(device) == 0   # this rank's GPU (the input buffers below must live where the context does)

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
           "pose_head": Td.reshape(-1)[:4].tolist(), "mailbox": g.peer_info()[2] if (world > 1 and exchange == "peer") else "", "exchange": exchange}
    res["device"] = device
    if world > 1 and exchange == "rccl":
        res["rccl_ranks"] = int(g.comm_info()[1])   # ncclCommCount of the communicator the frames ran on
    if selftest is not None:
        res["selftest"] = {"ok": True, "lm_exchange_us": selftest[0], "cov_exchange_us": selftest[1]}
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




if __name__ == "__main__":
    a = sys.argv[1:]
    rank_main(int(a[0]), int(a[1]), a[2], a[3], int(a[4]), float(a[5]), int(a[6]) if len(a) > 6 else 0, a[7] if len(a) > 7 else "peer")
