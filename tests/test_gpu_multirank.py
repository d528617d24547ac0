"""GPU tests of the N > 1 path on a one-GPU box.

* K5 sharded by query point (every rank sorts the whole cloud and searches 1/W of the Morton-sorted queries): the union of the
  ranks' covariance slices equals the unsharded result, bit for bit (test hook rolo_set_shard_knn: no communicator, no all-gather);
* the whole collective schedule — ncclAllGather of the covariances, ncclAllReduce of the 32 fp64 per pass — with TWO ranks in two
  processes that share the one device. RCCL may refuse two ranks on one device ("Duplicate GPU"); the test then skips and says
  so: the one-rank communicator test (test_gpu_registration.py::test_rccl_path_on_one_rank) still drives every collective call.
"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from rolo_amd import synth  # noqa: E402

pytestmark = pytest.mark.gpu

G = -np.asarray(synth.PREV_STEP_T)
L0 = G * 0.97


def _pair():
    src, tgt, _ = synth.dense_pair("os1-64", col_stride=4)
    return src, tgt


@pytest.mark.parametrize("world", [2, 3, 8])
def test_knn_sharded_by_query_point_union_is_exact(world):
    from rolo_amd.rotvgicp import RotVGICP
    from rolo_amd._lib import lib, check
    src, tgt = _pair()
    ref = RotVGICP(); ref.setResolution(1.0); ref.setInputTarget(tgt); ref.setInputSource(src)
    ref.computeCovariances()
    cs, ct = ref.getSourceCovariances(), ref.getTargetCovariances()
    seen_s = np.zeros(src.shape[0], int); seen_t = np.zeros(tgt.shape[0], int)
    ns, nt = src.shape[0], tgt.shape[0]

    def slice_size(n, r):   # equal slices of whole 256-query workgroups of the Morton-sorted positions (padding sorts last)
        blocks = (16 * ((n + 15) // 16) + 255) // 256   # whole leaves of 16 points
        chunk = -(-blocks // world) * 256
        return max(0, min((r + 1) * chunk, n) - r * chunk)

    for r in range(world):
        g = RotVGICP(); g.setResolution(1.0)
        g.setInputTarget(tgt); g.setInputSource(src)
        # poison the covariance buffers, then let the rank compute its slice only
        g.setSourceCovariances(np.full((ns, 4, 4), np.nan)); g.setTargetCovariances(np.full((nt, 4, 4), np.nan))
        check(lib().rolo_set_shard_knn(g._h, 1), "rolo_set_shard_knn")
        check(lib().rolo_set_shard(g._h, r, world), "rolo_set_shard")
        g.computeCovariances()
        gs, gt = g.getSourceCovariances(), g.getTargetCovariances()
        own_s = ~np.isnan(gs[:, 0, 0]); own_t = ~np.isnan(gt[:, 0, 0])
        assert int(own_s.sum()) == slice_size(ns, r) and int(own_t.sum()) == slice_size(nt, r)
        assert np.array_equal(gs[own_s], cs[own_s]) and np.array_equal(gt[own_t], ct[own_t])   # bit for bit
        seen_s += own_s; seen_t += own_t
        g.close()
    assert (seen_s == 1).all() and (seen_t == 1).all()   # the slices tile the clouds exactly


def _rank_main(rank, world, uid_path, out_path):
    import time
    import numpy as np
    from rolo_amd.rotvgicp import RotVGICP
    src, tgt = _pair()
    if rank == 0:
        uid = RotVGICP.comm_unique_id()
        with open(uid_path + ".tmp", "wb") as f:
            f.write(uid)
        os.replace(uid_path + ".tmp", uid_path)
    else:
        t0 = time.time()
        while not os.path.exists(uid_path):
            if time.time() - t0 > 60:
                raise SystemExit(3)
            time.sleep(0.05)
        uid = open(uid_path, "rb").read()
    g = RotVGICP(0); g.setResolution(1.0)
    try:
        g.comm_init(uid, rank, world)
    except Exception as e:  # RCCL refuses duplicate devices on some builds
        np.save(out_path, np.array([np.nan]))
        open(out_path + ".err", "w").write(repr(e))
        return
    res = []
    for _ in range(2):
        g.setInputTarget(tgt); g.setInputSource(src)
        g.register_async(None, np.zeros(3), G, L0)
        Tf, Td, t = g.register_wait()
        res.append(np.concatenate([Td.reshape(-1), t]))
    cov = g.getSourceCovariances()
    np.save(out_path, np.concatenate([np.concatenate(res), cov.reshape(-1)[:4096]]))


def test_two_ranks_one_device_rccl_allgather_allreduce(tmp_path):
    import subprocess
    from rolo_amd.rotvgicp import RotVGICP
    world = 2
    uid_path = str(tmp_path / "uid.bin")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", NCCL_DEBUG="WARN")
    code = "import sys; sys.path.insert(0, %r); from tests.test_gpu_multirank import _rank_main; _rank_main(int(sys.argv[1]), %d, %r, sys.argv[2])" % (ROOT, world, uid_path)
    procs = [subprocess.Popen([sys.executable, "-c", code, str(r), str(tmp_path / f"out{r}.npy")], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
             for r in range(world)]
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=150)[0].decode(errors="replace"))
    except subprocess.TimeoutExpired:
        for p in procs:
            p.kill()
        pytest.skip("two RCCL ranks on one device did not rendezvous within 150 s on this box (needs two GPUs)")
    if any(p.returncode != 0 for p in procs):
        pytest.skip("two RCCL ranks on one device: rank process failed: " + " | ".join(o[-300:] for o in outs))
    res = [np.load(tmp_path / f"out{r}.npy") for r in range(world)]
    if any(np.isnan(r[0]) for r in res):
        pytest.skip("RCCL refuses two ranks on one device here: " + open(str(tmp_path / "out0.npy") + ".err").read()[:300])
    src, tgt = _pair()
    ref = RotVGICP(); ref.setResolution(1.0); ref.setInputTarget(tgt); ref.setInputSource(src)
    ref.register_async(None, np.zeros(3), G, L0); Tf0, Td0, t0 = ref.register_wait()
    want = np.concatenate([Td0.reshape(-1), t0])
    for r in res:
        for k in range(2):
            got = r[19 * k:19 * (k + 1)]
            assert np.abs(got - want).max() < 1e-11
        # the all-gathered covariances are the unsharded ones, bit for bit
        assert np.array_equal(r[38:], ref.getSourceCovariances().reshape(-1)[:4096])
    assert np.array_equal(res[0], res[1])  # identical LM decisions on both ranks


# ---- peer exchange (rolo_peer_*): the sharded path without a collective library -------------------------------------------------------
def _peer_rank_body(g, rank, world, exchange_handles, frames):
    """What every rank runs: export, swap handles, connect, then `frames` whole-frame registrations + the stage-level calls.
    Returns a flat vector of everything that must agree across the ranks and with the unsharded run."""
    from rolo_amd._lib import lib
    src, tgt = _pair()
    g.setResolution(1.0)
    h = g.peer_export(world, src.shape[0] + tgt.shape[0])
    handles = exchange_handles(rank, h)
    g.peer_connect(handles, rank, world)
    assert g.peer_info()[:2] == (rank, world)
    us = g.peer_selftest(4)   # known words through both exchanges before the first frame (collective: every rank calls it)
    assert us[0] > 0 and us[1] > 0
    out = []
    for _ in range(frames):   # frames 1 and 2 eager (the schedule length settles), frame 3 captured, frame 4 replayed from the hipGraph
        g.setInputTarget(tgt.copy()); g.setInputSource(src.copy())   # a new array object: setInput* really uploads, K5 runs every frame
        g.register_async(None, np.zeros(3), G, L0)
        Tf, Td, t = g.register_wait()
        out.append(np.concatenate([Td.reshape(-1), t, [g.last_stats.n_passes, g.last_translation_stats.n_passes]]))
    # stage-level evaluations go through peer_allreduce_kernel
    T = np.eye(4); T[:3, :3] = synth.rpy_to_R(0.002, -0.001, 0.004)
    e, H, b = g.so3_linearize(T)
    e2 = g.compute_error(T)
    out.append(np.concatenate([H.reshape(-1), b, [e, e2]]))
    cov = np.concatenate([g.getSourceCovariances().reshape(-1), g.getTargetCovariances().reshape(-1)])
    cnt = g.counters()
    assert cnt["frames"] == frames and cnt["graph_replays"] >= 1 and cnt["topup_frames"] == 0, cnt   # the sharded frame does replay its hipGraph
    return np.concatenate(out), cov


def _unsharded_reference(frames):
    from rolo_amd.rotvgicp import RotVGICP
    src, tgt = _pair()
    ref = RotVGICP(); ref.setResolution(1.0)
    out = []
    for _ in range(frames):
        ref.setInputTarget(tgt.copy()); ref.setInputSource(src.copy())
        ref.register_async(None, np.zeros(3), G, L0); Tf, Td, t = ref.register_wait()
        out.append(np.concatenate([Td.reshape(-1), t, [ref.last_stats.n_passes, ref.last_translation_stats.n_passes]]))
    T = np.eye(4); T[:3, :3] = synth.rpy_to_R(0.002, -0.001, 0.004)
    e, H, b = ref.so3_linearize(T); e2 = ref.compute_error(T)
    out.append(np.concatenate([H.reshape(-1), b, [e, e2]]))
    cov = np.concatenate([ref.getSourceCovariances().reshape(-1), ref.getTargetCovariances().reshape(-1)])
    ref.close()
    return np.concatenate(out), cov


def _check_against_unsharded(res, frames):
    want, wcov = _unsharded_reference(frames)
    for vec, cov in res:
        assert np.array_equal(cov, wcov)                      # exchanged covariances: the unsharded ones, bit for bit
        rel = np.abs(vec - want) / np.maximum(np.abs(want), 1.0)
        assert rel[:-14].max() < 1e-9, rel[:-14].max()        # poses, translations, pass counts (sums differ in the order of additions only)
        assert rel[-14:].max() < 1e-9, rel[-14:].max()        # H, b, err of the stage-level calls
    for vec, cov in res[1:]:
        assert np.array_equal(vec, res[0][0])                 # every rank took bit-identical LM decisions


@pytest.mark.parametrize("world", [2])
def test_peer_exchange_contexts_of_one_process(world):
    """Two contexts of ONE process on the one device, one host thread per rank (the handles resolve through the process-local registry).
    (Not more than two: HIP deals a process's streams to four hardware queues, a third context's stream would sit BEHIND the first
    one's on the same queue and its kernels could not start while that one polls for them — ranks of a real job are processes.)"""
    import threading
    from rolo_amd.rotvgicp import RotVGICP
    bar = threading.Barrier(world)
    table = [None] * world
    res = [None] * world
    errs = []

    def exchange(rank, h):
        table[rank] = h
        bar.wait(timeout=60)
        return list(table)

    def body(rank):
        try:
            g = RotVGICP(0)
            res[rank] = _peer_rank_body(g, rank, world, exchange, frames=4)
            bar.wait(timeout=120)   # nobody frees its mailbox while a peer may still write into it
            g.close()
        except Exception as e:  # noqa: BLE001
            errs.append((rank, repr(e)))
            bar.abort()

    th = [threading.Thread(target=body, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=300)
    assert not errs, errs
    _check_against_unsharded(res, 4)


def _peer_proc_main(rank, world, dirpath):
    import time
    from rolo_amd.rotvgicp import RotVGICP

    def exchange(r, h):
        with open(os.path.join(dirpath, f"h{r}.tmp"), "wb") as f:
            f.write(h)
        os.replace(os.path.join(dirpath, f"h{r}.tmp"), os.path.join(dirpath, f"h{r}.bin"))
        hs = []
        for q in range(world):
            p = os.path.join(dirpath, f"h{q}.bin"); t0 = time.time()
            while not os.path.exists(p):
                if time.time() - t0 > 120:
                    raise SystemExit(3)
                time.sleep(0.02)
            hs.append(open(p, "rb").read())
        return hs

    g = RotVGICP(0)
    vec, cov = _peer_rank_body(g, rank, world, exchange, frames=4)
    np.save(os.path.join(dirpath, f"vec{rank}.npy"), vec); np.save(os.path.join(dirpath, f"cov{rank}.npy"), cov)
    kind = g.peer_info()[2]
    open(os.path.join(dirpath, f"done{rank}"), "w").write(kind)
    t0 = time.time()   # keep the mailbox alive until every rank is done
    while not all(os.path.exists(os.path.join(dirpath, f"done{q}")) for q in range(world)) and time.time() - t0 < 120:
        time.sleep(0.02)
    g.close()


@pytest.mark.parametrize("world", [2, 3, 4, 8])
def test_ranks_in_processes_one_device_peer_exchange(tmp_path, world):
    """W processes (up to the 8 of a node: every slot of the mailbox layout), hipIpc handles, one device: the whole N > 1 path — K5 by query slice + peer-written covariance exchange, passes by
    source shard + mailbox all-reduce in the controller — against the unsharded registration."""
    import subprocess
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", ROLO_PEER_TIMEOUT_MS="20000")
    code = "import sys; sys.path.insert(0, %r); from tests.test_gpu_multirank import _peer_proc_main; _peer_proc_main(int(sys.argv[1]), %d, %r)" % (ROOT, world, str(tmp_path))
    procs = [subprocess.Popen([sys.executable, "-c", code, str(r)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(world)]
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=240)[0].decode(errors="replace"))
    except subprocess.TimeoutExpired:
        for p in procs:
            p.kill()
        pytest.fail("peer ranks did not finish within 240 s")
    assert all(p.returncode == 0 for p in procs), " | ".join(o[-600:] for o in outs)
    res = [(np.load(tmp_path / f"vec{r}.npy"), np.load(tmp_path / f"cov{r}.npy")) for r in range(world)]
    _check_against_unsharded(res, 4)


def test_peer_connect_refuses_a_mailbox_of_another_size():
    """Every rank pushes whole segments into every peer's exchange area: a peer that exported for fewer points must be refused at connect"""
    from rolo_amd.rotvgicp import RotVGICP
    from rolo_amd._lib import RoloError
    a, b = RotVGICP(0), RotVGICP(0)
    ha, hb = a.peer_export(2, 100000), b.peer_export(2, 20000)
    with pytest.raises(RoloError) as ei:
        a.peer_connect([ha, hb], 0, 2)
    assert ei.value.code == -1 and "max_points" in str(ei.value)
    hb = b.peer_export(2, 100000)   # exported again with matching arguments: fine
    a.peer_connect([ha, hb], 0, 2); b.peer_connect([ha, hb], 1, 2)
    assert a.peer_info()[:2] == (0, 2) and b.peer_info()[:2] == (1, 2)
    a.close(); b.close()


def test_peer_timeout_is_an_error_not_a_hang(monkeypatch):
    """A rank whose peer never shows up gives up after ROLO_PEER_TIMEOUT_MS with ROLO_ECOMM (-9)."""
    import time
    from rolo_amd.rotvgicp import RotVGICP
    from rolo_amd._lib import RoloError
    monkeypatch.setenv("ROLO_PEER_TIMEOUT_MS", "150")
    src, tgt = _pair()
    a, b = RotVGICP(0), RotVGICP(0)
    for g in (a, b):
        g.setResolution(1.0)
    ha, hb = a.peer_export(2, src.shape[0] + tgt.shape[0]), b.peer_export(2, src.shape[0] + tgt.shape[0])
    a.peer_connect([ha, hb], 0, 2)   # b never connects, never runs
    a.setInputTarget(tgt); a.setInputSource(src)
    t0 = time.time()
    with pytest.raises(RoloError) as ei:
        a.register_async(None, np.zeros(3), G, L0)
        a.register_wait()
    assert time.time() - t0 < 20
    assert ei.value.code == -9
    a.close(); b.close()


# ---- BASELINE configs[3] at its own size: the 262 144-point frame (128 x 2048), leaf 0.5 m, 20 iterations, sharded over W rank processes ---------
def _digest(a):
    import hashlib
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def _config3_rank_main(rank, world, dirpath):
    """one rank of the sharded 262k-point frame: loads the pair the parent generated, runs 4 frames (eager, eager, captured, replayed),
    leaves pose / translation / pass counts per frame and digests of the exchanged covariances"""
    import json
    import time
    from rolo_amd.rotvgicp import RotVGICP
    src = np.load(os.path.join(dirpath, "src.npy")); tgt = np.load(os.path.join(dirpath, "tgt.npy"))

    def exchange(r, h):
        with open(os.path.join(dirpath, f"h{r}.tmp"), "wb") as f:
            f.write(h)
        os.replace(os.path.join(dirpath, f"h{r}.tmp"), os.path.join(dirpath, f"h{r}.bin"))
        hs = []
        for q in range(world):
            p = os.path.join(dirpath, f"h{q}.bin"); t0 = time.time()
            while not os.path.exists(p):
                if time.time() - t0 > 180:
                    raise SystemExit(3)
                time.sleep(0.02)
            hs.append(open(p, "rb").read())
        return hs

    g = RotVGICP(0); g.setResolution(0.5); g.setFixedIterations(20)
    g.peer_connect(exchange(rank, g.peer_export(world, src.shape[0] + tgt.shape[0])), rank, world)
    st = g.peer_selftest(4)
    frames = []
    for _ in range(4):
        g.setInputTarget(tgt.copy()); g.setInputSource(src.copy())
        g.register_async(None, np.zeros(3), G, L0)
        Tf, Td, t = g.register_wait()
        frames.append(dict(Td=Td.reshape(-1).tolist(), t=t.tolist(), rot_outer=g.last_stats.n_outer, passes=[g.last_stats.n_passes, g.last_translation_stats.n_passes]))
    res = dict(rank=rank, frames=frames, counters=g.counters(), selftest_us=list(st), mailbox=g.peer_info()[2],
               cov_src=_digest(g.getSourceCovariances()), cov_tgt=_digest(g.getTargetCovariances()))
    json.dump(res, open(os.path.join(dirpath, f"res{rank}.json"), "w"))
    open(os.path.join(dirpath, f"done{rank}"), "w").close()
    t0 = time.time()   # keep the mailbox alive until every rank is done
    while not all(os.path.exists(os.path.join(dirpath, f"done{q}")) for q in range(world)) and time.time() - t0 < 180:
        time.sleep(0.02)
    g.close()


def _rot_angle(Ra, Rb):
    c = (np.trace(Ra.T @ Rb) - 1.0) / 2.0
    return float(np.arccos(np.clip(c, -1.0, 1.0))) if c < 1.0 - 1e-12 else float(np.sqrt(max(0.0, 2.0 * (1.0 - c))))


@pytest.mark.parametrize("world", [2, 8])
def test_sharded_262k_frame_matches_oracle(tmp_path, world):
    """BASELINE configs[3] ("OS1-128 262k pts point-sharded across 8 x MI355X") at its workload: W rank processes over hipIpc handles (one device here,
    one GPU each on a node), K5 by query slice + peer-written covariance exchange, passes by source shard + mailbox all-reduce in the controller,
    hipGraph replayed — every rank's pose against the ORACLE (<= 1e-5 rad / <= 1e-4 m, the north-star bar), exchanged covariances bit-identical to the
    unsharded ones, all ranks bit-identical to each other. Reference loop that is split: rot_vgicp_impl.hpp:313-382."""
    import json
    import subprocess
    from oracle import pyorc
    from rolo_amd.rotvgicp import RotVGICP
    src, tgt, _ = synth.dense_pair("os1-128x2048")
    assert src.shape[0] == 262144 and tgt.shape[0] == 262144
    np.save(tmp_path / "src.npy", src); np.save(tmp_path / "tgt.npy", tgt)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", ROLO_PEER_TIMEOUT_MS="60000")
    code = "import sys; sys.path.insert(0, %r); from tests.test_gpu_multirank import _config3_rank_main; _config3_rank_main(int(sys.argv[1]), %d, %r)" % (ROOT, world, str(tmp_path))
    procs = [subprocess.Popen([sys.executable, "-c", code, str(r)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(world)]
    # while the ranks run: the oracle on the same frame (CPU), then the unsharded HIP run once they are done
    o = pyorc.Reg(pyorc.default_params(voxel_type=1, voxel_resolution=0.5, fixed_iterations=20)); o.set_target(tgt); o.set_source(src)
    rc, _, Td_o, _, _ = o.align(); assert rc == 0
    rc, t_o, _ = o.compute_translation(np.zeros(3), G, L0); assert rc == 0
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=420)[0].decode(errors="replace"))
    except subprocess.TimeoutExpired:
        for p in procs:
            p.kill()
        pytest.fail("sharded ranks did not finish within 420 s")
    assert all(p.returncode == 0 for p in procs), " | ".join(o_[-600:] for o_ in outs)
    res = [json.load(open(tmp_path / f"res{r}.json")) for r in range(world)]
    ref = RotVGICP(); ref.setResolution(0.5); ref.setFixedIterations(20)
    ref.setInputTarget(tgt); ref.setInputSource(src)
    ref.register_async(None, np.zeros(3), G, L0); _, Td_u, t_u = ref.register_wait()
    cs, ct = _digest(ref.getSourceCovariances()), _digest(ref.getTargetCovariances())
    ref.close()
    for r in res:
        assert r["cov_src"] == cs and r["cov_tgt"] == ct                       # exchanged covariances = the unsharded ones, bit for bit
        assert r["counters"]["frames"] == 4 and r["counters"]["graph_replays"] >= 1 and r["counters"]["topup_frames"] == 0, r["counters"]
        for f in r["frames"]:
            Td = np.array(f["Td"]).reshape(4, 4); t = np.array(f["t"])
            assert f["rot_outer"] == 20
            assert _rot_angle(Td[:3, :3], Td_o[:3, :3]) <= 1e-5 and np.abs(t - t_o).max() <= 1e-4      # vs the ORACLE
            assert np.abs(Td - Td_u).max() < 1e-9 and np.abs(t - t_u).max() < 1e-9                      # vs the unsharded HIP run (order of additions)
        assert r["frames"] == res[0]["frames"]                                  # every rank, every frame: bit-identical decisions


def _distinct_rank_main(rank, world, dirpath):
    """a DIFFERENT pair of equal size every frame; rank 1 never uses the hipGraph (use_graph = 0), rank 0 captures on its third frame — the ranks' captures
    are out of step, which a host-side choice of the exchange area baked into a capture did not survive (round 3's advisor finding)"""
    import time
    from rolo_amd.rotvgicp import RotVGICP
    pairs = [synth.dense_pair("os1-64", col_stride=4, origin=synth.pool_origin(i)) for i in range(3)]

    def exchange(r, h):
        with open(os.path.join(dirpath, f"h{r}.tmp"), "wb") as f:
            f.write(h)
        os.replace(os.path.join(dirpath, f"h{r}.tmp"), os.path.join(dirpath, f"h{r}.bin"))
        hs = []
        for q in range(world):
            p = os.path.join(dirpath, f"h{q}.bin"); t0 = time.time()
            while not os.path.exists(p):
                if time.time() - t0 > 120:
                    raise SystemExit(3)
                time.sleep(0.02)
            hs.append(open(p, "rb").read())
        return hs

    g = RotVGICP(0); g.setResolution(1.0)
    if rank == 1:
        g.setUseGraph(False)
    n = pairs[0][0].shape[0] + pairs[0][1].shape[0]
    g.peer_connect(exchange(rank, g.peer_export(world, n)), rank, world)
    out = []
    for k in range(7):
        src, tgt, _ = pairs[k % 3]
        g.setInputTarget(tgt.copy()); g.setInputSource(src.copy())
        g.register_async(None, np.zeros(3), G, L0)
        Tf, Td, t = g.register_wait()
        cov = np.concatenate([g.getSourceCovariances().reshape(-1), g.getTargetCovariances().reshape(-1)])
        out.append(np.concatenate([Td.reshape(-1), t, [float(int(_digest(cov)[:12], 16))]]))
    cnt = g.counters()
    np.save(os.path.join(dirpath, f"vec{rank}.npy"), np.concatenate(out + [np.array([cnt["graph_replays"], cnt["eager_frames"]], float)]))
    open(os.path.join(dirpath, f"done{rank}"), "w").close()
    t0 = time.time()
    while not all(os.path.exists(os.path.join(dirpath, f"done{q}")) for q in range(world)) and time.time() - t0 < 120:
        time.sleep(0.02)
    g.close()


def test_peer_exchange_distinct_frames_with_captures_out_of_step(tmp_path):
    import subprocess
    from rolo_amd.rotvgicp import RotVGICP
    world = 2
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", ROLO_PEER_TIMEOUT_MS="20000")
    code = "import sys; sys.path.insert(0, %r); from tests.test_gpu_multirank import _distinct_rank_main; _distinct_rank_main(int(sys.argv[1]), %d, %r)" % (ROOT, world, str(tmp_path))
    procs = [subprocess.Popen([sys.executable, "-c", code, str(r)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(world)]
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=240)[0].decode(errors="replace"))
    except subprocess.TimeoutExpired:
        for p in procs:
            p.kill()
        pytest.fail("peer ranks did not finish within 240 s")
    assert all(p.returncode == 0 for p in procs), " | ".join(o[-600:] for o in outs)
    res = [np.load(tmp_path / f"vec{r}.npy") for r in range(world)]
    assert res[0][-2] >= 1 and res[1][-2] == 0 and res[1][-1] == 7     # rank 0 replayed its graph, rank 1 ran every frame eagerly
    pairs = [synth.dense_pair("os1-64", col_stride=4, origin=synth.pool_origin(i)) for i in range(3)]
    assert len({p[0].shape[0] for p in pairs}) == 1 and not np.array_equal(pairs[0][0], pairs[1][0])
    want = []
    ref = RotVGICP(); ref.setResolution(1.0)
    for k in range(7):
        src, tgt, _ = pairs[k % 3]
        ref.setInputTarget(tgt.copy()); ref.setInputSource(src.copy())
        ref.register_async(None, np.zeros(3), G, L0); _, Td, t = ref.register_wait()
        cov = np.concatenate([ref.getSourceCovariances().reshape(-1), ref.getTargetCovariances().reshape(-1)])
        want.append(np.concatenate([Td.reshape(-1), t, [float(int(_digest(cov)[:12], 16))]]))
    ref.close()
    want = np.concatenate(want)
    for r in res:
        got = r[:-2].reshape(7, 20); w = want.reshape(7, 20)
        assert np.array_equal(got[:, 19], w[:, 19])                     # every frame's exchanged covariances are THAT frame's, bit for bit
        assert np.abs(got[:, :19] - w[:, :19]).max() < 1e-9
    assert np.array_equal(res[0][:-2], res[1][:-2])


def test_peer_selftest_needs_a_connection():
    from rolo_amd.rotvgicp import RotVGICP
    from rolo_amd._lib import RoloError
    g = RotVGICP(0)
    with pytest.raises(RoloError) as ei:
        g.peer_selftest(2)
    assert ei.value.code == -5 and "not connected" in str(ei.value)
    g.close()
