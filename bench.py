#!/usr/bin/env python3
"""bench.py — scans/sec of the MI355X-native ROLO scan-matching hot path (contract: see the task statement).

One "step" = one complete registration of a synthetic OS1-128 frame pair (128 x 1024 = 131 072 points each,
the "128k-pt frame" of BASELINE.json) on every registration context kept in flight (default 4 per GPU): per-point
20-NN covariances of both clouds, target voxel-hash build (UNIFORM leaf 0.5 m), SO(3) LM stage forced to exactly 20
outer iterations, continuous-time translation LM stage — all enqueued on the context's HIP stream with the two clouds
already resident in HBM when the timed region starts.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--mode replicas|shard] [--no-cpu] [--sensor os1-128]

Timing: W untimed warm-up steps, then rounds of EXACTLY K steps each, every round bracketed by barrier +
torch.cuda.synchronize() on both sides and reduced with MAX over the ranks; at least 5 rounds and at least 0.5 s of
timed work; `value` / `ms_per_step` are the MEDIAN round (all rounds are listed under "rounds_ms_per_step").

N > 1: `python bench.py --gpus N` re-executes itself under torch.distributed.run (one rank per GPU, RCCL); launched by
torch.distributed.run directly it reads RANK / LOCAL_RANK / WORLD_SIZE. Rank 0 prints the single JSON line.
  replicas (value)   — the path partitions by frame: every rank registers its own frames, no data-path collective,
                       "scaling": "weak"; value = total frames / max-over-ranks time.
  sharded (extra)    — BASELINE configs[3]: ONE OS1-128 2048-column frame (262 144 points), K5 sharded by query point
                       with an all-gather of the covariances, source points of the LM passes sharded with one RCCL
                       all-reduce of 32 fp64 per pass; reported under "sharded" at N > 1.
  config5 (extra)    — BASELINE configs[4]: 512 distinct OS1-64 pairs resident in HBM streamed through the contexts
                       (pairs dealt round-robin to the ranks), hipGraph replay; "config5" in the JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
MIN_TIMED_S = 3.0       # the headline's timed rounds add up to at least this (they run first, before any CPU leg)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--sensor", default="os1-128")
    ap.add_argument("--leaf", type=float, default=0.5)
    ap.add_argument("--mode", default="replicas", choices=["replicas", "shard"])
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replay (profiling runs)")
    ap.add_argument("--no-shard-leg", action="store_true", help="N>1: skip the point-sharded 262k-point leg")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="length of the bounded cpu_baseline sample")
    ap.add_argument("--streams", type=int, default=4,
                    help="registration contexts (HIP streams) kept in flight per GPU; a step is then one frame pair per stream")
    ap.add_argument("--pair-search", default="on", choices=["on", "off"],
                    help="search source and target of a frame in one chain of launches (rolo_params.overlap_knn, the library default)")
    ap.add_argument("--no-pipeline", action="store_true", help="skip the raw-frame -> pose pipeline leg")
    ap.add_argument("--pipeline-only", action="store_true", help="run only the pipeline leg and print its dict (profiling runs)")
    ap.add_argument("--no-config5", action="store_true", help="skip the 512-pair OS1-64 leg (BASELINE configs[4])")
    ap.add_argument("--no-layout-check", action="store_true", help="skip the re-measurement with contexts re-created after foreign streams")
    ap.add_argument("--no-side-legs", action="store_true", help="skip convergence_driven / iteration_sweep / host_clouds (profiling runs)")
    ap.add_argument("--config5-pairs", type=int, default=512)
    ap.add_argument("--single-round", action="store_true", help="one timed round only (profiling runs)")
    ap.add_argument("--pool", type=int, default=8,
                    help="DISTINCT frame pairs resident in HBM that the contexts rotate through (pair 0 = the nominal pair of SURVEY 8d; "
                         "1 = every context registers the same pair over and over, round 2's headline)")
    ap.add_argument("--shard-exchange", default="both", choices=["peer", "rccl", "both", "peer-inproc", "rccl-inproc"],
                    help="sharded leg (N > 1): peer = mailbox exchange inside the controller kernel (rolo_peer_*); rccl = ncclAllGather + ncclAllReduce per pass "
                         "(north_star's form); both (default) = one after the other. Either runs its ranks in CHILD processes (handles / unique id through files, no "
                         "torch: a crash or a hang there cannot take the bench line down); peer-inproc / rccl-inproc = the same inside the torch.distributed ranks. "
                         "--mode shard uses peer-inproc or rccl-inproc")
    ap.add_argument("--sharded-children-test", type=int, default=0,
                    help="self-test on a one-GPU box: run the sharded leg's child ranks (this many) all on device 0, print their results and exit")
    ap.add_argument("--load-hint", default="auto", choices=["auto", "idle", "busy"],
                    help="rolo_set_load_hint of every context: auto = per frame from the device's load (the product default); profiling runs with one context "
                         "pass busy to see the kernels the multi-context headline runs")
    ap.add_argument("--batch", type=int, default=int(os.environ.get("ROLO_BENCH_BATCH", "1")),
                    help="frame pairs per registration call (rolo_batch_*: shared LM launches); 1 = one operator per frame")
    return ap.parse_args()


def usable_cores() -> int:
    """Host cores this process may actually use: affinity mask capped by the cgroup CPU quota (the GPU boxes show
    256 CPUs but run under a 16-CPU quota; 256 OpenMP threads there are 60x slower than 16)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except Exception:
        pass
    return max(1, n)


def _free_port() -> int:
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def maybe_spawn(args):
    """`python bench.py --gpus N` without a launcher: re-execute under torch.distributed.run, one rank per GPU."""
    if args.gpus <= 1 or "WORLD_SIZE" in os.environ:
        return
    import torch
    avail = torch.cuda.device_count()
    if avail < args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus}: only {avail} GPU(s) visible on this node")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    raise SystemExit(subprocess.call(cmd, env=env))


def pipeline_leg(args, torch, device):
    """Secondary measurement, not `value`: K1 -> K13 per frame on raw OS1-128 frames resident in HBM (feature thinning,
    POLAR voxels 0.175/0.175/2.0 — what the ROS nodes do), one frame at a time as LidarOdometry must."""
    from rolo_amd import synth
    from rolo_amd.frontend import front_params
    from rolo_amd.odometry import LidarOdometry
    sensor = args.sensor
    S = synth.SENSORS[sensor]
    fp = front_params(n_scan=S[0], horizon_scan=S[1])
    R = np.eye(3); t = np.zeros(3); frames = []
    for k in range(5):  # short trajectory, replayed back and forth so consecutive frames stay neighbours
        fr = synth.make_frame(sensor, R, t, synth.SEED + k)
        frames.append((torch.from_numpy(np.ascontiguousarray(fr.xyz, np.float32)).cuda(),
                       torch.from_numpy(np.ascontiguousarray(fr.ring, np.uint16).view(np.int16)).cuda(), fr.xyz.shape[0], fr.xyz.shape[1]))
        t = t + R @ np.array([0.3, 0.02 * k, 0.0]); R = R @ synth.rpy_to_R(np.deg2rad(0.3), np.deg2rad(-0.2), np.deg2rad(2.0))
    order = [0, 1, 2, 3, 4, 3, 2, 1]
    res = {"workload": f"{sensor} raw frames ({frames[0][2]} points) -> projection -> features -> RotVGICP (POLAR voxels) -> pose"}
    # frame-at-a-time (latency), pipelined (rolo_odom_submit of frame k+1 before rolo_odom_collect of frame k: K1-K4 on
    # their own stream overlap the registration), and pipelined with covariance hand-over between frames (off by default)
    for name, piped, reuse in (("frame_latency_ms", 0, 0), ("scans_per_s", 1, 0), ("scans_per_s_reuse_covariances", 1, 1)):
        od = LidarOdometry(device, 0.3)
        od.setOption(LidarOdometry.REUSE_COVARIANCES, reuse)
        stamp = 100.0; cnt = None

        def submit(i):
            nonlocal stamp
            x, r, n_raw, stride = frames[order[i % len(order)]]
            stamp += 0.1
            od.submit(fp, stamp, x.data_ptr(), r.data_ptr(), n_raw=n_raw, stride=stride)

        def run(i0, i1):
            nonlocal cnt
            if piped:
                submit(i0)
            for i in range(i0, i1):
                if piped:
                    if i + 1 < i1:
                        submit(i + 1)
                else:
                    submit(i)
                rc, _, _, _, cnt = od.collect()
        run(0, 1); od.odometryHandler(stamp + 0.05)
        run(1, 6)
        torch.cuda.synchronize()
        nfr = 80; reps = []; i0 = 6
        import gc
        for _ in range(5):   # median of five rounds of 80 frames (round 2 timed ONE round of 40 frames = 25 ms of work: a single CPython cyclic
            gc.collect(); gc.disable()   # collection — ~40 ms with torch imported — inside it made a leg read 650 instead of 2000 frames/s)
            t0 = time.perf_counter()
            run(i0, i0 + nfr); i0 += nfr
            torch.cuda.synchronize()
            reps.append(time.perf_counter() - t0)
            gc.enable()
        dt = float(np.median(reps))
        res[name] = 1e3 * dt / nfr if name.endswith("_ms") else nfr / dt
        res.setdefault("rounds", {})[name] = [round(1e3 * r / nfr, 4) if name.endswith("_ms") else round(nfr / r, 1) for r in reps]
        res["features_per_frame"] = int(cnt[1] + cnt[2]); res["valid_points"] = int(cnt[0])
        res.setdefault("schedule", {})[name] = od.reg.counters()
        od.close()
    return res


def backend_leg(args, device):
    """Secondary measurement, not `value`: SURVEY 8f.4, the back end's scan-to-submap optimisation (rolo_scan2map_optimize) — a sub-map of
    the corner / surface features of 5 key frames, the features of a 6th frame registered to it from a perturbed pose; host arrays in,
    pose out (uploads, both sub-map trees, all Gauss-Newton iterations)."""
    from rolo_amd import synth
    from rolo_amd.backend import Scan2Map
    from rolo_amd.frontend import FrontEnd, front_params
    from rolo_amd.rotvgicp import RotVGICP
    from scipy.spatial.transform import Rotation
    sensor = args.sensor
    S = synth.SENSORS[sensor]
    fp = front_params(n_scan=S[0], horizon_scan=S[1])
    ctx = RotVGICP(device); fe = FrontEnd(ctx, fp)
    R = np.eye(3); t = np.zeros(3); mc, ms, last = [], [], None
    for k in range(6):
        fr = synth.make_frame(sensor, R, t, synth.SEED + k)
        pr = fe.project(fr.xyz, fr.ring); ex = fe.extract(pr["n"])
        c, su = ex["corner"].copy(), ex["surface"].copy()
        if k < 5:
            for a, dst in ((c, mc), (su, ms)):
                a[:, :3] = (a[:, :3].astype(np.float64) @ R.T + t).astype(np.float32); dst.append(a)
        else:
            last = (c, su, R.copy(), t.copy())
        t = t + R @ np.array([0.3, 0.02 * k, 0.0]); R = R @ synth.rpy_to_R(np.deg2rad(0.3), np.deg2rad(-0.2), np.deg2rad(2.0))
    ctx.close()
    mc = np.concatenate(mc); ms = np.concatenate(ms)
    corner, surf, R6, t6 = last
    truth = np.concatenate([Rotation.from_matrix(R6).as_euler("xyz"), t6]).astype(np.float32)
    guess = (truth + np.array([0.004, -0.003, 0.01, 0.06, -0.04, 0.02], np.float32)).astype(np.float32)
    g = Scan2Map(device)
    for _ in range(2):
        tf = g.scan2MapOptimization(corner, surf, mc, ms, guess)
    reps = 5
    t0 = time.perf_counter()
    for _ in range(reps):
        tf = g.scan2MapOptimization(corner, surf, mc, ms, guess)
    dt = (time.perf_counter() - t0) / reps
    st = g.last_stats
    # the sub-map resident in the context (rolo_scan2map_set_submap once per key-frame set): a call = one upload of the scan's features + the iterations
    t0 = time.perf_counter(); g.setSubmap(mc, ms); dt_set = time.perf_counter() - t0
    for _ in range(2):
        tf_r = g.scan2MapOptimization(corner, surf, None, None, guess)
    t0 = time.perf_counter()
    for _ in range(reps):
        tf_r = g.scan2MapOptimization(corner, surf, None, None, guess)
    dt_r = (time.perf_counter() - t0) / reps
    out = {"workload": f"{sensor}: {corner.shape[0]} corner + {surf.shape[0]} surface features against a sub-map of {mc.shape[0]} + {ms.shape[0]} (5 key frames)",
           "ms_per_call": 1e3 * dt_r, "ms_per_call_with_submap_upload_and_tree_build": 1e3 * dt, "ms_set_submap": 1e3 * dt_set, "same_result": bool(np.array_equal(tf, tf_r)),
           "iterations": int(st.iterations), "converged": int(st.converged),
           "pose_error_vs_truth": {"rot_rad": float(np.abs(tf[:3] - truth[:3]).max()), "trans_m": float(np.abs(tf[3:] - truth[3:]).max())}}
    g.close()
    return out


def _pool_pair(a):
    """pair i of the headline pool: the nominal motion of SURVEY 8d seen from stand point i of the hall (i = 0: the nominal pair itself)"""
    from rolo_amd import synth
    sensor, seed, i = a
    src, tgt, _ = synth.dense_pair(sensor, seed=seed + 2 * i, origin=synth.pool_origin(i))
    return src, tgt


def cpu_model() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def shim_leg(tmp="/tmp"):
    """Secondary measurement, not `value`: the LITERAL unchanged caller — src/lidarOdometry.cpp:460-494 constructs fast_gicp::RotVGICP inside
    scanRegeistration, once per frame — through the C++ drop-in class (include/rot_vgicp_hip.hpp) in a C++-only process: milliseconds per
    frame with the object constructed per frame (context pool behind the constructor), with one persistent object, and what an unpooled
    rolo_ctx_create + rolo_ctx_destroy would add. VLP-16 pair, POLAR voxels, the reference's convergence rule, host clouds in (PCIe included)."""
    import tempfile
    from rolo_amd import synth
    d = tempfile.mkdtemp(prefix="shim_", dir=tmp)
    exe = os.path.join(d, "shim_demo")
    cmd = ["g++", "-std=c++17", "-O2", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "shim_demo.cpp"), "-o", exe,
           "-L", os.path.join(ROOT, "rolo_amd"), "-lrolo_hip", "-Wl,-rpath," + os.path.join(ROOT, "rolo_amd"), "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.run(cmd, check=True, capture_output=True)
    src, tgt, _ = synth.dense_pair("vlp16")
    src.astype(np.float32).tofile(os.path.join(d, "s.bin")); tgt.astype(np.float32).tofile(os.path.join(d, "t.bin"))
    r = subprocess.run([exe, "loop", os.path.join(d, "s.bin"), os.path.join(d, "t.bin"), "50"], capture_output=True, text=True, check=True)
    a, b, c, sa, sb = r.stdout.split()
    return {"workload": f"vlp16 dense pair ({src.shape[0]} points), POLAR voxels, convergence-driven, host clouds, C++-only process, 50 frames",
            "ms_per_frame_object_constructed_per_frame": float(a), "ms_per_frame_persistent_object": float(b),
            "ms_unpooled_ctx_create_destroy": float(c), "results_identical_across_frames": sa == "1" and sb == "1"}


def sharded_children(world, sensor, frames, leaf, timeout=420, one_device=False, exchange="peer"):
    """BASELINE configs[3] through the peer exchange (or RCCL: exchange="rccl") with one CHILD process per GPU (python -m rolo_amd.peerbench: the C ABI only,
    mailbox handles / the RCCL unique id and barriers through files). Returns the per-rank dicts or {"error": ...}; a crashed or hung child is killed and
    reported, never propagated."""
    import tempfile
    d = tempfile.mkdtemp(prefix="rolo_sharded_")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "GROUP_RANK", "ROLE_RANK", "LOCAL_WORLD_SIZE", "TORCHELASTIC_RUN_ID"):
        env.pop(k, None)
    procs = [subprocess.Popen([sys.executable, "-m", "rolo_amd.peerbench", str(r), str(world), d, sensor, str(frames), str(leaf), str(0 if one_device else r), exchange], env=env, cwd=ROOT,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(world)]
    t0 = time.time(); out = [b""] * world
    try:
        for i, p in enumerate(procs):
            out[i] = p.communicate(timeout=max(5.0, timeout - (time.time() - t0)))[0]
    except subprocess.TimeoutExpired:
        for p in procs:
            if p.poll() is None:
                p.kill()
        return {"error": f"sharded child ranks did not finish within {timeout} s"}
    if any(p.returncode != 0 for p in procs):
        return {"error": "sharded child rank failed: " + " | ".join(f"rank {i} rc {p.returncode}: {o[-300:].decode(errors='replace')}" for i, (p, o) in enumerate(zip(procs, out)) if p.returncode != 0)}
    return [json.load(open(os.path.join(d, f"res{r}.json"))) for r in range(world)]


def _config5_pair(i):
    """pair i of BASELINE configs[4]: seed 20260926 + i, motion drawn U(+-2 deg), U(+-0.4 m)"""
    from rolo_amd import synth
    g = np.random.default_rng(synth.SEED + i)
    rpy = g.uniform(-2.0, 2.0, 3); t = g.uniform(-0.4, 0.4, 3)
    src, tgt, _ = synth.dense_pair("os1-64", seed=synth.SEED + i, rpy_deg=tuple(rpy), t=tuple(t))
    return src, tgt


def config5_leg(args, torch, dist, rank, world, local_rank, new_ctx, barrier):
    """BASELINE configs[4]: 512 DISTINCT OS1-64 pairs (leaf 1.0 m, 20 iterations) resident in HBM, dealt round-robin to the ranks,
    streamed through the contexts of each rank (hipGraph replay after the second pair of a context)."""
    import multiprocessing as mp
    from rolo_amd import synth
    npairs = args.config5_pairs
    mine = list(range(rank, npairs, world))
    tg = time.perf_counter()
    with mp.get_context("spawn").Pool(min(usable_cores(), max(1, len(mine)))) as pool:
        pairs = pool.map(_config5_pair, mine, chunksize=2)
    gen_s = time.perf_counter() - tg
    dev = [(torch.from_numpy(s).cuda(), torch.from_numpy(t).cuda(), s.shape[0], t.shape[0]) for s, t in pairs]
    guess = -np.asarray(synth.PREV_STEP_T, np.float64); last = guess * 0.97; zero3 = np.zeros(3)
    ctxs = [new_ctx(leaf=1.0) for _ in range(max(args.streams, 1))]
    results = [None] * len(dev)
    npass = np.zeros(len(dev), int)

    def enqueue(g, k):
        s, t, ns, nt = dev[k]
        g.setInputTargetDevice(t.data_ptr(), nt, 4); g.setInputSourceDevice(s.data_ptr(), ns, 4)
        g.register_async(None, zero3, guess, last, 0.1, 0.1, 0.3)

    def sweep(store):
        nxt = 0; inflight = []
        for g in ctxs:
            if nxt < len(dev):
                enqueue(g, nxt); inflight.append((g, nxt)); nxt += 1
        while inflight:
            g, k = inflight.pop(0)
            Tf, Td, t = g.register_wait()
            npass[k] = g.last_stats.n_passes + g.last_translation_stats.n_passes
            if store:
                results[k] = (Td.copy(), t.copy())
            if nxt < len(dev):
                enqueue(g, nxt); inflight.append((g, nxt)); nxt += 1

    sweep(False)  # warm-up sweep: allocations, graph capture
    c0 = [g.counters() for g in ctxs]
    import gc
    times = []
    for rep in range(3):
        gc.collect(); gc.disable()
        barrier(); t0 = time.perf_counter()
        sweep(rep == 0)
        barrier(); dt = time.perf_counter() - t0
        gc.enable()
        if world > 1:
            tt = torch.tensor([dt], dtype=torch.float64, device="cuda"); dist.all_reduce(tt, op=dist.ReduceOp.MAX); dt = float(tt.item())
        times.append(dt)
    dt = float(np.median(times))
    out = {"workload": f"{npairs} distinct os1-64 pairs (65 536 rays each, seeds {synth.SEED}+i, motions U(+-2 deg), U(+-0.4 m)), UNIFORM leaf 1.0 m, "
                       f"20 SO(3) LM iterations + CT translation, resident in HBM, {len(ctxs)} contexts per GPU, hipGraph replay",
           "scans_per_s_batch512": npairs / dt, "sweep_ms": [1e3 * x for x in times], "pairs_per_rank": len(mine), "generation_s": gen_s}
    c1 = [g.counters() for g in ctxs]
    out["passes_per_pair"] = {"min": int(npass.min()), "median": float(np.median(npass)), "max": int(npass.max())}
    # over the three timed sweeps: how many frames replayed the graph, were re-captured (the schedule length follows the pairs' needs),
    # or needed host round trips because the schedule was too short
    out["schedule_timed_sweeps"] = {k_: int(sum(b_[k_] - a_[k_] for a_, b_ in zip(c0, c1))) for k_ in ("frames", "graph_replays", "graph_captures", "eager_frames", "topup_frames")}
    if rank == 0 and not args.no_cpu:
        from oracle import pyorc
        p = pyorc.default_params(voxel_type=pyorc.VOXEL_UNIFORM, voxel_resolution=1.0, fixed_iterations=20, num_threads=usable_cores())
        worst_r = worst_t = 0.0
        sample = list(range(0, len(mine), max(1, len(mine) // 16)))[:16]
        for k in sample:
            o = pyorc.Reg(p); o.set_target(pairs[k][1]); o.set_source(pairs[k][0])
            rc, _, Td, _, _ = o.align(); rc2, to, _ = o.compute_translation(np.zeros(3), guess, last)
            Tg, tg_ = results[k]
            dR = Tg[:3, :3] @ Td[:3, :3].T
            worst_r = max(worst_r, float(np.arccos(np.clip((np.trace(dR) - 1) / 2, -1, 1)))); worst_t = max(worst_t, float(np.abs(tg_ - to).max()))
        out["parity_sample"] = {"pairs": [mine[k] for k in sample], "max_rot_err_rad": worst_r, "max_trans_err_m": worst_t,
                                "bar": "<= 1e-5 rad, <= 1e-4 m vs the oracle"}
    for g in ctxs:
        g.close()
    return out


def cpu_baseline_legs(args, src, tgt, guess, last):
    """The oracle (CPU restatement of the reference, same OpenMP structure) on this box's host cores: the headline workload with all
    usable cores (`cpu_baseline`), BASELINE configs[0] (VLP-16 pair, POLAR voxels, convergence-driven, ONE thread), per-stage ms."""
    from oracle import pyorc
    from rolo_amd import synth
    cores = usable_cores()

    def sample(p, s, t, seconds, cap):
        stages = {"set_inputs": 0.0, "covariances_kdtree_knn_svd": 0.0, "voxelmap": 0.0, "rotation_lm": 0.0, "translation_lm": 0.0}
        t0 = time.perf_counter(); frames = 0
        while True:
            ta = time.perf_counter(); o = pyorc.Reg(p); o.set_target(t); o.set_source(s)
            tb = time.perf_counter(); o.compute_covariances()
            tc = time.perf_counter(); o.build_voxelmap()
            td = time.perf_counter(); o.align()
            te = time.perf_counter(); o.compute_translation(np.zeros(3), guess, last)
            tf = time.perf_counter()
            for k, v in zip(stages, (tb - ta, tc - tb, td - tc, te - td, tf - te)):
                stages[k] += v
            frames += 1
            cdt = time.perf_counter() - t0
            if cdt >= seconds or frames >= cap:
                break
        return frames, cdt, {k: 1e3 * v / frames for k, v in stages.items()}

    p = pyorc.default_params(voxel_type=pyorc.VOXEL_UNIFORM, voxel_resolution=args.leaf, fixed_iterations=20, num_threads=cores)
    frames, cdt, st = sample(p, src, tgt, args.cpu_seconds, 200)
    out = {"cpu_baseline": {"value": frames / cdt, "unit": "scans/s", "cores": cores, "cpu_model": cpu_model(), "kind": "port",
                            "sample": f"{frames} frame pair(s) of the same workload (kd-tree build, 20-NN covariances, voxel map, 20 SO(3) LM iterations, "
                                      f"CT translation) on oracle/librolo_oracle.so, OMP threads = {cores} (cgroup CPU quota of this box; os.cpu_count() = "
                                      f"{os.cpu_count()}), {cdt:.1f} s", "stage_ms": st}}
    s16, t16, _ = synth.dense_pair("vlp16")
    p1 = pyorc.default_params(polar_resolution=(0.175, 0.175, 2.0), num_threads=1)
    f1, c1, st1 = sample(p1, s16, t16, min(args.cpu_seconds, 6.0), 100)
    out["cpu_baseline_1thread"] = {"value": f1 / c1, "unit": "scans/s", "cores": 1, "cpu_model": cpu_model(), "kind": "port",
                                   "sample": f"BASELINE configs[0]: {f1} VLP-16 pair(s) ({s16.shape[0]} points), POLAR voxels 0.175/0.175/2.0, reference convergence "
                                             f"rule, 1 OMP thread, {c1:.1f} s", "stage_ms": st1}
    return out


def main():
    args = parse()
    if args.sharded_children_test > 1:
        kinds_ = ["peer", "rccl"] if args.shard_exchange == "both" else [args.shard_exchange]   # (rccl: refuses two ranks on one device — the point is a clean error record)
        print(json.dumps({k_: sharded_children(args.sharded_children_test, "os1-128x2048", max(5, args.steps // 2), args.leaf, one_device=True, exchange=k_,
                                               timeout=420 if k_ == "peer" else 90) for k_ in kinds_ if k_ in ("peer", "rccl")}))
        return
    maybe_spawn(args)
    import torch  # device memory, streams and torch.distributed only
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: rolo_amd has no CPU fallback")
    torch.cuda.set_device(local_rank)
    if world > 1:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    from rolo_amd import synth
    from rolo_amd.rotvgicp import RotVGICP, RotVGICPBatch
    if args.pipeline_only:
        print(json.dumps(pipeline_leg(args, torch, local_rank)))
        return

    # ---- synthetic inputs (rank-specific seed in replicas mode: independent frames) ----
    seed = synth.SEED + (1000 * rank if args.mode == "replicas" else 0)
    npool = max(1, args.pool) if args.mode == "replicas" else 1
    if npool > 1:
        import multiprocessing as mp
        with mp.get_context("spawn").Pool(min(usable_cores(), npool)) as pool_:
            pairs = pool_.map(_pool_pair, [(args.sensor, seed, i) for i in range(npool)])
    else:
        pairs = [_pool_pair((args.sensor, seed, 0))]
    src, tgt = pairs[0]
    n = src.shape[0]
    d_pool = [(torch.from_numpy(s_).cuda(), torch.from_numpy(t_).cuda(), s_.shape[0]) for s_, t_ in pairs]
    assert all(p_[2] == n for p_ in d_pool)
    d_src, d_tgt = d_pool[0][0], d_pool[0][1]
    guess = -np.asarray(synth.PREV_STEP_T, np.float64)
    last = guess * 0.97

    def new_ctx(alone=False, g=None, leaf=None, iters=None):
        g = g or RotVGICP(local_rank)
        g.setResolution(args.leaf if leaf is None else leaf)
        g.setFixedIterations(int(os.environ.get("ROLO_BENCH_ITERS", "20")) if iters is None else iters)   # 0: convergence-driven, the reference's own loop (lsq_registration_impl.hpp:161)
        g.setOverlapKnn(args.pair_search == "on")
        g.setUseGraph(not args.no_graph)
        g.setLoadHint({"auto": -1, "idle": 0, "busy": 1}[args.load_hint])
        return g

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    zero3 = np.zeros(3)

    class Batch:
        """B frame pairs behind the enqueue / register_wait interface of one operator"""
        def __init__(self, B):
            self.b = RotVGICPBatch(B, local_rank)
            for m in self.b.members:
                new_ctx(g=m)
            self.guess = np.tile(guess, (B, 1)); self.last = np.tile(last, (B, 1)); self.zero = np.zeros((B, 3))

        def enqueue(self, ds, dt_, npts):
            for m in self.b.members:
                m.setInputTargetDevice(dt_.data_ptr(), npts, 4)
                m.setInputSourceDevice(ds.data_ptr(), npts, 4)
            self.b.register_async(None, self.zero, self.guess, self.last, 0.1, 0.1, 0.3)

        def register_wait(self):
            self.b.register_wait()
            self.last_stats = self.b.members[0].last_stats; self.last_translation_stats = self.b.members[0].last_translation_stats

    def enqueue(g, data):
        ds, dt_, npts = data
        if isinstance(g, Batch):
            return g.enqueue(ds, dt_, npts)
        g.setInputTargetDevice(dt_.data_ptr(), npts, 4)
        g.setInputSourceDevice(ds.data_ptr(), npts, 4)
        g.register_async(None, zero3, guess, last, 0.1, 0.1, 0.3)

    pass_log = []   # (rotation passes, translation passes) of every frame waited for while `pass_log_on`
    pass_log_on = [False]
    cursor = [0]    # next pair of the pool

    def run_steps(gs, k, data):
        """k steps; one step = one frame pair on every context of `gs`. Contexts are serviced round-robin
        (wait for a context's frame, immediately enqueue its next one) so the GPU always has work queued.
        data = a list of resident pairs: every enqueue takes the NEXT pair of the pool, so the contexts in flight hold different
        pairs at any time and a context sees a different pair every frame."""
        if not isinstance(gs, (list, tuple)):
            gs = [gs]
        if k <= 0:
            return

        def nxt():
            if not isinstance(data, list):
                return data
            d = data[cursor[0] % len(data)]; cursor[0] += 1
            return d
        for g in gs:
            enqueue(g, nxt())
        for it in range(k):
            for g in gs:
                g.register_wait()
                if pass_log_on[0] and not isinstance(g, Batch):
                    pass_log.append((g.last_stats.n_passes, g.last_translation_stats.n_passes, g.last_stats.n_cost_only + g.last_translation_stats.n_cost_only,
                                     g.last_stats.n_outer, g.last_translation_stats.n_outer))
                if it + 1 < k:
                    enqueue(g, nxt())

    def timed_round(g, steps, data):
        import gc
        # CPython's cyclic GC walks every tracked object (torch is imported: ~40 ms per full collection) and fired once
        # per ~140 frames in the middle of the timed loop; collect now, keep it off while timing
        gc.collect()
        gc.disable()
        barrier()
        t0 = time.perf_counter()
        run_steps(g, steps, data)
        barrier()
        dt = time.perf_counter() - t0
        gc.enable()
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    def timed(g, steps, warmup, data, min_s=None):
        """rounds of exactly `steps` steps; >= 5 rounds and >= 0.5 s in total (the round times are max-over-ranks values, so every
        rank takes the same decisions); returns (median round time, all round times)"""
        run_steps(g, warmup, data)
        rounds = []
        while True:
            rounds.append(timed_round(g, steps, data))
            # >= 5 rounds and >= 3 s of timed work (round 4: 0.5 s — 18 rounds x 28 ms inside a 40 s run, which a driver-side 5-second GPU-busy sampler cannot see)
            if args.single_round or (len(rounds) >= 5 and (sum(rounds) >= (MIN_TIMED_S if min_s is None else min_s) or len(rounds) >= 400)):
                break
        return float(np.median(rounds)), rounds

    data = d_pool if len(d_pool) > 1 else (d_src, d_tgt, n)
    g = new_ctx()
    ctxs = [g] + [new_ctx() for _ in range(max(args.streams, 1) - 1)]
    B = max(args.batch, 1)
    if B > 1 and not (args.mode == "shard" and world > 1):
        ctxs = [Batch(B) for _ in range(max(args.streams, 1))]
    else:
        B = 1
    rccl_ranks = None

    def connect(gc, kind, max_points):
        """point-shard context gc over the ranks: kind "peer" = rolo_peer_* (64-byte hipIpc handles all-gathered through torch.distributed),
        "rccl" = rolo_comm_init. Returns what the library reports back."""
        if kind == "peer":
            h = gc.peer_export(world, max_points)
            mine = torch.frombuffer(bytearray(h), dtype=torch.uint8).cuda()
            allh = [torch.empty(64, dtype=torch.uint8, device="cuda") for _ in range(world)]
            dist.all_gather(allh, mine)
            gc.peer_connect([bytes(t_.cpu().numpy().tobytes()) for t_ in allh], rank, world)
            r_, w_, kind_ = gc.peer_info()
            return {"exchange": "peer mailboxes (rolo_peer_*)", "ranks": w_, "mailbox_memory": kind_}
        uid = [RotVGICP.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(uid, src=0)
        gc.comm_init(uid[0], rank, world)
        return {"exchange": "RCCL (ncclAllGather + ncclAllReduce per pass)", "ranks": gc.comm_info()[1]}

    shard_info = None
    if args.mode == "shard" and world > 1:
        ctxs = [g]
        shard_info = connect(g, "rccl" if args.shard_exchange in ("rccl", "rccl-inproc", "both") else "peer", 2 * n)
        rccl_ranks = shard_info["ranks"]
    pass_log_on[0] = True
    dt, rounds = timed(ctxs, args.steps, args.warmup, data)
    pass_log_on[0] = False
    # how many of a frame's passes were cost-only (a pass right after a rejected trial stops after the trial's cost: LmState::lin_skip): counted ON THE
    # DEVICE by the controller that consumes such a pass (rolo_stats::n_cost_only), over every frame of the timed rounds — round 5 inferred it from the LM trace of one
    # untimed frame per pool pair
    cost_only = None
    if pass_log and len(pass_log[0]) > 2:
        co_ = np.array([p_[2] for p_ in pass_log])
        cost_only = {"median": float(np.median(co_)), "min": int(co_.min()), "max": int(co_.max()), "frames": int(co_.shape[0]),
                     "what": "passes that evaluated the trial cost alone (SURVEY 8d's error pass), counted on the device (rolo_stats.n_cost_only) over the frames of the timed rounds"}
    frames_total = args.steps * len(ctxs) * B * (world if args.mode == "replicas" else 1)
    value = frames_total / dt
    rs, ts = ctxs[0].last_stats, ctxs[0].last_translation_stats
    passes = rs.n_passes + ts.n_passes

    out = {
        "metric": "scans/sec (128k-pt frame, 20 GN iters)",
        "value": value,
        "unit": "scans/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * dt / args.steps,
        "frames_per_step": len(ctxs) * B * (world if args.mode == "replicas" else 1),
        "timed_rounds": len(rounds),
        "rounds_ms_per_step": [round(1e3 * r / args.steps, 4) for r in rounds],
        "higher_is_better": True,
        "scaling": "weak" if args.mode == "replicas" else "strong",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": f"{args.sensor} dense frame pair, {n} pts/cloud, k=20 PLANE covariances, UNIFORM voxel leaf "
                               f"{args.leaf} m, 20 SO(3) LM iterations + CT translation LM", "mode": args.mode,
                   "parallelism": f"{args.mode}{world}" + (" (one frame pair per rank and context, no data-path collective)" if args.mode == "replicas" else
                                                            " (K5 by query point + covariance exchange, passes by source point + exchange of 32 fp64 per pass)"),
                   "streams_per_gpu": len(ctxs), "frames_per_call": B, "hip_graph": not args.no_graph, "rot_outer": rs.n_outer, "trans_outer": ts.n_outer,
                   "passes_per_frame": passes, "n_correspondences": rs.n_correspondences},
    }
    if shard_info is not None:
        out["config"]["shard"] = shard_info
    # what the frames of the timed rounds needed (distinct pairs need different numbers of LM trials): fused pass launches per frame,
    # and how often the first launch schedule of a frame was too short (rolo_register_wait then tops up through host round trips)
    if pass_log:
        pl = np.array(pass_log)
        tot = pl[:, 0] + pl[:, 1]
        cnt = [c_.counters() for c_ in ctxs if not isinstance(c_, Batch)]
        out["config"]["inputs"] = (f"{len(d_pool)} distinct resident frame pairs rotated through the contexts (pair 0 = the nominal pair of SURVEY 8d; pair i = the same "
                                   "motion from stand point i of the hall, own noise)") if len(d_pool) > 1 else "one resident frame pair registered by every context"
        out["config"]["passes_per_frame_stats"] = {"min": int(tot.min()), "median": float(np.median(tot)), "max": int(tot.max()),
                                                   "rotation": {"min": int(pl[:, 0].min()), "max": int(pl[:, 0].max())},
                                                   "translation": {"min": int(pl[:, 1].min()), "max": int(pl[:, 1].max())}, "frames_logged": int(pl.shape[0])}
        out["config"]["schedule"] = {k_: int(sum(c_[k_] for c_ in cnt)) for k_ in ("frames", "graph_replays", "graph_captures", "eager_frames", "topup_frames")}
        # host side of a frame (round 5's verdict, item 2a): std::chrono inside rolo_register_async (the hipGraphLaunch) and rolo_register_wait, summed over every frame the
        # contexts saw (warm-up, captures and eager frames included); one Python thread drives all contexts, so enqueue + wait_other per frame must stay well below the frame time
        nfr_ = max(sum(c_["frames"] for c_ in cnt), 1)
        out["host_enqueue_us_per_frame"] = 1e-3 * sum(c_.get("host_enqueue_ns", 0) for c_ in cnt) / nfr_
        out["config"]["host_us_per_frame"] = {"enqueue": out["host_enqueue_us_per_frame"], "wait_blocked": 1e-3 * sum(c_.get("host_wait_blocked_ns", 0) for c_ in cnt) / nfr_,
                                              "wait_other": 1e-3 * sum(c_.get("host_wait_other_ns", 0) for c_ in cnt) / nfr_, "frame_time_us": 1e6 / value,
                                              "what": "host time inside rolo_register_async (graph launch) / blocked in rolo_register_wait's event wait / the rest of rolo_register_wait, per frame; one host thread drives all contexts"}
        passes = int(round(float(np.median(tot))))
        out["config"]["passes_per_frame"] = passes
        if cost_only:
            out["config"]["cost_only_passes"] = cost_only

    # ---- layout check: throughput must not depend on WHEN the contexts are created relative to other streams of the process ----------------
    # (HIP deals streams to its four hardware queues in creation order; rounds 1-3 depended on it: 2140 vs 2940 scans/s for the same four contexts.
    # The library now takes its streams from a per-device bank created in bursts — rolo_ctx_create / api.hip — so contexts closed and re-created
    # after FOREIGN streams appeared must measure the same.)
    if world == 1 and args.mode == "replicas" and B == 1 and not args.no_layout_check:
        try:
            foreign = [torch.cuda.Stream() for _ in range(3)]   # three streams the library did not create: every non-trivial shift of a 4-queue deal
            for c_ in ctxs:
                c_.close()
            ctxs = [new_ctx() for _ in range(len(ctxs))]
            g = ctxs[0]
            dt2, rounds2 = timed(ctxs, args.steps, args.warmup, data)
            v2 = args.steps * len(ctxs) / dt2
            out["layout_check"] = {"scans_per_s_contexts_recreated_after_foreign_streams": v2, "ratio": v2 / value, "foreign_streams": len(foreign),
                                   "what": "the headline's contexts closed, three foreign HIP streams created, the same number of contexts created again, the same timed rounds"}
        except Exception as e:  # pragma: no cover
            out["layout_check"] = {"error": repr(e)}

    # ---- frame-level HBM figures of BASELINE.json's metric ("scans/sec ...; achieved HBM GB/s") -------------------------------------
    # algorithmic: SURVEY.md 8d's bytes of one frame — 360 B/pt covariances for both clouds, 136 B/pt + 96 B/voxel map build, and
    # 104 B/pt per FUSED pass launch (a launch evaluates the trial cost and the next linearisation in one sweep over the same
    # records, so it is priced once, as DESIGN.md §4 and `roofline` do) x scans/s.
    # measured: the PMC-summed fabric traffic of one frame ((2 FETCH_SIZE + WRITE_SIZE) KB over every kernel of a single-context
    # eager frame, the newest profiles/rNN/pmc_traffic.json) x scans/s — "rocprof-reported achieved HBM GB/s".
    try:
        from rolo_amd._lib import lib as _rl
        V = max(int(_rl().rolo_num_voxels(g._h)), 0)
        n_cost = int(round(cost_only["median"])) if cost_only else 0   # SURVEY 8d prices an error pass like a linearising one: 104 B/pt either way
        frame_bytes = 360.0 * 2 * n + 136.0 * n + 96.0 * V + 104.0 * n * passes
        fh = {"algorithmic_bytes_per_frame": frame_bytes, "achieved": frame_bytes * value / 1e9, "peak": HBM_PEAK_GBS * world,
              "unit": "GB/s", "frac": frame_bytes * value / 1e9 / (HBM_PEAK_GBS * world), "voxels": V, "fused_pass_launches": passes, "cost_only_pass_launches": n_cost}
        from rolo_amd.profile import pmc_traffic_file
        pmc, pmc_desc = pmc_traffic_file()
        if pmc and os.path.exists(pmc) and args.sensor == "os1-128":
            tr = json.load(open(pmc))
            nfr = max(tr.get("knn_walk_kernel", {}).get("launches", 0) + tr.get("knn_walk_sub_kernel", {}).get("launches", 0), 1)   # one search launch per frame, whichever walk kernel the size picks
            # (the 1 GiB device-to-device copies of this script's own HBM copy test show up as __amd_rocclr_copyBuffer: not part of a frame)
            fabric = sum(v["hbm_bytes_per_launch"] * v["launches"] for k, v in tr.items()
                         if isinstance(v, dict) and "launches" in v and not k.startswith("__amd_rocclr_copyBuffer")) / nfr
            fh.update({"fabric_bytes_per_frame": fabric, "achieved_fabric_GBps": fabric * value / 1e9, "frac_fabric": fabric * value / 1e9 / (HBM_PEAK_GBS * world),
                       "traffic_source": pmc_desc})
        out["frame_hbm"] = fh
    except Exception as e:  # pragma: no cover
        out["frame_hbm"] = {"error": repr(e)}

    # ---- measured-achievable HBM rate next to the nominal peak: device-to-device copy of 1 GiB ----
    try:
        if args.single_round:
            raise RuntimeError("skipped in profiling runs")   # keeps the 1 GiB copies out of the rocprofv3 summaries
        a = torch.empty(1 << 30, dtype=torch.uint8, device="cuda"); b = torch.empty_like(a)
        b.copy_(a); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            b.copy_(a)
        torch.cuda.synchronize()
        out["hbm_measured_copy_GBps"] = 5 * 2 * (1 << 30) / (time.perf_counter() - t0) / 1e9
        del a, b
    except Exception as e:  # pragma: no cover
        out["hbm_measured_copy_GBps"] = None

    # ---- single-frame latency: one context alone ----
    if (len(ctxs) > 1 or B > 1) and args.mode == "replicas":
        gl = new_ctx(alone=True)
        lsteps = max(5, min(args.steps, 20))
        run_steps(gl, 3, data)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run_steps(gl, lsteps, data)
        torch.cuda.synchronize()
        out["config"]["single_frame_latency_ms"] = 1e3 * (time.perf_counter() - t0) / lsteps
        gl.close()

    # ---- roofline of the dominant kernel + per-kernel timing, measured live with HIP events on the ctx stream ----
    try:
        from rolo_amd import profile
        cursor[0] = 0   # one profiled step per pair of the pool, in pool order (all pairs weigh equally in the averages; step 0 = the nominal pair)
        # the profiled context runs ALONE; pin the kernels the timed region ran (several contexts in flight => the busy-device choice, rolo_set_load_hint)
        busy = len(ctxs) > 1 and args.load_hint != "idle"
        g.setLoadHint(1 if busy else {"auto": -1, "idle": 0, "busy": 1}[args.load_hint])
        out["roofline"] = profile.roofline(g, lambda: run_steps(g, 1, data), n, n, passes, HBM_PEAK_GBS, reps=len(d_pool) if len(d_pool) > 1 else 3)
        out["roofline"]["regime"] = ("busy device: the kernels the timed region ran (other contexts' frames in flight)" if busy else "idle device")
        if busy:   # the same frames with the idle-device choice (what a caller with ONE context in flight gets: the latency form)
            g.setLoadHint(0); cursor[0] = 0
            r0 = profile.roofline(g, lambda: run_steps(g, 1, data), n, n, passes, HBM_PEAK_GBS, reps=len(d_pool) if len(d_pool) > 1 else 3)
            out["roofline"]["idle_device"] = {k_: r0[k_] for k_ in ("kernel", "achieved", "frac", "avg_launch_ms")}   # (no PMC pass of this kernel is committed: profiles/r04 has round 4's)
        g.setLoadHint({"auto": -1, "idle": 0, "busy": 1}[args.load_hint])
        ps = out["roofline"].get("avg_launch_ms_per_profiled_step") or []
        if len(d_pool) > 1 and ps:
            out["roofline"]["nominal_pair"] = {"avg_launch_ms": ps[0], "frac": out["roofline"]["algorithmic_bytes_per_launch"] / (ps[0] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                               "note": "pair 0 of the pool = SURVEY 8d's nominal frame pair (what rounds 1 and 2 profiled); the stand-point pairs' searches are heavier"}
    except Exception as e:  # pragma: no cover
        out["roofline"] = {"error": repr(e)}

    # ---- instruction-issue bound next to the HBM one (round 4's verdict: a frac of 0.06 with a traffic ratio of 0.3 says "cached and issue / latency-bound") ----
    try:
        from rolo_amd.profile import valu_issue
        out["valu_issue"] = valu_issue(value / max(world, 1))
    except Exception as e:  # pragma: no cover
        out["valu_issue"] = {"error": repr(e)}

    # ---- N>1: BASELINE configs[3] — ONE 262 144-point frame sharded over the ranks (never allowed to take the main number down) ----
    if world > 1 and args.mode == "replicas" and not args.no_shard_leg:
        try:
            s2, t2, _ = synth.dense_pair("os1-128x2048", seed=synth.SEED)
            d2 = (torch.from_numpy(s2).cuda(), torch.from_numpy(t2).cuda(), s2.shape[0])
            stp = max(5, args.steps // 2)
            kinds = ["peer", "rccl"] if args.shard_exchange == "both" else [args.shard_exchange]
            out["sharded"] = {"workload": f"os1-128x2048 dense frame pair, {s2.shape[0]} pts/cloud, leaf {args.leaf} m, 20 SO(3) LM iterations + CT translation", "scaling": "strong",
                              "note": "one frame: Hilbert sort / BVH / voxel map replicated on every rank, K5 searched by 1/W of the queries + exchange of the 48 B/pt "
                                      "covariances, LM passes over 1/W of the source points + exchange of 32 fp64 per pass (peer: mailbox words written by the controller kernel "
                                      "itself, summed in rank order, hipGraph replay; rccl: ncclAllGather + reduce launch + ncclAllReduce + controller launch per pass, eager). "
                                      "Both exchanges run one CHILD process per GPU while the torch.distributed ranks wait on the host"}

            def host_wait(tag):
                """everybody waits for rank 0 on the HOST (the rendezvous store), not in an RCCL kernel: while the child ranks own the GPUs the
                parents must not keep a collective's kernel spinning on them"""
                try:
                    store = dist.distributed_c10d._get_default_store()
                    if rank == 0:
                        store.set(tag, "1")
                    else:
                        import datetime
                        store.wait([tag], datetime.timedelta(seconds=900))
                except Exception:   # no store to be had: a flag file on this node (one node by contract)
                    flag = os.path.join("/tmp", f"{tag}_{os.environ.get('MASTER_PORT', '0')}")
                    if rank == 0:
                        open(flag, "w").close()
                    else:
                        t0_ = time.time()
                        while not os.path.exists(flag) and time.time() - t0_ < 900:
                            time.sleep(0.05)

            def children_leg(kind):
                """rank 0 runs the W child ranks and assembles their records; the other parents wait on the host"""
                leg = None
                torch.cuda.synchronize()
                if rank == 0:
                    res = sharded_children(world, "os1-128x2048", stp, args.leaf, exchange=kind, timeout=300 if kind == "peer" else 180)   # (a lost peer gives up after ROLO_PEER_TIMEOUT_MS = 10 s by itself; RCCL has no such bound)
                    if isinstance(res, dict):
                        leg = res
                    else:
                        ms = max(r_["ms_per_frame"] for r_ in res)
                        leg = {"value": 1e3 / ms, "unit": "scans/s", "ms_per_frame": ms, "ms_per_frame_by_rank": [round(r_["ms_per_frame"], 4) for r_ in res],
                               "passes": res[0]["passes"], "schedule": res[0]["counters"], "ranks": len(res),
                               "exchange": ("peer mailboxes (rolo_peer_*), hipGraph replay" if kind == "peer" else "RCCL: ncclAllGather (covariances) + ncclAllReduce of the 32 fp64 sums per LM pass, eager launches")
                                           + ", one child process per GPU",
                               "per_launch_us_rank0": {k_: res[0][k_]["mean"] for k_ in res[0] if k_.endswith("_us")},
                               "ranks_agree": all(r_["pose_head"] == res[0]["pose_head"] for r_ in res)}
                        if kind == "peer":
                            leg["mailbox_memory"] = res[0]["mailbox"]
                            leg["selftest"] = {"ok": all(r_.get("selftest", {}).get("ok") for r_ in res),
                                               "lm_exchange_us_by_rank": [round(r_["selftest"]["lm_exchange_us"], 2) for r_ in res],
                                               "cov_exchange_us_by_rank": [round(r_["selftest"]["cov_exchange_us"], 2) for r_ in res],
                                               "lm_exchange_us_max": max(r_["selftest"]["lm_exchange_us"] for r_ in res),
                                               "cov_exchange_us_max": max(r_["selftest"]["cov_exchange_us"] for r_ in res),
                                               "what": "rolo_peer_selftest on every rank before the first frame: 16 all-reduces of 32 known fp64 through the LM mailboxes + one covariance-segment push of known words, verified on every rank"}
                        else:
                            leg["rccl_ranks"] = min(r_.get("rccl_ranks", 0) for r_ in res)   # ncclCommCount as every rank's communicator reports it
                host_wait(f"rolo_sharded_children_done_{kind}")
                barrier()
                return leg

            for kind in kinds:
                if kind in ("peer", "rccl"):
                    leg = children_leg(kind)
                    if rank == 0:
                        out["sharded"][kind] = leg
                        if kind == "rccl" and isinstance(leg, dict) and "rccl_ranks" in leg:
                            rccl_ranks = leg["rccl_ranks"]
                    continue
                try:
                    gs = new_ctx(alone=True)
                    info = connect(gs, "peer" if kind == "peer-inproc" else "rccl", 2 * s2.shape[0])
                    dts, rds = timed(gs, stp, 2, d2)
                    leg = {"value": stp / dts, "unit": "scans/s", "ms_per_frame": 1e3 * dts / stp, "passes": gs.last_stats.n_passes + gs.last_translation_stats.n_passes,
                           "schedule": gs.counters()}
                    leg.update(info)
                    if kind == "rccl-inproc":
                        rccl_ranks = info.get("ranks")
                    barrier()
                    gs.close()
                except Exception as e:  # pragma: no cover
                    leg = {"error": repr(e)}
                out["sharded"][kind] = leg
                barrier()
            out["sharded"]["rccl_ranks"] = rccl_ranks   # how many ranks an RCCL communicator of this run really had (None: no RCCL leg ran / it failed)
            # DESIGN.md section 6's estimate next to the measurement (from one-device rows: only K5 scales with W, the sort / tree / voxel map are replicated,
            # the LM chain is latency-bound): 8 GPUs ~0.7 ms per frame against 1.07 ms on one = ~1.5 x
            out["sharded"]["estimate"] = {"ms_per_frame_8_gpus": 0.7, "ms_per_frame_1_gpu": 1.07, "speedup_8_gpus": 1.5, "source": "DESIGN.md section 6 (estimated from one-device measurements in round 3)"}
            out["config"]["rccl_ranks"] = rccl_ranks
            out["sharded"]["product_answer"] = ("throughput: the frame-parallel replicas (`value`: one frame pair per GPU and context, no data-path collective) — the LM chain of ONE frame is a string "
                                                "of latency-bound launches that sharding cannot shorten, only the neighbour search scales with the ranks (DESIGN.md section 6: ~1.5 x on 8 GPUs "
                                                "expected); the sharded form is for a frame that does not fit one GPU or whose latency matters more than the node's throughput, and there the peer "
                                                "exchange (no library call, no extra launch per pass) is the product path, RCCL the cross-check")
            first = out["sharded"].get(kinds[0]) or {}
            if "value" in first:   # the headline of this leg = the first exchange kind asked for
                out["sharded"].update({"value": first["value"], "unit": "scans/s", "ms_per_frame": first["ms_per_frame"], "ranks": first.get("ranks")})
            # the same frame on ONE rank's GPU alone, for the strong-scaling ratio (rank 0 only runs it while the others wait)
            barrier()
            if rank == 0:
                g1 = new_ctx(alone=True)
                run_steps(g1, 2, d2); torch.cuda.synchronize(); t0 = time.perf_counter(); run_steps(g1, stp, d2); torch.cuda.synchronize()
                out["sharded"]["one_gpu_same_frame_scans_per_s"] = stp / (time.perf_counter() - t0)
                g1.close()
            barrier()
        except Exception as e:  # pragma: no cover
            out["sharded"] = {"error": repr(e)}

    # ---- the two regimes a ROLO user runs, next to the forced-20 headline (round 5's verdict, item 3) + the iteration sweep ----------------------------------
    # `value` above forces 20 outer iterations (BASELINE.json's metric); the reference itself stops at convergence (lsq_registration_impl.hpp:161-176) and hands
    # HOST clouds to setInputTarget / setInputSource (src/lidarOdometry.cpp:466-467). Same pool, same contexts layout, short timed rounds (>= 5 rounds, >= 1 s).
    def side_leg(iters, host=None, nctx=None):
        """scans/s of `nctx` contexts in flight with `iters` forced iterations (0 = convergence-driven); host = None: device-resident pairs, "pinned" / "pageable": the pair
        handed over as host arrays through rolo_set_target / rolo_set_source, the upload inside the timed region"""
        nctx = n_side_ctx if nctx is None else nctx
        cs = [new_ctx(iters=iters) for _ in range(nctx)]
        log0 = len(pass_log)
        try:
            if host is None:
                pass_log_on[0] = True
                dt_, rr_ = timed(cs, args.steps, args.warmup, data, min_s=1.0)
                pass_log_on[0] = False
            else:
                hp = host_pools[host]
                k_ = [0]

                def enq(g_):
                    s_, t_ = hp[k_[0] % len(hp)]; k_[0] += 1
                    g_.setInputTarget(t_); g_.setInputSource(s_)       # host arrays: hipMemcpyAsync H2D + pack on the context's stream, inside the timed region
                    g_.register_async(None, zero3, guess, last, 0.1, 0.1, 0.3)

                def steps_(k):
                    for g_ in cs:
                        enq(g_)
                    for it_ in range(k):
                        for g_ in cs:
                            g_.register_wait()
                            if it_ + 1 < k:
                                enq(g_)
                steps_(args.warmup)
                rr_ = []
                import gc
                while True:
                    gc.collect(); gc.disable(); barrier(); t0_ = time.perf_counter(); steps_(args.steps); barrier(); rr_.append(time.perf_counter() - t0_); gc.enable()
                    if args.single_round or (len(rr_) >= 5 and (sum(rr_) >= 1.0 or len(rr_) >= 100)):
                        break
                dt_ = float(np.median(rr_))
            r_ = {"scans_per_s": args.steps * nctx / dt_ * (world if args.mode == "replicas" else 1), "contexts": nctx, "timed_rounds": len(rr_)}
            if host is None and len(pass_log) > log0:
                pl_ = np.array(pass_log[log0:])
                r_.update({"passes_per_frame": {"median": float(np.median(pl_[:, 0] + pl_[:, 1])), "min": int((pl_[:, 0] + pl_[:, 1]).min()), "max": int((pl_[:, 0] + pl_[:, 1]).max())},
                           "cost_only_passes_median": float(np.median(pl_[:, 2])),
                           "rot_outer": {"median": float(np.median(pl_[:, 3])), "min": int(pl_[:, 3].min()), "max": int(pl_[:, 3].max())},
                           "trans_outer": {"median": float(np.median(pl_[:, 4])), "min": int(pl_[:, 4].min()), "max": int(pl_[:, 4].max())}})
                del pass_log[log0:]
            return r_
        finally:
            pass_log_on[0] = False
            for c_ in cs:
                c_.close()

    if args.mode == "replicas" and B == 1 and not args.no_side_legs:
        # the headline's contexts are closed first: their streams go back to the library's bank and the legs' contexts take the SAME streams — contexts created next to
        # four idle ones get another burst of streams, which HIP may deal to its hardware queues differently (measured: 10 forced iterations 2715 scans/s beside the idle
        # headline contexts against 3811 on their streams; DESIGN section 8 "stream bank")
        n_side_ctx = len(ctxs)
        for c_ in ctxs:
            try:
                c_.close()
            except Exception:
                pass
        try:
            out["convergence_driven"] = dict(side_leg(0), what="the same pool and contexts with fixed_iterations = 0: both stages stop at the reference's own convergence tests "
                                                               "(lsq_registration_impl.hpp:161-176, :63-73) — what a ROLO node runs; `value` forces 20 rotation iterations")
        except Exception as e:  # pragma: no cover
            out["convergence_driven"] = {"error": repr(e)}
        try:
            sweep = {"20": {"scans_per_s": value}}
            for it_ in (10, 2):
                sweep[str(it_)] = side_leg(it_)
            v20, v10, v2 = sweep["20"]["scans_per_s"], sweep["10"]["scans_per_s"], sweep["2"]["scans_per_s"]
            p20 = passes; p10 = sweep["10"].get("passes_per_frame", {}).get("median", 0); p2 = sweep["2"].get("passes_per_frame", {}).get("median", 0)
            # per-trial cost under load = d(frame time) / d(passes); what is left at zero passes = the search + map + build share of a frame
            per_trial_us = 1e6 * (1.0 / v20 - 1.0 / v2) / max(p20 - p2, 1)
            out["iteration_sweep"] = dict(sweep, per_trial_us_under_load=per_trial_us, frame_share_without_lm_ms=1e3 * (1.0 / v20) - 1e-3 * per_trial_us * p20,
                                          what="forced rotation iterations 20 / 10 / 2 with the headline's contexts: the slope prices one LM trial (pass + controller launch) under load, "
                                               "the intercept the search + map + build of a frame (frame time = 1 / scans_per_s: the contexts overlap)")
        except Exception as e:  # pragma: no cover
            out["iteration_sweep"] = {"error": repr(e)}
        try:
            host_pools = {}
            for kind_ in ("pageable", "pinned"):
                hp_ = []
                for s_, t_ in [pairs[i_ % len(pairs)] for i_ in range(5)]:   # (five: with four contexts in flight a context then sees a different array every frame — the same array again is a no-op, rot_vgicp_impl.hpp:113-115)
                    if kind_ == "pinned":
                        hp_.append((torch.from_numpy(np.ascontiguousarray(s_)).pin_memory().numpy(), torch.from_numpy(np.ascontiguousarray(t_)).pin_memory().numpy()))
                    else:
                        hp_.append((np.ascontiguousarray(s_).copy(), np.ascontiguousarray(t_).copy()))
                host_pools[kind_] = hp_
            hc = {}
            for kind_ in ("pinned", "pageable"):
                for nc_ in (1, n_side_ctx):
                    hc[f"{kind_}_{nc_}ctx"] = side_leg(None, host=kind_, nctx=nc_)
            hc["bytes_uploaded_per_frame"] = 2 * n * 16
            hc["what"] = ("the 131 072-point pair handed over as HOST arrays through rolo_set_target / rolo_set_source (lidarOdometry.cpp:466-467: what the unchanged caller does), "
                          "H2D inside the timed region, 20 forced iterations; pinned = page-locked caller buffers, pageable = plain malloc'd arrays (the runtime stages them)")
            out["host_clouds"] = hc
        except Exception as e:  # pragma: no cover
            out["host_clouds"] = {"error": repr(e)}

    # ---- BASELINE configs[4]: 512 distinct OS1-64 pairs streamed through the contexts ----
    if args.mode == "replicas" and not args.no_config5:
        # the headline's contexts go first: HIP deals a process's streams to its four hardware queues in creation order, and four MORE contexts
        # next to four idle ones land in the layout that measured 2.1 k instead of 2.9 k scans/s — config5 ran like that until round 3
        for c_ in ctxs:
            try:
                (c_.b if isinstance(c_, Batch) else c_).close()
            except Exception:
                pass
        try:
            out["config5"] = config5_leg(args, torch, dist, rank, world, local_rank, new_ctx, barrier)
        except Exception as e:  # pragma: no cover
            out["config5"] = {"error": repr(e)}

    # ---- drop-in pipeline leg: raw frames -> pose through the fused node cores (rolo_odom_frame), production settings ----
    if world == 1 and not args.no_pipeline:
        try:
            out["pipeline"] = pipeline_leg(args, torch, local_rank)
        except Exception as e:  # pragma: no cover
            out["pipeline"] = {"error": repr(e)}

        try:
            out["backend"] = backend_leg(args, local_rank)
        except Exception as e:  # pragma: no cover
            out["backend"] = {"error": repr(e)}
        try:
            out["cpp_operator_per_frame"] = shim_leg()
        except Exception as e:  # pragma: no cover
            out["cpp_operator_per_frame"] = {"error": repr(e)}

    # ---- CPU baselines: the oracle on this box's host cores (rank 0, N = 1 only) ----
    if rank == 0 and world == 1 and not args.no_cpu:
        out.update(cpu_baseline_legs(args, src, tgt, guess, last))

    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
