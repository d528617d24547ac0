#!/usr/bin/env python3
"""bench.py — scans/sec of the MI355X-native ROLO scan-matching hot path (contract: see the task statement).

One "step" = one complete registration of a synthetic OS1-128 frame pair (128 x 1024 = 131 072 points each,
the "128k-pt frame" of BASELINE.json): per-point 20-NN covariances of both clouds, target voxel-hash build
(UNIFORM leaf 0.5 m), SO(3) LM stage forced to exactly 20 outer iterations, continuous-time translation LM
stage — all enqueued on one HIP stream with the two clouds already resident in HBM when the timed region starts.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--mode replicas|shard] [--no-cpu] [--sensor os1-128]

N > 1 (launched by torch.distributed.run, one rank per GPU):
  replicas (default) — the path partitions by frame: every rank registers its own frames, no data-path
                       collective, "scaling": "weak"; value = total frames / max-over-ranks time.
  shard              — one frame, source points sharded over the ranks, one RCCL all-reduce of <= 32 fp64 per LM
                       pass; reported under "sharded" next to the replicas number.
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured achievable copy rate


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
    ap.add_argument("--shard-leg", action="store_true",
                    help="N>1: also time ONE frame with its source points sharded over the ranks (RCCL all-reduce of 32 fp64 per LM pass); "
                         "opt-in because this container has a single GPU and that collective path has only run in the 2-rank gloo test")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="length of the bounded cpu_baseline sample")
    ap.add_argument("--streams", type=int, default=4,
                    help="registration contexts (HIP streams) kept in flight per GPU; a step is then one frame pair per stream")
    ap.add_argument("--pair-search", default="on", choices=["on", "off"],
                    help="search source and target of a frame in one chain of launches (rolo_params.overlap_knn, the library default)")
    ap.add_argument("--no-pipeline", action="store_true", help="skip the raw-frame -> pose pipeline leg")
    ap.add_argument("--pipeline-only", action="store_true", help="run only the pipeline leg and print its dict (profiling runs)")
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
        nfr = 40
        t0 = time.perf_counter()
        run(6, 6 + nfr)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        res[name] = 1e3 * dt / nfr if name.endswith("_ms") else nfr / dt
        res["features_per_frame"] = int(cnt[1] + cnt[2]); res["valid_points"] = int(cnt[0])
        od.close()
    return res


def main():
    args = parse()
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
    seed = synth.SEED + (rank if args.mode == "replicas" else 0)
    src, tgt, _ = synth.dense_pair(args.sensor, seed=seed)
    n = src.shape[0]
    d_src = torch.from_numpy(src).cuda()
    d_tgt = torch.from_numpy(tgt).cuda()
    guess = -np.asarray(synth.PREV_STEP_T, np.float64)
    last = guess * 0.97

    def new_ctx(alone=False, g=None):
        g = g or RotVGICP(local_rank)
        g.setResolution(args.leaf)
        g.setFixedIterations(int(os.environ.get("ROLO_BENCH_ITERS", "20")))
        g.setOverlapKnn(args.pair_search == "on")
        g.setUseGraph(not args.no_graph)
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

        def enqueue(self):
            for m in self.b.members:
                m.setInputTargetDevice(d_tgt.data_ptr(), n, 4)
                m.setInputSourceDevice(d_src.data_ptr(), n, 4)
            self.b.register_async(None, self.zero, self.guess, self.last, 0.1, 0.1, 0.3)

        def register_wait(self):
            self.b.register_wait()
            self.last_stats = self.b.members[0].last_stats; self.last_translation_stats = self.b.members[0].last_translation_stats

    def enqueue(g):
        if isinstance(g, Batch):
            return g.enqueue()
        g.setInputTargetDevice(d_tgt.data_ptr(), n, 4)
        g.setInputSourceDevice(d_src.data_ptr(), n, 4)
        g.register_async(None, zero3, guess, last, 0.1, 0.1, 0.3)

    def run_steps(gs, k):
        """k steps; one step = one frame pair on every context of `gs`. Contexts are serviced round-robin
        (wait for a context's frame, immediately enqueue its next one) so the GPU always has work queued."""
        if not isinstance(gs, (list, tuple)):
            gs = [gs]
        if k <= 0:
            return
        for g in gs:
            enqueue(g)
        for it in range(k):
            for g in gs:
                g.register_wait()
                if it + 1 < k:
                    enqueue(g)

    def timed(g, steps, warmup):
        import gc
        run_steps(g, warmup)
        # CPython's cyclic GC walks every tracked object (torch is imported: ~40 ms per full collection) and fired once
        # per ~140 frames in the middle of the timed loop; collect now, keep it off while timing
        gc.collect()
        gc.disable()
        barrier()
        t0 = time.perf_counter()
        run_steps(g, steps)
        barrier()
        dt = time.perf_counter() - t0
        gc.enable()
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    g = new_ctx()
    ctxs = [g] + [new_ctx() for _ in range(max(args.streams, 1) - 1)]
    B = max(args.batch, 1)
    if B > 1 and not (args.mode == "shard" and world > 1):
        ctxs = [Batch(B) for _ in range(max(args.streams, 1))]
    else:
        B = 1
    if args.mode == "shard" and world > 1:
        ctxs = [g]
        uid = [RotVGICP.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(uid, src=0)
        g.comm_init(uid[0], rank, world)
    dt = timed(ctxs, args.steps, args.warmup)
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
        "frames_per_step": len(ctxs) * B,
        "higher_is_better": True,
        "scaling": "weak" if args.mode == "replicas" else "strong",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": f"{args.sensor} dense frame pair, {n} pts/cloud, k=20 PLANE covariances, UNIFORM voxel leaf "
                               f"{args.leaf} m, 20 SO(3) LM iterations + CT translation LM", "mode": args.mode,
                   "parallelism": f"{args.mode}{world}", "streams_per_gpu": len(ctxs), "frames_per_call": B, "hip_graph": not args.no_graph, "rot_outer": rs.n_outer, "trans_outer": ts.n_outer,
                   "passes_per_frame": passes, "n_correspondences": rs.n_correspondences},
    }

    # ---- frame-level HBM figure of BASELINE.json's metric ("scans/sec ...; achieved HBM GB/s"): SURVEY.md 8d's ALGORITHMIC bytes of
    # one frame (360 B/pt covariances for both clouds, 136 B/pt + 96 B/voxel map build, 104 B/pt per linearise and per error
    # evaluation; a fused pass is one of each except the first of a stage) x scans/s. `roofline` below stays the dominant kernel.
    try:
        from rolo_amd._lib import lib as _rl
        V = max(int(_rl().rolo_num_voxels(g._h)), 0)
        n_eval = sum(2 * st.n_passes - 1 for st in (rs, ts))
        frame_bytes = 360.0 * 2 * n + 136.0 * n + 96.0 * V + 104.0 * n * n_eval
        out["frame_hbm"] = {"algorithmic_bytes_per_frame": frame_bytes, "achieved": frame_bytes * value / 1e9, "peak": HBM_PEAK_GBS * world,
                            "unit": "GB/s", "frac": frame_bytes * value / 1e9 / (HBM_PEAK_GBS * world), "voxels": V, "linearise_plus_error_evaluations": n_eval}
    except Exception as e:  # pragma: no cover
        out["frame_hbm"] = {"error": repr(e)}

    # ---- single-frame latency: one context alone ----
    if (len(ctxs) > 1 or B > 1) and args.mode == "replicas":
        gl = new_ctx(alone=True)
        lsteps = max(5, min(args.steps, 20))
        run_steps(gl, 3)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run_steps(gl, lsteps)
        torch.cuda.synchronize()
        out["config"]["single_frame_latency_ms"] = 1e3 * (time.perf_counter() - t0) / lsteps
        gl.close()

    # ---- roofline of the dominant kernel + per-kernel timing, measured live with HIP events on the ctx stream ----
    try:
        from rolo_amd import profile
        out["roofline"] = profile.roofline(g, lambda: run_steps(g, 1), n, n, passes, HBM_PEAK_GBS)
    except Exception as e:  # pragma: no cover
        out["roofline"] = {"error": repr(e)}

    # ---- optional: point-sharded leg at N>1 (never allowed to take the main number down with it) ----
    if world > 1 and args.mode == "replicas" and args.shard_leg:
        try:
            src0, tgt0, _ = synth.dense_pair(args.sensor, seed=synth.SEED)
            d_src.copy_(torch.from_numpy(src0)); d_tgt.copy_(torch.from_numpy(tgt0))
            gs = new_ctx(alone=True)
            uid = [RotVGICP.comm_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(uid, src=0)
            gs.comm_init(uid[0], rank, world)
            dts = timed(gs, args.steps, args.warmup)
            out["sharded"] = {"value": args.steps / dts, "unit": "scans/s", "ms_per_step": 1e3 * dts / args.steps,
                              "note": "one frame, source points sharded over the ranks, RCCL all-reduce of 32 fp64 per LM pass "
                                      "(kNN / voxel map replicated on every rank)"}
        except Exception as e:  # pragma: no cover
            out["sharded"] = {"error": repr(e)}

    # ---- drop-in pipeline leg: raw frames -> pose through the fused node cores (rolo_odom_frame), production settings ----
    if world == 1 and not args.no_pipeline:
        try:
            out["pipeline"] = pipeline_leg(args, torch, local_rank)
        except Exception as e:  # pragma: no cover
            out["pipeline"] = {"error": repr(e)}

    # ---- CPU baseline: the oracle (CPU restatement of the reference, same OpenMP structure) on this box's host cores ----
    if rank == 0 and world == 1 and not args.no_cpu:
        from oracle import pyorc
        cores = usable_cores()
        p = pyorc.default_params(voxel_type=pyorc.VOXEL_UNIFORM, voxel_resolution=args.leaf, fixed_iterations=20, num_threads=cores)
        t0 = time.perf_counter()
        frames = 0
        while True:  # bounded sample: about 10 s of CPU work
            o = pyorc.Reg(p)
            o.set_target(tgt); o.set_source(src)
            o.align()
            o.compute_translation(np.zeros(3), guess, last)
            frames += 1
            cdt = time.perf_counter() - t0
            if cdt >= args.cpu_seconds or frames >= 200:
                break
        out["cpu_baseline"] = {"value": frames / cdt, "unit": "scans/s", "cores": cores, "kind": "port",
                               "sample": f"{frames} frame pair(s) of the same workload (kd-tree build, 20-NN covariances, voxel map, 20 SO(3) "
                                         f"LM iterations, CT translation) on oracle/librolo_oracle.so, OMP threads = {cores} "
                                         f"(cgroup CPU quota of this box; os.cpu_count() = {os.cpu_count()}), {cdt:.1f} s"}

    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
