"""Build librolo_hip.so (gfx950) in-tree with hipcc. Used by __graft_entry__.build() and by hand:

    python -m rolo_amd.build [--force]
"""
from __future__ import annotations

import concurrent.futures as cf
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "librolo_hip.so")
SOURCES = ["api.hip", "knn_cov.hip", "voxelmap.hip", "passes.hip", "misc.hip", "front.hip", "odometry.hip", "fusion.hip", "scan2map.hip", "peer.hip"]
HEADERS = ["rolo_internal.hpp", "dev_math.hpp", "voxel_dev.hpp", "knn_walk.hpp", "knn_packet.hpp", "polar_exact.hpp", "polar_exact_consts.hpp", "peer_dev.hpp", "polar_f32.hpp", "lm_begin.hpp", "load_learner.hpp", os.path.join("..", "..", "include", "rolo_hip.h"),
           os.path.join("..", "..", "include", "rolo_fusion.h")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = os.environ.get("ROLO_EXTRA_FLAGS", "").split() + ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-Wall", "-Wno-unused-function",
         "-Wno-unused-result"]


# float32 paths whose results must be bit-identical to the CPU statement (kNN distances and their pruning bounds,
# pcl::transformPointCloud, range-image projection, curvature): no FMA contraction (HIP's __fmul_rn/__fadd_rn are
# plain operators that the compiler is otherwise free to fuse)
EXTRA = {"knn_cov.hip": ["-ffp-contract=off"], "scan2map.hip": ["-ffp-contract=off"], "misc.hip": ["-ffp-contract=off"], "front.hip": ["-ffp-contract=off"]}
# (front.hip was built at -O2 through round 3: hipcc 7.2 -O3 died in the backend — "Illegal instruction detected: Operand has incorrect register class" — on the
# round-1 form of extract_kernel's serial greedy walk; that function was rewritten in rounds 2-3 and the file has compiled at -O3 since: profiles/tools/README.md)


def _stale(out, deps):
    return (not os.path.exists(out)) or any(os.path.getmtime(d) > os.path.getmtime(out) for d in deps if os.path.exists(d))


def build(force: bool = False, verbose: bool = False, tag: str = "", extra_flags=()) -> str:
    """tag / extra_flags: A/B experiment builds (librolo_hip_<tag>.so, objects under csrc/_obj_<tag>/); the product is tag ''."""
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    objs, jobs = [], []
    objdir = os.path.join(CSRC, "_obj_" + tag) if tag else CSRC
    lib = os.path.join(HERE, f"librolo_hip_{tag}.so") if tag else LIB
    os.makedirs(objdir, exist_ok=True)
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(objdir, s.replace(".hip", ".o"))
        objs.append(obj)
        if force or _stale(obj, [src] + hdrs):
            jobs.append([HIPCC] + FLAGS + list(extra_flags) + EXTRA.get(s, []) + ["-c", src, "-o", obj])
    if jobs:
        with cf.ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 4)) as ex:
            for cmd, res in zip(jobs, ex.map(lambda c: subprocess.run(c, capture_output=True, text=True), jobs)):
                if verbose or res.returncode != 0:
                    sys.stderr.write(" ".join(cmd) + "\n" + res.stdout + res.stderr)
                if res.returncode != 0:
                    raise RuntimeError("hipcc failed for " + cmd[-3])
    if force or jobs or _stale(lib, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs + ["-ldl"]
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            sys.stderr.write(" ".join(cmd) + "\n" + res.stdout + res.stderr)
            raise RuntimeError("link failed")
    return lib


if __name__ == "__main__":
    _tag = next((a.split("=", 1)[1] for a in sys.argv if a.startswith("--tag=")), "")
    _xf = [a.split("=", 1)[1] for a in sys.argv if a.startswith("--flag=")]
    print(build(force="--force" in sys.argv, verbose=True, tag=_tag, extra_flags=_xf))
