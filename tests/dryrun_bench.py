"""TEST INFRASTRUCTURE — a CPU dry run of bench.py's multi-rank control flow (round 4's verdict, item 4b): the first run on an 8-GPU node must not die in Python.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port P tests/dryrun_bench.py --gpus 8 ...

runs bench.main() UNCHANGED on W gloo ranks with
  * torch.cuda reduced to no-ops (tensors stay on the host, `nccl` becomes `gloo`),
  * rolo_amd.rotvgicp.RotVGICP replaced by a stand-in with the same surface whose frames are registered by the ORACLE (small clouds) or answered with the
    identity (clouds above 40 k points: the dry run checks the flow, not numbers),
so that everything around the device is exercised as on the node: rank / seed handling, the input pool, the timed rounds with their barriers and MAX
all-reduce, the sharded leg's child processes (they fail at rolo_ctx_create here: the error must land in the JSON, both exchanges), the host-side waits, the
config5 deal, the JSON assembly on rank 0. Nothing here is reachable from the product or from bench.py itself: only tests/test_multirank_cpu.py starts it.
"""
import ctypes as C
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("OMP_NUM_THREADS", "1")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

# ---- torch.cuda -> host ------------------------------------------------------------------------------------------------------------------
torch.cuda.is_available = lambda: True
torch.cuda.set_device = lambda *a, **k: None
torch.cuda.synchronize = lambda *a, **k: None
torch.cuda.Stream = lambda *a, **k: types.SimpleNamespace()
torch.Tensor.cuda = lambda self, *a, **k: self
torch.Tensor.pin_memory = lambda self, *a, **k: self
_tensor, _empty = torch.tensor, torch.empty


def _strip(f):
    def g(*a, **k):
        if k.get("device") in ("cuda",) or isinstance(k.get("device"), torch.device):
            k.pop("device")
        return f(*a, **k)
    return g


torch.tensor, torch.empty = _strip(_tensor), _strip(_empty)
_init = dist.init_process_group


def _init_gloo(backend=None, **kw):
    kw.pop("device_id", None)
    return _init("gloo", **kw)


dist.init_process_group = _init_gloo

# ---- the device stand-in -----------------------------------------------------------------------------------------------------------------
from oracle import pyorc  # noqa: E402  (tests may use the oracle; this file lives under tests/)
import rolo_amd.rotvgicp as rv  # noqa: E402


class _Stats(types.SimpleNamespace):
    pass


class FakeRotVGICP:
    """the surface bench.py drives; results from the oracle for small clouds"""
    def __init__(self, device=0):
        self._h = None
        self.leaf, self.fixed = 1.0, 0
        self.src = self.tgt = None
        self.pending = None
        self.n_frames = 0
        self.last_stats = _Stats(n_passes=0, n_outer=0, n_correspondences=0, converged=1, n_cost_only=0)
        self.last_translation_stats = _Stats(n_passes=0, n_outer=0, n_cost_only=0)

    def setResolution(self, leaf): self.leaf = float(leaf)
    def setFixedIterations(self, n): self.fixed = int(n)
    def setOverlapKnn(self, on): pass
    def setUseGraph(self, on): pass
    def setLoadHint(self, mode): pass
    def close(self): pass

    @staticmethod
    def _view(ptr, n, stride):
        return np.ctypeslib.as_array(C.cast(C.c_void_p(int(ptr)), C.POINTER(C.c_float)), shape=(int(n), int(stride))).copy()

    def setInputTargetDevice(self, ptr, n, stride): self.tgt = self._view(ptr, n, stride)
    def setInputSourceDevice(self, ptr, n, stride): self.src = self._view(ptr, n, stride)
    def setInputTarget(self, cloud): self.tgt = np.ascontiguousarray(cloud, np.float32)   # (the host_clouds leg hands host arrays over)
    def setInputSource(self, cloud): self.src = np.ascontiguousarray(cloud, np.float32)

    def register_async(self, guess, t0, g, l, dtn=0.1, dtn1=0.1, lam=0.3):
        assert self.pending is None, "a registration is already in flight on this context"
        self.pending = (np.asarray(t0, np.float64), np.asarray(g, np.float64), np.asarray(l, np.float64))

    def register_wait(self):
        assert self.pending is not None, "no registration in flight"
        t0, g, l = self.pending
        self.pending = None
        self.n_frames += 1
        n = self.src.shape[0]
        if n <= 40000:
            o = pyorc.Reg(pyorc.default_params(voxel_type=pyorc.VOXEL_UNIFORM, voxel_resolution=self.leaf, fixed_iterations=self.fixed, num_threads=1))
            o.set_target(self.tgt); o.set_source(self.src)
            rc, Tf, Td, it, cv = o.align()
            rc2, t, tit = o.compute_translation(t0, g, l)
            assert rc == 0 and rc2 == 0
            self.last_stats = _Stats(n_passes=int(it) + 1, n_outer=int(it), n_correspondences=int(o.correspondences()[0].shape[0]), converged=int(cv), n_cost_only=0)
            self.last_translation_stats = _Stats(n_passes=int(tit) + 1, n_outer=int(tit), n_cost_only=0)
            return Tf, Td, t
        self.last_stats = _Stats(n_passes=21, n_outer=20, n_correspondences=n, converged=1, n_cost_only=14)
        self.last_translation_stats = _Stats(n_passes=9, n_outer=2, n_cost_only=6)
        return np.eye(4, dtype=np.float32), np.eye(4), np.zeros(3)

    def trace(self):
        return []

    def counters(self):
        return dict(frames=self.n_frames, graph_replays=0, graph_captures=0, eager_frames=self.n_frames, topup_frames=0, sync_chunks=0, hint_rot=0, hint_trans=0, walk_lanes=1)


rv.RotVGICP = FakeRotVGICP

import bench  # noqa: E402

if __name__ == "__main__":
    bench.main()
