"""The resident LM kernel with N contexts in flight (a -DROLO_LMP_STATS build): phases of its trials AND its admission — how long workgroup 0 waits, from its own start,
until all workgroups of the launch are resident (they can only be placed on CUs whose register files other kernels' wavefronts have left).
    ROLO_HIP_LIB=rolo_amd/librolo_hip_lmpstats.so python profiles/tools/lmpstats_load.py [contexts=4]"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from rolo_amd import synth  # noqa: E402
from rolo_amd._lib import lib  # noqa: E402
from rolo_amd.rotvgicp import RotVGICP  # noqa: E402

nctx = int(sys.argv[1]) if len(sys.argv) > 1 else 4
src, tgt, _ = synth.dense_pair("os1-128", seed=synth.SEED)
d = (torch.from_numpy(src).cuda(), torch.from_numpy(tgt).cuda(), src.shape[0])
G = -np.asarray(synth.PREV_STEP_T, np.float64)
gs = []
for _ in range(nctx):
    g = RotVGICP(0); g.setResolution(0.5); g.setFixedIterations(20); gs.append(g)
ft = lib().rolo_debug_lmp_times; ft.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
fa = lib().rolo_debug_lmp_admission; fa.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
bt = (C.c_ulonglong * 8)(); ba = (C.c_ulonglong * 4)()


def enq(g):
    g.setInputTargetDevice(d[1].data_ptr(), d[2], 4); g.setInputSourceDevice(d[0].data_ptr(), d[2], 4)
    g.register_async(None, np.zeros(3), G, G * 0.97)


for g in gs: enq(g)
K = 120
for it in range(K):
    if it == 20:
        for g in gs: g.register_wait()
        ft(bt, 1); fa(ba, 1); t0 = time.perf_counter()
        for g in gs: enq(g)
    for g in gs:
        g.register_wait()
        if it + 1 < K: enq(g)
dt = time.perf_counter() - t0
ft(bt, 0); fa(ba, 0)
n = max(bt[7], 1); na = max(ba[1], 1)
print("contexts %d: %.0f frames/s | per trial us: body+reduce %.2f  exchange %.2f  step %.2f  total %.2f (cost-only: %.2f %.2f %.2f) | per launch us: admission %.1f (max %.1f), whole kernel %.1f, trials %.1f | bails %s" % (
    nctx, (K - 20) * nctx / dt, bt[0] / n / 100, bt[1] / n / 100, bt[2] / n / 100, (bt[0] + bt[1] + bt[2]) / n / 100, bt[3] / max(bt[6], 1) / 100, bt[4] / max(bt[6], 1) / 100, bt[5] / max(bt[6], 1) / 100,
    ba[0] / na / 100, ba[2] / 100, ba[3] / na / 100, (bt[0] + bt[1] + bt[2]) / na / 100, [g.counters()["persist_bails"] for g in gs]))
