"""Phases of the resident LM kernel's trials (a -DROLO_LMP_STATS build: python -m rolo_amd.build --tag=lmpstats --flag=-DROLO_LMP_STATS), one context alone:
    ROLO_HIP_LIB=rolo_amd/librolo_hip_lmpstats.so ROLO_LM_FUSED=2 ROLO_LM_PERSIST_WGS=<64|128|256> python profiles/tools/lmpstats.py"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from rolo_amd import synth  # noqa: E402
from rolo_amd._lib import lib  # noqa: E402
from rolo_amd.rotvgicp import RotVGICP  # noqa: E402

src, tgt, _ = synth.dense_pair("os1-128", seed=synth.SEED)
d = (torch.from_numpy(src).cuda(), torch.from_numpy(tgt).cuda(), src.shape[0])
G = -np.asarray(synth.PREV_STEP_T, np.float64)
g = RotVGICP(0); g.setResolution(0.5); g.setFixedIterations(20); g.setFusedLm(int(os.environ.get("ROLO_LM_FUSED", "2")))
f = lib().rolo_debug_lmp_times; f.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
buf = (C.c_ulonglong * 8)()
for it in range(12):
    if it == 4:
        f(buf, 1)
    g.setInputTargetDevice(d[1].data_ptr(), d[2], 4); g.setInputSourceDevice(d[0].data_ptr(), d[2], 4)
    g.register_async(None, np.zeros(3), G, G * 0.97); g.register_wait()
f(buf, 0)
n = max(buf[7], 1)
print("trials", buf[7], "per trial us: body+reduce %.2f  exchange %.2f  step %.2f  total %.2f" % (buf[0] / n / 100, buf[1] / n / 100, buf[2] / n / 100, (buf[0] + buf[1] + buf[2]) / n / 100),
      "| cost-only trials %d: body+reduce %.2f  exchange %.2f  step %.2f | the others: body+reduce %.2f  exchange %.2f  step %.2f |" % (
          buf[6], buf[3] / max(buf[6], 1) / 100, buf[4] / max(buf[6], 1) / 100, buf[5] / max(buf[6], 1) / 100,
          (buf[0] - buf[3]) / max(n - buf[6], 1) / 100, (buf[1] - buf[4]) / max(n - buf[6], 1) / 100, (buf[2] - buf[5]) / max(n - buf[6], 1) / 100),
      "passes", g.last_stats.n_passes, g.last_translation_stats.n_passes, "cost-only", g.last_stats.n_cost_only + g.last_translation_stats.n_cost_only)
