import sys, os, time, ctypes as C; sys.path.insert(0, '.')   # run from the repository root
import numpy as np
from rolo_amd import synth, _lib, profile
from rolo_amd.rotvgicp import RotVGICP
import torch
tag = os.environ.get("ROLO_HIP_LIB", "default").split("librolo_hip")[-1]
for sensor, stride in (("os1-128", 1), ("os1-128", 3), ("os1-64", 1)):
    src, tgt, _ = synth.dense_pair(sensor, col_stride=stride)
    g = RotVGICP(); g.setResolution(0.5)
    ds = torch.from_numpy(src).cuda(); dt = torch.from_numpy(tgt).cuda()
    def step():
        g.setInputTargetDevice(dt.data_ptr(), tgt.shape[0], 4); g.setInputSourceDevice(ds.data_ptr(), src.shape[0], 4)
        g.computeCovariances()
    for _ in range(3): step()
    acc = profile.kernel_times(g, step, reps=5)
    print(tag, sensor, src.shape[0], "build %.4f walk %.4f tail %.4f ms" % tuple(float(np.mean([r.sum() for r in acc[k]])) for k in ("knn_build", "knn_walk", "knn_tail")))
    g.close()
