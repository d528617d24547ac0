import sys, ctypes as C; sys.path.insert(0, '.')   # run from the repository root
import numpy as np, torch
from rolo_amd import synth, _lib
from rolo_amd.frontend import front_params
from rolo_amd.odometry import LidarOdometry
L = _lib.lib()
f = L.rolo_debug_extract_times; f.argtypes = [C.c_void_p]
S = synth.SENSORS["os1-128"]; fp = front_params(n_scan=S[0], horizon_scan=S[1])
R = np.eye(3); t = np.zeros(3); frames = []
for k in range(3):
    fr = synth.make_frame("os1-128", R, t, synth.SEED + k)
    frames.append((torch.from_numpy(np.ascontiguousarray(fr.xyz, np.float32)).cuda(), torch.from_numpy(np.ascontiguousarray(fr.ring, np.uint16).view(np.int16)).cuda(), fr.xyz.shape[0], fr.xyz.shape[1]))
    t = t + R @ np.array([0.3, 0.02 * k, 0.0])
od = LidarOdometry(0, 0.3); stamp = 100.0
for i in range(6):
    x, r, n_raw, stride = frames[i % 3]; stamp += 0.1
    od.submit(fp, stamp, x.data_ptr(), r.data_ptr(), n_raw=n_raw, stride=stride); od.collect()
    if i == 0: od.odometryHandler(stamp + 0.05)
out = np.zeros((128, 8), np.uint64); f(out.ctypes.data)
d = np.diff(out[:, :7].astype(np.int64), axis=1) / 2100.0   # ~2.1 GHz shader clock -> us
names = ["load window + keys", "sector sort (6 x seg)", "12 pick stages + scan lists", "write back + bbox + grid keys", "voxel-grid sort", "run heads + centroids"]
for k in range(6): print("%-32s mean %7.1f us  max %7.1f" % (names[k], d[:, k].mean(), d[:, k].max()))
print("total mean %.1f us max %.1f" % (d.sum(1).mean(), d.sum(1).max()))
