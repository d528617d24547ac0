import sys, ctypes as C; sys.path.insert(0, '.')   # run from the repository root
import numpy as np, torch
from rolo_amd import synth, _lib
from rolo_amd.rotvgicp import RotVGICP
L = _lib.lib(); f = L.rolo_debug_pass_times; f.argtypes = [C.c_void_p]
src, tgt, _ = synth.dense_pair("os1-128")
ds = torch.from_numpy(src).cuda(); dt = torch.from_numpy(tgt).cuda(); n = src.shape[0]
g = RotVGICP(); g.setResolution(0.5); g.setFixedIterations(20)
z = np.zeros(3)
for it in range(6):
    g.setInputTargetDevice(dt.data_ptr(), n, 4); g.setInputSourceDevice(ds.data_ptr(), n, 4)
    g.register_async(None, z, z, z, 0.1, 0.1, 0.3); g.register_wait()
out = np.zeros(8, np.uint64); f(out.ctypes.data)
k = float(out[7]); names = ["first load (stage flag of the state) back", "points, covariances, voxel records in + arithmetic", "block reduction + row store", "-"]
for i in range(4): print("%-50s %7.0f cycles  %.2f us" % (names[i], out[i] / k, out[i] / k / 2100.0))
print("launches with a step:", int(k))
