import sys; sys.path.insert(0, '.')   # run from the repository root
import numpy as np
from rolo_amd import synth
from rolo_amd.rotvgicp import RotVGICP
from oracle import pyorc
rng = np.random.default_rng(7)
src0, tgt0, _ = synth.dense_pair("os1-128")
sizes = [21, 22, 63, 64, 65, 127, 129, 255, 256, 257, 1000, 1023, 1025, 4095, 4097, 16385, 50000, 100003, 131071]
bad = 0
for n in sizes + [int(x) for x in rng.integers(30, 131072, 12)]:
    si = np.sort(rng.choice(src0.shape[0], n, replace=False)); ti = np.sort(rng.choice(tgt0.shape[0], max(21, n - int(rng.integers(0, 7))), replace=False))
    src = np.ascontiguousarray(src0[si]); tgt = np.ascontiguousarray(tgt0[ti])
    g = RotVGICP(); g.setResolution(1.0); g.setInputTarget(tgt); g.setInputSource(src)
    g.buildVoxelMap(); k, c, m, v = g.voxels(); o = np.lexsort(k.T[::-1]); ref = (k[o], c[o], m[o], v[o])
    cov_ref = g.getTargetCovariances().copy()
    g.close()
    for rep in range(2):   # second repetition replays / recaptures with the same sizes
        g = RotVGICP(); g.setResolution(1.0); g.setInputTarget(tgt); g.setInputSource(src)
        z = np.zeros(3)
        for it in range(3 if rep else 1):
            if it: g.setInputTarget(tgt); g.setInputSource(src)
            g.register_async(None, z, z, z, 0.1, 0.1, 0.3)
            try: g.register_wait()
            except Exception as ex: print('  n', n, 'register:', str(ex)[-60:])
        k, c, m, v = g.voxels(); o = np.lexsort(k.T[::-1])
        ok = all(np.array_equal(a, b) for a, b in zip(ref, (k[o], c[o], m[o], v[o]))) and np.array_equal(cov_ref, g.getTargetCovariances())
        g.close()
        if not ok: bad += 1; print("MISMATCH n", n, "rep", rep)
    if n <= 4097:
        idx_o, d2_o = pyorc.knn(src, 20)
        g = RotVGICP(); g.setResolution(1.0); g.setInputTarget(tgt); g.setInputSource(src)
        idx_g, d2_g = g.knn(0); g.close()
        if not (np.array_equal(idx_o, idx_g) and np.array_equal(d2_o, d2_g)): bad += 1; print("KNN MISMATCH n", n)
print("stress done, mismatches:", bad)
