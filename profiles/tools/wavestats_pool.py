import sys, os, ctypes as C; sys.path.insert(0, '.')
import numpy as np
from rolo_amd import synth, _lib
from rolo_amd.rotvgicp import RotVGICP
L = _lib.lib()
fr = L.rolo_debug_wave_records; fr.argtypes = [C.c_void_p]
for i in (4, 6, 0):
    src, tgt, _ = synth.dense_pair("os1-128", seed=synth.SEED + 2 * i, origin=synth.pool_origin(i))
    g = RotVGICP(); g.setResolution(0.5)
    for it in range(3):
        g.setInputTarget(tgt.copy()); g.setInputSource(src.copy()); g.computeCovariances()
    rec = np.zeros((16384, 8), np.uint32); fr(rec.ctypes.data)
    nw = 4 * 2 * ((src.shape[0] + 255) // 256); rec = rec[:nw]
    st = rec[:, 4].astype(np.int64); en = rec[:, 5].astype(np.int64); ok = en != 0
    t0 = st[ok].min(); st = (st - t0) / 100.0; en = (en - t0) / 100.0; dur = en - st
    r = np.linalg.norm(src[:, :3], axis=1)
    print("pair", i, "origin", synth.pool_origin(i), "range percentiles 10/50/90: %.1f %.1f %.1f m; share within 3 m: %.3f" % (*np.percentile(r, [10, 50, 90]), (r < 3).mean()))
    for name, x in (("nodes", rec[:, 0]), ("leaves", rec[:, 1]), ("ins", rec[:, 2]), ("dur", dur), ("end", en)):
        x = x.astype(np.float64)[ok]
        print(f"  {name:7s} mean {x.mean():8.1f} p50 {np.percentile(x,50):8.1f} p90 {np.percentile(x,90):8.1f} p99 {np.percentile(x,99):8.1f} max {x.max():8.1f}")
    # which waves are heavy: by cloud (first half = source) and by the range of their queries is not known here; report by wave index decile
    lv = rec[:, 1].astype(np.float64)
    print("  leaves per wave by decile of wave index:", [round(float(c.mean()), 1) for c in np.array_split(lv, 10)])
    np.save('gpurun_out/wave_rec_pool%d.npy' % i, rec)
    g.close()
