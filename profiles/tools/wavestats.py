import sys, os, ctypes as C; sys.path.insert(0, '.')   # run from the repository root
import numpy as np
from rolo_amd import synth, _lib
from rolo_amd.rotvgicp import RotVGICP
L = _lib.lib()
fr = L.rolo_debug_wave_records; fr.argtypes = [C.c_void_p]
for sensor, stride in (("os1-128", 1), ("os1-128", 3)):
    src, tgt, _ = synth.dense_pair(sensor, col_stride=stride)
    g = RotVGICP(); g.setResolution(0.5)
    for it in range(3):
        g.setInputTarget(tgt); g.setInputSource(src); g.computeCovariances()
    rec = np.zeros((16384, 8), np.uint32)
    fr(rec.ctypes.data)
    QPB = 4 * int(os.environ.get('PACKET', '64')); nblk = 2 * ((src.shape[0] + QPB - 1) // QPB)
    nw = 4 * nblk
    rec = rec[:nw]
    np.save("gpurun_out/wave_rec_%d.npy" % stride, rec)
    st = rec[:, 4].astype(np.int64); en = rec[:, 5].astype(np.int64)
    ok = en != 0
    t0 = st[ok].min(); st = (st - t0) / 100.0; en = (en - t0) / 100.0
    dur = en - st
    print(sensor, src.shape[0], "waves", nw, "valid", int(ok.sum()))
    for name, x in (("nodes", rec[:, 0]), ("leaves", rec[:, 1]), ("ins", rec[:, 2]), ("push", rec[:, 3]), ("start", st), ("end", en), ("dur", dur)):
        x = x.astype(np.float64)[ok]
        print(f"{name:7s} mean {x.mean():9.1f} p50 {np.percentile(x,50):9.1f} p90 {np.percentile(x,90):9.1f} p99 {np.percentile(x,99):9.1f} max {x.max():9.1f}")
    # the insert alone: how many of the 64 lanes are live in an execution (the executions are the UNION over the lanes of the candidates that
    # beat a lane's bound), and what a per-lane queue would execute instead (per leaf the MAXIMUM over the lanes of the candidates accepted)
    ins = rec[:, 2].astype(np.float64)[ok]; lanes = rec[:, 6].astype(np.float64)[ok]; mx = rec[:, 7].astype(np.float64)[ok]
    print("insert executions per wave %.1f, live lanes per execution %.1f of 64 (%.1f %%), max-per-lane executions per wave %.1f (%.2f x the union)" % (
        ins.mean(), lanes.sum() / max(ins.sum(), 1), 100 * lanes.sum() / max(64 * ins.sum(), 1), mx.mean(), mx.sum() / max(ins.sum(), 1)))
    A = np.c_[rec[:, 0], rec[:, 1], rec[:, 2], np.ones(nw)].astype(np.float64)[ok]
    coef, *_ = np.linalg.lstsq(A, dur[ok], rcond=None)
    print("dur_us ~ %.3f*nodes + %.3f*leaves + %.3f*ins + %.1f   (corr of fit %.3f)" % (*coef, np.corrcoef(A @ coef, dur[ok])[0, 1]))
    for t in range(0, int(en[ok].max()) + 40, 40):
        print("t=%4d us running %d" % (t, int(((st <= t) & (en > t) & ok).sum())))
    order = np.argsort(-en)[:12]
    print("last waves (wid, start, dur, nodes, leaves, ins):", [(int(i), round(float(st[i]), 1), round(float(dur[i]), 1), int(rec[i, 0]), int(rec[i, 1]), int(rec[i, 2])) for i in order])
    g.close()
