"""Per-wavefront records of the cooperative walk (build: python -m rolo_amd.build --tag=knnstats --flag=-DROLO_KNN_STATS; run with ROLO_HIP_LIB=rolo_amd/librolo_hip_knnstats.so):
when a wavefront finishes its own packet, when it stops helping, how many leaves it scored in either role, who published (and how deep a stack), how many help
sessions ran and whether the workgroup ran out of result slots. POOL_PAIR selects the stand point (0 = the nominal pair)."""
import sys, os, ctypes as C; sys.path.insert(0, '.')
import numpy as np
from rolo_amd import synth, _lib
from rolo_amd.rotvgicp import RotVGICP
L = _lib.lib()
fr = L.rolo_debug_wave_records; fr.argtypes = [C.c_void_p]
i = int(os.environ.get("POOL_PAIR", "0")); stride = int(os.environ.get("STRIDE", "1"))
src, tgt, _ = synth.dense_pair("os1-128", seed=synth.SEED + 2 * i, origin=synth.pool_origin(i), col_stride=stride)
g = RotVGICP(); g.setResolution(0.5)
for it in range(3):
    g.setInputTarget(tgt); g.setInputSource(src); g.computeCovariances()
rec = np.zeros((16384, 8), np.uint32); fr(rec.ctypes.data)
packets = 2 * ((src.shape[0] + 63) // 64)
NW = int(os.environ.get("ROLO_KNN_COOP_NW", "0")) or (16 if packets >= 4096 else (8 if packets >= 2048 else 4))
G4 = 2 * ((src.shape[0] + 255) // 256); nblk = (G4 + NW // 4 - 1) // (NW // 4); nw = nblk * NW
rec = rec[:nw]
t0 = rec[:, 4].astype(np.int64); t1 = rec[:, 5].astype(np.int64); t2 = rec[:, 6].astype(np.int64)
base = t0.min(); t0 = (t0 - base) / 100.0; t1 = (t1 - base) / 100.0; t2 = (t2 - base) / 100.0
donor = (rec[:, 3] & 1) != 0; sp = (rec[:, 3] >> 8) & 0xff; sess = (rec[:, 3] >> 16) & 0x7fff; full = (rec[:, 3] >> 31) != 0   # sp: entries published sess = (rec[:, 3] >> 16) & 0x7fff; full = (rec[:, 3] >> 31) != 0
own = rec[:, 1].astype(float); helped = rec[:, 2].astype(float)
print(f"pair {i} stride {stride}: {src.shape[0]} pts, NW {NW}, waves {nw}, budget {os.environ.get('ROLO_KNN_BUDGET', 'default')}")
print(f"donors {donor.mean() * 100:.1f} % (entries published: mean {sp[donor].mean() if donor.any() else 0:.1f}, max {sp.max()}); help sessions per wave {sess.mean():.2f}; waves that met full slots {full.mean() * 100:.1f} %")
print(f"leaves: own packet mean {own.mean():.1f} (donors {own[donor].mean() if donor.any() else 0:.1f}, p99 {np.percentile(own, 99):.0f}, max {own.max():.0f}); while helping mean {helped.mean():.1f}; total per wave {(own + helped).mean():.1f}")
for name, x in (("start", t0), ("own packet done", t1), ("helping done", t2)):
    print(f"{name:16s} mean {x.mean():7.1f} p50 {np.percentile(x, 50):7.1f} p90 {np.percentile(x, 90):7.1f} p99 {np.percentile(x, 99):7.1f} max {x.max():7.1f} us")
blk_end = t2.reshape(nblk, NW).max(1); blk_own = t1.reshape(nblk, NW).max(1)
print(f"workgroup end: mean {blk_end.mean():.1f} p90 {np.percentile(blk_end, 90):.1f} max {blk_end.max():.1f} us; last own-packet per workgroup: mean {blk_own.mean():.1f} max {blk_own.max():.1f}")
work = (own + helped).reshape(nblk, NW).sum(1)
print(f"leaves per workgroup: mean {work.mean():.0f} max {work.max():.0f} (x{work.max() / work.mean():.2f})")
worst = np.argsort(-blk_end)[:5]
for b in worst:
    sl = slice(b * NW, (b + 1) * NW)
    print(f"  wg {b}: end {blk_end[b]:.1f} us, own-done {np.round(t1[sl], 0).astype(int).tolist()}, own leaves {own[sl].astype(int).tolist()}, helped {helped[sl].astype(int).tolist()}, donors {donor[sl].astype(int).tolist()}, sp {sp[sl].tolist()}, full {int(full[sl].any())}")
g.close()
