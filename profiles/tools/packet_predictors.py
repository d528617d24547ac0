import sys; sys.path.insert(0, '.')
import numpy as np
from rolo_amd import synth
exec(open('profiles/tools/heavy_analysis.py').read().split("src, tgt, _ = synth.dense_pair")[0])
i = 1
src, tgt, _ = synth.dense_pair("os1-128", seed=synth.SEED + 2 * i, origin=synth.pool_origin(i))
rec = np.load("gpurun_out/wave_rec_pool%d.npy" % i)
def xcd_block(b, G):
    GROUP, RUN = 512, 64
    if b >= G // GROUP * GROUP: return b
    grp, o = divmod(b, GROUP)
    return grp * GROUP + (o & 7) * RUN + (o >> 3)
G = 1024
inv = {xcd_block(b, G): b for b in range(G)}
feats = []; leaves = []
for which, cloud in enumerate((src, tgt)):
    p = cloud[:, :3].astype(np.float32)
    mn = p.min(0); ext = (p.max(0) - mn).max(); sc = np.float32(1024.0) / np.float32(ext)
    q = np.clip(((p - mn) * sc).astype(np.int64), 0, 1023)
    key = hilbert30(q[:, 0], q[:, 1], q[:, 2]) >> 4
    ps = p[np.argsort(key, kind="stable")]
    for pk in range(len(ps) // 64):
        pts = ps[pk * 64:(pk + 1) * 64]
        step = np.linalg.norm(np.diff(pts, axis=0), axis=1)
        diag = np.linalg.norm(pts.max(0) - pts.min(0))
        med = np.median(step)
        ld = np.array([np.linalg.norm(pts[l * 16:(l + 1) * 16].max(0) - pts[l * 16:(l + 1) * 16].min(0)) for l in range(4)])
        feats.append((diag, diag / med, (step > 8 * med).sum(), ld.sum() / med, ld.max() / med, np.sort(step)[-1] / med))
        w = inv[which * 512 + pk // 4] * 4 + pk % 4
        leaves.append(rec[w, 1])
feats = np.array(feats); leaves = np.array(leaves, float)
names = ["diag", "diag/medstep", "n jumps>8med", "sum leafdiag/medstep", "max leafdiag/medstep", "maxstep/medstep"]
thr_heavy = np.sort(leaves)[-40]
print("heaviest 40 packets: leaves >=", thr_heavy, "; p50", np.median(leaves), "p99", np.percentile(leaves, 99))
for k, nm in enumerate(names):
    f = feats[:, k]
    order = np.argsort(-f)
    for top in (40, 80, 160, 320):
        sel = order[:top]
        print("%-24s top %3d by feature: catches %2d of the 40 heaviest; mean leaves of selected %.1f; heaviest uncaught %d" % (nm, top, (leaves[sel] >= thr_heavy).sum(), leaves[sel].mean(), np.delete(leaves, sel).max()))
