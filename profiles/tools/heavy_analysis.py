# offline: which packets of the walk are heavy? wave records from the stats build (gpurun_out/wave_rec_1.npy) vs packet geometry
import sys; sys.path.insert(0, '.')   # run from the repository root
import numpy as np
from rolo_amd import synth
from scipy.spatial import cKDTree
def expand10(v):
    v = v & 0x3ff
    v = (v | (v << 16)) & 0x030000ff
    v = (v | (v << 8)) & 0x0300f00f
    v = (v | (v << 4)) & 0x030c30c3
    v = (v | (v << 2)) & 0x09249249
    return v
def hilbert30(x, y, z):
    X = [x.astype(np.uint32).copy(), y.astype(np.uint32).copy(), z.astype(np.uint32).copy()]
    Q = 512
    while Q > 1:
        P = np.uint32(Q - 1)
        for i in range(3):
            m = (X[i] & Q) != 0
            X[0] = np.where(m, X[0] ^ P, X[0])
            t = (X[0] ^ X[i]) & P
            t = np.where(m, 0, t).astype(np.uint32)
            X[0] = X[0] ^ t; X[i] = X[i] ^ t
        Q >>= 1
    X[1] ^= X[0]; X[2] ^= X[1]
    t = np.zeros_like(X[0]); Q = 512
    while Q > 1:
        t = np.where((X[2] & Q) != 0, t ^ np.uint32(Q - 1), t); Q >>= 1
    X[0] ^= t; X[1] ^= t; X[2] ^= t
    return (expand10(X[0]) << 2) | (expand10(X[1]) << 1) | expand10(X[2])
src, tgt, _ = synth.dense_pair("os1-128")
rec = np.load("gpurun_out/wave_rec_1.npy")
def xcd_block(b, G):   # runs of 64 blocks dealt round-robin (G > 512)
    GROUP, RUN = 512, 64
    if b >= G // GROUP * GROUP: return b
    grp, o = divmod(b, GROUP)
    return grp * GROUP + (o & 7) * RUN + (o >> 3)
G = 1024
out = []
for which, cloud in enumerate((src, tgt)):
    p = cloud[:, :3].astype(np.float32)
    mn = p.min(0); ext = (p.max(0) - mn).max(); sc = np.float32(1024.0) / np.float32(ext)
    q = np.clip(((p - mn) * sc).astype(np.int64), 0, 1023)
    key = hilbert30(q[:, 0], q[:, 1], q[:, 2])
    order = np.argsort(key, kind="stable")
    ps = p[order]
    tree = cKDTree(p); d, _ = tree.query(ps, k=21); r20 = d[:, 20]
    n = ps.shape[0]
    for pk in range(n // 64):
        pts = ps[pk * 64:(pk + 1) * 64]
        diag = np.linalg.norm(pts.max(0) - pts.min(0))
        leafd = np.mean([np.linalg.norm(pts[l * 16:(l + 1) * 16].max(0) - pts[l * 16:(l + 1) * 16].min(0)) for l in range(4)])
        r = r20[pk * 64:(pk + 1) * 64]
        out.append((which, pk, diag, leafd, r.mean(), r.max(), np.linalg.norm(pts.mean(0))))
out = np.array(out)
# map packet -> wave id: blk (logical) = which*512 + pk//4 ; wave = pk%4 ; physical block index b with xcd_block(b)=blk
inv = {xcd_block(b, G): b for b in range(G)}
wid = np.array([inv[int(w) * 512 + int(pk) // 4] * 4 + int(pk) % 4 for w, pk in out[:, :2]])
nodes, leaves, ins = rec[wid, 0].astype(float), rec[wid, 1].astype(float), rec[wid, 2].astype(float)
dur = (rec[wid, 5].astype(np.int64) - rec[wid, 4].astype(np.int64)) / 100.0
feat = {"diag": out[:, 2], "leaf diag": out[:, 3], "diag/leafdiag": out[:, 2] / out[:, 3], "r20 mean": out[:, 4], "r20 max": out[:, 5], "diag/r20mean": out[:, 2] / out[:, 4],
        "(diag+2 r20max)^3/r20mean^3": ((out[:, 2] + 2 * out[:, 5]) / out[:, 4]) ** 3, "range": out[:, 6]}
for k, v in feat.items():
    print("%-28s corr with leaves %.3f  with log: %.3f" % (k, np.corrcoef(v, leaves)[0, 1], np.corrcoef(np.log(v), np.log(leaves))[0, 1]))
top = np.argsort(-leaves)[:12]
for i in top: print("wave %5d leaves %3d nodes %3d dur %6.1f | diag %.2f leafdiag %.2f ratio %.1f r20 mean %.2f max %.2f range %.1f" % (wid[i], leaves[i], nodes[i], dur[i], out[i, 2], out[i, 3], out[i, 2] / out[i, 3], out[i, 4], out[i, 5], out[i, 6]))
print("median: diag %.2f leafdiag %.2f ratio %.1f r20 %.2f" % (np.median(out[:, 2]), np.median(out[:, 3]), np.median(out[:, 2] / out[:, 3]), np.median(out[:, 4])))
# how well would a threshold on ratio pick the heavy ones?
ratio = out[:, 2] / out[:, 3]
for thr in (3, 4, 5, 6):
    sel = ratio > thr
    print("ratio > %d: %4d packets (%.1f %%), catches %d of the 41 heaviest (leaves >= %d), mean leaves of selected %.1f" % (thr, sel.sum(), 100 * sel.mean(), (sel & (leaves >= np.sort(leaves)[-41])).sum(), np.sort(leaves)[-41], leaves[sel].mean()))
