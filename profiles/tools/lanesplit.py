"""CPU only: how separable are the heavy packets of the walk BY LANES? For every packet of 64 curve-adjacent queries: the leaves it must visit given its lanes' final
20-NN radii (the walk's measured leaf count is at that minimum, leafq.py), and the same for its lane halves and quarters. If a heavy packet's queries sit on scattered
fragments, a quarter of the lanes needs about a quarter of the leaves and handing lane groups to idle wavefronts would cut its time almost in proportion; if they sit in
one dense neighbourhood, every group still needs most of the leaves and nothing is gained.   python profiles/tools/lanesplit.py [pool pair]"""
import sys; sys.path.insert(0, '.')
import numpy as np
from rolo_amd import synth
from scipy.spatial import cKDTree
exec(open('profiles/tools/heavy_analysis.py').read().split("src, tgt, _ = synth.dense_pair")[0])
i = int(sys.argv[1]) if len(sys.argv) > 1 else 4
src, tgt, _ = synth.dense_pair("os1-128", seed=synth.SEED + 2 * i, origin=synth.pool_origin(i))
p = tgt[:, :3].astype(np.float32)
mn = p.min(0); ext = (p.max(0) - mn).max(); sc = np.float32(1024.0) / np.float32(ext)
q = np.clip(((p - mn) * sc).astype(np.int64), 0, 1023)
key = hilbert30(q[:, 0], q[:, 1], q[:, 2]) >> 4
order = np.argsort(key, kind="stable")
tree = cKDTree(p); d, _ = tree.query(p, k=20); r20 = d[:, 19]
ps = p[order]; r = r20[order]
n = len(ps); nl = n // 16
lo = ps[:nl * 16].reshape(nl, 16, 3).min(1); hi = ps[:nl * 16].reshape(nl, 16, 3).max(1)
ctr = (lo + hi) / 2; half = (hi - lo) / 2
lt = cKDTree(ctr); maxhalf = np.linalg.norm(half, axis=1).max()
rows = []
for pk in range(0, n // 64):
    qs = ps[pk * 64:(pk + 1) * 64]; rs = r[pk * 64:(pk + 1) * 64]
    cand = np.array(lt.query_ball_point(qs.mean(0), np.linalg.norm(qs - qs.mean(0), axis=1).max() + rs.max() + maxhalf))
    dd = np.maximum(np.abs(qs[:, None, :] - ctr[cand][None]) - half[cand][None], 0)
    hit = (dd ** 2).sum(2) <= (rs[:, None] ** 2)          # 64 x cand
    full = hit.any(0).sum()
    halves = [hit[a:a + 32].any(0).sum() for a in (0, 32)]
    quarters = [hit[a:a + 16].any(0).sum() for a in (0, 16, 32, 48)]
    rows.append((full, max(halves), sum(halves), max(quarters), sum(quarters)))
R = np.array(rows, float)
full = R[:, 0]
print(f"pair {i}: {len(R)} packets; leaves per packet mean {full.mean():.1f} p99 {np.percentile(full, 99):.0f} max {full.max():.0f}")
for name, sel in (("all packets", full > 0), ("packets above 34 leaves", full > 34), ("packets above 45 leaves", full > 45), ("the 40 heaviest", full >= np.sort(full)[-40])):
    S = R[sel]
    print(f"{name:26s} n {sel.sum():5d}: full {S[:, 0].mean():6.1f} | largest half {S[:, 1].mean():6.1f} ({(S[:, 1] / S[:, 0]).mean():.2f} of full), both halves {S[:, 2].mean():6.1f} | "
          f"largest quarter {S[:, 3].mean():6.1f} ({(S[:, 3] / S[:, 0]).mean():.2f}), all quarters {S[:, 4].mean():6.1f}")
