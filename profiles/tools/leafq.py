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
order_h = np.argsort(key, kind="stable")
tree = cKDTree(p); d, _ = tree.query(p, k=20); r20_all = d[:, 19]

def kd_refine(order, block, leaf=16):
    """inside consecutive blocks of `block` points of the given order: recursive median split on the widest axis down to leaves of 16"""
    out = order.copy()
    def rec(idx):
        if len(idx) <= leaf: return idx
        pts = p[idx]; ax = np.argmax(pts.max(0) - pts.min(0))
        o = np.argsort(pts[:, ax], kind="stable"); h = (len(idx) // 2 + leaf - 1) // leaf * leaf if len(idx) > 2 * leaf else leaf
        h = len(idx) // 2 // leaf * leaf or leaf
        return np.concatenate([rec(idx[o[:h]]), rec(idx[o[h:]])])
    for b in range(0, len(order), block):
        out[b:b + block] = rec(order[b:b + block])
    return out

def evaluate(order, name):
    ps = p[order]; r = r20_all[order]
    n = len(ps); nl = n // 16
    lo = ps[:nl * 16].reshape(nl, 16, 3).min(1); hi = ps[:nl * 16].reshape(nl, 16, 3).max(1)
    diag = np.linalg.norm(hi - lo, axis=1)
    # per packet: leaves whose box is within reach of ANY lane's final sphere (minimum leaf visits of the packet walk)
    ctr = (lo + hi) / 2; half = (hi - lo) / 2
    lt = cKDTree(ctr); maxhalf = np.linalg.norm(half, axis=1).max()
    cnt = []
    per_lane = []
    for pk in range(0, n // 64, 7):   # sample every 7th packet
        qs = ps[pk * 64:(pk + 1) * 64]; rs = r[pk * 64:(pk + 1) * 64]
        cand = lt.query_ball_point(qs.mean(0), np.linalg.norm(qs - qs.mean(0), axis=1).max() + rs.max() + maxhalf)
        cand = np.array(cand)
        dd = np.maximum(np.abs(qs[:, None, :] - ctr[cand][None]) - half[cand][None], 0)   # 64 x cand x 3
        d2 = (dd ** 2).sum(2)
        hit = d2 <= (rs[:, None] ** 2)
        cnt.append(hit.any(0).sum()); per_lane.append(hit.sum(1).mean())
    cnt = np.array(cnt)
    print("%-28s leaf diag p50 %.2f p90 %.2f p99 %.2f max %.2f | min leaves per packet: mean %.1f p50 %.0f p90 %.0f p99 %.0f max %d | per lane %.1f" % (
        name, *np.percentile(diag, [50, 90, 99, 100]), cnt.mean(), *np.percentile(cnt, [50, 90, 99]), cnt.max(), np.mean(per_lane)))
evaluate(order_h, "hilbert")
for blk in (256, 1024, 4096):
    evaluate(kd_refine(order_h, blk), "hilbert + kd inside %d" % blk)
evaluate(kd_refine(order_h, len(order_h)), "full kd (median splits)")
