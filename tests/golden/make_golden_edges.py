"""Plants float32 points whose POLAR voxel coordinate lies within 5e-13 bins of a bin edge (vmp_voxel.hpp:208-211, resolution 0.175 / 0.175 / 2.0) — where a libm
that is a few ulp off files a point under the neighbouring key. Writes tests/golden/polar_edge_points.npz: the points, the keys of the C++ oracle (glibc) and, for
the record, how many candidates were dropped because glibc itself does not return the correctly rounded atan2 / acos there (decided with the double-double
Newton refinement of rolo_amd/csrc/polar_exact.hpp restated in numpy — scratch-free: the helper lives in this file). Run from the repository root (~1 min)."""
import os, re, sys
import numpy as np
sys.path.insert(0, os.getcwd())
from oracle import pyorc

RES = (0.175, 0.175, 2.0)
src = open('rolo_amd/csrc/polar_exact_consts.hpp').read()


def arr(name):
    m = re.search(name + r"\[\d+\]\[2\] = \{(.*?)\};", src).group(1)
    return np.array([float.fromhex(x) for x in re.findall(r"-?0x[0-9a-f.]+p[+-]\d+", m)]).reshape(-1, 2)


SIN_K, COS_K, SIN_C, COS_C = arr("SIN_K"), arr("COS_K"), arr("SIN_C"), arr("COS_C")
PI8_H = float.fromhex(re.search(r"PI8_H = (\S+),", src).group(1)); PI8_L = float.fromhex(re.search(r"PI8_L = (\S+);", src).group(1))
INV = float.fromhex(re.search(r"INV_PI8 = (\S+);", src).group(1))


def split(a):
    t = 134217729.0 * a; hi = t - (t - a); return hi, a - hi


def two_prod(a, b):
    p = a * b; ah, al = split(a); bh, bl = split(b); return p, ((ah * bh - p) + ah * bl + al * bh) + al * bl


def two_sum(a, b):
    s = a + b; bb = s - a; return s, (a - (s - bb)) + (b - bb)


def qts(a, b):
    s = a + b; return s, b - (s - a)


def add(a, b):
    sh, sl = two_sum(a[0], b[0]); th, tl = two_sum(a[1], b[1]); sl = sl + th; sh, sl = qts(sh, sl); sl = sl + tl; return qts(sh, sl)


def neg(a): return (-a[0], -a[1])


def mul(a, b):
    ph, pl = two_prod(a[0], b[0]); pl = pl + (a[0] * b[1] + a[1] * b[0]); return qts(ph, pl)


def mul_d(a, b):
    ph, pl = two_prod(a[0], b); pl = a[1] * b + pl; return qts(ph, pl)


def sincos_dd(a):
    k = np.rint(a * INV).astype(int)
    r = add((a, np.zeros_like(a)), neg(mul_d((PI8_H + 0 * a, PI8_L + 0 * a), k.astype(float))))
    r2 = mul(r, r)
    ps = (SIN_C[11, 0] + 0 * a, SIN_C[11, 1] + 0 * a); pc = (COS_C[11, 0] + 0 * a, COS_C[11, 1] + 0 * a)
    for i in range(10, -1, -1):
        ps = add(mul(ps, r2), (SIN_C[i, 0] + 0 * a, SIN_C[i, 1] + 0 * a)); pc = add(mul(pc, r2), (COS_C[i, 0] + 0 * a, COS_C[i, 1] + 0 * a))
    sr = add(r, mul(r, mul(ps, r2))); cr = add((1.0 + 0 * a, 0 * a), mul(pc, r2))
    ka = np.abs(k); sg = np.where(k < 0, -1.0, 1.0)
    sk = (sg * SIN_K[ka, 0], sg * SIN_K[ka, 1]); ck = (COS_K[ka, 0], COS_K[ka, 1])
    return add(mul(sk, cr), mul(ck, sr)), add(mul(ck, cr), neg(mul(sk, sr)))


def atan2_cr(y, x):
    a0 = np.arctan2(y, x); s, c = sincos_dd(a0)
    num = add(mul_d(c, y), neg(mul_d(s, x))); den = x * c[0] + y * s[0]
    return a0 + (num[0] + num[1]) / den


def acos_cr(v):
    b0 = np.arccos(v); s, c = sincos_dd(b0); num = add(c, (-v, 0 * v))
    return np.where(s[0] == 0, b0, b0 + (num[0] + num[1]) / np.where(s[0] == 0, 1.0, s[0]))


def keys_cr(p):
    x, y, z = (p[:, i].astype(np.float64) for i in range(3))
    r = np.sqrt((x * x + y * y) + z * z)
    return np.stack([np.floor((atan2_cr(y, x) + np.pi) / RES[0]), np.floor(acos_cr(z / r) / RES[1]), np.floor(r / RES[2])], 1).astype(np.int32)


def main():
    rng = np.random.default_rng(20260927)
    found = []
    M = 4_000_000
    # theta edges
    want_t, want_p, want_r = 620, 360, 60
    got = 0
    while got < want_t:
        k = rng.integers(1, 36, M); th = k * RES[0] - np.pi
        rho = rng.uniform(3.0, 45.0, M)
        x = (rho * np.cos(th)).astype(np.float32); y = (rho * np.sin(th)).astype(np.float32)
        q = (np.arctan2(y.astype(np.float64), x.astype(np.float64)) + np.pi) / RES[0]
        hit = np.abs(q - np.rint(q)) < 5e-13
        n = int(hit.sum())
        if n:
            z = rng.uniform(-3.0, 6.0, n).astype(np.float32)
            found.append(np.stack([x[hit], y[hit], z], 1)); got += n
    got = 0
    while got < want_p:
        k = rng.integers(2, 17, M); ph = k * RES[1]
        r = rng.uniform(3.0, 45.0, M); az = rng.uniform(-np.pi, np.pi, M)
        x = (r * np.sin(ph) * np.cos(az)).astype(np.float32); y = (r * np.sin(ph) * np.sin(az)).astype(np.float32); z = (r * np.cos(ph)).astype(np.float32)
        xd, yd, zd = x.astype(np.float64), y.astype(np.float64), z.astype(np.float64)
        rr = np.sqrt((xd * xd + yd * yd) + zd * zd)
        q = np.arccos(zd / rr) / RES[1]
        hit = np.abs(q - np.rint(q)) < 5e-13
        n = int(hit.sum())
        if n:
            found.append(np.stack([x[hit], y[hit], z[hit]], 1)); got += n
    got = 0
    while got < want_r:   # r / 2.0 an integer to the last bits: sqrt and the division are IEEE on both sides — no hazard, planted for completeness
        k = rng.integers(2, 20, M); r = 2.0 * k
        az = rng.uniform(-np.pi, np.pi, M); el = rng.uniform(-0.3, 0.3, M)
        x = (r * np.cos(el) * np.cos(az)).astype(np.float32); y = (r * np.cos(el) * np.sin(az)).astype(np.float32); z = (r * np.sin(el)).astype(np.float32)
        xd, yd, zd = x.astype(np.float64), y.astype(np.float64), z.astype(np.float64)
        q = np.sqrt((xd * xd + yd * yd) + zd * zd) / RES[2]
        hit = np.abs(q - np.rint(q)) < 5e-13
        n = int(hit.sum())
        if n:
            found.append(np.stack([x[hit], y[hit], z[hit]], 1)); got += n
    pts = np.concatenate(found).astype(np.float32)
    pts4 = np.concatenate([pts, np.ones((pts.shape[0], 1), np.float32)], 1)
    k_or = pyorc.voxel_keys(pts4, 0, polar_res=RES)
    k_cr = keys_cr(pts)
    same = (k_or == k_cr).all(1)
    print(f"candidates {pts.shape[0]}; the oracle's libm (glibc) disagrees with the correctly rounded functions on {int((~same).sum())}: dropped")
    keep = pts4[same]
    np.savez_compressed("tests/golden/polar_edge_points.npz", points=keep, keys=k_or[same], dropped_glibc_not_correctly_rounded=np.array([int((~same).sum())]),
                        candidates=np.array([pts.shape[0]]))
    print("kept", keep.shape[0])


if __name__ == "__main__":
    main()
