#!/usr/bin/env python3
"""Generate tests/golden/front_*.npz (run in the authoring container only).

Source of truth = oracle/twin_front.py, the independent numpy statement of the projection / feature extraction (the
reference itself cannot be built or imported here; DESIGN.md "Oracle"). Inputs: seeded synthetic raw frames
(rolo_amd.synth), thinned to keep the fixtures small, with duplicates, out-of-range points, an unknown ring and NaNs mixed
in. Each .npz holds the raw input and every output of K1-K4 (the extracted cloud as `owner`: the raw index behind every
valid pixel — its coordinates are the raw ones, its intensity ring * z).

    python tests/golden/make_golden_front.py
"""
import os
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from rolo_amd import synth  # noqa: E402
from oracle import twin_front as tw  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
CASES = {"front_vlp16": ("vlp16", 16, 1800, 4), "front_os64": ("os1-64", 64, 1024, 8)}


def main():
    for name, (sensor, n_scan, H, col_stride) in CASES.items():
        fr = synth.make_frame(sensor, synth.rpy_to_R(0.01, -0.02, 0.3), np.array([0.4, -0.2, 0.05]), synth.SEED + 7)
        xyz = np.array(fr.xyz, np.float32); ring = np.array(fr.ring, np.uint16)
        # the synthetic frames are in firing order (ring fastest): keep every col_stride-th column
        keep = (np.arange(xyz.shape[0]) // n_scan) % col_stride == 0
        xyz, ring = xyz[keep], ring[keep]
        rs = np.random.RandomState(20260926)
        dup = rs.choice(xyz.shape[0], 200, replace=False)                    # later duplicates of a pixel must lose
        xyz = np.concatenate([xyz, xyz[dup] * np.float32(1.001)]); ring = np.concatenate([ring, ring[dup]])
        xyz[rs.choice(xyz.shape[0], 30, replace=False)] *= np.float32(0.01)  # below lidarMinRange
        ring[rs.choice(xyz.shape[0], 10, replace=False)] = n_scan + 3        # ring outside the image
        xyz[rs.choice(xyz.shape[0], 10, replace=False), 1] = np.nan          # refused by the node; dropped by the arithmetic
        p = tw.project(xyz, ring, n_scan, H)
        e = tw.extract_features(p, n_scan)
        np.savez_compressed(os.path.join(OUT, name + ".npz"), xyz=xyz, ring=ring, n_scan=n_scan, horizon_scan=H,
                            point_col_ind=p["point_col_ind"], point_range=p["point_range"], start_ring=p["start_ring"],
                            end_ring=p["end_ring"], owner=p["owner"].astype(np.int32), curvature=e["curvature"], picked=e["picked"].astype(np.int8),
                            label=e["label"].astype(np.int8), corner=e["corner"], surface=e["surface"])
        print(name, "raw", xyz.shape[0], "valid", p["n"], "corner", e["corner"].shape[0], "surface", e["surface"].shape[0])


if __name__ == "__main__":
    main()
