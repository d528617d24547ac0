#!/usr/bin/env python3
"""tests/golden/<case>.npz -> a directory of raw little-endian arrays + params.txt, the input of tools/dump_reference_golden.cpp
(which runs in a ROLO workspace, where numpy's .npz reader is not a given).

    python tools/export_golden_inputs.py [case ...] --out /tmp/rolo_golden_inputs
"""
import argparse
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = ["vlp16_polar", "os64_uniform"]


def export(case, outdir):
    z = np.load(os.path.join(ROOT, "tests", "golden", case + ".npz"))
    d = os.path.join(outdir, case); os.makedirs(d, exist_ok=True)
    np.ascontiguousarray(z["source"], "<f4").tofile(os.path.join(d, "source.f32"))
    np.ascontiguousarray(z["target"], "<f4").tofile(os.path.join(d, "target.f32"))
    for k in ("T_probe", "T_probe2", "T_probe6", "t_guess", "t_last", "t_probe"):
        np.ascontiguousarray(z[k], "<f8").tofile(os.path.join(d, k + ".f64"))
    with open(os.path.join(d, "params.txt"), "w") as f:
        f.write(f"voxel_type {int(z['voxel_type'])}\nleaf {float(z['leaf'])}\npolar_theta 0.175\npolar_phi 0.175\npolar_r 2.0\nct_lambda 0.3\n")
    print(case, "->", d, f"({z['source'].shape[0]} + {z['target'].shape[0]} points)")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("cases", nargs="*", default=CASES)
    ap.add_argument("--out", default="/tmp/rolo_golden_inputs")
    a = ap.parse_args()
    for c in a.cases:
        export(c, a.out)
