#!/usr/bin/env python3
"""the .npy files tools/dump_reference_golden.cpp wrote for one case -> tests/golden/ref_<case>.npz (what
tests/test_oracle_vs_reference_dump.py looks for).

    python tools/pack_reference_dump.py <case> <outputs_dir>
"""
import glob
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

if __name__ == "__main__":
    case, d = sys.argv[1], sys.argv[2]
    # the third-party versions the reference was built with decide last bits on this path (tools/README.md, "Versions the dump must name")
    vf = os.path.join(d, "versions.txt")
    need = ("eigen", "pcl", "flann", "compiler")
    if not os.path.exists(vf):
        raise SystemExit(f"{vf} is missing: write one line per library — " + ", ".join(k + "=<version>" for k in need) + " (tools/README.md)")
    versions = dict(l.strip().split("=", 1) for l in open(vf) if "=" in l)
    missing = [k for k in need if k not in versions]
    if missing:
        raise SystemExit(f"{vf} does not name: {', '.join(missing)}")
    arrs = {os.path.splitext(os.path.basename(p))[0]: np.load(p) for p in glob.glob(os.path.join(d, "*.npy"))}
    arrs["align_T_f"] = arrs["align_T_f"].T.copy()   # Eigen stores column-major
    arrs["versions"] = np.array(sorted(f"{k}={v}" for k, v in versions.items()))
    out = os.path.join(ROOT, "tests", "golden", f"ref_{case}.npz")
    np.savez_compressed(out, **arrs)
    print(out, sorted(arrs), os.path.getsize(out), "bytes")
