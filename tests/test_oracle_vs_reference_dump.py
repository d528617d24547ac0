"""CPU: the C++ oracle against dumps of the REAL reference (tools/dump_reference_golden.cpp, run in a ROLO workspace; recipe in
tools/README.md). No such dump exists in this repository — the reference cannot be built in this image — so the tests skip and say so;
dropping tests/golden/ref_<case>.npz next to the twins' fixtures activates them. This is the route from "parity unpinned" to pinned."""
import glob
import os

import numpy as np
import pytest

from oracle import pyorc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFS = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "ref_*.npz")))
CASES = [os.path.basename(p)[4:-4] for p in REFS]


def test_reference_dumps_present_or_say_why():
    if not REFS:
        pytest.skip("no tests/golden/ref_*.npz: the reference (Eigen + PCL + FLANN + ROS) cannot be built in this image; tools/README.md has the "
                    "three commands that produce the dumps in a ROLO workspace — parity stays 'unpinned' until then")


def _oracle(case):
    z = np.load(os.path.join(ROOT, "tests", "golden", case + ".npz"))
    kw = dict(polar_resolution=(0.175, 0.175, 2.0)) if int(z["voxel_type"]) == 0 else dict(voxel_type=pyorc.VOXEL_UNIFORM, voxel_resolution=float(z["leaf"]))
    o = pyorc.Reg(pyorc.default_params(num_threads=1, **kw))
    o.set_target(z["target"]); o.set_source(z["source"])
    return z, o


def rel(a, b):
    return np.abs(np.asarray(a) - np.asarray(b)).max() / max(np.abs(np.asarray(b)).max(), 1e-300)


@pytest.mark.parametrize("case", CASES or ["<none>"])
def test_oracle_matches_the_reference_stage_by_stage(case):
    if not REFS:
        pytest.skip("no reference dump (see test_reference_dumps_present_or_say_why)")
    r = np.load(os.path.join(ROOT, "tests", "golden", f"ref_{case}.npz"))
    z, o = _oracle(case)
    o.compute_covariances()
    # K5: FLANN's tie order among equidistant neighbours is arbitrary, the covariance is order-independent up to rounding
    assert rel(o.source_covariances(), r["src_cov"]) < 1e-9 and rel(o.target_covariances(), r["tgt_cov"]) < 1e-9
    # K6: keys exact; per-point voxel statistics
    assert np.array_equal(o.target_voxel_keys(), r["tgt_keys"])
    keys, counts, means, covs = o.voxels()
    lut = {tuple(k): i for i, k in enumerate(keys)}
    idx = np.array([lut[tuple(k)] for k in r["tgt_keys"]])
    assert np.array_equal(counts[idx], r["tgt_vox_count"]) and rel(means[idx], r["tgt_vox_mean"]) < 1e-12 and rel(covs[idx], r["tgt_vox_cov"]) < 1e-9
    # K7 + K8 / K10 / K9 / K11
    e, H, b = o.so3_linearize(z["T_probe"])
    assert rel(e, r["so3_err"]) < 1e-9 and rel(H, r["so3_H"]) < 1e-9 and rel(b, r["so3_b"]) < 1e-9
    found, ckeys = o.correspondences()
    src_idx = np.nonzero(found[:, 0])[0]
    assert np.array_equal(src_idx, np.sort(r["corr_src"])) and np.array_equal(ckeys[src_idx, 0], r["corr_vox_keys"][np.argsort(r["corr_src"], kind="stable")])
    assert rel(o.compute_error(z["T_probe2"]), r["err_probe2"]) < 1e-9
    et, Ht, bt = o.t3_linearize(z["t_probe"], z["t_guess"], z["t_last"], 0.1, 0.1, 0.3)
    assert rel(et, r["t3_err"]) < 1e-9 and rel(Ht, r["t3_H"]) < 1e-9 and rel(bt, r["t3_b"]) < 1e-9
    # compute_t_error's out-of-bounds store — "initial value retained" is the restated default; a toolchain whose stray store lands
    # differently shows up HERE and only here
    ev = o.compute_t_error(z["t_probe"], z["t_guess"], z["t_last"], 0.1, 0.1, 0.3)
    assert rel(ev, r["t3_err_variant"]) < 1e-9, "SURVEY Q2: this toolchain's compute_t_error differs from the restated 'initial value retained' variant"
    e6, H6, b6 = o.linearize(z["T_probe6"])
    assert rel(e6, r["lin6_err"]) < 1e-9 and rel(H6, r["lin6_H"]) < 1e-9 and rel(b6, r["lin6_b"]) < 1e-9
    # K12: the two-stage solve
    z2, o2 = _oracle(case)
    rc, Tf, Td, it, cv = o2.align()
    assert rc == 0 and it == int(r["align_iters"][0]) and int(cv) == int(r["align_iters"][1])
    dR = Tf[:3, :3].astype(np.float64) @ r["align_T_f"][:3, :3].astype(np.float64).T
    assert np.arccos(np.clip((np.trace(dR) - 1) / 2, -1, 1)) <= 1e-5
    rc2, t, _ = o2.compute_translation(np.zeros(3), z["t_guess"], z["t_last"])
    assert rc2 == 0 and np.abs(t - r["trans_final"]).max() <= 1e-4


def test_dump_tool_io_half_round_trips(tmp_path):
    """tools/dump_reference_golden.cpp cannot be built here (it includes the reference's own headers, which need Eigen + PCL), but the half of it that needs
    nothing of the reference can: the raw-array reader, the parameter file and the .npy writer (-DDUMP_IO_SELFTEST). The inputs come from the recipe's own
    exporter (tools/export_golden_inputs.py), numpy reads back what the tool wrote."""
    import subprocess
    import sys
    exe = str(tmp_path / "dump_io_selftest")
    r = subprocess.run(["g++", "-std=c++14", "-O1", "-Wall", "-Werror", "-DDUMP_IO_SELFTEST", os.path.join(ROOT, "tools", "dump_reference_golden.cpp"), "-o", exe],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    case = "vlp16_polar"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "export_golden_inputs.py"), case, "--out", str(tmp_path / "inputs")], capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    ind = tmp_path / "inputs" / case
    r = subprocess.run([exe, str(ind)], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.startswith("ok"), r.stdout + r.stderr
    z = np.load(os.path.join(ROOT, "tests", "golden", case + ".npz"))
    assert np.array_equal(np.load(ind / "echo_source.npy"), np.asarray(z["source"], np.float32).reshape(-1, 4))
    T = np.load(ind / "echo_T.npy")
    assert T.shape == (4, 4) and T.dtype == np.float64 and np.array_equal(T, np.fromfile(ind / "T_probe.f64", np.float64).reshape(4, 4))
    assert np.load(ind / "echo_ints.npy").tolist()[1:] == [7, -3] and np.load(ind / "echo_scalar.npy").shape == (1,)


def test_dump_tool_reference_half_type_checks():
    """TYPE CHECK ONLY — it pins nothing (round 4's verdict, item 8). The reference-facing half of tools/dump_reference_golden.cpp (the subclass that opens the
    reference's protected stage functions, the dump of every per-stage field) compiled with -fsyntax-only against tests/cpp/mock_ref: declaration-only
    transcriptions of the members of fast_gicp::RotVGICP / VmfVoxelMap / Eigen / PCL the tool uses (tests/cpp/mock_ref/README.md). No body, no link, no run — so the
    tool cannot rot unnoticed (this check found two unused locals the first time it ran); the authority remains the real build in a ROLO workspace, tools/README.md."""
    import subprocess
    r = subprocess.run(["g++", "-std=c++14", "-fsyntax-only", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "tests", "cpp", "mock_ref"),
                        os.path.join(ROOT, "tools", "dump_reference_golden.cpp")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-4000:]
    # the mocks are declarations only: nothing in them may define a function body that computes (a stand-in BUILD of the reference is not allowed)
    for dirpath, _, files in os.walk(os.path.join(ROOT, "tests", "cpp", "mock_ref")):
        for f in files:
            if f.endswith(".md"):
                continue
            txt = open(os.path.join(dirpath, f)).read()
            assert "return " not in txt and "DECLARATION-ONLY" in txt, os.path.join(dirpath, f)
