#!/usr/bin/env python3
"""Generate the committed golden fixtures under tests/golden/ (run in the authoring container only).

Source of truth for the fixtures = oracle/twin.py, the independent numpy/scipy statement of the reference's
registration algorithm (the reference itself — C++ on Eigen/PCL/ROS — can be neither built nor imported here;
see DESIGN.md "Oracle"). Inputs come from rolo_amd.synth (seeded). Each .npz holds inputs and per-stage expected
outputs so that both the C++ oracle (CPU tests) and the HIP path (GPU tests) can be checked stage by stage.

    python tests/golden/make_golden.py
"""
import os
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from rolo_amd import synth  # noqa: E402
from oracle.twin import Twin  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))

CASES = {
    # name: (sensor, synth kwargs, twin kwargs)
    "vlp16_polar": ("vlp16", dict(col_stride=8), dict(voxel_type="polar", polar_res=(0.175, 0.175, 2.0))),
    "os64_uniform": ("os1-64", dict(col_stride=8, ring_stride=2), dict(voxel_type="uniform", leaf=1.0)),
    "vlp16_polar_fixed20": ("vlp16", dict(col_stride=8), dict(voxel_type="polar", polar_res=(0.175, 0.175, 2.0),
                                                              fixed_iterations=20)),
    "vlp16_polar_q2": ("vlp16", dict(col_stride=8), dict(voxel_type="polar", polar_res=(0.175, 0.175, 2.0),
                                                         q2_intended=True)),
}


def run_case(name, sensor, skw, tkw):
    src, tgt, (R, t) = synth.dense_pair(sensor, **skw)
    tw = Twin(src, tgt, **tkw)
    n = src.shape[0]
    sub = np.arange(0, n, 9)  # covariance samples
    # stage checks at a fixed, non-trivial pose
    Tprobe = np.eye(4)
    Tprobe[:3, :3] = synth.rpy_to_R(0.004, -0.007, 0.02)
    e_p, H_p, b_p = tw.so3_linearize(Tprobe)
    corr_src, corr_vox_keys = tw.c_src.astype(np.int32), tw.vox_keys[tw.c_vox]
    Tprobe2 = Tprobe.copy(); Tprobe2[:3, :3] = synth.rpy_to_R(0.0045, -0.0065, 0.021)
    e_p2 = tw.compute_error(Tprobe2)
    # translation-stage pieces on the correspondences cached by that so3_linearize
    g = -np.asarray(synth.PREV_STEP_T)  # Translation after the rotation stage = forward prediction (pure translation guess)
    l = g * 0.97
    tprobe = np.array([0.01, -0.004, 0.002])
    et, Ht, bt = tw.t3(tprobe, g, l, 0.1, 0.1, np.float32(0.3), False)
    et_err = tw.t3(tprobe, g, l, 0.1, 0.1, np.float32(0.3), True, want_H=False)
    Tp6 = Tprobe.copy(); Tp6[:3, 3] = (0.01, -0.02, 0.005)
    e6, H6, b6 = tw.linearize6(Tp6)
    # full solve
    tw.trace.clear()
    x0, it, conv, hist = tw.align()
    t_fin, t_it, thist = tw.compute_translation(np.zeros(3), g, l, 0.1, 0.1, np.float32(0.3))
    trace = np.array([[a for a in rec] for rec in tw.trace], float)
    np.savez_compressed(
        os.path.join(OUT, name + ".npz"),
        source=src, target=tgt, truth_R=R, truth_t=t,
        cov_sub=sub.astype(np.int32), src_cov_sub=tw.src_cov[sub], tgt_cov_sub=tw.tgt_cov[sub],
        src_knn_sub=tw.src_knn[sub], tgt_keys=tw.tgt_keys.astype(np.int16),
        vox_keys=tw.vox_keys.astype(np.int16), vox_count=tw.vox_count, vox_mean=tw.vox_mean,
        vox_cov_sub=tw.vox_cov[::5],
        T_probe=Tprobe, so3_err=e_p, so3_H=H_p, so3_b=b_p, corr_src=corr_src, corr_vox_keys=corr_vox_keys.astype(np.int16),
        T_probe2=Tprobe2, err_probe2=e_p2, T_probe6=Tp6, lin6_err=e6, lin6_H=H6, lin6_b=b6,
        align_T=x0, align_iters=it, align_converged=conv,
        align_y0=np.array([h[0] for h in hist]), align_H=np.array([h[1] for h in hist]), align_b=np.array([h[2] for h in hist]),
        t_guess=g, t_last=l, t_probe=tprobe, t3_err=et, t3_H=Ht, t3_b=bt, t3_err_variant=et_err,
        trans_final=t_fin, trans_iters=t_it,
        trans_y0=np.array([h[0] for h in thist]), trace=trace,
        fixed_iterations=tkw.get("fixed_iterations", 0), q2_intended=int(tkw.get("q2_intended", False)),
        voxel_type=0 if tkw.get("voxel_type") == "polar" else 1, leaf=tkw.get("leaf", 1.0),
    )
    print(name, "n=", n, "V=", tw.vox_keys.shape[0], "Nc(probe)=", len(corr_src), "iters", it, conv, "t_iters", t_it,
          "size KB", os.path.getsize(os.path.join(OUT, name + ".npz")) // 1024)


if __name__ == "__main__":
    for name, (sensor, skw, tkw) in CASES.items():
        run_case(name, sensor, skw, tkw)
