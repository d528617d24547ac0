"""Search build / walk / tail per launch (HIP events) over the bench's POOL of distinct frame pairs and at the pipeline's cloud size — the figure VERDICT r03 item 2
is measured on. ROLO_KNN_BUDGET=<leaves> (0: the plain walk) selects the walk; one process per setting:

    for b in 0 20 28 36; do ROLO_KNN_BUDGET=$b python profiles/tools/knnbench_pool.py; done
"""
import os, sys, json; sys.path.insert(0, '.')   # run from the repository root
import numpy as np
from concurrent.futures import ProcessPoolExecutor
from rolo_amd import synth, profile


def pair(a):
    sensor, i, stride = a
    s, t, _ = synth.dense_pair(sensor, seed=synth.SEED + 2 * i, origin=synth.pool_origin(i), col_stride=stride)
    return s, t


if __name__ == "__main__":
    import torch
    from rolo_amd.rotvgicp import RotVGICP
    npool = int(os.environ.get("POOL", "8"))
    cases = [("os1-128", i, 1) for i in range(npool)] + [("os1-128", i, 3) for i in range(min(npool, 4))] + [("os1-64", 0, 1)]
    with ProcessPoolExecutor(max_workers=min(len(cases), os.cpu_count() or 4)) as ex:
        pairs = list(ex.map(pair, cases))
    out = {"budget": os.environ.get("ROLO_KNN_BUDGET", "default"), "rows": []}
    for (sensor, i, stride), (src, tgt) in zip(cases, pairs):
        g = RotVGICP(); g.setResolution(0.5)
        ds = torch.from_numpy(src).cuda(); dt = torch.from_numpy(tgt).cuda()

        def step():
            g.setInputTargetDevice(dt.data_ptr(), tgt.shape[0], 4); g.setInputSourceDevice(ds.data_ptr(), src.shape[0], 4)
            g.computeCovariances()
        for _ in range(3):
            step()
        acc = profile.kernel_times(g, step, reps=6)
        row = {"sensor": sensor, "pool": i, "stride": stride, "n": int(src.shape[0])}
        for k in ("knn_build", "knn_walk", "knn_tail"):
            row[k] = float(np.mean([r.sum() for r in acc[k]]))
        out["rows"].append(row)
        g.close()
    full = [r["knn_walk"] for r in out["rows"] if r["stride"] == 1 and r["sensor"] == "os1-128"]
    third = [r["knn_walk"] for r in out["rows"] if r["stride"] == 3]
    out["walk_ms_pool_mean_2x131072"] = float(np.mean(full)); out["walk_ms_nominal_2x131072"] = full[0]
    out["walk_ms_mean_2x43776"] = float(np.mean(third))
    out["walk_ms_2x65536"] = out["rows"][-1]["knn_walk"]
    print(json.dumps(out))
