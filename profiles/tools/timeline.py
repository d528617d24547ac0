import sys, os, time, ctypes as C; sys.path.insert(0, '.')   # run from the repository root
os.environ["ROLO_STAMP"] = "1"
import numpy as np, torch
from rolo_amd import synth, _lib
from rolo_amd.rotvgicp import RotVGICP
L = _lib.lib()
fs = L.rolo_debug_stamps; fs.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]; fs.restype = C.c_int
src, tgt, _ = synth.dense_pair("os1-128")
ds = torch.from_numpy(src).cuda(); dt = torch.from_numpy(tgt).cuda(); n = src.shape[0]
zero3 = np.zeros(3); guess = -np.asarray(synth.PREV_STEP_T, np.float64); last = guess * 0.97
def handle(g):
    for name in ("_ctx", "ctx", "_h", "handle"):
        if hasattr(g, name): return getattr(g, name)
    raise RuntimeError(dir(g))
for nctx in (1, 4):
    gs = []
    for _ in range(nctx):
        g = RotVGICP(); g.setResolution(0.5); g.setFixedIterations(20); g.setOverlapKnn(True); g.setUseGraph(True); gs.append(g)
    def enq(g):
        g.setInputTargetDevice(dt.data_ptr(), n, 4); g.setInputSourceDevice(ds.data_ptr(), n, 4)
        g.register_async(None, zero3, guess, last, 0.1, 0.1, 0.3)
    for _ in range(6):
        for g in gs: enq(g)
        for g in gs: g.register_wait()
    K = 30; recs = []
    t0 = time.perf_counter()
    for g in gs: enq(g)
    for it in range(K):
        for ci, g in enumerate(gs):
            g.register_wait()
            out = (C.c_ulonglong * 8)(); fs(handle(g), out); recs.append((ci, it, list(out)[:5]))
            if it + 1 < K: enq(g)
    t1 = time.perf_counter()
    print("contexts %d: %.0f frames/s (%.3f ms per frame)" % (nctx, K * nctx / (t1 - t0), 1e3 * (t1 - t0) / (K * nctx)))
    A = np.array([r[2] for r in recs], dtype=np.int64)
    resident = bool((A[:, 3] < A[:, 2]).any())   # the resident LM kernel (fused_lm = 2) has no stamp between its stages: slot 3 is never written
    if resident:
        A[:, 3] = A[:, 2]
    d = np.diff(A, axis=1) / 100.0
    names = ["search (build+walk+tail)", "voxel finalize + begin", "rotation LM", "translation LM"] if not resident else ["search (build+walk+tail)", "voxel finalize + begin", "(resident kernel: no stamp)", "LM, both stages (resident)"]
    for k in range(4): print("  %-28s mean %8.1f us  p50 %8.1f  max %8.1f" % (names[k], d[5:, k].mean(), np.median(d[5:, k]), d[5:, k].max()))
    if nctx > 1:   # how long a context's stream waits for its next frame (end of frame k -> first stamp of frame k + 1)
        gaps = []
        for ci in range(nctx):
            rows = [st for c_, it, st in recs if c_ == ci]
            gaps += [(rows[k + 1][0] - rows[k][4]) / 100.0 for k in range(5, len(rows) - 1)]
        print("  end of a frame -> start of the context's next: mean %.1f us  p50 %.1f  max %.1f" % (np.mean(gaps), np.median(gaps), np.max(gaps)))
    tot = (A[:, 4] - A[:, 0]) / 100.0
    print("  frame start->end             mean %8.1f us" % tot[5:].mean())
    if nctx > 1:
        base = A[8 * nctx:, 0].min()
        for ci, it, st in recs[8 * nctx: 8 * nctx + 3 * nctx]:
            print("   ctx %d frame %2d: " % (ci, it) + "  ".join("%8.1f" % ((x - base) / 100.0) for x in (st[:3] + st[4:] if resident else st)))
    for g in gs: g.close()
