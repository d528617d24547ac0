import csv, sys, glob
for d in sys.argv[1:]:
    f = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)
    if not f: print(d, "no stats"); continue
    for r in csv.DictReader(open(f[0])):
        if any(k in r["Name"] for k in ("knn_tail", "knn_walk_kernel", "lm_persist", "sort_scatter", "morton", "voxel_finalize", "leaf_kernel")):
            print(d.split("/")[-1], r["Name"].split("(")[0][-50:], r["Calls"], "avg us %.1f" % (float(r["AverageNs"]) / 1e3))
