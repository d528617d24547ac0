import csv, sys, glob
for d in sys.argv[1:]:
    f = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)
    if not f: print(d, "no stats"); continue
    for r in csv.DictReader(open(f[0])):
        if "knn_tail" in r["Name"] or "knn_walk_kernel" in r["Name"] or "lm_persist" in r["Name"]:
            print(d.split("/")[-1], r["Name"].split("(")[0][-50:], r["Calls"], "avg us %.1f" % (float(r["AverageNs"]) / 1e3))
