import csv, glob, sys, collections
for d in sys.argv[1:]:
    agg = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "conv_pipe" not in r["Kernel_Name"] and "conv_gather" not in r["Kernel_Name"]:
                continue
            a = agg[r["Counter_Name"]]
            a[0] += float(r["Counter_Value"]); a[1] += 1
    print(d, {k: round(v[0] / max(v[1], 1)) for k, v in sorted(agg.items())})
