import csv, sys, glob, collections
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: [0, 0.0])
        for row in csv.DictReader(open(f)):
            if "zgemm" not in row.get("Kernel_Name", ""): continue
            k = row["Counter_Name"]; acc[k][0] += 1; acc[k][1] += float(row["Counter_Value"])
        for k, (n, v) in sorted(acc.items()):
            print(f"{k:32s} launches={n:3d} mean={v/n:.4g}")
