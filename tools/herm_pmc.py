"""HBM traffic of the Hermitian reduction's matvec kernel (csrc/herm.hip, herm_col2_kernel) from two rocprofv3 PMC passes
(FETCH_SIZE, WRITE_SIZE; units and the gfx950 correction as MI355X_MICROARCH.md prescribes: KB, FETCH doubled) against its
algorithmic bytes sum_i 8 (n - i - 1) (n - i) -- the lower triangle of the trailing matrix, read once per column (round 4;
the full square, 16 (n - i - 1)^2, in round 3) -- and its time from a kernel trace.

    python tools/herm_pmc.py <n> <fetch_dir> <write_dir> <trace_db>
"""
import csv
import glob
import sqlite3
import sys

n = int(sys.argv[1])
tot = {}
cnt = {}
for d, cname in ((sys.argv[2], "FETCH_SIZE"), (sys.argv[3], "WRITE_SIZE")):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != cname:
                continue
            k = "col2" if "herm_col2" in r["Kernel_Name"] else "col1" if "herm_col1" in r["Kernel_Name"] else "col3" if "herm_col3" in r["Kernel_Name"] else None
            if k:
                tot[(k, cname)] = tot.get((k, cname), 0.0) + float(r["Counter_Value"])
                cnt[(k, cname)] = cnt.get((k, cname), 0) + 1
db = sqlite3.connect(sys.argv[4])
ms = {}
for name, s, e in db.execute("select name, start, end from kernels"):
    for k in ("col1", "col2", "col3"):
        if "herm_" + k in name:
            ms[k] = ms.get(k, 0.0) + (e - s) * 1e-6
alg = sum(8.0 * (n - i - 1) * (n - i) for i in range(n - 1))
print(f"n = {n}: herm_col2_kernel (Hermitian matvec with the trailing matrix + panel dots), {cnt.get(('col2', 'FETCH_SIZE'), 0)} launches")
for k in ("col1", "col2", "col3"):
    f = tot.get((k, "FETCH_SIZE"), 0.0) * 1024.0 * 2.0
    w = tot.get((k, "WRITE_SIZE"), 0.0) * 1024.0
    t = ms.get(k, 0.0)
    line = f"  herm_{k}: fetched {f / 1e9:8.2f} GB (FETCH_SIZE x 2), written {w / 1e9:7.2f} GB, kernel time {t:8.1f} ms"
    if t > 0:
        line += f", {(f + w) / t / 1e9:6.2f} TB/s by the counters"
    if k == "col2":
        line += f"; algorithmic matrix bytes (lower triangle) {alg / 1e9:.2f} GB = {alg / t / 1e9 if t else 0:.2f} TB/s, traffic / algorithmic = {(f + w) / alg:.2f}"
    print(line)
