"""HBM traffic per kernel from rocprofv3 PMC passes (one counter per pass: FETCH_SIZE, WRITE_SIZE).

    python tools/pmc_traffic.py <lu_trace.txt> <pass_dir> [<pass_dir> ...]

Units and corrections as MI355X_MICROARCH.md prescribes: the counters are in KB, and gfx950 counts a
128-B read request as 64 B, so FETCH_SIZE is doubled; WRITE_SIZE is taken as reported.

The zgemm launches of the LU are attributed to their K class by ORDER: the library (run with
MAUS_LU_TRACE=<lu_trace.txt>) logs "M N K batch" for every trailing-update launch, and the dispatches of
the LU zgemm instantiations appear in the counter file in the same order.  Output: a per-kernel table on
stdout and gpurun_out/zgemm_pmc_traffic.json for the K>=256 launches (the roofline kernel of bench.py).
"""
import collections
import re
import csv
import glob
import json
import os
import sys

LU_ZGEMM = re.compile(r"zgemm_kernel<[^>]*, true>|zgemm3m_dma_kernel<\d+, \d+, \d+, \d+, true,")       # TILED = true: the LU's instantiations
trace_path, dirs = sys.argv[1], sys.argv[2:]
trace = [tuple(int(x) for x in line.split()) for line in open(trace_path) if line.strip()]


def klass(name):
    if "lu_diaginv" in name:            # round 4: the inverted diagonal blocks belong to the triangular solves
        return "trsm"
    for key in ("zgemm", "lu_panel", "laswp", "trsm", "build_h", "backsolve", "mt_jump", "init_perm"):
        if key in name:
            return key
    return "other"


acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
big = collections.defaultdict(lambda: [0, 0.0, 0.0])          # counter -> [launches, KB, algorithmic bytes]
for d in dirs:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        rows = list(csv.DictReader(open(f)))
        rows.sort(key=lambda r: int(r["Dispatch_Id"]))
        # LU trailing-update launches = the zgemm instantiations on the tile-major workspace (last template argument
        # TILED = true); the population products are launched outside the LU and are not in the trace
        lu_rows = [r for r in rows if LU_ZGEMM.search(r["Kernel_Name"])]
        per_counter = collections.defaultdict(list)
        for r in lu_rows:
            per_counter[r["Counter_Name"]].append(r)
        for cname, rs in per_counter.items():
            if len(rs) != len(trace):
                print(f"# {f}: {cname}: {len(rs)} LU zgemm dispatches vs {len(trace)} trace lines -- K attribution skipped")
                continue
            for r, (M, N, K, G) in zip(rs, trace):
                if K >= 256:
                    b = big[cname]
                    b[0] += 1
                    b[1] += float(r["Counter_Value"])
                    b[2] += 16.0 * (M * K + K * N + 2.0 * M * N) * G
        for r in rows:
            a = acc[klass(r["Kernel_Name"])][r["Counter_Name"]]
            a[0] += 1
            a[1] += float(r["Counter_Value"])

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
per_class = {}
for k, cs in acc.items():
    if "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
        per_class[k] = {"launches": cs["FETCH_SIZE"][0], "fetch_bytes": cs["FETCH_SIZE"][1] * 1024.0 * 2.0,
                        "write_bytes": cs["WRITE_SIZE"][1] * 1024.0}
        per_class[k]["hbm_bytes"] = per_class[k]["fetch_bytes"] + per_class[k]["write_bytes"]
matrices = max((t[3] for t in trace), default=0)
json.dump({"unit": f"bytes per bench step (one sweep of {matrices} solves, MAUS_LU_STREAMS=1)", "matrices": matrices, "classes": per_class,
           "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950), WRITE_SIZE as reported; separate --pmc passes"},
          open(os.path.join(root, "gpurun_out", "pmc_traffic_per_kernel.json"), "w"), indent=1)
for k, cs in acc.items():
    parts = []
    for c, (n, v) in sorted(cs.items()):
        b = v * 1024.0 * (2.0 if c == "FETCH_SIZE" else 1.0)
        parts.append(f"{c}: launches={n} total={b / 1e9:.1f} GB (corrected) per-launch={b / n / 1e6:.1f} MB")
    print(k, "|", " | ".join(parts))

if "FETCH_SIZE" in big and "WRITE_SIZE" in big:
    f, w = big["FETCH_SIZE"], big["WRITE_SIZE"]
    out = {"kernel": "zgemm3m_dma_kernel (64x64 / 64x32 tiles, LDS-DMA staged 3M), K>=256 launches of the LU trailing updates", "matrices": matrices,
           "command": "MAUS_LU_STREAMS=1 MAUS_LU_TRACE=<file> rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate passes) --kernel-trace "
                      "-- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-isolated --no-small-batch --skip-diagnosis --kernel-events off",
           "launches": f[0], "fetch_bytes_per_launch": f[1] * 1024.0 * 2.0 / f[0], "write_bytes_per_launch": w[1] * 1024.0 / w[0],
           "algorithmic_bytes_per_launch": f[2] / f[0],
           "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts 128-B requests as 64 B); WRITE_SIZE as reported; KB units"}
    out["hbm_bytes_per_launch"] = out["fetch_bytes_per_launch"] + out["write_bytes_per_launch"]
    print("K>=256 zgemm launches:", json.dumps(out, indent=1))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    json.dump(out, open(os.path.join(root, "gpurun_out", "zgemm_pmc_traffic.json"), "w"), indent=1)
