"""Average duration / rate of the K>=256 zgemm launches from a rocprofv3 kernel trace.

    python tools/k256_durations.py <lu_trace.txt> <dir with *kernel_trace.csv>

The rocprofv3 --stats summary groups launches by kernel NAME, and the LU's zgemm instantiation serves
every K from 64 to 512.  The library's MAUS_LU_TRACE log ("M N K batch" per trailing-update launch, in
launch order) lets the per-dispatch trace be split by K, so that bench.py's roofline (K>=256 launches)
can be checked against the profiler's own timestamps.
"""
import csv
import glob
import re
import sys

LU_ZGEMM = re.compile(r"zgemm_kernel<[^>]*, true>|zgemm3m_dma_kernel<\d+, \d+, \d+, \d+, true,")       # TILED = true: the LU's instantiations
trace = [tuple(int(x) for x in line.split()) for line in open(sys.argv[1]) if line.strip()]
for f in glob.glob(sys.argv[2] + "/**/*kernel_trace.csv", recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if LU_ZGEMM.search(r["Kernel_Name"])]      # TILED = true: the LU's instantiations
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    if len(rows) != len(trace):
        print(f"{f}: {len(rows)} LU zgemm dispatches vs {len(trace)} trace lines")
        continue
    acc = {}
    ivs = {}
    for r, (M, N, K, G) in zip(rows, trace):
        key = "K>=256" if K >= 256 else f"K={K}"
        a = acc.setdefault(key, [0, 0.0, 0.0])
        a[0] += 1
        a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6
        a[2] += 8.0 * M * N * K * G
        ivs.setdefault(key, []).append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    print(f)
    for key, (n, ms, fl) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
        # union of the launches' intervals: the time during which at least one launch of the class was executing
        # (what bench.py's roofline.achieved divides by; equal to the total when nothing overlaps)
        un, cs, ce = 0.0, None, None
        for s0, e0 in sorted(ivs[key]):
            if ce is None:
                cs, ce = s0, e0
            elif s0 <= ce:
                ce = max(ce, e0)
            else:
                un += ce - cs
                cs, ce = s0, e0
        un = (un + (ce - cs)) * 1e-6
        print(f"  {key:8s} launches={n:4d} total={ms:9.2f} ms avg={ms / n:8.3f} ms  {fl / ms * 1e-9:6.2f} TFLOP/s (8MNK) | "
              f"union={un:9.2f} ms  {fl / un * 1e-9:6.2f} TFLOP/s")
