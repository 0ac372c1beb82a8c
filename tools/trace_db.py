"""Per-dispatch durations of a rocprofv3 kernel trace (the rocpd SQLite database rocprofv3 writes by default) grouped by
(kernel, grid, workgroup) for the LAST maus_shifted_lu_solve call of the traced run: which launches of a kernel are short and
which are long.

    rocprofv3 --kernel-trace -d gpurun_out/x -o t -- python3 tools/lu_batch_rates.py 181
    python tools/trace_db.py gpurun_out/x/t_results.db [name-filter ...]
"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    m = re.match(r"(?:void )?([A-Za-z_0-9]+)(<[^(]*>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else name[:60]


db = sqlite3.connect(sys.argv[1])
filters = sys.argv[2:]
rows = db.execute("select name, start, end, grid_x, grid_y, workgroup_x from kernels order by start").fetchall()
builds = [i for i, r in enumerate(rows) if "build_h" in r[0]]
j = builds[-1]
while j > 0 and "mt_" in rows[j - 1][0]:
    j -= 1
seg = rows[j:]
acc = {}
busy = 0.0
for n, s, e, gx, gy, wx in seg:
    k = (short(n), gx // max(1, wx), gy, wx)
    a = acc.setdefault(k, [0, 0.0])
    a[0] += 1
    a[1] += (e - s) * 1e-6
    busy += (e - s) * 1e-6
span = (seg[-1][2] - seg[0][1]) * 1e-6
print(f"last call: {len(seg)} dispatches, span {span:.2f} ms, sum of kernel durations {busy:.2f} ms, idle between kernels {span - busy:.2f} ms")
byk = {}
for (nm, gx, gy, wx), (n, ms) in acc.items():
    byk.setdefault(nm, []).append((gx, gy, wx, n, ms))
for nm, lst in sorted(byk.items(), key=lambda kv: -sum(x[4] for x in kv[1])):
    print(f"{nm}: {sum(x[3] for x in lst)} launches, {sum(x[4] for x in lst):.2f} ms")
    if not filters or any(f in nm for f in filters):
        for gx, gy, wx, n, ms in sorted(lst, key=lambda x: -x[4])[:10]:
            print(f"      grid {gx:5d} x {gy:4d} wg {wx:4d}: n={n:4d} total={ms:8.2f} ms avg={ms / n * 1e3:8.1f} us")
