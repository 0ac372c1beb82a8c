# Round-3 profiles of the side paths (final tree): GMRES loop bodies (configs[2]) and the Hermitian decomposition (configs[3])
set -e
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_r03b
rm -rf $O && mkdir -p $O
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c3 -- python3 bench.py --config c3 --steps 6 --no-cpu-baseline > $O/c3.json 2> $O/c3.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c5 -- python3 bench.py --config c5 --steps 6 --no-cpu-baseline > $O/c5.json 2> $O/c5.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/herm -- python3 tools/herm_eigh_time.py 8192 > $O/herm.txt 2> $O/herm.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/herm_fetch -- python3 tools/herm_eigh_time.py 4096 > $O/herm_fetch.txt 2> $O/herm_fetch.err
python3 - $O <<'PY'
import csv, glob, re, sys
O = sys.argv[1]
def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    m = re.match(r"(?:void )?([A-Za-z_0-9]+)(<[^(]*>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else name[:60]
for tag in ("c3", "c5", "herm"):
    for f in glob.glob(f"{O}/{tag}/**/*kernel_stats.csv", recursive=True):
        rows = list(csv.DictReader(open(f)))
        tot = sum(float(r["TotalDurationNs"]) for r in rows)
        print(f"## {tag}: kernel time {tot * 1e-6:.1f} ms")
        for r in rows[:8]:
            print(f"   {short(r['Name'])[:100]:100s} calls={r['Calls']:>6s} total_ms={float(r['TotalDurationNs']) * 1e-6:9.2f} avg_us={float(r['AverageNs']) * 1e-3:9.1f} {float(r['Percentage']):5.1f} %")
acc = {}
for f in glob.glob(f"{O}/herm_fetch/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        a = acc.setdefault(k, [0, 0.0, 0.0])
        a[0] += 1; a[1] += float(r["Counter_Value"]); a[2] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
print("## herm n=4096, FETCH_SIZE (KB, doubled for gfx950) per kernel")
for k, (n, kb, ns) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:6]:
    print(f"   {k[:80]:80s} launches={n:6d} fetched={2 * kb * 1024 / 1e9:8.2f} GB in {ns * 1e-6:8.1f} ms = {2 * kb * 1024 / max(ns, 1):6.2f} GB/s x1e0".replace(" GB/s x1e0", " GB/ms (= TB/s)"))
PY
find $O -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
