cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03
mkdir -p $O
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_herm -o t -- python3 tools/herm_eigh_time.py 8192 > $O/trace_herm.log 2>&1
tail -2 $O/trace_herm.log | cut -c1-300
f=$(find $O/trace_herm -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv,sys,re
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    nm=re.sub(r"\(anonymous namespace\)::","",r["Name"]); nm=re.match(r"(?:void )?([A-Za-z_0-9]+(<[^(]*>)?)",nm).group(1)
    print(f'{nm[:90]:90s} calls={r["Calls"]:>6s} total_ms={float(r["TotalDurationNs"])*1e-6:9.1f} avg_us={float(r["AverageNs"])*1e-3:9.1f}')
PY
find $O/trace_herm -name "*.csv" -size +5M -delete
