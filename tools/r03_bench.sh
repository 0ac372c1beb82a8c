cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03
mkdir -p $O
timeout -k 10 400 python bench.py > $O/bench_c1.json 2> $O/bench_c1.err && python tools/bench_summary.py $O/bench_c1.json
for c in c2 c3 c5; do
  timeout -k 10 300 python bench.py --config $c > $O/bench_$c.json 2> $O/bench_$c.err || tail -5 $O/bench_$c.err
  python - $O/bench_$c.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r=d["roofline"]
    print(d["metric"], "value", round(d["value"],1), "step_tflops", round(d["step_tflops"],2), "roof", r["kernel"][:50], round(r["achieved"],2), r["unit"], "frac", round(r["frac"],3), "kernel_ms", d.get("kernel_ms_profiled_pass"), d.get("gmres"))
except Exception as e:
    print(sys.argv[1], "unreadable", e)
PY
done
rocprofv3 -L > $O/counters.txt 2>&1 || true
grep -o -E "\b(SQ_[A-Z0-9_]*(MFMA|LDS|BUSY|WAVE_CYCLES|WAIT|ACTIVE_INST)[A-Z0-9_]*|GRBM_[A-Z_]+|TCC_EA0?_[A-Z_]+)\b" $O/counters.txt | sort -u | tr '\n' ' ' > $O/counters_short.txt; wc -c $O/counters_short.txt
