# round 3, final tree: whole GPU suite, smoke, the driver-shaped bench and the side configurations
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03
mkdir -p $O
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > $O/final8_tests.txt 2>&1
echo "pytest rc=$?" >> $O/final8_tests.txt
tail -4 $O/final8_tests.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout -k 10 500 python bench.py > $O/final8_c1.json 2> $O/final8_c1.err && python tools/bench_summary.py $O/final8_c1.json
for c in c2 c3 c4 c5; do
  timeout -k 10 400 python bench.py --config $c > $O/final8_$c.json 2> $O/final8_$c.err || tail -5 $O/final8_$c.err
  python - $O/final8_$c.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r=d["roofline"]
    print(d["metric"][-5:], "value", round(d["value"],1), "step_frac", round(d["step_frac_of_mfma_peak"],3), "roof", r["kernel"][:40], round(r["achieved"],2), r["unit"], "frac", round(r["frac"],3), "cpu", d.get("cpu_baseline",{}).get("value"), "build_s", d["config"]["solver_build_s"])
except Exception as e:
    print(sys.argv[1], "unreadable", e)
PY
done
