# round 4: selected tests + bench
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04
T=${1:-quick}
shift
mkdir -p $O
timeout -k 10 900 python -m pytest "$@" -x -q -m gpu --durations=6 > $O/${T}_tests.txt 2>&1
echo "pytest rc=$?" >> $O/${T}_tests.txt
tail -14 $O/${T}_tests.txt
grep -q "rc=0" $O/${T}_tests.txt || { echo TESTS FAILED; exit 1; }
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
( time timeout -k 10 600 python bench.py > $O/${T}_c1.json 2> $O/${T}_c1.err ) 2>&1 | grep real
python tools/bench_summary.py $O/${T}_c1.json
python - $O/${T}_c1.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("small", d.get("small_batch_rates"), "step_exec_frac", round(d["step_frac_of_mfma_peak_executed"],3), "roof", round(d["roofline"]["frac"],3))
for k,v in (d.get("side_configs") or {}).items():
    if isinstance(v, dict):
        print(k, {kk: (round(vv,2) if isinstance(vv,float) else vv) for kk,vv in v.items() if kk in ("value","ms_per_step","steps","error","wall_s_of_this_side_run","step_frac_of_mfma_peak","solver_build_s","stderr_tail")}, (v.get("roofline") or {}).get("frac"), (v.get("cpu_baseline") or {}).get("value"))
    else:
        print(k, v)
PY
