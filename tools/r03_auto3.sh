cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03
mkdir -p $O
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/auto3_tests.txt 2>&1; echo "pytest rc=$?" >> $O/auto3_tests.txt; tail -3 $O/auto3_tests.txt
timeout -k 10 300 python tools/body_probe.py 11 > $O/body_probe_c2.txt 2>&1; tail -5 $O/body_probe_c2.txt
for c in c2 c4 c5; do
timeout -k 10 300 python bench.py --config $c --no-cpu-baseline > $O/auto3_$c.json 2> $O/auto3_$c.err || tail -3 $O/auto3_$c.err
python - $O/auto3_$c.json $c <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], "value", round(d["value"],1), "step_frac", round(d["step_frac_of_mfma_peak"],3), [round(p["ms"],1) for p in d["per_step"]][:12])
PY
done
