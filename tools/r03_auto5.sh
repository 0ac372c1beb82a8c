cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03; mkdir -p $O
for c in c3 c3 c2 c4 c5; do
timeout -k 10 300 python bench.py --config $c --no-cpu-baseline > $O/cap_$c.json 2> $O/cap_$c.err || tail -3 $O/cap_$c.err
python - $O/cap_$c.json $c <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], "value", round(d["value"],1), "step_frac", round(d["step_frac_of_mfma_peak"],3), [round(p["ms"],1) for p in d["per_step"]][:12], "roof", round(d["roofline"]["achieved"],1))
PY
done
