cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03
mkdir -p $O
timeout -k 10 300 python tools/body_probe.py 11 > $O/body_probe_c2.txt 2>&1; cat $O/body_probe_c2.txt | tail -12
for r in 1 2; do
timeout -k 10 300 python bench.py --config c2 --no-cpu-baseline > $O/auto2_c2_$r.json 2> $O/auto2_c2_$r.err || tail -3 $O/auto2_c2_$r.err
python - $O/auto2_c2_$r.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("c2 value", round(d["value"],1), "step_frac", round(d["step_frac_of_mfma_peak"],3), [round(p["ms"],1) for p in d["per_step"]])
PY
done
