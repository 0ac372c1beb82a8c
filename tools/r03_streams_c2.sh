# c2 / c3 with sub-batch streams: at n = 1024 the panel and the small levels are latency / HBM bound and the big zgemm is not
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03
mkdir -p $O
for c in c2; do
 for s in 1 2 3 4; do
  MAUS_LU_STREAMS=$s timeout -k 10 300 python bench.py --config $c --no-cpu-baseline > $O/st_${c}_$s.json 2> $O/st_${c}_$s.err || tail -3 $O/st_${c}_$s.err
  python - $O/st_${c}_$s.json $c $s <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], "streams", sys.argv[3], "value", round(d["value"],1), "step_frac", round(d["step_frac_of_mfma_peak"],3), [round(p["ms"],1) for p in d["per_step"]])
PY
 done
done
for s in 1 2; do
  MAUS_LU_STREAMS=$s timeout -k 10 300 python bench.py --config c3 --no-cpu-baseline > $O/st_c3_$s.json 2> $O/st_c3_$s.err || tail -3 $O/st_c3_$s.err
  python - $O/st_c3_$s.json c3 $s <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], "streams", sys.argv[3], "value", round(d["value"],1), "step_frac", round(d["step_frac_of_mfma_peak"],3), [round(p["ms"],1) for p in d["per_step"]][:8])
PY
done
