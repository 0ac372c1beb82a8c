cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03; mkdir -p $O
timeout -k 10 800 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_gmres.py tests/test_gpu_step_parity.py -x -q -m gpu > $O/tile_tests.txt 2>&1; echo "pytest rc=$?" >> $O/tile_tests.txt; tail -4 $O/tile_tests.txt
for c in c4 c3 c5; do
timeout -k 10 300 python bench.py --config $c --no-cpu-baseline > $O/tile_$c.json 2> $O/tile_$c.err || tail -3 $O/tile_$c.err
python - $O/tile_$c.json $c <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], "value", round(d["value"],1), [round(p["ms"],1) for p in d["per_step"]][:6])
PY
done
