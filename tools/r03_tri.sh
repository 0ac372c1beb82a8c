cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_herm_eigh.py tests/test_gpu_cond_estimate.py tests/test_gpu_evolve.py -x -q -m gpu > $O/tri_tests.txt 2>&1; echo "pytest rc=$?" >> $O/tri_tests.txt; tail -6 $O/tri_tests.txt
for c in c3 c5; do
timeout -k 10 300 python bench.py --config $c --no-cpu-baseline > $O/tri_$c.json 2> $O/tri_$c.err || tail -3 $O/tri_$c.err
python - $O/tri_$c.json $c <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], "value", round(d["value"],1), "build_s", d["config"]["solver_build_s"], d["config"].get("condition_number"))
PY
done
