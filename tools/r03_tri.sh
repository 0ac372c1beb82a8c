cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_herm_eigh.py tests/test_gpu_cond_estimate.py tests/test_gpu_dist_nccl.py -x -q -m gpu > $O/tri_tests.txt 2>&1; echo "pytest rc=$?" >> $O/tri_tests.txt; tail -6 $O/tri_tests.txt
timeout -k 10 300 python bench.py --config c4 --no-cpu-baseline > $O/tri_c4.json 2> $O/tri_c4.err || tail -3 $O/tri_c4.err
python - $O/tri_c4.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("c4 value", round(d["value"],1), [round(p["ms"],1) for p in d["per_step"]], "build_s", d["config"]["solver_build_s"], d["config"].get("condition_number"))
PY
timeout -k 10 300 python tools/c4_run.py 8192 128 2>&1 | tail -8 | cut -c1-250
