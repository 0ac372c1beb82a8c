cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_herm_eigh.py tests/test_gpu_dist_nccl.py tests/test_gpu_full_size.py -x -q > $O/herm_tests.txt 2>&1; tail -3 $O/herm_tests.txt
timeout -k 10 600 python tools/c4_run.py 8192 128 > $O/c4_device_eigh.txt 2>&1; cat $O/c4_device_eigh.txt | cut -c1-300
timeout -k 10 600 python bench.py --config c4 --no-cpu-baseline 2> $O/c4_bench.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print(d['metric'][-5:], 'value', round(d['value'],1), 'build_s', d['config']['solver_build_s'], 'warmup_s', d['config']['warmup_s'], 'per_step', [(s['ms'],s['active']) for s in d['per_step']], r['kernel'][:40], round(r['frac'],3))"
