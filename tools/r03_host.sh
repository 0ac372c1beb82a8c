cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03
mkdir -p $O
timeout -k 10 200 python tools/host_profile.py c5 3 > $O/host_c5.txt 2>&1; tail -42 $O/host_c5.txt
timeout -k 10 200 python tools/host_profile.py c2 5 > $O/host_c2.txt 2>&1; tail -42 $O/host_c2.txt
for t in 0 1 2; do
  echo "MAUS_POPGEMM_TILE=$t"
  MAUS_POPGEMM_TILE=$t timeout -k 10 300 python bench.py --config c3 --no-cpu-baseline 2> $O/c3_tile$t.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('c3 value', round(d['value'],1), 'gemm TF', round(r['achieved'],2), 'avg launch ms', round(r['avg_launch_ms'],3), d['kernel_ms_profiled_pass'])"
done
