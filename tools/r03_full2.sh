cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03
mkdir -p $O
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > $O/full2_tests.txt 2>&1
echo "pytest rc=$?" >> $O/full2_tests.txt
tail -5 $O/full2_tests.txt
grep -q "rc=0" $O/full2_tests.txt || echo TESTS FAILED
LU_BATCH_KERNELS=1 timeout -k 10 300 python tools/lu_batch_rates.py 32 181 256 > $O/full2_rates.txt 2>&1; cat $O/full2_rates.txt
for c in c2 c5; do
  timeout -k 10 300 python bench.py --config $c --no-cpu-baseline 2> $O/full2_$c.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print(d['metric'][-5:], 'value', round(d['value'],1), 'step_frac', round(d['step_frac_of_mfma_peak'],3), 'dom', r['kernel'][:30], round(r['frac'],3), d['kernel_ms_profiled_pass'], [s['ms'] for s in d['per_step']][:8])"
done
