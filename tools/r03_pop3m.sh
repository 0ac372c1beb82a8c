cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03
mkdir -p $O
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > $O/pop3m_tests.txt 2>&1
echo "pytest rc=$?" >> $O/pop3m_tests.txt
tail -4 $O/pop3m_tests.txt
for v in 1 0; do
for c in c3 c5; do
  MAUS_POPGEMM_3M=$v timeout -k 10 300 python bench.py --config $c --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('pop3m=$v', d['metric'][-5:], 'value', round(d['value'],1), 'step_frac', round(d['step_frac_of_mfma_peak'],3), 'gemm', round(r['achieved'],1), round(r.get('achieved_algorithmic_8mnk',0),1), d['kernel_ms_profiled_pass'], [s['ms'] for s in d['per_step']][:6])"
done; done
