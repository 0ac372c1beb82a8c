cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_full_size.py -x -q > $O/c2_tests.txt 2>&1; tail -3 $O/c2_tests.txt
LU_N=1024 LU_BATCH_KERNELS=1 timeout -k 10 300 python tools/lu_batch_rates.py 256 331 > $O/c2_rates.txt 2>&1; cat $O/c2_rates.txt
LU_N=1024 MAUS_PANEL_RS=0 LU_BATCH_KERNELS=1 timeout -k 10 300 python tools/lu_batch_rates.py 256 331 >> $O/c2_rates.txt 2>&1; tail -2 $O/c2_rates.txt
for nbo in 256 384; do echo "NBO=$nbo"; LU_N=1024 MAUS_LU_NBO=$nbo LU_BATCH_KERNELS=1 timeout -k 10 300 python tools/lu_batch_rates.py 271 2>&1 | tail -1; done
LU_N=1024 timeout -k 10 300 rocprofv3 --kernel-trace -d $O/trace_c2 -o t -- python3 tools/lu_batch_rates.py 271 > $O/trace_c2.log 2>&1
LU_BATCH_KERNELS=1 timeout -k 10 300 python tools/lu_batch_rates.py 181 256 2>&1 | tail -2
timeout -k 10 300 python bench.py --config c2 --no-cpu-baseline 2> $O/c2b.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print(d['metric'][-5:], 'value', round(d['value'],1), 'step_frac', round(d['step_frac_of_mfma_peak'],3), d['kernel_ms_profiled_pass'])"
