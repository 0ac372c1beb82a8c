# round 4: back substitution with 32 loads per lane in flight -- tests, rates
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04
T=${1:-bs}
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_robustness.py tests/test_gpu_step_parity.py tests/test_gpu_full_size.py tests/test_gpu_n8192.py -x -q -m gpu > $O/${T}_tests.txt 2>&1
echo "pytest rc=$?" >> $O/${T}_tests.txt
tail -3 $O/${T}_tests.txt
grep -q "rc=0" $O/${T}_tests.txt || { echo TESTS FAILED; exit 1; }
LU_BATCH_KERNELS=1 timeout -k 10 300 python tools/lu_batch_rates.py 8 32 64 181 256 > $O/${T}_rates.txt 2>&1; cat $O/${T}_rates.txt
LU_N=1024 LU_BATCH_KERNELS=1 timeout -k 10 200 python tools/lu_batch_rates.py 32 256 2>&1 | grep "G="
LU_N=8192 LU_BATCH_KERNELS=1 timeout -k 10 200 python tools/lu_batch_rates.py 16 2>&1 | grep "G="
