# round 4: multi-workgroup panel with tagged granules -- kernel / robustness tests, small-batch rates
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04
T=${1:-mw}
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_robustness.py tests/test_gpu_full_size.py -x -q -m gpu > $O/${T}_tests.txt 2>&1
echo "pytest rc=$?" >> $O/${T}_tests.txt
tail -5 $O/${T}_tests.txt
grep -q "rc=0" $O/${T}_tests.txt || { echo TESTS FAILED; exit 1; }
LU_BATCH_KERNELS=1 timeout -k 10 300 python tools/lu_batch_rates.py 8 16 32 64 128 > $O/${T}_rates.txt 2>&1; cat $O/${T}_rates.txt
