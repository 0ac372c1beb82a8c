# round 4: pivot-step instruction count -- bitwise before / after digests, kernel tests, rates
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04
T=${1:-piv}
mkdir -p $O
MAUS_LIB=$GRAFT_REPO_ROOT/tools/bin/libmaus_hip_old.so python tools/lu_digest.py > $O/${T}_digest_before.txt 2>&1
python tools/lu_digest.py > $O/${T}_digest_after.txt 2>&1
if cmp -s $O/${T}_digest_before.txt $O/${T}_digest_after.txt; then echo "DIGESTS EQUAL"; else echo "DIGESTS DIFFER"; diff $O/${T}_digest_before.txt $O/${T}_digest_after.txt; fi
cat $O/${T}_digest_after.txt
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_robustness.py -x -q -m gpu > $O/${T}_tests.txt 2>&1
echo "pytest rc=$?" >> $O/${T}_tests.txt
tail -3 $O/${T}_tests.txt
LU_BATCH_KERNELS=1 timeout -k 10 300 python tools/lu_batch_rates.py 32 181 256 > $O/${T}_rates.txt 2>&1; cat $O/${T}_rates.txt
LU_N=1024 LU_BATCH_KERNELS=1 timeout -k 10 200 python tools/lu_batch_rates.py 256 2>&1 | grep "G="
