# round 3: pipelined 4-wave H build (build_h_mt4_kernel) + tree plan -- bit-equality tests, then time by sub-stream count
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_mt19937.py tests/test_gpu_kernels.py -x -q > $O/mt4_tests.txt 2>&1 || { tail -30 $O/mt4_tests.txt; exit 1; }
tail -3 $O/mt4_tests.txt
rm -f $O/mt4_rates.txt
for s in 0 4 8 16; do
  echo "## MAUS_MT_SUBSTREAMS=$s (0 = plan's rule)" >> $O/mt4_rates.txt
  MAUS_MT_SUBSTREAMS=$s LU_BATCH_KERNELS=1 timeout -k 10 300 python tools/lu_batch_rates.py 32 181 256 >> $O/mt4_rates.txt 2>> $O/mt4_rates.err
done
cat $O/mt4_rates.txt
timeout -k 10 400 rocprofv3 --kernel-trace -d $O/trace_mt4 -o t -- python3 tools/lu_batch_rates.py 181 > $O/trace_mt4.log 2>&1
