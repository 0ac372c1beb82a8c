# round 3: wave-synchronous H build (build_h_mtw_kernel) -- bit-equality tests, then its time by sub-stream count
set -e
O=gpurun_out/r03
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_mt19937.py -x -q > $O/mtw_tests.txt 2>&1 || { tail -30 $O/mtw_tests.txt; exit 1; }
tail -3 $O/mtw_tests.txt
for s in 0 8 16 32; do
  echo "## MAUS_MT_SUBSTREAMS=$s (0 = plan's rule)" >> $O/mtw_rates.txt
  MAUS_MT_SUBSTREAMS=$s LU_BATCH_KERNELS=1 timeout -k 10 300 python tools/lu_batch_rates.py 32 181 256 >> $O/mtw_rates.txt 2>> $O/mtw_rates.err
done
echo "## old kernel" >> $O/mtw_rates.txt
MAUS_BUILD_MT_OLD=1 LU_BATCH_KERNELS=1 timeout -k 10 300 python tools/lu_batch_rates.py 32 181 256 >> $O/mtw_rates.txt 2>> $O/mtw_rates.err
cat $O/mtw_rates.txt
