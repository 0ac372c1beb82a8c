set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_mt19937.py -x -q > $O/mt4b_tests.txt 2>&1 || { tail -30 $O/mt4b_tests.txt; exit 1; }
tail -3 $O/mt4b_tests.txt
rm -f $O/mt4b_rates.txt
for s in 0 4 8 16; do
  echo "## MAUS_MT_SUBSTREAMS=$s (0 = plan's rule)" >> $O/mt4b_rates.txt
  MAUS_MT_SUBSTREAMS=$s LU_BATCH_KERNELS=1 timeout -k 10 300 python tools/lu_batch_rates.py 32 181 256 >> $O/mt4b_rates.txt 2>> $O/mt4b_rates.err
done
echo "## MAUS_PANEL_PW8=1" >> $O/mt4b_rates.txt
MAUS_PANEL_PW8=1 LU_BATCH_KERNELS=1 timeout -k 10 300 python tools/lu_batch_rates.py 181 256 >> $O/mt4b_rates.txt 2>> $O/mt4b_rates.err
cat $O/mt4b_rates.txt
