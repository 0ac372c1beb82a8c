# round 4: Hermitian reduction with three launches per column -- tests, timing, kernel stats, PMC passes
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_herm_eigh.py tests/test_gpu_n8192.py -x -q -m gpu > $O/herm_tests.txt 2>&1
echo "pytest rc=$?" >> $O/herm_tests.txt
tail -4 $O/herm_tests.txt
grep -q "rc=0" $O/herm_tests.txt || { echo TESTS FAILED; exit 1; }
MAUS_HERM_TIMING=1 timeout -k 10 300 python tools/herm_eigh_time.py 4096 8192 > $O/herm_time.txt 2>&1; cat $O/herm_time.txt
rm -rf $O/hs && mkdir -p $O/hs
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/hs -- python3 tools/herm_eigh_time.py 8192 > $O/herm_stats.out 2>&1
find $O/hs -name "*kernel_stats.csv" -exec cp {} $O/herm_kernel_stats_8192.csv \;
rm -rf $O/hs
head -14 $O/herm_kernel_stats_8192.csv | cut -c1-60,200-400
for cn in FETCH_SIZE WRITE_SIZE; do
  rm -rf $O/hp_$cn && mkdir -p $O/hp_$cn
  timeout -k 10 400 rocprofv3 --pmc $cn --kernel-include-regex "herm_col" --output-format csv -d $O/hp_$cn -- python3 tools/herm_eigh_time.py 4096 > $O/herm_pmc_$cn.out 2>&1 || { echo "PMC pass $cn failed"; tail -5 $O/herm_pmc_$cn.out; exit 1; }
done
rm -rf $O/ht && mkdir -p $O/ht
timeout -k 10 300 rocprofv3 --kernel-trace -d $O/ht -o t -- python3 tools/herm_eigh_time.py 4096 > $O/herm_trace.out 2>&1
DB=$(find $O/ht -name "*.db" | head -1)
python3 tools/herm_pmc.py 4096 $O/hp_FETCH_SIZE $O/hp_WRITE_SIZE $DB > $O/herm_pmc.txt 2>&1; cat $O/herm_pmc.txt
rm -rf $O/ht $O/hp_FETCH_SIZE $O/hp_WRITE_SIZE
