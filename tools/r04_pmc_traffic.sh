# round 4: HBM traffic per kernel class (separate FETCH_SIZE / WRITE_SIZE passes of one 256-solve step) + herm timing
set -e
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_r04t
rm -rf $O && mkdir -p $O
cd $R
P="bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-isolated --no-small-batch --no-side --skip-diagnosis --kernel-events off"
rm -f $O/lu_trace_pmc.txt
MAUS_LU_TRACE=$O/lu_trace_pmc.txt rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_FETCH_SIZE -- python3 $P > $O/pmc_fetch.json 2> $O/pmc_fetch.err
rm -f $O/lu_trace_pmc.txt
MAUS_LU_TRACE=$O/lu_trace_pmc.txt rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_WRITE_SIZE -- python3 $P > $O/pmc_write.json 2> $O/pmc_write.err
python3 tools/pmc_traffic.py $O/lu_trace_pmc.txt $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE > $O/pmc_traffic_per_kernel.txt
cp gpurun_out/pmc_traffic_per_kernel.json gpurun_out/zgemm_pmc_traffic.json $O/ 2>/dev/null || true
find $O -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
cat $O/pmc_traffic_per_kernel.txt
MAUS_HERM_TIMING=1 python tools/herm_eigh_time.py 8192 2>&1 | grep -E "enqueue|n=8192"
rm -rf $R/gpurun_out/r04/hs && mkdir -p $R/gpurun_out/r04/hs
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r04/hs -- python3 tools/herm_eigh_time.py 8192 > /dev/null 2>&1
find $R/gpurun_out/r04/hs -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/r04/herm_kernel_stats_8192.csv \;
rm -rf $R/gpurun_out/r04/hs
head -8 $R/gpurun_out/r04/herm_kernel_stats_8192.csv | cut -c1-50,180-330
