# kernel-trace + stats of the driver-shaped bench on the final tree of round 3 (part (1) of tools/r03_profile.sh)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_r03f
rm -rf $O && mkdir -p $O
cd $R
B="bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-isolated --no-small-batch --skip-diagnosis"
MAUS_LU_TRACE=$O/lu_trace.txt timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $B > $O/stats.json 2> $O/stats.err
python3 tools/k256_durations.py $O/lu_trace.txt $O/stats > $O/zgemm_durations_by_k.txt
cat $O/zgemm_durations_by_k.txt
find $O -name "*kernel_stats.csv" -exec sh -c 'cp "$1" "$2/bench_kernel_stats.csv"' _ {} $O \;
find $O -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
rm -f $O/lu_trace.txt
tail -1 $O/stats.json | cut -c1-300
head -8 $O/bench_kernel_stats.csv
