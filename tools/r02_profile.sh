# Round-2 profile collection on the GPU box (see profiles/README.md).  Outputs under gpurun_out/prof_r02/.
set -e
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_r02
rm -rf $O && mkdir -p $O
cd $R
B="bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-isolated --no-small-batch --skip-diagnosis"
# (1) kernel-trace + stats, default (three sub-batch streams)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/default -- python3 $B > $O/default.json 2> $O/default.err
# (2) the same on one stream, with the library's launch log (splits the zgemm dispatches by K)
MAUS_LU_STREAMS=1 MAUS_LU_TRACE=$O/lu_trace_single.txt rocprofv3 --kernel-trace --stats --output-format csv -d $O/single -- python3 $B > $O/single.json 2> $O/single.err
python3 tools/k256_durations.py $O/lu_trace_single.txt $O/single > $O/zgemm_durations_by_k_single_stream.txt
# (3) HBM traffic: separate --pmc passes (FETCH_SIZE, WRITE_SIZE) of ONE step of 256 solves on one stream
P="bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-isolated --no-small-batch --skip-diagnosis --kernel-events off"
MAUS_LU_STREAMS=1 MAUS_LU_TRACE=$O/lu_trace_pmc.txt rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_FETCH_SIZE -- python3 $P > $O/pmc_fetch.json 2> $O/pmc_fetch.err
rm -f $O/lu_trace_pmc.txt
MAUS_LU_STREAMS=1 MAUS_LU_TRACE=$O/lu_trace_pmc.txt rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_WRITE_SIZE -- python3 $P > $O/pmc_write.json 2> $O/pmc_write.err
python3 tools/pmc_traffic.py $O/lu_trace_pmc.txt $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE > $O/pmc_traffic_per_kernel.txt
cp gpurun_out/pmc_traffic_per_kernel.json gpurun_out/zgemm_pmc_traffic.json $O/ 2>/dev/null || true
# keep only the small summaries (the traces are hundreds of MB)
find $O -name "*kernel_stats.csv" -exec sh -c 'cp "$1" "$2/$(basename $(dirname $(dirname "$1")))_kernel_stats.csv"' _ {} $O \;
find $O -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
ls -la $O
