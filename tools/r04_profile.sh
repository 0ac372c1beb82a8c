# Round-4 profile collection on the GPU box (see profiles/README.md).  Outputs under gpurun_out/prof_r04/.
set -e
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_r04
rm -rf $O && mkdir -p $O
cd $R
B="bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-isolated --no-small-batch --no-side --skip-diagnosis"
# (1) kernel-trace + stats of the bench (one stream: the default), with the library's launch log (splits the zgemm dispatches by K)
MAUS_LU_TRACE=$O/lu_trace.txt rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $B > $O/stats.json 2> $O/stats.err
python3 tools/k256_durations.py $O/lu_trace.txt $O/stats > $O/zgemm_durations_by_k.txt
# (2) counters: separate --pmc passes of ONE step of 256 solves
P="bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-isolated --no-small-batch --no-side --skip-diagnosis --kernel-events off"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_A -- python3 $P > $O/pmc_A.json 2> $O/pmc_A.err
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_MFMA GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_B -- python3 $P > $O/pmc_B.json 2> $O/pmc_B.err
python3 tools/pmc_mfma.py $O/pmc_A $O/pmc_B $O/pmc_mfma_lds_per_kernel.json > $O/pmc_mfma_lds_per_kernel.txt
cat $O/pmc_mfma_lds_per_kernel.txt
rm -f $O/lu_trace_pmc.txt
MAUS_LU_TRACE=$O/lu_trace_pmc.txt rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_FETCH_SIZE -- python3 $P > $O/pmc_fetch.json 2> $O/pmc_fetch.err
rm -f $O/lu_trace_pmc.txt
MAUS_LU_TRACE=$O/lu_trace_pmc.txt rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_WRITE_SIZE -- python3 $P > $O/pmc_write.json 2> $O/pmc_write.err
python3 tools/pmc_traffic.py $O/lu_trace_pmc.txt $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE > $O/pmc_traffic_per_kernel.txt
cp gpurun_out/pmc_traffic_per_kernel.json gpurun_out/zgemm_pmc_traffic.json $O/ 2>/dev/null || true
# keep only the small summaries (the traces are hundreds of MB)
find $O -name "*kernel_stats.csv" -exec sh -c 'cp "$1" "$2/$(basename $(dirname $(dirname "$1")))_kernel_stats.csv"' _ {} $O \;
find $O -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
ls -la $O
