# round 4: configs[1] after the two-window jump kernel (the tile-shape check of the K >= 256 updates that ran here used a temporary build switch: profiles/r04_population_products.txt)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04
T=${1:-c2b}
mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_mt19937.py -x -q > $O/${T}_tests.txt 2>&1; tail -3 $O/${T}_tests.txt
LU_N=1024 LU_BATCH_KERNELS=1 timeout -k 10 200 python tools/lu_batch_rates.py 32 256 > $O/${T}_rates.txt 2>&1; cat $O/${T}_rates.txt
rm -rf $O/trace && mkdir -p $O/trace
LU_N=1024 timeout -k 10 300 rocprofv3 --kernel-trace -d $O/trace -o t -- python3 tools/lu_batch_rates.py 256 > $O/${T}_trace.out 2>&1
DB=$(find $O/trace -name "*.db" | head -1)
python3 tools/trace_db.py $DB trsm panel zgemm > $O/${T}_trace_last_call.txt 2>&1
rm -rf $O/trace
head -30 $O/${T}_trace_last_call.txt
timeout -k 10 300 python bench.py --config c2 --no-cpu-baseline > $O/${T}_bench.json 2> $O/${T}_bench.err; python tools/bench_summary.py $O/${T}_bench.json
