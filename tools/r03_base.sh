# round 3 (re-entry): baseline of the committed tree -- per-kernel breakdown of one LU call and the driver-shaped bench
set -e
O=gpurun_out/r03
mkdir -p $O
MAUS_LU_STREAMS=1 LU_BATCH_KERNELS=1 timeout -k 10 300 python tools/lu_batch_rates.py 32 64 181 256 > $O/base_kernels.txt 2> $O/base_kernels.err
cat $O/base_kernels.txt
timeout -k 10 400 python bench.py > $O/base_bench.json 2> $O/base_bench.err
python tools/bench_summary.py $O/base_bench.json 2>/dev/null || tail -c 1500 $O/base_bench.json
