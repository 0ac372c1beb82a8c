set -e
MAUS_LU_PARTITION=4 timeout -k 10 300 python -m pytest tests/test_gpu_bench_path.py -m gpu -q -x -k "two_streams or status_codes" > gpurun_out/part_tests.log 2>&1 || { tail -30 gpurun_out/part_tests.log; exit 1; }
tail -1 gpurun_out/part_tests.log
for k in 4 3 2; do
MAUS_LU_PARTITION=$k timeout -k 10 300 python bench.py --no-cpu-baseline --no-small-batch --no-isolated --steps 12 --warmup 5 > gpurun_out/part$k.json 2> gpurun_out/part$k.err
python tools/bench_summary.py gpurun_out/part$k.json | cut -c1-120
done
