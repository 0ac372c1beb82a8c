python -m pytest tests/test_gpu_dist_nccl.py -x -q > gpurun_out/r02_nccl_test.log 2>&1 || true
grep -n "Error\|error" gpurun_out/r02_nccl_test.log | head -10; tail -3 gpurun_out/r02_nccl_test.log
python bench.py > gpurun_out/r02_bench_final.json 2> gpurun_out/r02_bench_final.err
tail -c 600 gpurun_out/r02_bench_final.json
