set -e
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_full_size.py -x -q > gpurun_out/r02_ip_kernel_tests.log 2>&1 || true
tail -5 gpurun_out/r02_ip_kernel_tests.log
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-small-batch"
$B > gpurun_out/r02_v_implicit1.json
MAUS_LU_IMPLICIT=0 $B > gpurun_out/r02_v_implicit0.json
python -m pytest tests/test_gpu_bench_path.py -x -q --durations=12 > gpurun_out/r02_bench_path_tests.log 2>&1 || true
tail -12 gpurun_out/r02_bench_path_tests.log
