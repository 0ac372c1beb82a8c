python -m pytest tests/test_gpu_dist_nccl.py -x -q > gpurun_out/r02_nccl_test.log 2>&1 || true
grep -n "Error\|error\|Traceback\|File \"<string>\"" gpurun_out/r02_nccl_test.log | head -20
tail -5 gpurun_out/r02_nccl_test.log
bash tools/r02_profile.sh > gpurun_out/r02_profile.log 2>&1 || tail -20 gpurun_out/r02_profile.log
tail -12 gpurun_out/r02_profile.log
