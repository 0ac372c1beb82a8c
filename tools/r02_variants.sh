python -m pytest tests/test_gpu_evolve.py -q > gpurun_out/r02_gputests4.log 2>&1
python -m pytest tests/test_gpu_bench_path.py -q -k "long" >> gpurun_out/r02_gputests4.log 2>&1
grep -n "passed\|failed\|^E  \|Error" gpurun_out/r02_gputests4.log | cut -c1-250 | head -20
B="python bench.py --pop 32 --steps 6 --warmup 2 --no-cpu-baseline --no-small-batch"
$B --kernel-events all > gpurun_out/r02_pop32_base.json
MAUS_LU_MIN_SUB=8 MAUS_LU_STREAMS=2 $B --no-isolated > gpurun_out/r02_pop32_s2.json
MAUS_LU_MIN_SUB=8 MAUS_LU_STREAMS=4 $B --no-isolated > gpurun_out/r02_pop32_s4.json
MAUS_LU_MIN_SUB=16 MAUS_LU_STREAMS=3 python bench.py --pop 64 --steps 6 --warmup 2 --no-cpu-baseline --no-small-batch --no-isolated > gpurun_out/r02_pop64_s3.json
python bench.py --pop 64 --steps 6 --warmup 2 --no-cpu-baseline --no-small-batch --no-isolated > gpurun_out/r02_pop64_base.json
timeout -k 10 700 python tools/c4_run.py 8192 128 > gpurun_out/r02_c4_8192.txt 2>&1
tail -8 gpurun_out/r02_c4_8192.txt
