python -m pytest tests -m gpu -q -x > gpurun_out/r02_gputests_full.log 2>&1
grep -n "passed\|failed\|^FAILED\|^E  " gpurun_out/r02_gputests_full.log | cut -c1-250 | head
python bench.py > gpurun_out/r02_bench_final.json 2> gpurun_out/r02_bench_final.err
python tools/bench_summary.py gpurun_out/r02_bench_final.json | cut -c1-300
