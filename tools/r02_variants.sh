python bench.py > gpurun_out/r02_bench_final.json 2> gpurun_out/r02_bench_final.err
python tools/bench_summary.py gpurun_out/r02_bench_final.json | cut -c1-300
bash tools/r02_profile.sh > gpurun_out/r02_profile.log 2>&1 || tail -20 gpurun_out/r02_profile.log
cat gpurun_out/prof_r02/zgemm_durations_by_k_single_stream.txt
