# round 4: the SVD power step reuses the residual's product -- tests, configs[4] bench
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04
T=${1:-av}
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_robustness.py tests/test_gpu_step_parity.py tests/test_gpu_evolve.py -x -q -m gpu > $O/${T}_tests.txt 2>&1
echo "pytest rc=$?" >> $O/${T}_tests.txt
tail -3 $O/${T}_tests.txt
grep -q "rc=0" $O/${T}_tests.txt || { grep -n "Error\|assert" $O/${T}_tests.txt | head -20; echo TESTS FAILED; exit 1; }
timeout -k 10 300 python bench.py --config c5 --no-cpu-baseline > $O/${T}_bench_c5.json 2> $O/${T}_bench_c5.err; python tools/bench_summary.py $O/${T}_bench_c5.json
timeout -k 10 300 python tools/host_profile.py c5 3 > $O/${T}_host_c5.txt 2>&1; head -12 $O/${T}_host_c5.txt
