# round 4: candidate state as structure-of-arrays: whole GPU suite, host profiles of c5 / c4, side benches
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04
T=${1:-soa}
mkdir -p $O
timeout -k 10 1000 python -m pytest tests -x -q -m gpu --durations=8 > $O/${T}_tests.txt 2>&1
echo "pytest rc=$?" >> $O/${T}_tests.txt
tail -16 $O/${T}_tests.txt
grep -q "rc=0" $O/${T}_tests.txt || { echo TESTS FAILED; exit 1; }
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout -k 10 300 python tools/host_profile.py c5 3 > $O/${T}_host_c5.txt 2>&1; head -30 $O/${T}_host_c5.txt
timeout -k 10 300 python tools/host_profile.py c4 3 > $O/${T}_host_c4.txt 2>&1; head -24 $O/${T}_host_c4.txt
for c in c5 c4 c2 c3; do
  timeout -k 10 300 python bench.py --config $c --no-cpu-baseline > $O/${T}_bench_$c.json 2> $O/${T}_bench_$c.err; python tools/bench_summary.py $O/${T}_bench_$c.json
done
