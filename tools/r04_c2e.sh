# round 4: configs[1]: sub-batch streams at n = 1024, 256 solves
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04
T=${1:-c2e}
mkdir -p $O
for s in 1 2 3 4; do
  echo "MAUS_LU_STREAMS=$s"
  MAUS_LU_STREAMS=$s LU_N=1024 timeout -k 10 200 python tools/lu_batch_rates.py 128 256
done > $O/${T}_rates.txt 2>&1; cat $O/${T}_rates.txt
timeout -k 10 300 python bench.py --config c2 --no-cpu-baseline > $O/${T}_bench.json 2> $O/${T}_bench.err; python tools/bench_summary.py $O/${T}_bench.json
MAUS_LU_STREAMS=2 timeout -k 10 300 python bench.py --config c2 --no-cpu-baseline > $O/${T}_bench2.json 2> $O/${T}_bench2.err; python tools/bench_summary.py $O/${T}_bench2.json
