# round 4 baseline: per-kernel-class LU times at 32 / 181 / 256 / 290 solves and the driver-shaped bench, one box
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04
mkdir -p $O
LU_BATCH_KERNELS=1 timeout -k 10 300 python tools/lu_batch_rates.py 32 181 256 290 > $O/base_rates.txt 2>&1; cat $O/base_rates.txt
timeout -k 10 400 python bench.py > $O/base_c1.json 2> $O/base_c1.err && python tools/bench_summary.py $O/base_c1.json
