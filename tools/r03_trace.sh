# round 3: per-dispatch trace of one 181-solve call (trsm / jump kernels by grid size)
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03
mkdir -p $O
MAUS_MT_SUBSTREAMS=16 timeout -k 10 400 rocprofv3 --kernel-trace -d $O/trace16 -o t -- python3 tools/lu_batch_rates.py 181 > $O/trace16.log 2>&1
python tools/trace_by_grid.py $O/trace16 > $O/trace16_by_grid.txt
cat $O/trace16_by_grid.txt
find $O/trace16 -name "*.csv" -size +20M -delete
